// Diagnostic only (not part of libradargrid_hip.so): what read bandwidth does an MI355X deliver to a kernel that
// streams one or two large arrays and does nothing else?  Gives the "measured ceiling" DESIGN.md prices K1 against.
#include <hip/hip_runtime.h>
#include <stdint.h>

// (1) one array, 16 bytes per lane per load, `unroll` independent loads in flight, grid-stride
template <int UNROLL>
__global__ __launch_bounds__(256) void read_linear(const float4* __restrict__ a, long n16, float* __restrict__ out) {
  float acc = 0.0f;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n16; i += stride) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;   // never true in practice: keeps the loads alive without a store stream
}

// (2) K1's skeleton: a wavefront owns a contiguous range of `per_wave` pairs of TWO arrays and walks it in 512-pair
// tiles with dword loads (256 contiguous bytes per wave-instruction), two tiles in flight
__global__ __launch_bounds__(256) void read_two_streams(const int* __restrict__ a, const float* __restrict__ b, long n,
                                                        int per_wave, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long beg = wave * per_wave;
  if (beg >= n) return;
  const long end = beg + per_wave < n ? beg + per_wave : n;
  float acc = 0.0f;
  int ci[8]; float cw[8]; int di[8]; float dw[8];
  auto load = [&](long t, int (&xi)[8], float (&xw)[8]) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      long k = t + it * 64 + lane; k = k < n ? k : n - 1;
      xi[it] = a[k]; xw[it] = b[k];
    }
  };
  long t = beg;
  load(t, ci, cw);
  for (; t < end; t += 1024) {
    if (t + 512 < end) load(t + 512, di, dw);
#pragma unroll
    for (int it = 0; it < 8; ++it) acc += cw[it] + (float)ci[it];
    if (t + 1024 < end) load(t + 1024, ci, cw);
    if (t + 512 < end) {
#pragma unroll
      for (int it = 0; it < 8; ++it) acc += dw[it] + (float)di[it];
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

extern "C" int bw_read_linear(const void* a, long bytes, int blocks, int unroll, float* out, void* stream) {
  const long n16 = bytes / 16;
  hipStream_t s = (hipStream_t)stream;
  if (unroll == 8) hipLaunchKernelGGL(read_linear<8>, dim3(blocks), dim3(256), 0, s, (const float4*)a, n16, out);
  else if (unroll == 4) hipLaunchKernelGGL(read_linear<4>, dim3(blocks), dim3(256), 0, s, (const float4*)a, n16, out);
  else hipLaunchKernelGGL(read_linear<2>, dim3(blocks), dim3(256), 0, s, (const float4*)a, n16, out);
  return (int)hipGetLastError();
}

extern "C" int bw_read_two_streams(const void* a, const void* b, long n, int per_wave, float* out, void* stream) {
  const long waves = (n + per_wave - 1) / per_wave;
  const long blocks = (waves + 3) / 4;
  hipLaunchKernelGGL(read_two_streams, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const int*)a,
                     (const float*)b, n, per_wave, out);
  return (int)hipGetLastError();
}
