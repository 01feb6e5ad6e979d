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

// (3) blocked: workgroup b reads the contiguous block [b * block16, (b + 1) * block16) front to back, 4 KiB per step and
// four steps in flight -- the access FRONT of the row-wise kernel (every resident workgroup sits in its own stretch of
// the array) without any of its arithmetic.  block16 = 256 (4 KiB) is the grid-stride pattern again.
__global__ __launch_bounds__(256) void read_blocked(const float4* __restrict__ a, long n16, long block16,
                                                    float* __restrict__ out) {
  const long beg = (long)blockIdx.x * block16;
  const long end = beg + block16 < n16 ? beg + block16 : n16;
  float acc = 0.0f;
  long i = beg + threadIdx.x;
  for (; i + 3 * 256 < end; i += 4 * 256) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < end; i += 256) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}

// (4) blocked read + a small write per workgroup: after reading its block, every `every`-th workgroup writes
// `wbytes * every` bytes (dense: at wr + group * wbytes * every) -- the row-wise kernel's 1 KiB of grid per 68 KiB of
// records, with the writes batched `every` workgroups at a time.  mode 1: the four 256-byte pieces of a workgroup go to
// four lines 8000 bytes apart (the bench grid's layout) instead of one dense KiB; mode -1: the same shape with every piece
// aligned to 256 bytes (lines 8192 bytes apart); mode >= 2: spread over that many slabs.
__global__ __launch_bounds__(256) void read_blocked_write(const float4* __restrict__ a, long n16, long block16,
                                                          float* __restrict__ wr, int wfloats, int every, int mode) {
  const long beg = (long)blockIdx.x * block16;
  const long end = beg + block16 < n16 ? beg + block16 : n16;
  float acc = 0.0f;
  long i = beg + threadIdx.x;
  for (; i + 3 * 256 < end; i += 4 * 256) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < end; i += 256) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
  if (blockIdx.x % every == 0) {
    const long base = (long)(blockIdx.x / every) * wfloats * every;
    if (mode == 0) {
      for (int k = threadIdx.x; k < wfloats * every; k += 256) wr[base + k] = acc;
    } else if (mode >= 2) {     // spread: workgroup b writes its piece into slab b % mode of `mode` equal slabs of the array
      const long per_slab = ((long)gridDim.x + mode - 1) / mode;
      const long piece = (long)(blockIdx.x % mode) * per_slab + blockIdx.x / mode;
      for (int k = threadIdx.x; k < wfloats; k += 256) wr[piece * wfloats + k] = acc;
    } else if (mode == -1) {   // the grid layout's shape with ALIGNED pieces: lines 2048 floats apart, 64 floats per piece
      const long grp = blockIdx.x / 32, col = blockIdx.x % 32;
      const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
      wr[(grp * 4 + w) * 2048 + col * 64 + lane] = acc;
    } else {            // 4 lines x 64 floats, lines 2000 floats apart, 32 workgroups per line group, 62-float pieces:
      const long grp = blockIdx.x / 32, col = blockIdx.x % 32;          // every piece straddles three 128-byte lines
      const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
      wr[(grp * 4 + w) * 2000 + col * 62 + lane] = acc;
    }
  }
}

extern "C" int bw_read_blocked_write(const void* a, long bytes, long block_bytes, float* wr, int wbytes, int every, int mode,
                                     void* stream) {
  const long n16 = bytes / 16, block16 = block_bytes / 16;
  const long blocks = (n16 + block16 - 1) / block16;
  hipLaunchKernelGGL(read_blocked_write, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)a, n16,
                     block16, wr, wbytes / 4, every, mode);
  return (int)hipGetLastError();
}

extern "C" int bw_read_blocked(const void* a, long bytes, long block_bytes, float* out, void* stream) {
  const long n16 = bytes / 16, block16 = block_bytes / 16;
  const long blocks = (n16 + block16 - 1) / block16;
  hipLaunchKernelGGL(read_blocked, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)a, n16, block16,
                     out);
  return (int)hipGetLastError();
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
