// Diagnostic only: how much does a gather instruction cost as a function of which 64 gate indices share it?
// out = sum over i of packed[idx[i]], idx read coalesced; only the ORDER of idx differs between runs.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void gather_sum(const int* __restrict__ idx, long n, const float* __restrict__ packed,
                                                  float* __restrict__ out) {
  float acc = 0.0f;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const int a = idx[i], b = idx[i + stride], c = idx[i + 2 * stride], d = idx[i + 3 * stride];
    const float va = packed[a], vb = packed[b], vc = packed[c], vd = packed[d];
    acc += (va + vb) + (vc + vd);
  }
  for (; i < n; i += stride) acc += packed[idx[i]];
  if (acc == 123.456f) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream_only(const int* __restrict__ idx, long n, float* __restrict__ out) {
  int acc = 0;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) acc += (idx[i] ^ idx[i + stride]) + (idx[i + 2 * stride] ^ idx[i + 3 * stride]);
  for (; i < n; i += stride) acc += idx[i];
  if (acc == 123456789) out[0] = (float)acc;
}

extern "C" int gp_gather_sum(const void* idx, long n, const void* packed, float* out, int blocks, void* stream) {
  hipLaunchKernelGGL(gather_sum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const int*)idx, n, (const float*)packed, out);
  return (int)hipGetLastError();
}
extern "C" int gp_stream_only(const void* idx, long n, float* out, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_only, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const int*)idx, n, out);
  return (int)hipGetLastError();
}
