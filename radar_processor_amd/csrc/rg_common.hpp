// Shared helpers for libradargrid_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "radargrid_hip.h"

namespace rg {

void set_error(const char* fmt, ...);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// checks the launch that was just enqueued; does not synchronise
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return RG_ELAUNCH;
  }
  return RG_OK;
}

constexpr int kWave = 64;        // CDNA4 wavefront
constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kNumXcd = 8;       // MI355X: 8 XCDs, workgroups dealt round-robin

// Bijective XCD-aware remap (cdna_hip_programming.md T1): workgroups b and b+8 share an XCD, so give
// every XCD one contiguous slab of the logical index space -> neighbouring voxel chunks (which gather the
// same gates) hit the same 4 MiB L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk / kNumXcd, r = nblk % kNumXcd;
  const unsigned xcd = bid % kNumXcd, k = bid / kNumXcd;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

__device__ __forceinline__ uint32_t f32_bits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float bits_f32(uint32_t v) { return __builtin_bit_cast(float, v); }

// One gather of a gate's packed field slots (layout of rg_pack_fields_f32: float32 [G][STRIDE], STRIDE in {1,2,4,8}).
template <int STRIDE>
__device__ __forceinline__ void load_packed(const float* __restrict__ p, unsigned g, float (&v)[STRIDE]) {
  if constexpr (STRIDE == 1) {
    v[0] = p[g];
  } else if constexpr (STRIDE == 2) {
    const float2 t = reinterpret_cast<const float2*>(p)[g];
    v[0] = t.x; v[1] = t.y;
  } else {
#pragma unroll
    for (int s = 0; s < STRIDE; s += 4) {
      const float4 t = reinterpret_cast<const float4*>(p)[(size_t)g * (STRIDE / 4) + s / 4];
      v[s] = t.x; v[s + 1] = t.y; v[s + 2] = t.z; v[s + 3] = t.w;
    }
  }
}

}  // namespace rg

#define RG_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      rg::set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)
