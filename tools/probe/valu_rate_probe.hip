// VALU issue-rate probe for gfx950: how many cycles does a wave64 instruction of each kind occupy a SIMD?
// Each wavefront runs ITER iterations of 8 independent instructions of one kind (inline asm: nothing is folded away);
// enough wavefronts (8 per SIMD) to keep every SIMD issuing.  Host side: tools/probe/valu_rate_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void valu_rate_kernel(float* out, int iters, float seed) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 a[8], b, c;
  float s[8];
  b = (f32x2){seed, seed * 0.5f};
  c = (f32x2){1.0f, 0.25f};
  unsigned m = 0x38003838u + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (f32x2){seed + i, seed - i}; s[i] = seed * i; }
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {          // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
      REP8(X)
#undef X
    } else if constexpr (KIND == 1) {   // v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if constexpr (KIND == 2) {   // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP8(X)
#undef X
    } else if constexpr (KIND == 3) {   // v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP8(X)
#undef X
    } else if constexpr (KIND == 4) {   // v_add_f32
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c.x));
      REP8(X)
#undef X
    } else if constexpr (KIND == 5) {   // v_mul_legacy_f32
#define X(i) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c.x));
      REP8(X)
#undef X
    } else if constexpr (KIND == 6) {   // v_cndmask_b32 (vcc)
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(c.x) : );
      REP8(X)
#undef X
    } else if constexpr (KIND == 7) {   // v_cvt_pk_f32_fp8
#define X(i) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(a[i]) : "v"(m));
      REP8(X)
#undef X
    } else if constexpr (KIND == 8) {   // v_cvt_f32_ubyte0
#define X(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(s[i]) : "v"(m));
      REP8(X)
#undef X
    } else if constexpr (KIND == 9) {   // v_pk_fma_f32 with a broadcast operand (op_sel_hi:[0,1,1]), as the row-wise kernel issues it
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if constexpr (KIND == 10) {  // v_exp_f32 (transcendental)
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
      REP8(X)
#undef X
    } else if constexpr (KIND == 11) {  // v_and_or_b32
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m), "v"(c.x));
      REP8(X)
#undef X
    } else if constexpr (KIND == 12) {  // v_mov_b32 dpp quad_perm
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(s[i]));
      REP8(X)
#undef X
    } else if constexpr (KIND == 13) {  // v_fma_f64
      double d[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = seed + i;
      double e = 1.000001, g = 0.5;
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(e), "v"(g));
      for (int k = 0; k < 1; ++k) { REP8(X) }
#undef X
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += (float)d[i];
    }
  }
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
  if (acc == 123.456f) out[threadIdx.x] = acc;    // practically never
}

extern "C" int valu_rate_probe(int kind, int blocks, int iters, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define L(K) case K: hipLaunchKernelGGL((valu_rate_kernel<K>), dim3(blocks), dim3(256), 0, s, out, iters, 1.5f); break;
  switch (kind) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) default: return -1; }
#undef L
  return (int)hipGetLastError();
}
