// K1  csr_apply -- the per-volume hot loop (replaces radar_grid/interpolate.py:69-104 and the per-field
// loop of :137-140).
//
// Roofline: HBM.  Algorithmic bytes per launch = 8*P (gate_idx + weight per pair) + sizeof(indptr)*(V+1)
//           + F*(5*G + 4*V)   (SURVEY.md §8(d)); the field gather is served by L2 / Infinity Cache.
//
// Mapping ("CSR-stream" at wavefront granularity, no workgroup barrier):
//   * one wavefront owns 64 consecutive voxel rows = one contiguous range of pairs [seg_b, seg_e);
//   * it walks that range in tiles: every lane loads 4 consecutive pairs with one dwordx4 for the indices
//     and one for the weights (1 KiB per wave-instruction, fully coalesced), gathers the packed field
//     value(s) of each gate with ONE load (mask folded into the value as a sentinel), and parks the masked
//     products (w*v, w) in the wave's private LDS tile;
//   * LDS is the transposition buffer from "pair order" to "row order": lane l then sums the part of ITS
//     row (r0 + l) that lies in the tile, in float64, straight out of LDS;
//   * after the last tile each lane writes its voxel: 64 consecutive floats per field (coalesced).
// Empty rows cost nothing, long rows only lengthen their own lane's loop, and there is no atomics / no
// inter-wave communication, so results are bit-reproducible run to run.
#include "rg_common.hpp"

namespace {

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

template <typename IndT, int NF, int STRIDE, int TILE>
__global__ __launch_bounds__(rg::kBlock) void csr_apply_kernel(
    const IndT* __restrict__ indptr, const int32_t* __restrict__ gidx, const float* __restrict__ wts,
    long n_vox, long n_pairs, const float* __restrict__ packed, unsigned last_gate, float fill,
    float* __restrict__ out) {
  static_assert(TILE % 256 == 0, "a wave loads 256 pairs per step");
  __shared__ float2 tile_all[rg::kBlock / rg::kWave][TILE * NF];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float2* tile = tile_all[wv];

  const unsigned blk = rg::xcd_remap(blockIdx.x, gridDim.x);
  const long r0 = ((long)blk * (rg::kBlock / rg::kWave) + wv) * 64;
  if (r0 >= n_vox) return;  // wave-uniform
  const long row = r0 + lane;
  const long ra = row < n_vox ? row : n_vox;
  const long rb = row + 1 < n_vox ? row + 1 : n_vox;
  const long rs = (long)indptr[ra];
  const long re = (long)indptr[rb];
  const long seg_b = (long)indptr[r0];
  const long seg_e = (long)indptr[r0 + 64 < n_vox ? r0 + 64 : n_vox];

  double acc_p[NF], acc_w[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) { acc_p[f] = 0.0; acc_w[f] = 0.0; }

  for (long t0 = seg_b & ~3L; t0 < seg_e; t0 += TILE) {
    // ---- stream phase: coalesced CSR loads, gather, masked products -> LDS -------------------------
#pragma unroll
    for (int it = 0; it < TILE / 256; ++it) {
      const int slot = it * 256 + lane * 4;
      const long j = t0 + slot;
      if (j < seg_e) {
        int id[4];
        float w[4];
        if (j + 4 <= n_pairs) {
          const int4 a = *reinterpret_cast<const int4*>(gidx + j);
          const float4 b = *reinterpret_cast<const float4*>(wts + j);
          id[0] = a.x; id[1] = a.y; id[2] = a.z; id[3] = a.w;
          w[0] = b.x; w[1] = b.y; w[2] = b.z; w[3] = b.w;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool in = j + k < n_pairs;
            id[k] = in ? gidx[j + k] : 0;
            w[k] = in ? wts[j + k] : 0.0f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned g = min((unsigned)id[k], last_gate);  // a corrupt index must not fault
          float v[STRIDE];
          load_packed<STRIDE>(packed, g, v);
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const bool ok = rg::f32_bits(v[f]) != RG_EXCLUDED_BITS;
            tile[(slot + k) * NF + f] = make_float2(ok ? w[k] * v[f] : 0.0f, ok ? w[k] : 0.0f);
          }
        }
      }
    }
    // LDS traffic of one wave is executed in order; the fences only pin the compiler.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- row phase: lane l sums the slice of row r0+l that lies inside this tile ------------------
    const long lo = (rs > t0 ? rs : t0) - t0;
    long hi = (re < t0 + TILE ? re : t0 + TILE) - t0;
    const int a = (int)lo;
    const int b = hi < 0 ? 0 : (int)hi;
    for (int j = a; j < b; ++j) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const float2 e = tile[j * NF + f];
        acc_p[f] += (double)e.x;
        acc_w[f] += (double)e.y;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  if (row < n_vox) {
#pragma unroll
    for (int f = 0; f < NF; ++f) out[(size_t)f * n_vox + row] = acc_w[f] > 0.0 ? (float)(acc_p[f] / acc_w[f]) : fill;
  }
}

template <typename IndT, int NF, int STRIDE, int TILE>
int launch(const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs, const float* packed,
           long n_gates, float fill, float* out, hipStream_t s) {
  const long chunks = (n_vox + 63) / 64;
  const long blocks = (chunks + 3) / 4;
  hipLaunchKernelGGL((csr_apply_kernel<IndT, NF, STRIDE, TILE>), dim3((unsigned)blocks), dim3(rg::kBlock), 0, s,
                     static_cast<const IndT*>(indptr), gidx, wts, n_vox, n_pairs, packed, (unsigned)(n_gates - 1),
                     fill, out);
  return rg::check_launch("rg_csr_apply_f32");
}

template <typename IndT>
int dispatch(int nf, const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs,
             const float* packed, long n_gates, float fill, float* out, hipStream_t s) {
  switch (nf) {
    case 1: return launch<IndT, 1, 1, 1024>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
    case 2: return launch<IndT, 2, 2, 512>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
    case 3: return launch<IndT, 3, 4, 256>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
    case 4: return launch<IndT, 4, 4, 256>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
    case 5: return launch<IndT, 5, 8, 256>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
    case 6: return launch<IndT, 6, 8, 256>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
    case 7: return launch<IndT, 7, 8, 256>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
    default: return launch<IndT, 8, 8, 256>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s);
  }
}

inline int stride_for(int nf) { return nf == 1 ? 1 : nf == 2 ? 2 : nf <= 4 ? 4 : 8; }

}  // namespace

extern "C" int rg_csr_apply_f32(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx,
                                const float* weights, int64_t n_vox, int64_t n_pairs, const float* packed,
                                int32_t n_fields, int32_t stride, int64_t n_gates, float fill_value, float* out,
                                rg_stream_t stream) {
  RG_REQUIRE(indptr && out, RG_EINVAL, "rg_csr_apply_f32: null indptr/out");
  RG_REQUIRE(n_vox >= 0 && n_pairs >= 0, RG_EINVAL, "rg_csr_apply_f32: negative size");
  RG_REQUIRE(n_fields >= 1 && n_fields <= RG_MAX_FIELDS, RG_EUNSUPPORTED, "rg_csr_apply_f32: n_fields=%d not in 1..%d",
             n_fields, RG_MAX_FIELDS);
  RG_REQUIRE(stride == stride_for(n_fields), RG_EINVAL, "rg_csr_apply_f32: stride=%d, expected %d for %d fields", stride,
             stride_for(n_fields), n_fields);
  RG_REQUIRE(n_pairs == 0 || (gate_idx && weights && packed && n_gates > 0), RG_EINVAL,
             "rg_csr_apply_f32: pairs present but gate_idx/weights/packed/n_gates missing");
  RG_REQUIRE(n_gates <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_apply_f32: n_gates exceeds int32 gate indices");
  RG_REQUIRE(n_vox <= 0x3FFFFFFFFFL, RG_EUNSUPPORTED, "rg_csr_apply_f32: n_vox too large for one launch");
  RG_REQUIRE(rg::aligned16(gate_idx) && rg::aligned16(weights) && rg::aligned16(packed), RG_EALIGN,
             "rg_csr_apply_f32: gate_idx, weights and packed must be 16-byte aligned");
  if (n_vox == 0) return RG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    return dispatch<int64_t>(n_fields, indptr, gate_idx, weights, n_vox, n_pairs, packed, n_gates, fill_value, out, s);
  return dispatch<int32_t>(n_fields, indptr, gate_idx, weights, n_vox, n_pairs, packed, n_gates, fill_value, out, s);
}
