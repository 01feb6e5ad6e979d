// K1  csr_apply -- the per-volume hot loop (replaces radar_grid/interpolate.py:69-104 and the per-field
// loop of :137-140).
//
// Roofline: HBM.  Algorithmic bytes per launch = 8*P (gate_idx + weight per pair) + sizeof(indptr)*(V+1)
//           + F*(5*G + 4*V)   (SURVEY.md §8(d)); the field gather is served by L2 / Infinity Cache.
//
// Mapping: "CSR-stream" at wavefront granularity -- coalesced dwordx4 streaming of (gate_idx, weight), one
// gather per pair with the mask folded into the value, LDS as the pair-order -> row-order transposition
// buffer, register prefetch of the next tile; details at csr_apply_kernel below.
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

// Streaming loads of the CSR: one dwordx4 of indices + one of weights per lane (4 pairs).  NT = non-temporal
// hint: the CSR is read exactly once, so it should not displace the packed field values (the gather's
// working set) from the XCD's L2.
template <bool NT>
__device__ __forceinline__ void load_csr_quad(const int32_t* __restrict__ gidx, const float* __restrict__ wts, long j,
                                              int4& ci, float4& cw) {
  using i4 = int __attribute__((ext_vector_type(4)));
  using f4 = float __attribute__((ext_vector_type(4)));
  if constexpr (NT) {
    const i4 a = __builtin_nontemporal_load(reinterpret_cast<const i4*>(gidx + j));
    const f4 b = __builtin_nontemporal_load(reinterpret_cast<const f4*>(wts + j));
    ci = make_int4(a.x, a.y, a.z, a.w);
    cw = make_float4(b.x, b.y, b.z, b.w);
  } else {
    ci = *reinterpret_cast<const int4*>(gidx + j);
    cw = *reinterpret_cast<const float4*>(wts + j);
  }
}

// One wavefront = 64 consecutive voxel rows = one contiguous pair range, walked in tiles of TILE pairs:
//
//   stream phase   every lane owns 4 consecutive pairs per 256-pair step: the (idx, w) quads were prefetched
//                  into registers one tile ahead; gather the packed field value(s) of the 4 gates, form the
//                  masked float32 products (w*v, w) and park them in the wave's private LDS tile; then issue
//                  the NEXT tile's CSR loads so they are in flight during the row phase;
//   row phase      LDS is the transposition buffer from pair order to row order.  4 lanes share a row: in
//                  pass p (p = 0..3) lane l works for row 16p + l/4 and sums every 4th element of that row's
//                  slice of the tile (float32 partials over <= TILE/4 terms, folded into float64 accumulators
//                  per tile), so ~11 rows of ~50 pairs keep 44 lanes busy for ~13 steps instead of 11 lanes
//                  for ~50.  Passes whose 16 rows do not touch the tile are skipped wave-uniformly.
//   epilogue       quad-reduce the 4 sub-lane accumulators, move row r0+l's result to lane l, one coalesced
//                  256-byte store per field.
//
// No workgroup barrier, no atomics, no inter-wave communication: results are bit-reproducible.
template <typename IndT, int NF, int STRIDE, int TILE, bool NT>
__global__ __launch_bounds__(rg::kBlock) void csr_apply_kernel(
    const IndT* __restrict__ indptr, const int32_t* __restrict__ gidx, const float* __restrict__ wts,
    long n_vox, long n_pairs, const float* __restrict__ packed, unsigned last_gate, float fill,
    float* __restrict__ out) {
  static_assert(TILE % 256 == 0, "a wave loads 256 pairs per step");
  constexpr int IT = TILE / 256;
  __shared__ float2 tile_all[rg::kBlock / rg::kWave][TILE * NF];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float2* tile = tile_all[wv];

  const unsigned blk = rg::xcd_remap(blockIdx.x, gridDim.x);
  const long r0 = ((long)blk * (rg::kBlock / rg::kWave) + wv) * 64;
  if (r0 >= n_vox) return;  // wave-uniform
  const long row = r0 + lane;
  const long seg_b = (long)indptr[r0];
  const long seg_e = (long)indptr[r0 + 64 < n_vox ? r0 + 64 : n_vox];
  const long tb = seg_b & ~3L;                       // 16-byte aligned origin of the tile walk
  const int span = (int)(seg_e - tb);                // pairs to walk (a 64-row chunk never holds 2^31 pairs)
  // row bounds as offsets from tb; lane l <-> row r0+l
  const int rs_o = (int)((long)indptr[row < n_vox ? row : n_vox] - tb);
  const int re_o = (int)((long)indptr[row + 1 < n_vox ? row + 1 : n_vox] - tb);
  // quad layout for the row phase: in pass p lane l serves row 16p + (l >> 2), sub-lane q = l & 3
  const int q = lane & 3;
  int qs[4], qe[4], ps[4], pe[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    qs[p] = __shfl(rs_o, 16 * p + (lane >> 2), 64);
    qe[p] = __shfl(re_o, 16 * p + (lane >> 2), 64);
    ps[p] = __builtin_amdgcn_readlane(rs_o, 16 * p);        // pair span of the pass's 16 rows (wave-uniform)
    pe[p] = __builtin_amdgcn_readlane(re_o, 16 * p + 15);
  }

  double acc_p[4][NF], acc_w[4][NF];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int f = 0; f < NF; ++f) { acc_p[p][f] = 0.0; acc_w[p][f] = 0.0; }

  if (span > 0) {
    // gate_idx / weights are readable up to the next multiple of 4 elements (C-ABI contract), so a clamped
    // quad address is always safe; slots outside [seg_b, seg_e) are computed but never read back.
    const long last_quad = (n_pairs - 1) & ~3L;
    int4 ci[IT];
    float4 cw[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const long j = tb + it * 256 + lane * 4;
      load_csr_quad<NT>(gidx, wts, j < last_quad ? j : last_quad, ci[it], cw[it]);
    }
    for (int t = 0; t < span; t += TILE) {
      // ---- stream phase -----------------------------------------------------------------------------
      float val[IT][4][STRIDE];
      float4 w_cur[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int id[4] = {ci[it].x, ci[it].y, ci[it].z, ci[it].w};
        w_cur[it] = cw[it];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          load_packed<STRIDE>(packed, min((unsigned)id[k], last_gate), val[it][k]);  // clamp: never fault
      }
      if (t + TILE < span) {  // prefetch the next tile's CSR quads; they fly during the row phase
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const long j = tb + t + TILE + it * 256 + lane * 4;
          load_csr_quad<NT>(gidx, wts, j < last_quad ? j : last_quad, ci[it], cw[it]);
        }
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const float w[4] = {w_cur[it].x, w_cur[it].y, w_cur[it].z, w_cur[it].w};
        const int slot = it * 256 + lane * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const bool ok = rg::f32_bits(val[it][k][f]) != RG_EXCLUDED_BITS;
            tile[(slot + k) * NF + f] = make_float2(ok ? w[k] * val[it][k][f] : 0.0f, ok ? w[k] : 0.0f);
          }
        }
      }
      // LDS traffic of one wave executes in order; the fences only pin the compiler.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- row phase --------------------------------------------------------------------------------
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (ps[p] < t + TILE && pe[p] > t) {  // wave-uniform: this pass's rows touch the tile
          const int a = (qs[p] > t ? qs[p] : t) - t;
          const int b = (qe[p] < t + TILE ? qe[p] : t + TILE) - t;
          float part_p[NF], part_w[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) { part_p[f] = 0.0f; part_w[f] = 0.0f; }
          for (int j = a + q; j < b; j += 4) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const float2 e = tile[j * NF + f];
              part_p[f] += e.x;
              part_w[f] += e.y;
            }
          }
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            acc_p[p][f] += (double)part_p[f];
            acc_w[p][f] += (double)part_w[f];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }

  // ---- epilogue: quad reduce, transpose back to lane == row, coalesced store -------------------------
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    float res = fill;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double sp = acc_p[p][f], sw = acc_w[p][f];
      sp += __shfl_xor(sp, 1, 64); sw += __shfl_xor(sw, 1, 64);
      sp += __shfl_xor(sp, 2, 64); sw += __shfl_xor(sw, 2, 64);
      const float r = sw > 0.0 ? (float)(sp / sw) : fill;
      const float moved = __shfl(r, 4 * (lane & 15), 64);    // row 16p + k lives in lane 4k
      if ((lane >> 4) == p) res = moved;
    }
    if (row < n_vox) out[(size_t)f * n_vox + row] = res;
  }
}

// XCD placement of the 64-row chunks (speed only, never correctness):
//   kXcdNone   workgroup b -> logical block b: neighbouring blocks land on different XCDs (round-robin deal),
//              perfectly balanced when pair density varies with height, every L2 sees the same gate window;
//   kXcdGroup  groups of 32 consecutive logical blocks stay on one XCD, groups rotate over the XCDs;
//   kXcdSlab   one contiguous eighth of the grid per XCD (best L2 locality, worst balance: top levels are sparse).
constexpr int kXcdNone = 0, kXcdGroup = 1, kXcdSlab = 2;

template <int MODE>
__device__ __forceinline__ unsigned place_block(unsigned bid, unsigned nblk) {
  if constexpr (MODE == kXcdSlab) {
    return rg::xcd_remap(bid, nblk);
  } else if constexpr (MODE == kXcdGroup) {
    constexpr unsigned S = 32;
    const unsigned super = S * rg::kNumXcd;
    const unsigned full = nblk / super * super;           // only whole super-groups are permuted (bijective)
    if (bid >= full) return bid;
    const unsigned base = bid / super * super, r = bid % super;
    return base + (r % rg::kNumXcd) * S + r / rg::kNumXcd;
  } else {
    return bid;
  }
}

using f32x2 = float __attribute__((ext_vector_type(2)));

// v3: lane-contiguous pair mapping.  In step `it` lane l handles pair t + 64*it + l, so one gather
// wave-instruction covers 64 CONSECUTIVE pairs (about one voxel row: a dozen short runs of consecutive range
// gates) instead of every 4th pair of 256 -- roughly 3x fewer cache lines per gather, which is what the
// texture-address unit is billed for.  The CSR itself is streamed with dword loads (256 contiguous bytes per
// wave-instruction, consecutive instructions consecutive), so no alignment or padding contract is needed.
template <typename IndT, int NF, int STRIDE, int TILE, int XCD>
__global__ __launch_bounds__(rg::kBlock) void csr_apply_kernel_v3(
    const IndT* __restrict__ indptr, const int32_t* __restrict__ gidx, const float* __restrict__ wts,
    long n_vox, long n_pairs, const float* __restrict__ packed, unsigned last_gate, float fill,
    float* __restrict__ out) {
  static_assert(TILE % 64 == 0, "a wave handles 64 pairs per step");
  constexpr int IT = TILE / 64;
  __shared__ f32x2 tile_all[rg::kBlock / rg::kWave][TILE * NF];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  f32x2* tile = tile_all[wv];

  const unsigned blk = place_block<XCD>(blockIdx.x, gridDim.x);
  const long r0 = ((long)blk * (rg::kBlock / rg::kWave) + wv) * 64;
  if (r0 >= n_vox) return;  // wave-uniform
  const long row = r0 + lane;
  const long seg_b = (long)indptr[r0];
  const long seg_e = (long)indptr[r0 + 64 < n_vox ? r0 + 64 : n_vox];
  const int span = (int)(seg_e - seg_b);             // a 64-row chunk never holds 2^31 pairs
  const int rs_o = (int)((long)indptr[row < n_vox ? row : n_vox] - seg_b);
  const int re_o = (int)((long)indptr[row + 1 < n_vox ? row + 1 : n_vox] - seg_b);
  const int q = lane & 3;
  int qs[4], qe[4], ps[4], pe[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    qs[p] = __shfl(rs_o, 16 * p + (lane >> 2), 64);
    qe[p] = __shfl(re_o, 16 * p + (lane >> 2), 64);
    ps[p] = __builtin_amdgcn_readlane(rs_o, 16 * p);
    pe[p] = __builtin_amdgcn_readlane(re_o, 16 * p + 15);
  }
  double acc_p[4][NF], acc_w[4][NF];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int f = 0; f < NF; ++f) { acc_p[p][f] = 0.0; acc_w[p][f] = 0.0; }

  if (span > 0) {
    const long last = n_pairs - 1;
    int ci[IT];
    float cw[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      long j = seg_b + it * 64 + lane;
      j = j < last ? j : last;                        // clamped slots are computed but never read back
      ci[it] = gidx[j];
      cw[it] = wts[j];
    }
    for (int t = 0; t < span; t += TILE) {
      // ---- stream phase -----------------------------------------------------------------------------
      float val[IT][STRIDE];
      float w_cur[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        w_cur[it] = cw[it];
        load_packed<STRIDE>(packed, min((unsigned)ci[it], last_gate), val[it]);   // clamp: never fault
      }
      if (t + TILE < span) {  // prefetch the next tile's CSR; in flight during the row phase
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          long j = seg_b + t + TILE + it * 64 + lane;
          j = j < last ? j : last;
          ci[it] = gidx[j];
          cw[it] = wts[j];
        }
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool ok = rg::f32_bits(val[it][f]) != RG_EXCLUDED_BITS;
          f32x2 e;
          e.x = ok ? w_cur[it] * val[it][f] : 0.0f;
          e.y = ok ? w_cur[it] : 0.0f;
          tile[(it * 64 + lane) * NF + f] = e;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- row phase: 4 lanes per row, 16 rows per pass ------------------------------------------------
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (ps[p] < t + TILE && pe[p] > t) {  // wave-uniform
          const int a = (qs[p] > t ? qs[p] : t) - t;
          const int b = (qe[p] < t + TILE ? qe[p] : t + TILE) - t;
          f32x2 part0[NF], part1[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) { part0[f] = (f32x2)(0.0f); part1[f] = (f32x2)(0.0f); }
          int j = a + q;
          for (; j + 4 < b; j += 8) {         // two elements per trip (one ds_read2_b64), two partial sums
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              part0[f] += tile[j * NF + f];
              part1[f] += tile[(j + 4) * NF + f];
            }
          }
          if (j < b) {
#pragma unroll
            for (int f = 0; f < NF; ++f) part0[f] += tile[j * NF + f];
          }
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const f32x2 s = part0[f] + part1[f];
            acc_p[p][f] += (double)s.x;
            acc_w[p][f] += (double)s.y;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }

#pragma unroll
  for (int f = 0; f < NF; ++f) {
    float res = fill;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double sp = acc_p[p][f], sw = acc_w[p][f];
      sp += __shfl_xor(sp, 1, 64); sw += __shfl_xor(sw, 1, 64);
      sp += __shfl_xor(sp, 2, 64); sw += __shfl_xor(sw, 2, 64);
      const float r = sw > 0.0 ? (float)(sp / sw) : fill;
      const float moved = __shfl(r, 4 * (lane & 15), 64);    // row 16p + k lives in lane 4k
      if ((lane >> 4) == p) res = moved;
    }
    if (row < n_vox) out[(size_t)f * n_vox + row] = res;
  }
}

template <typename IndT, int NF, int STRIDE, int TILE, int XCD>
int launch_v3(const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs, const float* packed,
              long n_gates, float fill, float* out, hipStream_t s) {
  const long chunks = (n_vox + 63) / 64;
  const long blocks = (chunks + 3) / 4;
  hipLaunchKernelGGL((csr_apply_kernel_v3<IndT, NF, STRIDE, TILE, XCD>), dim3((unsigned)blocks), dim3(rg::kBlock), 0, s,
                     static_cast<const IndT*>(indptr), gidx, wts, n_vox, n_pairs, packed, (unsigned)(n_gates - 1),
                     fill, out);
  return rg::check_launch("rg_csr_apply_f32");
}

template <typename IndT, int NF, int STRIDE, int TILE, bool NT>
int launch(const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs, const float* packed,
           long n_gates, float fill, float* out, hipStream_t s) {
  const long chunks = (n_vox + 63) / 64;
  const long blocks = (chunks + 3) / 4;
  hipLaunchKernelGGL((csr_apply_kernel<IndT, NF, STRIDE, TILE, NT>), dim3((unsigned)blocks), dim3(rg::kBlock), 0, s,
                     static_cast<const IndT*>(indptr), gidx, wts, n_vox, n_pairs, packed, (unsigned)(n_gates - 1),
                     fill, out);
  return rg::check_launch("rg_csr_apply_f32");
}

template <typename IndT>
int dispatch(int nf, int variant, const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs,
             const float* packed, long n_gates, float fill, float* out, hipStream_t s) {
#define RG_K1(NF_, ST_, TILE_, NT_) \
  launch<IndT, NF_, ST_, TILE_, NT_>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s)
#define RG_K3(NF_, ST_, TILE_, XCD_) \
  launch_v3<IndT, NF_, ST_, TILE_, XCD_>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s)
  if (nf == 1) {  // tuning variants exist for the single-field kernel only
    switch (variant) {
      case 1: return RG_K3(1, 1, 512, kXcdNone);
      case 2: return RG_K3(1, 1, 512, kXcdSlab);
      case 3: return RG_K1(1, 1, 512, true);        // v2: quad mapping, slab placement
      case 4: return RG_K3(1, 1, 256, kXcdNone);
      case 5: return RG_K3(1, 1, 1024, kXcdNone);
      default: return RG_K3(1, 1, 512, kXcdGroup);
    }
  }
  switch (nf) {
    case 2: return RG_K3(2, 2, 512, kXcdGroup);
    case 3: return RG_K3(3, 4, 256, kXcdGroup);
    case 4: return RG_K3(4, 4, 256, kXcdGroup);
    case 5: return RG_K3(5, 8, 256, kXcdGroup);
    case 6: return RG_K3(6, 8, 256, kXcdGroup);
    case 7: return RG_K3(7, 8, 256, kXcdGroup);
    default: return RG_K3(8, 8, 256, kXcdGroup);
  }
#undef RG_K3
#undef RG_K1
}

inline int stride_for(int nf) { return nf == 1 ? 1 : nf == 2 ? 2 : nf <= 4 ? 4 : 8; }

}  // namespace

extern "C" int rg_csr_apply_f32_ex(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx,
                                   const float* weights, int64_t n_vox, int64_t n_pairs, const float* packed,
                                   int32_t n_fields, int32_t stride, int64_t n_gates, float fill_value, float* out,
                                   int32_t variant, rg_stream_t stream) {
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
    return dispatch<int64_t>(n_fields, variant, indptr, gate_idx, weights, n_vox, n_pairs, packed, n_gates, fill_value,
                             out, s);
  return dispatch<int32_t>(n_fields, variant, indptr, gate_idx, weights, n_vox, n_pairs, packed, n_gates, fill_value, out,
                           s);
}

extern "C" int rg_csr_apply_f32(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx,
                                const float* weights, int64_t n_vox, int64_t n_pairs, const float* packed,
                                int32_t n_fields, int32_t stride, int64_t n_gates, float fill_value, float* out,
                                rg_stream_t stream) {
  return rg_csr_apply_f32_ex(indptr, indptr_is_i64, gate_idx, weights, n_vox, n_pairs, packed, n_fields, stride, n_gates,
                             fill_value, out, 0, stream);
}
