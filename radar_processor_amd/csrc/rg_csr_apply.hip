// K1  csr_apply -- the per-volume hot loop (replaces radar_grid/interpolate.py:69-104 and the per-field
// loop of :137-140).
//
// Roofline: HBM.  Algorithmic bytes per launch = 8*P (gate_idx + weight per pair) + sizeof(indptr)*(V+1)
//           + F*(5*G + 4*V)   (SURVEY.md §8(d)); the field gather is served by L1 / L2 / Infinity Cache.
//
// Mapping: "CSR-stream" at wavefront granularity.  One wavefront = 64 consecutive voxel rows = one contiguous
// pair range, walked in tiles of TILE pairs with a two-deep software pipeline:
//
//   products(t)    lane l handles pair t + 64*it + l (lane-contiguous): masked float32 products
//                  (w*v, w) of tile t go to the wave's private LDS tile -- its gathered values were requested
//                  one iteration earlier;
//   gather(t+1)    one load per pair from the packed field array (mask folded into the value): 64 CONSECUTIVE
//                  pairs per wave-instruction = about one voxel row = a dozen short runs of consecutive range
//                  gates, i.e. few cache lines per instruction;
//   stream(t+2)    coalesced dword loads of gate_idx and weights (256 contiguous bytes per wave-instruction,
//                  consecutive instructions consecutive) -- in flight for a whole iteration;
//   row phase(t)   LDS is the transposition buffer from pair order to row order.  4 lanes share a row: in
//                  pass p (p = 0..3) lane l works for row 16p + l/4 and sums every 4th element of that row's
//                  slice of the tile (float32 partials of <= TILE/4 terms, folded into per-row accumulators
//                  once per tile); passes whose 16 rows do not touch the tile are skipped wave-uniformly;
//   epilogue       quad-reduce the 4 sub-lane accumulators in float64, move row r0+l's result to lane l, one
//                  coalesced 256-byte store per field.
//
// Empty rows cost nothing, long rows only lengthen their own quad's loop; no workgroup barrier, no atomics, no
// inter-wave communication, so results are bit-reproducible run to run.
#include <type_traits>

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

// XCD placement of the 64-row chunks (speed only, never correctness):
//   kXcdNone   workgroup b -> logical block b: neighbouring blocks land on different XCDs (round-robin deal),
//              perfectly balanced when pair density varies with height, every L2 sees the same gate window;
//   kXcdGroup  groups of 32 consecutive logical blocks stay on one XCD, groups rotate over the XCDs;
//   kXcdSlab   one contiguous eighth of the grid per XCD (best L2 locality, worst balance: top levels are
//              sparse, so the XCDs that own them idle -- measured 30 % slower on the bench grid).
constexpr int kXcdNone = 0, kXcdGroup = 1, kXcdSlab = 2;

template <int MODE>
__device__ __forceinline__ unsigned place_block(unsigned bid, unsigned nblk) {
  if constexpr (MODE == kXcdSlab) {
    return rg::xcd_remap(bid, nblk);
  } else if constexpr (MODE == kXcdGroup) {
    constexpr unsigned S = 32;
    const unsigned super = S * rg::kNumXcd;
    const unsigned full = nblk / super * super;  // only whole super-groups are permuted (bijective)
    if (bid >= full) return bid;
    const unsigned base = bid / super * super, r = bid % super;
    return base + (r % rg::kNumXcd) * S + r / rg::kNumXcd;
  } else {
    return bid;
  }
}

// tuning / diagnostic flags (template parameter FLAGS)
constexpr int kNoGather = 1;     // timing-only ablation: skip the gather (results are wrong by construction)
constexpr int kNoRows = 2;       // timing-only ablation: skip the row phase
constexpr int kNonTemporal = 4;  // stream the CSR with the nt cache policy
constexpr int kAcc32 = 8;        // per-row accumulators in float32 instead of float64 (fewer VGPRs)
constexpr int kFlatOrder = 16;   // gather at the top of the iteration (no gather-ahead pipelining)

using f32x2 = float __attribute__((ext_vector_type(2)));

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

template <typename IndT, int NF, int STRIDE, int TILE, int XCD, int FLAGS>
__global__ __launch_bounds__(rg::kBlock) void csr_apply_kernel(
    const IndT* __restrict__ indptr, const int32_t* __restrict__ gidx, const float* __restrict__ wts,
    long n_vox, long n_pairs, const float* __restrict__ packed, unsigned last_gate, float fill,
    float* __restrict__ out) {
  static_assert(TILE % 64 == 0, "a wave handles 64 pairs per step");
  constexpr int IT = TILE / 64;
  constexpr bool NT = (FLAGS & kNonTemporal) != 0;
  using acc_t = typename std::conditional<(FLAGS & kAcc32) != 0, float, double>::type;
  __shared__ f32x2 tile_all[rg::kBlock / rg::kWave][TILE * NF];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: chunk bounds live in SGPRs
  f32x2* tile = tile_all[wv];

  const unsigned blk = place_block<XCD>(blockIdx.x, gridDim.x);
  const long r0 = ((long)blk * (rg::kBlock / rg::kWave) + wv) * 64;
  if (r0 >= n_vox) return;  // wave-uniform
  const long row = r0 + lane;
  const long seg_b = (long)indptr[r0];
  const long seg_e = (long)indptr[r0 + 64 < n_vox ? r0 + 64 : n_vox];
  const int span = (int)(seg_e - seg_b);  // a 64-row chunk never holds 2^31 pairs
  // row bounds as offsets into the chunk's pair range; lane l <-> row r0 + l
  const int rs_o = (int)((long)indptr[row < n_vox ? row : n_vox] - seg_b);
  const int re_o = (int)((long)indptr[row + 1 < n_vox ? row + 1 : n_vox] - seg_b);
  // quad layout of the row phase: in pass p lane l serves row 16p + (l >> 2), sub-lane q = l & 3
  const int q = lane & 3;
  int qs[4], qe[4], ps[4], pe[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    qs[p] = __shfl(rs_o, 16 * p + (lane >> 2), 64);
    qe[p] = __shfl(re_o, 16 * p + (lane >> 2), 64);
    ps[p] = __builtin_amdgcn_readlane(rs_o, 16 * p);  // pair span of the pass's 16 rows (wave-uniform)
    pe[p] = __builtin_amdgcn_readlane(re_o, 16 * p + 15);
  }
  acc_t acc_p[4][NF], acc_w[4][NF];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int f = 0; f < NF; ++f) { acc_p[p][f] = 0; acc_w[p][f] = 0; }

  if (span > 0) {
    // Wave-uniform bases + 32-bit lane offsets: one VGPR per load address.  Slots past the end of the arrays
    // are clamped (computed but never read back by the row phase).
    const int32_t* __restrict__ gi = gidx + seg_b;
    const float* __restrict__ wi = wts + seg_b;
    const long tail = n_pairs - 1 - seg_b;
    const int kmax = tail < 0x3FFFFFFF ? (int)tail : 0x3FFFFFFF;
    int ci[IT];
    float cw[IT];
    float w_n[IT];
    float val_n[IT][STRIDE];

    auto stream = [&](int t) {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int k = min(t + it * 64 + lane, kmax);
        ci[it] = stream_load<NT>(gi + k);
        cw[it] = stream_load<NT>(wi + k);
      }
    };
    auto gather = [&]() {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        w_n[it] = cw[it];
        if constexpr ((FLAGS & kNoGather) != 0) {
#pragma unroll
          for (int f = 0; f < STRIDE; ++f) val_n[it][f] = rg::bits_f32((unsigned)ci[it] & 0x3FFFFFFFu);
        } else {
          load_packed<STRIDE>(packed, min((unsigned)ci[it], last_gate), val_n[it]);  // clamp: never fault
        }
      }
    };

    stream(0);
    if constexpr ((FLAGS & kFlatOrder) == 0) {
      gather();
      if (TILE < span) stream(TILE);
    }
    for (int t = 0; t < span; t += TILE) {
      if constexpr ((FLAGS & kFlatOrder) != 0) {
        gather();
        if (t + TILE < span) stream(t + TILE);
      }
      // ---- products of tile t -> LDS ---------------------------------------------------------------
#pragma unroll
      for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool ok = rg::f32_bits(val_n[it][f]) != RG_EXCLUDED_BITS;
          f32x2 e;
          e.x = ok ? w_n[it] * val_n[it][f] : 0.0f;
          e.y = ok ? w_n[it] : 0.0f;
          tile[(it * 64 + lane) * NF + f] = e;
        }
      }
      if constexpr ((FLAGS & kFlatOrder) == 0) {
        // ---- gather for tile t+1, CSR stream for tile t+2: both in flight during the row phase ---------
        if (t + TILE < span) {
          gather();
          if (t + 2 * TILE < span) stream(t + 2 * TILE);
        }
      }
      // LDS traffic of one wave executes in order; the fences only pin the compiler.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- row phase: 4 lanes per row, 16 rows per pass --------------------------------------------
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (((FLAGS & kNoRows) != 0) ? (p == 0 && t == 0) : (ps[p] < t + TILE && pe[p] > t)) {  // wave-uniform
          const int a = (qs[p] > t ? qs[p] : t) - t;
          const int b = (qe[p] < t + TILE ? qe[p] : t + TILE) - t;
          f32x2 part0[NF], part1[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) { part0[f] = (f32x2)(0.0f); part1[f] = (f32x2)(0.0f); }
          int j = a + q;
          for (; j + 4 < b; j += 8) {  // two elements per trip, two independent partial sums
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
            acc_p[p][f] += (acc_t)s.x;
            acc_w[p][f] += (acc_t)s.y;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }

  // ---- epilogue: quad reduce in float64, transpose back to lane == row, coalesced store ----------------
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    float res = fill;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double sp = (double)acc_p[p][f], sw = (double)acc_w[p][f];
      sp += __shfl_xor(sp, 1, 64); sw += __shfl_xor(sw, 1, 64);
      sp += __shfl_xor(sp, 2, 64); sw += __shfl_xor(sw, 2, 64);
      const float r = sw > 0.0 ? (float)(sp / sw) : fill;
      const float moved = __shfl(r, 4 * (lane & 15), 64);  // row 16p + k lives in lane 4k
      if ((lane >> 4) == p) res = moved;
    }
    if (row < n_vox) out[(size_t)f * n_vox + row] = res;
  }
}

template <typename IndT, int NF, int STRIDE, int TILE, int XCD, int FLAGS>
int launch(const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs, const float* packed,
           long n_gates, float fill, float* out, hipStream_t s) {
  const long chunks = (n_vox + 63) / 64;
  const long blocks = (chunks + 3) / 4;
  hipLaunchKernelGGL((csr_apply_kernel<IndT, NF, STRIDE, TILE, XCD, FLAGS>), dim3((unsigned)blocks), dim3(rg::kBlock), 0,
                     s, static_cast<const IndT*>(indptr), gidx, wts, n_vox, n_pairs, packed, (unsigned)(n_gates - 1),
                     fill, out);
  return rg::check_launch("rg_csr_apply_f32");
}

template <typename IndT>
int dispatch(int nf, int variant, const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs,
             const float* packed, long n_gates, float fill, float* out, hipStream_t s) {
#define RG_K1(NF_, ST_, TILE_, XCD_, FLAGS_) \
  launch<IndT, NF_, ST_, TILE_, XCD_, FLAGS_>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s)
  if (nf == 1) {  // tuning variants (tools/tune_k1.py) exist for the single-field kernel only
    switch (variant) {
      case 1: return RG_K1(1, 1, 512, kXcdNone, kAcc32);
      case 2: return RG_K1(1, 1, 512, kXcdGroup, kAcc32);
      case 3: return RG_K1(1, 1, 512, kXcdGroup, 0);
      case 4: return RG_K1(1, 1, 256, kXcdNone, kAcc32);
      case 5: return RG_K1(1, 1, 512, kXcdNone, kFlatOrder);   // no gather-ahead pipelining
      case 6: return RG_K1(1, 1, 512, kXcdNone, kNonTemporal);
      case 7: return RG_K1(1, 1, 384, kXcdNone, 0);
      case 8: return RG_K1(1, 1, 512, kXcdSlab, 0);
      case 11: return RG_K1(1, 1, 512, kXcdNone, kNoGather);   // timing-only ablations
      case 12: return RG_K1(1, 1, 512, kXcdNone, kNoRows);
      case 13: return RG_K1(1, 1, 512, kXcdNone, kNoGather | kNoRows);
      default: return RG_K1(1, 1, 512, kXcdNone, 0);
    }
  }
  switch (nf) {
    case 2: return RG_K1(2, 2, 512, kXcdNone, 0);
    case 3: return RG_K1(3, 4, 256, kXcdNone, 0);
    case 4: return RG_K1(4, 4, 256, kXcdNone, 0);
    case 5: return RG_K1(5, 8, 256, kXcdNone, 0);
    case 6: return RG_K1(6, 8, 256, kXcdNone, 0);
    case 7: return RG_K1(7, 8, 256, kXcdNone, 0);
    default: return RG_K1(8, 8, 256, kXcdNone, 0);
  }
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
  RG_REQUIRE(rg::aligned16(packed), RG_EALIGN, "rg_csr_apply_f32: packed must be 16-byte aligned");
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
