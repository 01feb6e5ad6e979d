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
//   row phase(t)   LDS is the transposition buffer from pair order to row order: the rows that touch the tile share
//                  the 64 lanes dynamically, per-row sums live in LDS (see the comment at the kernel);
//   epilogue       combine, divide in float64, coalesced store of out[f][row].
//
// Empty rows cost nothing, long rows only lengthen their own lanes' loop; no workgroup barrier, no atomics, no
// inter-wave communication, so results are bit-reproducible run to run.
#include "rg_common.hpp"

// 8- and 16-byte buffer loads by intrinsic name: this compiler's __builtin_amdgcn_raw_buffer_load_b64 / _b128 return
// the first dword in every element (seen in the generated code), the intrinsics themselves are fine.  Declared at
// namespace scope: a name bound to an intrinsic must not have internal linkage.
using rg_f32x2 = float __attribute__((ext_vector_type(2)));
using rg_f32x4 = float __attribute__((ext_vector_type(4)));
__device__ rg_f32x2 rg_buffer_load_v2f32(__amdgpu_buffer_rsrc_t, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v2f32");
__device__ rg_f32x4 rg_buffer_load_v4f32(__amdgpu_buffer_rsrc_t, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v4f32");

namespace {

using rg::load_packed;

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
constexpr int kWpb1 = 256, kWpb2 = 512, kWpb8 = 768;  // dyn kernel only: waves per workgroup (default 4)
constexpr int kStages3 = 1024;   // dyn kernel only: three CSR tiles in flight per wavefront instead of two
constexpr int wpb_of(int flags) { return (flags & 768) == 256 ? 1 : (flags & 768) == 512 ? 2 : (flags & 768) == 768 ? 8 : 4; }

using f32x2 = float __attribute__((ext_vector_type(2)));

// The kernel: the phases above, with
//   * a DYNAMIC row phase: per tile the rows that actually touch it (a contiguous range, found with one ballot) share
//     the 64 lanes -- L = the largest power of two <= 64 / rows (at least `stride`) lanes per row, split into stride
//     field slots x L/stride interleaved element streams -- and the per-row sums live in a small LDS array instead of
//     per-pass registers, so lane utilisation does not depend on how many rows a tile holds and neither registers nor
//     code grow with the field count;
//   * BUFFER loads: the CSR tile and the packed fields are read through buffer resources, whose hardware range check
//     returns 0 for anything past the end of the array.  That removes every address clamp and all 64-bit address
//     arithmetic from the loop: a tile's resource is rebuilt in SGPRs (base + t, remaining bytes), the lane offset
//     is the constant 4 * lane and the 8 loads of a tile differ only in the instruction's immediate offset;
//   * a software pipeline unrolled by two (see the comment inside).
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int kRsrcRaw32 = 0x00020000;   // gfx9 buffer resource word 3: DATA_FORMAT = 32, untyped dword access

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, long bytes) {   // `base` and `bytes` wave-uniform
  const unsigned nb = bytes >= 0xFFFFFFFFL ? 0xFFFFFFFFu : bytes <= 0 ? 0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)nb, kRsrcRaw32);
}


template <int STRIDE>
__device__ __forceinline__ void buffer_load_packed(rsrc_t r, unsigned gate, float (&v)[STRIDE]) {
  const int off = (int)(gate * (4u * STRIDE));   // a corrupt index lands out of range -> the load returns 0
  if constexpr (STRIDE == 1) {
    v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  } else if constexpr (STRIDE == 2) {
    const rg_f32x2 t = rg_buffer_load_v2f32(r, off, 0, 0);
    v[0] = t.x; v[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < STRIDE; q += 4) {
      const rg_f32x4 t = rg_buffer_load_v4f32(r, off + 4 * q, 0, 0);
      v[q] = t.x; v[q + 1] = t.y; v[q + 2] = t.z; v[q + 3] = t.w;
    }
  }
}

template <typename IndT, int NF, int STRIDE, int TILE, int XCD, int FLAGS>
__global__ __launch_bounds__(64 * wpb_of(FLAGS)) void csr_apply_dyn_kernel(
    const IndT* __restrict__ indptr, const int32_t* __restrict__ gidx, const float* __restrict__ wts,
    long n_vox, long n_pairs, const float* __restrict__ packed, unsigned last_gate, float fill,
    float* __restrict__ out) {
  static_assert(TILE % 64 == 0, "a wave handles 64 pairs per step");
  constexpr int IT = TILE / 64;
  static_assert(IT * 256 <= 4096, "the tile's loads are told apart by a 12-bit immediate offset");
  constexpr int kLgStride = STRIDE == 1 ? 0 : STRIDE == 2 ? 1 : STRIDE == 4 ? 2 : 3;
  constexpr int WPB = wpb_of(FLAGS);
  constexpr int NST = (FLAGS & kStages3) ? 3 : 2;   // tiles of CSR in flight per wavefront
  __shared__ f32x2 tile_all[WPB][TILE * STRIDE];
  __shared__ f32x2 rowacc_all[WPB][64 * STRIDE];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x2* tile = tile_all[wv];
  f32x2* rowacc = rowacc_all[wv];

  const unsigned blk = place_block<XCD>(blockIdx.x, gridDim.x);
  const long r0 = ((long)blk * WPB + wv) * 64;
  if (r0 >= n_vox) return;  // wave-uniform
  const long row = r0 + lane;
  const long seg_b = (long)indptr[r0];
  const long seg_e = (long)indptr[r0 + 64 < n_vox ? r0 + 64 : n_vox];
  const int span = (int)(seg_e - seg_b);
  const int rs_o = (int)((long)indptr[row < n_vox ? row : n_vox] - seg_b);
  const int re_o = (int)((long)indptr[row + 1 < n_vox ? row + 1 : n_vox] - seg_b);
#pragma unroll
  for (int f = 0; f < STRIDE; ++f) rowacc[lane * STRIDE + f] = (f32x2)(0.0f);

  if (span > 0) {
    // Software pipeline over the tiles.  Stage k mod NST holds tile k's indices and weights, value set k mod NST
    // its gathered field values; one step of tile k issues the gather of tile k+1 (its indices were streamed NST-1
    // steps ago), turns tile k into products, streams tile k+NST into the stage it just freed and runs tile k's
    // row phase while all of that is in flight.  Two rules keep the compiler's wait counts exact:
    //  * the loop is unrolled by NST, so every load lands in the register it is consumed from -- a rotating
    //    register set would be copied, and a copy has to wait for the load it copies;
    //  * every step issues the same loads, unconditionally.  The buffer resources end at the chunk's last pair, so
    //    loads past it are dropped by the range check (no memory traffic, they return 0) instead of being branched
    //    around -- a branch would make the number of loads in flight path-dependent and force full waits.
    struct Stage {
      int ci[IT];
      float cw[IT];
    };
    struct Values {
      float v[IT][STRIDE];
    };
    Stage st[NST];
    Values vl[NST];
    const int32_t* __restrict__ gi = gidx + seg_b;
    const float* __restrict__ wi = wts + seg_b;
    const int lane4 = lane * 4;
    const rsrc_t rp = make_rsrc(packed, ((long)last_gate + 1) * (4 * STRIDE));

    auto stream = [&](Stage& sg, int t) {   // t wave-uniform: the resources live in SGPRs
      const rsrc_t ri = make_rsrc(gi + t, ((long)span - t) * 4);
      const rsrc_t rw = make_rsrc(wi + t, ((long)span - t) * 4);
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        sg.ci[it] = __builtin_amdgcn_raw_buffer_load_b32(ri, lane4 + it * 256, 0, 0);
        sg.cw[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, lane4 + it * 256, 0, 0));
      }
    };
    auto gather = [&](const Stage& sg, Values& val) {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        if constexpr ((FLAGS & kNoGather) != 0) {   // timing-only ablation
#pragma unroll
          for (int f = 0; f < STRIDE; ++f) val.v[it][f] = __builtin_bit_cast(float, sg.ci[it]);
        } else {
          buffer_load_packed<STRIDE>(rp, (unsigned)sg.ci[it], val.v[it]);
        }
      }
    };
    auto step = [&](int t, Stage& cur, const Stage& nxt, const Values& val, Values& val_nxt) {
      gather(nxt, val_nxt);
      // ---- products of tile t -> LDS ---------------------------------------------------------------
#pragma unroll
      for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int f = 0; f < STRIDE; ++f) {  // padding slots hold the sentinel -> (0, 0)
          const bool ok = f < NF && rg::f32_bits(val.v[it][f]) != RG_EXCLUDED_BITS;
          f32x2 e;
          e.x = ok ? cur.cw[it] * val.v[it][f] : 0.0f;
          e.y = ok ? cur.cw[it] : 0.0f;
          tile[(it * 64 + lane) * STRIDE + f] = e;
        }
      }
      stream(cur, t + NST * TILE);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- dynamic row phase ---------------------------------------------------------------------------
      const unsigned long long act = __ballot(re_o > rs_o && re_o > t && rs_o < t + TILE);
      if (act != 0) {  // wave-uniform
        const int ra = __builtin_ctzll(act), rb = 63 - __builtin_clzll(act);
        const int nact = rb - ra + 1;
        int lg = 31 - __builtin_clz(64 / nact);          // lanes per row = 2^lg <= 64 / rows
        lg = lg < kLgStride ? kLgStride : lg;            // ... but at least one lane per field slot
        const int rpr = 64 >> lg;                        // rows per round
        const int rin = lane & ((1 << lg) - 1);
        const int fslot = rin & (STRIDE - 1);
        const int sub = rin >> kLgStride, nsub = 1 << (lg - kLgStride);
        for (int rbase = ra; rbase <= rb; rbase += rpr) {
          const int myrow = rbase + (lane >> lg);
          const bool live = myrow <= rb;
          const int qs = __shfl(rs_o, myrow & 63, 64);
          const int qe = __shfl(re_o, myrow & 63, 64);
          const int a = (qs > t ? qs : t) - t;
          const int b = live ? (qe < t + TILE ? qe : t + TILE) - t : a;
          f32x2 part0 = (f32x2)(0.0f), part1 = (f32x2)(0.0f);
          int j = a + sub;
          for (; j + nsub < b; j += 2 * nsub) {  // two elements per trip, two independent partial sums
            part0 += tile[j * STRIDE + fslot];
            part1 += tile[(j + nsub) * STRIDE + fslot];
          }
          if (j < b) part0 += tile[j * STRIDE + fslot];
          f32x2 sum = part0 + part1;
          for (int m = STRIDE; m < (1 << lg); m <<= 1) {  // fold the interleaved streams of one field slot
            sum.x += __shfl_xor(sum.x, m, 64);
            sum.y += __shfl_xor(sum.y, m, 64);
          }
          if (live && sub == 0) rowacc[myrow * STRIDE + fslot] += sum;  // one owner per (row, slot): plain RMW
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    stream(st[0], 0);
    gather(st[0], vl[0]);
#pragma unroll
    for (int u = 1; u < NST; ++u) stream(st[u], u * TILE);
    for (int t = 0; t < span;) {
#pragma unroll
      for (int u = 0; u < NST; ++u) {
        step(t, st[u], st[(u + 1) % NST], vl[u], vl[(u + 1) % NST]);
        t += TILE;
        if (t >= span) break;
      }
    }
  }

  // ---- epilogue: lane == row again; one coalesced 256-byte store per field ------------------------------
  if (row < n_vox) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const f32x2 s = rowacc[lane * STRIDE + f];
      out[(size_t)f * n_vox + row] = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
    }
  }
}

template <typename IndT, int NF, int STRIDE, int TILE, int XCD, int FLAGS>
int launch_dyn(const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs, const float* packed,
               long n_gates, float fill, float* out, hipStream_t s) {
  constexpr int WPB = wpb_of(FLAGS);
  const long chunks = (n_vox + 63) / 64;
  const long blocks = (chunks + WPB - 1) / WPB;
  hipLaunchKernelGGL((csr_apply_dyn_kernel<IndT, NF, STRIDE, TILE, XCD, FLAGS>), dim3((unsigned)blocks), dim3(64 * WPB),
                     0, s, static_cast<const IndT*>(indptr), gidx, wts, n_vox, n_pairs, packed, (unsigned)(n_gates - 1),
                     fill, out);
  return rg::check_launch("rg_csr_apply_f32");
}

template <typename IndT>
int dispatch(int nf, int variant, const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long n_pairs,
             const float* packed, long n_gates, float fill, float* out, hipStream_t s) {
#define RG_KD(NF_, ST_, TILE_, XCD_, FLAGS_) \
  launch_dyn<IndT, NF_, ST_, TILE_, XCD_, FLAGS_>(indptr, gidx, wts, n_vox, n_pairs, packed, n_gates, fill, out, s)
  if (nf == 1) {  // tuning variants (tools/tune_k1.py) exist for the single-field kernel only; 0 = what ships
    switch (variant) {
      case 8: return RG_KD(1, 1, 512, kXcdSlab, 0);       // XCD placement
      case 16: return RG_KD(1, 1, 512, kXcdGroup, 0);
      case 18: return RG_KD(1, 1, 256, kXcdNone, 0);      // tile size
      case 23: return RG_KD(1, 1, 448, kXcdNone, 0);
      case 9: return RG_KD(1, 1, 512, kXcdNone, 0);
      case 17: return RG_KD(1, 1, 640, kXcdNone, 0);
      case 19: return RG_KD(1, 1, 512, kXcdNone, kWpb1);  // waves per workgroup
      case 20: return RG_KD(1, 1, 512, kXcdNone, kWpb2);
      case 21: return RG_KD(1, 1, 512, kXcdNone, kWpb8);
      case 22: return RG_KD(1, 1, 512, kXcdNone, kStages3);   // three CSR tiles in flight
      case 24: return RG_KD(1, 1, 384, kXcdNone, kStages3);
      case 25: return RG_KD(1, 1, 256, kXcdNone, kStages3);
      case 28: return RG_KD(1, 1, 512, kXcdNone, kNoGather);  // timing-only ablation: no field gather
      default: return RG_KD(1, 1, 384, kXcdNone, 0);   // 384-pair tiles: same speed as 512 or slightly better, 72 VGPRs
    }
  }
  switch (nf) {
    case 2: return RG_KD(2, 2, 512, kXcdNone, 0);
    case 3: return RG_KD(3, 4, 256, kXcdNone, 0);
    case 4: return RG_KD(4, 4, 256, kXcdNone, 0);
    case 5: return RG_KD(5, 8, 128, kXcdNone, 0);
    case 6: return RG_KD(6, 8, 128, kXcdNone, 0);
    case 7: return RG_KD(7, 8, 128, kXcdNone, 0);
    default: return RG_KD(8, 8, 128, kXcdNone, 0);
  }
#undef RG_KD
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
  RG_REQUIRE(n_gates * stride * 4 <= 0xFFFFFFFFL, RG_EUNSUPPORTED,
             "rg_csr_apply_f32: the packed fields (%ld gates x %d slots) exceed the 4 GiB one buffer resource addresses",
             (long)n_gates, stride);
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
