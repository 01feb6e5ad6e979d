// K1  csr_apply -- the per-volume hot loop (replaces radar_grid/interpolate.py:69-104 and the per-field
// loop of :137-140).
//
// Roofline: HBM.  Algorithmic bytes per launch = 8*P (gate_idx + weight per pair) + sizeof(indptr)*(V+1)
//           + F*(5*G + 4*V)   (SURVEY.md §8(d)); the field gather is served by L1 / L2 / Infinity Cache.
//
// Mapping: "CSR-stream" at wavefront granularity.  One wavefront = one SEGMENT = up to 64 consecutive voxel rows of one
// grid line (a line = `line_len` consecutive rows = one (z, y) row of nx voxels; segments never straddle lines, so the
// compact kernel of rg_csr_compact.hip can stack the segments of neighbouring lines into 2-D patches and still add
// exactly the same numbers in the same order) = one contiguous pair range, walked in tiles of TILE pairs with a
// two-deep software pipeline:
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
#include "rg_row_phase.hpp"

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
[[maybe_unused]] constexpr int kWpb1 = 256, kWpb2 = 512, kWpb8 = 768;  // dyn kernel only: waves per workgroup (default 4)
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
// Segments a line's wavefronts are rotated by per line (see the kernel).  Measured on the bench grid (32 segments per
// line, 4 wavefronts per workgroup): 0 -> 14.1 ms, 1 -> 12.6, 2 -> 13.3, 3 -> 12.9, 4 -> 14.0, 5 -> 13.0, 8 -> 18.0.
constexpr unsigned kSegmentRotation = 1;

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
    long n_vox, long line_len, unsigned segs_per_line, unsigned n_segs, unsigned rot_step,
    const float* __restrict__ packed, unsigned last_gate, float fill, float* __restrict__ out) {
  static_assert(TILE % 64 == 0, "a wave handles 64 pairs per step");
  constexpr int IT = TILE / 64;
  static_assert(IT * 256 <= 4096, "the tile's loads are told apart by a 12-bit immediate offset");
  constexpr int WPB = wpb_of(FLAGS);
  constexpr int NST = (FLAGS & kStages3) ? 3 : 2;   // tiles of CSR in flight per wavefront
  __shared__ __attribute__((aligned(16))) float tile_all[WPB][TILE * rg::tile_floats(NF, STRIDE)];
  __shared__ f32x2 rowacc_all[WPB][64 * NF];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tile = tile_all[wv];
  f32x2* rowacc = rowacc_all[wv];

  const unsigned blk = place_block<XCD>(blockIdx.x, gridDim.x);
  // segment index: line-major, segs_per_line segments of <= 64 rows per line.  32-bit arithmetic: a 64-bit division
  // costs a few hundred instructions per wavefront on this ISA.
  const unsigned seg0 = blk * WPB + wv;
  if (seg0 >= n_segs) return;  // wave-uniform
  const unsigned line = seg0 / segs_per_line;
  // rotate the segments of a line by rot_step per line: workgroups are dealt to the 8 XCDs round-robin, so with
  // segs_per_line a multiple of 8*WPB every XCD would own a fixed x-slab of the grid for the whole launch -- the slabs'
  // pair counts differ by 2x between the grid's edge and its centre, and the XCDs that finish early idle (speed only)
  const unsigned rot = seg0 - line * segs_per_line + (line * rot_step) % segs_per_line;
  const unsigned sx = rot >= segs_per_line ? rot - segs_per_line : rot;
  // a line's segments are balanced: the first line_len % segs_per_line of them hold one row more than the others
  const unsigned seg_base = (unsigned)(line_len / segs_per_line), seg_extra = (unsigned)(line_len % segs_per_line);
  const long r0 = (long)line * line_len + (long)(sx * seg_base + (sx < seg_extra ? sx : seg_extra));
  const int nrows = (int)(seg_base + (sx < seg_extra ? 1u : 0u));   // <= 64
  const long row = r0 + lane;
  const long seg_b = (long)indptr[r0];
  const long seg_e = (long)indptr[r0 + nrows];
  const int span = (int)(seg_e - seg_b);
  const int rs_o = (int)((long)indptr[r0 + (lane < nrows ? lane : nrows)] - seg_b);
  const int re_o = (int)((long)indptr[r0 + (lane + 1 < nrows ? lane + 1 : nrows)] - seg_b);
#pragma unroll
  for (int f = 0; f < NF; ++f) rowacc[lane * NF + f] = (f32x2)(0.0f);

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
      // ---- products of tile t -> LDS (layout and arithmetic: rg_row_phase.hpp) -----------------------------
#pragma unroll
      for (int it = 0; it < IT; ++it) rg::store_products<NF, STRIDE>(tile, TILE, it * 64 + lane, cur.cw[it], val.v[it]);
      stream(cur, t + NST * TILE);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- dynamic row phase (shared with rg_csr_compact_apply_f32: same lane split, same float32 adds) --------
      rg::row_phase<NF, STRIDE, TILE>(tile, rowacc, t, rs_o, re_o, lane);
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
  if (lane < nrows) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const f32x2 s = rowacc[lane * NF + f];
      out[(size_t)f * n_vox + row] = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
    }
  }
}

template <typename IndT, int NF, int STRIDE, int TILE, int XCD, int FLAGS>
int launch_dyn(const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long line_len, const float* packed,
               long n_gates, float fill, float* out, hipStream_t s) {
  constexpr int WPB = wpb_of(FLAGS);
  const long segs_per_line = (line_len + 63) / 64;
  const long n_segs = (n_vox / line_len) * segs_per_line;
  if (n_segs > 0xFFFFFFF0L) {
    rg::set_error("rg_csr_apply_f32: %ld segments exceed one launch", n_segs);
    return RG_EUNSUPPORTED;
  }
  const long blocks = (n_segs + WPB - 1) / WPB;
  hipLaunchKernelGGL((csr_apply_dyn_kernel<IndT, NF, STRIDE, TILE, XCD, FLAGS>), dim3((unsigned)blocks), dim3(64 * WPB),
                     0, s, static_cast<const IndT*>(indptr), gidx, wts, n_vox, line_len, (unsigned)segs_per_line,
                     (unsigned)n_segs, kSegmentRotation, packed,
                     (unsigned)(n_gates - 1), fill, out);
  return rg::check_launch("rg_csr_apply_f32");
}

template <typename IndT>
int dispatch(int nf, int variant, const void* indptr, const int32_t* gidx, const float* wts, long n_vox, long line_len,
             const float* packed, long n_gates, float fill, float* out, hipStream_t s) {
#define RG_KD(NF_, ST_, TILE_, XCD_, FLAGS_) \
  launch_dyn<IndT, NF_, ST_, TILE_, XCD_, FLAGS_>(indptr, gidx, wts, n_vox, line_len, packed, n_gates, fill, out, s)
  // `variant`: 0 = what ships.  128/256/384/512 select the pipeline tile explicitly (A/B timing, and the bit-identity
  // tests of rg_csr_compact_apply_f32 with a non-default tile); the other codes are single-field tuning variants
  // (tools/tune_k1.py).
  if (nf == 1) {
    switch (variant) {
#ifdef RG_EXPERIMENTS   // tuning variants and the timing-only ablation: experiment builds (tools/build_experiments.py) only
      case 8: return RG_KD(1, 1, 512, kXcdSlab, 0);       // XCD placement
      case 16: return RG_KD(1, 1, 512, kXcdGroup, 0);
      case 18: return RG_KD(1, 1, 256, kXcdNone, 0);
      case 23: return RG_KD(1, 1, 448, kXcdNone, 0);
      case 9: return RG_KD(1, 1, 512, kXcdNone, 0);
      case 17: return RG_KD(1, 1, 640, kXcdNone, 0);
      case 19: return RG_KD(1, 1, 512, kXcdNone, kWpb1);  // waves per workgroup
      case 20: return RG_KD(1, 1, 512, kXcdNone, kWpb2);
      case 21: return RG_KD(1, 1, 512, kXcdNone, kWpb8);
      case 22: return RG_KD(1, 1, 512, kXcdNone, kStages3);   // three CSR tiles in flight
      case 24: return RG_KD(1, 1, 384, kXcdNone, kStages3);
      case 25: return RG_KD(1, 1, 256, kXcdNone, kStages3);
      case 28: return RG_KD(1, 1, 512, kXcdNone, kNoGather);  // timing-only ablation: no field gather (wrong results)
#endif
      case 128: return RG_KD(1, 1, 128, kXcdNone, 0);     // pipeline tile (right answers; another order of the float32 adds)
      case 256: return RG_KD(1, 1, 256, kXcdNone, 0);
      case 512: return RG_KD(1, 1, 512, kXcdNone, 0);
      default: return RG_KD(1, 1, 384, kXcdNone, 0);   // 384-pair tiles: same speed as 512 or slightly better, 72 VGPRs
    }
  }
  switch (nf) {
    case 2:
      switch (variant) {
        case 128: return RG_KD(2, 2, 128, kXcdNone, 0);
        case 256: return RG_KD(2, 2, 256, kXcdNone, 0);
        case 512: return RG_KD(2, 2, 512, kXcdNone, 0);
        default: return RG_KD(2, 2, 384, kXcdNone, 0);
      }
    case 3:
      switch (variant) {
        case 128: return RG_KD(3, 4, 128, kXcdNone, 0);
        case 192: return RG_KD(3, 4, 192, kXcdNone, 0);
        case 256: return RG_KD(3, 4, 256, kXcdNone, 0);
        case 320: return RG_KD(3, 4, 320, kXcdNone, 0);
        // config 2 / bench grid, ms: 256 -> 2.48 / 17.5, 320 -> 2.37 / 16.9, 384 -> 2.39 / 17.1; 384 = 2 x 192 pairs is
        // also what the packed stream of the compact kernel needs (same tile = same bits)
        default: return RG_KD(3, 4, 384, kXcdNone, 0);
      }
    case 4:
      switch (variant) {
        case 128: return RG_KD(4, 4, 128, kXcdNone, 0);
        case 256: return RG_KD(4, 4, 256, kXcdNone, 0);
        case 320: return RG_KD(4, 4, 320, kXcdNone, 0);
        default: return RG_KD(4, 4, 384, kXcdNone, 0);
      }
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
                                   const float* weights, int64_t n_vox, int64_t n_pairs, int64_t line_len,
                                   const float* packed, int32_t n_fields, int32_t stride, int64_t n_gates,
                                   float fill_value, float* out, int32_t variant, rg_stream_t stream) {
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
#ifndef RG_EXPERIMENTS
  RG_REQUIRE(variant == 0 || variant == 128 || variant == 192 || variant == 256 || variant == 320 || variant == 384 ||
                 variant == 512, RG_EINVAL,
             "rg_csr_apply_f32_ex: variant must be 0 (default) or a pipeline tile of 128, 192, 256, 320, 384 or 512 pairs");
#endif
  if (n_vox == 0) return RG_OK;
  if (line_len <= 0) line_len = n_vox;   // no grid lines known: one line, plain 64-row segments
  RG_REQUIRE(n_vox % line_len == 0, RG_EINVAL, "rg_csr_apply_f32: n_vox=%ld is not a multiple of line_len=%ld",
             (long)n_vox, (long)line_len);
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    return dispatch<int64_t>(n_fields, variant, indptr, gate_idx, weights, n_vox, line_len, packed, n_gates, fill_value,
                             out, s);
  return dispatch<int32_t>(n_fields, variant, indptr, gate_idx, weights, n_vox, line_len, packed, n_gates, fill_value, out,
                           s);
}

extern "C" int rg_csr_apply_f32(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx,
                                const float* weights, int64_t n_vox, int64_t n_pairs, int64_t line_len,
                                const float* packed, int32_t n_fields, int32_t stride, int64_t n_gates, float fill_value,
                                float* out, rg_stream_t stream) {
  return rg_csr_apply_f32_ex(indptr, indptr_is_i64, gate_idx, weights, n_vox, n_pairs, line_len, packed, n_fields, stride,
                             n_gates, fill_value, out, 0, stream);
}
