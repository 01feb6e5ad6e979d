// K1p  rg_csr_compact_apply_columns_f32: the row-wise kernel of rg_csr_compact.hip with workgroups that are PERSISTENT over a
// COLUMN of chunks -- the same (line group, segment) patch through consecutive grid levels -- and an optional products
// epilogue (COLMAX / first-argmax in registers, selected levels stored as planes) that removes the 3-D grid's round trip
// through HBM.  radar_grid/interpolate.py:69-104 (the masked weighted mean), :137-140 (several fields, one pass);
// radar_grid/products.py:361-412 (CAPPI needs two levels), :462-490 (column maximum over a level window).
//
// Why: callers that keep 2-D products only (BASELINE configs 3 and 5) no longer write F x 4 x V bytes of grid and read them
// back -- on this part writes mixed into a streaming read cost ten times their stand-alone price.
//
// What ships is the COLUMN MODE of the row-wise kernel itself (csr_compact_rowwise_kernel<..., COLS = true> in
// rg_csr_compact.hip): the same code from a chunk's row pointers to its row sums, one chunk after the other behind a
// barrier, with 4-12 more VGPRs for the running maxima.  This file holds the entry point, the merge of level pieces, and --
// under -DRG_EXPERIMENTS only -- the two attempts of round 4 to ALSO take the per-chunk chain of dependent loads (row
// pointers / offsets -> dictionary -> field gathers -> barrier -> records) off the critical path, both bit-identical,
// both measured slower than the kernel they were meant to beat (profiles/r04_columns_*.json, EXPERIMENTS.md):
//   LOADER  (lanes_hint + 1000)  a fifth wavefront gathers chunk k+1's window into a SECOND LDS window and publishes chunk
//           k+2's metadata in an LDS ring while four stream chunk k; the streamers request chunk k+1's first records before
//           they finish chunk k, so nothing but one barrier happens at a boundary.  1.6-2.6x slower: the second window and
//           the fifth wavefront halve the resident streaming wavefronts per CU (8-12 instead of 20-24), and this kernel's
//           speed is its bytes in flight.
//   WALK    (lanes_hint + 2000)  four wavefronts, one window; chunk k+1's row pointers / record range / dictionary range
//           are requested when chunk k starts to stream and wait in registers, and at the boundary the first records are
//           requested BEFORE the window is filled.  7-11 % slower with the 3-D store (12-36 more VGPRs than the one-chunk
//           kernel cost a wavefront per SIMD; capped to its occupancy the allocator spills), break-even products-only.
//
// Arithmetic: per row exactly the row-wise kernel's -- same lanes per row (from the segment's mean row length), same
// batches of KPRE records, same two chains per lane, same butterfly, same float64 division -- so the 3-D grid is the same
// BITS as rg_csr_compact_apply_packed_f32 (tile = 0) and as oracle.csr_apply_rowwise_order (tests assert both).
// Products: lane == row of a streaming wavefront sees the levels of its (y, x) column in ascending order, so the column
// maximum follows np.fmax.reduce (first of equal values wins, NaN ignored) and the arg is the first level attaining it,
// -1 for an all-NaN column -- the contract of rg_column_reduce_f32 (csrc/rg_products.hip), bit for bit.  Levels
// [keep_lo, keep_lo + n_keep) are stored as planes (CAPPI's two levels; the caller blends them with rg_cappi_lerp_f32).
// With z_pieces > 1 a column is cut into level ranges handled by different workgroups (more workgroups on small grids);
// the partial (max, arg) planes are merged in ascending level order by a second tiny kernel -- associative, bit-exact.
//
// Roofline: HBM.  Bytes per launch = the row-wise kernel's minus what is not stored: 16*R + 8*(S+1) + 4*D + 8*(C+1) +
// ip*(V+1) + F*5*G + F*4*V [only if out] + F*4*Vxy*(n_keep [+ 2 if colmax]).
#include "rg_compact_layout.hpp"

#ifdef RG_EXPERIMENTS
namespace {

constexpr int kLoaderWave = kH;                 // wavefronts 0 .. kH-1 stream, wavefront kH loads
constexpr int kColThreads = 64 * (kH + 1);
constexpr int kMetaRing = 3;
constexpr int kFillBatchCols = 8;               // window entries per loader lane and batch (two memory latencies per batch)

struct SegMeta {        // one segment (= streaming wavefront) of a chunk, published by the loader
  int rs[65];           // rs[l] = first pair of row l relative to the segment's first pair; rs[l >= nrows] = span
  int nrows;            // 0: this line lies past the end of the plane
  int rec_n;            // records of the segment
  int w_lo;             // split chunk: this wavefront's dictionary is entries [w_lo, w_hi) of the chunk's; else [0, nd_all)
  int w_hi;
  int pad_;
  long rec_b;           // first record
  long r0;              // first row (flat voxel index)
};
struct ChunkMeta {
  SegMeta seg[kH];
  long d0;              // dict_ptr[chunk]
  int nd_all;           // dictionary entries (header included when split)
  int plane;            // grid level
};
static_assert(sizeof(SegMeta) % 8 == 0 && sizeof(ChunkMeta) % 16 == 0, "LDS image: 16-byte aligned ring slots");

// The read-only arrays are separate `const __restrict__` kernel parameters on purpose: only then may the compiler fetch
// wave-uniform values (row pointers of a segment's ends, record and dictionary ranges) with scalar loads into SGPRs.
// Packed into a by-value struct they lose `noalias`, every such value is fetched with a vector load and occupies a VGPR
// (two for a 64-bit one) for as long as it lives -- measured: +12 VGPRs for the metadata held one chunk ahead.
struct ColumnsPointers {
  const void* indptr;
  const int64_t* dict_ptr;
  const int32_t* dict;
  const float* packed;
  const rg_u32x4* rec;
  const int64_t* rec_ptr;
  const int32_t* order;   // optional: workgroup -> piece * n_cols + column (heaviest first); null = identity
  float* out;             // [F][n_vox] or null
  float* planes;          // [F][n_keep][n_xy] or null
  float* col_val;         // [pieces][F][n_xy] (pieces == 1: the caller's plane) or null
  int32_t* col_arg;       // same shape, or null
};
struct ColumnsArgs {      // everything that is not a pointer
  ChunkGrid cg;
  long n_vox, n_xy;
  unsigned last_gate, w_base, n_cols;
  float fill;
  int window_cap, lanes_hint, rec_order, pieces, keep_lo, n_keep, col_lo, col_hi;
};

template <int NF>
constexpr int window_entry_floats() { return RowwiseConfig<NF>::narrow ? 3 : NF == 1 ? 2 : stride_for(NF); }
struct Best {
  float v;     // NaN = nothing seen yet
  int idx;     // -1 = nothing seen yet
};

// np.fmax.reduce in level order (rg_products.hip: step<true>): the first level starts the reduction, later ones replace it
// only when strictly greater; NaN never wins.
__device__ __forceinline__ void colmax_step(Best& acc, float v, int z) {
  if (acc.idx < 0) {
    if (!isnan(v)) { acc.v = v; acc.idx = z; }
  } else {
    const bool keep = acc.v >= v || isnan(v);
    if (!keep) { acc.v = v; acc.idx = z; }
  }
}

__device__ __forceinline__ void lds_barrier() {
  // LDS traffic of this wavefront done, then the workgroup barrier.  No vmcnt wait: record loads requested for the next
  // chunk stay in flight across it, and so do the grid stores (nothing in the workgroup reads them back).
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// WALK: the register allocator is held to the occupancy of the one-chunk kernel (workgroups per CU = wavefronts per SIMD:
// 6 / 6 / 5 / 4 for 1-4 fields).  Left alone it hoists per-lane addresses out of the chunk loop and ends 20-30 VGPRs
// higher -- one or two wavefronts per SIMD fewer, and this kernel's speed is its bytes in flight.
#ifndef RG_COLUMNS_WAVES
#define RG_COLUMNS_WAVES(NF_) ((NF_) <= 2 ? 6 : (NF_) == 3 ? 5 : 4)
#endif
template <typename IndT, int NF, int STRIDE, bool PRODUCTS, bool LOADER>
__global__ __launch_bounds__(LOADER ? kColThreads : 64 * kH, LOADER ? 1 : RG_COLUMNS_WAVES(NF)) void csr_columns_kernel(
    const IndT* __restrict__ indptr, const int64_t* __restrict__ p_dict_ptr, const int32_t* __restrict__ p_dict,
    const float* __restrict__ p_packed, const rg_u32x4* __restrict__ p_rec, const int64_t* __restrict__ p_rec_ptr,
    const int32_t* __restrict__ p_order, float* __restrict__ p_out, float* __restrict__ p_planes,
    float* __restrict__ p_col_val, int32_t* __restrict__ p_col_arg, const ColumnsArgs a) {
  static_assert(NF >= 1 && NF <= 4 && (STRIDE == 1 || STRIDE == 2 || STRIDE == 4), "passes of 1-4 fields");
  using Cfg = RowwiseConfig<NF>;
  constexpr int KPRE = Cfg::kpre;
  constexpr bool kNarrow = Cfg::narrow, kRegs = Cfg::regs;
  constexpr bool kPremask = NF == 1;            // the window holds (value, 1) / (0, 0): see the row-wise kernel
  constexpr int WS = window_entry_floats<NF>();
  constexpr int KS = 2;                         // chains per lane (the row-wise kernel's order)
  // LDS: [metadata ring (LOADER)] [row sums (1-2 fields)] [window 0] [window 1 (LOADER)] -- one array: everything dynamic
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ChunkMeta* const meta = reinterpret_cast<ChunkMeta*>(lds);
  constexpr int kMetaFloats = LOADER ? (int)sizeof(ChunkMeta) * kMetaRing / 4 : 0;
  f32x2* const rowacc_all = reinterpret_cast<f32x2*>(lds + kMetaFloats);
  constexpr int kRowaccFloats = kRegs ? 0 : kH * 64 * NF * 2;
  float* const window0 = lds + kMetaFloats + kRowaccFloats;
  const int win_floats = ((a.window_cap + 1) * WS + 3) & ~3;     // one entry beyond window_cap: the all-EXCLUDED sentinel

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const ChunkGrid& cg = a.cg;

  // ---- this workgroup's column piece ---------------------------------------------------------------------------
  const unsigned item = p_order ? (unsigned)p_order[blockIdx.x] : blockIdx.x;
  const unsigned piece = item / a.n_cols, q = item - piece * a.n_cols;
  const unsigned yg = q / cg.nsx, col = q - yg * cg.nsx;
  // columns rotated per line group, as the one-chunk kernels rotate theirs: consecutive workgroups go to consecutive
  // XCDs, and without it a column of the grid would stay on one XCD for the whole launch
  unsigned sx = col + (yg * cg.rot_step) % cg.nsx;
  sx = sx >= cg.nsx ? sx - cg.nsx : sx;
  const int nz = (int)cg.n_planes;
  const int z0 = (int)((long)piece * nz / a.pieces), z1 = (int)((long)(piece + 1) * nz / a.pieces);
  const int n_chunks = z1 - z0;                  // >= 1: the host keeps pieces <= planes

  // Everything about chunk k of this workgroup that comes from memory, as one wavefront needs it.  `slot_of`: where the
  // segment's records are (include/radargrid_hip.h: RG_REC_ORDER_*).
  struct Raw {
    long seg_b, rec_b, r0, d0;   // first pair / first record / first row of the segment; dictionary offset of the chunk
    unsigned mine_s, mine_e;     // lane == row: LOW 32 bits of the absolute first pair of the row and of the next one
                                 // (a segment holds fewer than 2^31 pairs: offsets inside it need no more)
    int rec_n, nrows, nd_all, w_lo, w_hi, plane, span;
  };
  auto chunk_of = [&](int k, unsigned& grp) -> unsigned {
    grp = (unsigned)(z0 + k) * cg.nyg + yg;
    return grp * cg.nsx + sx;
  };
  auto slot_of = [&](unsigned grp, const Segment& sg, int w) -> long {
    if (a.rec_order != RG_REC_ORDER_DISPATCH) return sg.seg;
    const unsigned shift = ((grp + cg.grp0) * cg.rot_step) % cg.nsx;
    const unsigned bcol = sx >= shift ? sx - shift : sx + cg.nsx - shift;     // the block column whose rotated column is sx
    return ((long)grp * cg.nsx + bcol) * kH + w;
  };
  auto low32 = [&](long row) -> unsigned {      // low half of indptr[row] (little endian): one dword load for either type
    if constexpr (sizeof(IndT) == 8) return reinterpret_cast<const unsigned*>(indptr)[2 * row];
    else return (unsigned)indptr[row];
  };
  auto fetch = [&](int k, int w, int lane_) -> Raw {   // w wave-uniform; every load is independent of every other
    unsigned grp;
    const unsigned chunk = chunk_of(k, grp);
    const Segment sg = chunk_segment(cg, chunk, w);
    Raw r;
    r.plane = z0 + k;
    r.nrows = sg.nrows;
    r.r0 = sg.r0;
    r.d0 = p_dict_ptr[chunk];
    r.nd_all = (int)(p_dict_ptr[chunk + 1] - r.d0);
    r.seg_b = r.rec_b = r.mine_s = r.mine_e = 0;
    r.rec_n = r.span = 0;
    r.w_lo = 0;
    r.w_hi = r.nd_all;
    if (sg.nrows) {                               // wave-uniform
      r.seg_b = (long)indptr[sg.r0];
      r.span = (int)((long)indptr[sg.r0 + sg.nrows] - r.seg_b);
      r.mine_s = low32(sg.r0 + (lane_ < sg.nrows ? lane_ : sg.nrows));
      r.mine_e = low32(sg.r0 + (lane_ + 1 < sg.nrows ? lane_ + 1 : sg.nrows));
      const long slot = slot_of(grp, sg, w);
      r.rec_b = p_rec_ptr[slot];
      r.rec_n = (int)(p_rec_ptr[slot + 1] - r.rec_b);
      if (r.nd_all > 65536) {                     // split chunk: one dictionary per wavefront behind a header
        r.w_lo = p_dict[r.d0 + w];
        r.w_hi = w + 1 < kH ? p_dict[r.d0 + w + 1] : r.nd_all;
      }
    }
    return r;
  };
  // the field window of a chunk: `threads` lanes (all of them wave-complete) write entries i, i + threads, ...
  auto fill_window = [&](float* __restrict__ window, long d0, int nd_all, int tid, int threads, auto batch_tag) {
    constexpr int kBatch = decltype(batch_tag)::value;
    const int last_entry = nd_all > 0 ? nd_all - 1 : 0;
    const int32_t* __restrict__ cd = nd_all > 0 ? p_dict + d0 : (const int32_t*)p_dict_ptr;   // never an empty dictionary
    for (int i0 = tid; i0 <= nd_all; i0 += threads * kBatch) {
      unsigned gate[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = i0 + u * threads;
        gate[u] = (unsigned)cd[i < last_entry ? i : last_entry];
      }
      float v[kBatch][STRIDE];
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        rg::load_packed<STRIDE>(p_packed, gate[u] < a.last_gate ? gate[u] : a.last_gate, v[u]);
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        // branch-free, as in the row-wise kernel: a lane past the end writes the sentinel into the sentinel's entry again
        const int i_raw = i0 + u * threads;
        const int i = i_raw < nd_all ? i_raw : nd_all;
        if (i_raw >= nd_all) {
#pragma unroll
          for (int s = 0; s < STRIDE; ++s) v[u][s] = __builtin_bit_cast(float, RG_EXCLUDED_BITS);
        }
        if constexpr (kNarrow) {
          window[i * 3] = v[u][0]; window[i * 3 + 1] = v[u][1]; window[i * 3 + 2] = v[u][2];
        } else if constexpr (kPremask) {
          const bool good = rg::f32_bits(v[u][0]) != RG_EXCLUDED_BITS;
          reinterpret_cast<f32x2*>(window)[i] = good ? (f32x2){v[u][0], 1.0f} : (f32x2){0.0f, 0.0f};
        } else if constexpr (STRIDE == 2) {
          reinterpret_cast<f32x2*>(window)[i] = (f32x2){v[u][0], v[u][1]};
        } else {
          reinterpret_cast<f32x4*>(window)[i] = (f32x4){v[u][0], v[u][1], v[u][2], v[u][3]};
        }
      }
    }
  };

  // =============================== the loader wavefront (LOADER variant only) =======================================
  if constexpr (LOADER) {
    if (wv == kLoaderWave) {
      struct DictRange { long d0; int nd; };
      auto publish = [&](int k) -> DictRange {    // metadata of this workgroup's chunk k -> ring slot k % 3
        ChunkMeta& m = meta[k % kMetaRing];
        DictRange dr{0, 0};
#pragma unroll
        for (int w = 0; w < kH; ++w) {
          const Raw r = fetch(k, w, lane);
          SegMeta& sm = m.seg[w];
          sm.rs[lane] = r.nrows ? (int)(r.mine_s - (unsigned)r.seg_b) : 0;
          if (lane == 0) {
            sm.rs[64] = r.span;
            sm.nrows = r.nrows;
            sm.rec_n = r.rec_n;
            sm.w_lo = r.w_lo;
            sm.w_hi = r.w_hi;
            sm.rec_b = r.rec_b;
            sm.r0 = r.r0;
            if (w == 0) {
              m.d0 = r.d0;
              m.nd_all = r.nd_all;
              m.plane = r.plane;
            }
          }
          dr = DictRange{r.d0, r.nd_all};
        }
        return dr;
      };
      auto fill = [&](int k, DictRange dr) {      // skipped when the dictionary does not fit: the streamers gather per pair
        if (dr.nd <= a.window_cap)
          fill_window(window0 + (k & 1) * win_floats, dr.d0, dr.nd, lane, 64, std::integral_constant<int, kFillBatchCols>{});
      };
      DictRange next = publish(0);                 // chunk k + 1 of the loop below
      DictRange after = n_chunks > 1 ? publish(1) : DictRange{0, 0};
      fill(0, next);
      lds_barrier();                               // B_init: metadata 0 and 1, window 0
      next = after;
      for (int k = 0; k < n_chunks; ++k) {         // the streamers work on chunk k
        if (k + 2 < n_chunks) after = publish(k + 2);
        if (k + 1 < n_chunks) fill(k + 1, next);
        next = after;
        lds_barrier();                             // B_k
      }
      return;
    }
  }

  // =============================== the streaming wavefronts ==========================================================
  f32x2* const rowacc = rowacc_all + (kRegs ? 0 : wv * 64 * NF);
  unsigned wmask = 0x3FFFFFFu;                     // 26-bit weight mask in a VGPR: (x & mask) | w_base is one v_and_or_b32
  asm volatile("" : "+v"(wmask));
  constexpr int kOutOfRange = 0x7FFFFFF0;          // byte offset no segment reaches: the load returns zeros

  struct Ctx {            // one chunk as this wavefront sees it
    int rs_o, re_o;       // lane == row: its pairs [rs_o, re_o) of the segment
    int trips_row;        // lane == first row of a round: the most records any row of the round gives one lane
    int nrows, lgl, rounds, plane, nd_all, nd_last;
    bool windowed;
    rsrc_t rr;            // the segment's records
    long r0, d0;
    const int32_t* cdict; // per-pair path: this wavefront's dictionary
    const float* window;  // the chunk's LDS window
  };
  auto u64 = [](long v) -> long {                  // a wave-uniform 64-bit value into SGPRs
    return ((long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  };
  auto make_ctx = [&](const Raw& r, const float* window) -> Ctx {
    Ctx c;
    c.nrows = __builtin_amdgcn_readfirstlane(r.nrows);
    const long seg_b = u64(r.seg_b);
    c.rs_o = c.nrows ? (int)(r.mine_s - (unsigned)seg_b) : 0;
    c.re_o = c.nrows ? (int)(r.mine_e - (unsigned)seg_b) : 0;
    const int rec_n = __builtin_amdgcn_readfirstlane(r.rec_n);
    const int w_lo = __builtin_amdgcn_readfirstlane(r.w_lo), w_hi = __builtin_amdgcn_readfirstlane(r.w_hi);
    c.r0 = u64(r.r0);
    c.d0 = u64(r.d0);
    c.nd_all = __builtin_amdgcn_readfirstlane(r.nd_all);
    c.plane = __builtin_amdgcn_readfirstlane(r.plane);
    c.windowed = c.nd_all <= a.window_cap;
    const int nd = w_hi - w_lo;
    c.nd_last = nd > 0 ? nd - 1 : 0;
    c.cdict = p_dict + c.d0 + w_lo;
    c.window = window;
    c.rr = make_rsrc(p_rec + u64(r.rec_b), (long)rec_n * 16);
    const int span = __builtin_amdgcn_readfirstlane(r.span);
    // lanes per row: from the segment's mean row length, exactly as the row-wise kernel chooses them
    int lgl;
    if (a.lanes_hint > 0 && a.lanes_hint <= 64) {
      lgl = 31 - __builtin_clz(a.lanes_hint);
    } else {
      const int target = a.lanes_hint > 70 ? a.lanes_hint - 70 : Cfg::target;
      const int mean_rec = c.nrows ? span / (3 * c.nrows) + 1 : 1;
      const int need = (mean_rec + target - 1) / target;
      lgl = need <= 1 ? 0 : 32 - __builtin_clz(need - 1);
    }
    c.lgl = __builtin_amdgcn_readfirstlane(lgl > 6 ? 6 : lgl);
    const int nl = 1 << c.lgl, rpr = 64 >> c.lgl;
    c.rounds = (c.nrows + rpr - 1) >> (6 - c.lgl);
    const unsigned q0_row = (unsigned)c.rs_o / 3u;
    const unsigned q1_row = c.re_o > c.rs_o ? ((unsigned)c.re_o + 2u) / 3u : q0_row;
    int trips = (int)((q1_row - q0_row + (unsigned)nl - 1u) >> c.lgl);
    for (int mm = 1; mm < rpr; mm <<= 1) {
      const int o = __shfl_xor(trips, mm, 64);
      trips = o > trips ? o : trips;
    }
    c.trips_row = trips;
    return c;
  };
  auto lds_raw = [&](int k) -> Raw {               // LOADER: what the loader published for this wavefront
    const ChunkMeta& m = meta[k % kMetaRing];
    const SegMeta& sm = m.seg[wv];
    Raw r;
    r.seg_b = 0;
    r.mine_s = (unsigned)sm.rs[lane];
    r.mine_e = (unsigned)sm.rs[lane + 1];
    r.nrows = sm.nrows;
    r.rec_n = sm.rec_n;
    r.w_lo = sm.w_lo;
    r.w_hi = sm.w_hi;
    r.rec_b = sm.rec_b;
    r.r0 = sm.r0;
    r.d0 = m.d0;
    r.nd_all = m.nd_all;
    r.plane = m.plane;
    r.span = sm.rs[64];
    return r;
  };

  struct Step {          // one batch of KPRE record loads per lane (see the row-wise kernel)
    int lo0;
    unsigned len;
    int rem, off0, myrow, rho, left;
    bool live;
  };
  auto setup = [&](const Ctx& c, int rho) -> Step {
    const int rpr = 64 >> c.lgl;
    Step r;
    r.rho = rho;
    r.myrow = rho * rpr + (lane >> c.lgl);
    r.live = r.myrow < c.nrows;
    const int qs = __shfl(c.rs_o, r.myrow & 63, 64);          // unconditional: dead lanes still supply their bounds
    const int qe_row = __shfl(c.re_o, r.myrow & 63, 64);
    const int qe = r.live ? qe_row : qs;
    const int q0 = (int)((unsigned)qs / 3u);
    const int q1 = qe > qs ? (int)(((unsigned)qe + 2u) / 3u) : q0;
    const int qq = q0 + (lane & ((1 << c.lgl) - 1));
    r.lo0 = qs - 3 * qq;
    r.len = (unsigned)(qe - qs);
    r.rem = q1 - qq;
    r.off0 = qq * 16;
    r.left = rho < c.rounds ? __builtin_amdgcn_readfirstlane(__shfl(c.trips_row, (rho * rpr) & 63, 64)) : 0;
    return r;
  };
  auto advance = [&](const Ctx& c, const Step& r) -> Step {
    if (r.left > KPRE) {
      Step n = r;
      n.lo0 -= 3 * (KPRE << c.lgl);
      n.rem -= KPRE << c.lgl;
      n.off0 += 16 * (KPRE << c.lgl);
      n.left -= KPRE;
      return n;
    }
    return setup(c, r.rho + 1);
  };
  auto issue = [&](const Ctx& c, const Step& r, rg_u32x4 (&regs)[KPRE]) {
#pragma unroll
    for (int k = 0; k < KPRE; ++k)
      regs[k] = rg_buffer_load_v4u32(c.rr, (k << c.lgl) < r.rem ? r.off0 + 16 * (k << c.lgl) : kOutOfRange, 0, 0);
  };

  float ap[KS][NF], aw[KS][NF];                    // the lane's two running chains
  float mine_p[NF], mine_w[NF];                    // kRegs: lane == row
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    mine_p[f] = mine_w[f] = 0.0f;
#pragma unroll
    for (int k = 0; k < KS; ++k) ap[k][f] = aw[k][f] = 0.0f;
  }
  Best best[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) { best[f].v = __builtin_nanf(""); best[f].idx = -1; }

  auto consume = [&](const Ctx& c, const Step& r, const rg_u32x4& q4, int k, auto wtag) {
    constexpr bool kWindowed = decltype(wtag)::value;
    const int lo = r.lo0 - 3 * (k << c.lgl);
    const unsigned len = (k << c.lgl) < r.rem ? r.len : 0u;
    float w[3];
    int pos[3];
    w[0] = __builtin_bit_cast(float, (q4.x & wmask) | a.w_base);
    w[1] = __builtin_bit_cast(float, (q4.y & wmask) | a.w_base);
    w[2] = __builtin_bit_cast(float, (q4.z & wmask) | a.w_base);
    pos[0] = (int)(q4.w & 0xFFFFu);
    pos[1] = (int)(q4.w >> 16);
    unsigned p2b = q4.y >> 26, p2c = q4.z >> 26;
    asm volatile("" : "+v"(p2b), "+v"(p2c));
    pos[2] = (int)((p2c << 12) | ((p2b << 6) | (q4.x >> 26)));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool mine = (unsigned)(i - lo) < len;
      const int p = kWindowed ? pos[i] : (pos[i] < c.nd_last ? pos[i] : c.nd_last);
      float v[STRIDE];
      if constexpr (kWindowed) {
        const int e = mine ? p : c.nd_all;         // not this row's pair: the all-EXCLUDED sentinel entry
        __builtin_assume((unsigned)e <= 65536u);
        if constexpr (kNarrow) {
          v[0] = c.window[e * 3]; v[1] = c.window[e * 3 + 1]; v[2] = c.window[e * 3 + 2];
        } else if constexpr (kPremask) {
          const f32x2 term = (f32x2){w[i], w[i]} * reinterpret_cast<const f32x2*>(c.window)[e];
          ap[k % KS][0] += term.x;
          aw[k % KS][0] += term.y;
          continue;
        } else if constexpr (STRIDE == 2) {
          const f32x2 x = reinterpret_cast<const f32x2*>(c.window)[e];
          v[0] = x.x; v[1] = x.y;
        } else {
          const f32x4 x = reinterpret_cast<const f32x4*>(c.window)[e];
          v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        }
      } else {
        const unsigned g0 = (unsigned)c.cdict[p];
        rg::load_packed<STRIDE>(p_packed, g0 < a.last_gate ? g0 : a.last_gate, v);
        if (!mine) {
#pragma unroll
          for (int s = 0; s < STRIDE; ++s) v[s] = __builtin_bit_cast(float, RG_EXCLUDED_BITS);
        }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {               // masked gate: contributes to neither sum (interpolate.py:78-79)
        const bool good = rg::f32_bits(v[f]) != RG_EXCLUDED_BITS;
        // one select per field and pair; w is never 0 (a zero weight is not codable), so 0 * x = +0 only for an excluded gate
        const float wf = good ? w[i] : 0.0f;
        ap[k % KS][f] += rg_fmul_legacy(wf, v[f]);
        aw[k % KS][f] += wf;
      }
    }
  };
  // `wtag`: values from the LDS window (std::true_type) or, for a chunk whose dictionary does not fit, per pair from
  // memory.  The one-chunk kernel's lesson holds here too: the choice is made once per chunk, OUTSIDE the step loop.
  auto process = [&](const Ctx& c, const Step& r, const rg_u32x4 (&regs)[KPRE], bool last, auto wtag) {
#pragma unroll
    for (int k = 0; k < KPRE; ++k)
      if (k < r.left) consume(c, r, regs[k], k, wtag);
    if (!last) return;
    float sv[2 * NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float sp = ap[0][f], sw = aw[0][f];          // chains in ascending order
      ap[0][f] = aw[0][f] = 0.0f;
#pragma unroll
      for (int k = 1; k < KS; ++k) {
        sp += ap[k][f];
        sw += aw[k][f];
        ap[k][f] = aw[k][f] = 0.0f;
      }
      sv[2 * f] = sp;
      sv[2 * f + 1] = sw;
    }
    const int nl = 1 << c.lgl, rpr = 64 >> c.lgl;
    rg::butterfly<2 * NF>(sv, nl);
    if constexpr (kRegs) {
      const int first = r.myrow - (lane >> c.lgl);           // the round's first row (wave-uniform)
      const bool take = lane >= first && lane < first + rpr;
      const int src = ((lane - first) << c.lgl) & 63;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const float gp = __shfl(sv[2 * f], src, 64), gw = __shfl(sv[2 * f + 1], src, 64);
        mine_p[f] = take ? gp : mine_p[f];
        mine_w[f] = take ? gw : mine_w[f];
      }
    } else if (r.live && (lane & (nl - 1)) == 0) {
#pragma unroll
      for (int f = 0; f < NF; ++f) rowacc[r.myrow * NF + f] = (f32x2){sv[2 * f], sv[2 * f + 1]};
    }
  };
  // lane == row: the weighted mean, the store(s) and the column products of one finished chunk
  auto finish = [&](const Ctx& c, int lane_) {
    if constexpr (!kRegs) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int z = c.plane;
    const bool keep = PRODUCTS && p_planes && z >= a.keep_lo && z < a.keep_lo + a.n_keep;      // wave-uniform
    const bool in_col = PRODUCTS && p_col_val && z >= a.col_lo && z <= a.col_hi;
    if (lane_ < c.nrows) {
      const long xy = c.r0 - (long)z * a.n_xy + lane_;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        f32x2 s;
        if constexpr (kRegs) s = (f32x2){mine_p[f], mine_w[f]};
        else s = rowacc[lane_ * NF + f];
        const float val = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : a.fill;
        if (p_out) p_out[(size_t)f * a.n_vox + c.r0 + lane_] = val;
        if constexpr (PRODUCTS) {
          if (keep) p_planes[((size_t)f * a.n_keep + (z - a.keep_lo)) * a.n_xy + xy] = val;
          if (in_col) colmax_step(best[f], val, z);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) mine_p[f] = mine_w[f] = 0.0f;
    if constexpr (!kRegs) {      // the next chunk's rounds overwrite the row sums: this chunk's reads come first
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };

  rg_u32x4 regs_a[KPRE], regs_b[KPRE];
  Ctx cur;
  Step sa, sb;
  if constexpr (!LOADER) {
    // ---- WALK: the one-chunk kernel's loop, chunk after chunk ------------------------------------------------------------
    // What a chunk needs from memory before its first record can be requested -- the lane's row pointers, the segment's
    // first pair and record range, the chunk's dictionary range -- is requested one chunk AHEAD (when the previous chunk
    // starts to stream) and simply waits in registers; where the segment lies (rows, level) is arithmetic and recomputed.
    struct Ahead {
      unsigned mine_s, mine_e;                     // the only per-lane state a workgroup carries from chunk to chunk
      long seg_b, rec_b, d0;                       // wave-uniform (scalar loads)
      int span, rec_n, nd_all;
    };
    auto request = [&](int k, int lane_) -> Ahead {
      const Raw r = fetch(k, wv, lane_);
      return Ahead{r.mine_s, r.mine_e, r.seg_b, r.rec_b, r.d0, r.span, r.rec_n, r.nd_all};
    };
    auto ctx_of = [&](int k, const Ahead& h) -> Ctx {
      unsigned grp;
      const unsigned chunk = chunk_of(k, grp);
      const Segment sg = chunk_segment(cg, chunk, wv);
      Raw r;
      r.plane = z0 + k;
      r.nrows = sg.nrows;
      r.r0 = sg.r0;
      r.mine_s = h.mine_s; r.mine_e = h.mine_e;
      r.seg_b = h.seg_b; r.rec_b = h.rec_b; r.d0 = h.d0;
      r.span = h.span; r.rec_n = h.rec_n; r.nd_all = h.nd_all;
      r.w_lo = 0;
      r.w_hi = h.nd_all;
      if (h.nd_all > 65536 && sg.nrows) {          // split chunk (rare): its header is read here, not ahead
        r.w_lo = p_dict[h.d0 + wv];
        r.w_hi = wv + 1 < kH ? p_dict[h.d0 + wv + 1] : h.nd_all;
      }
      return make_ctx(r, window0);
    };
    Ahead ahead = request(0, lane);
    for (int kc = 0; kc < n_chunks; ++kc) {
      // The chunk number the body sees is opaque to the optimiser: knowing that consecutive chunks lie one plane apart, it
      // turns every per-lane address of the body (row pointers, the F output rows, the kept planes) into a 64-bit
      // induction variable that lives across the whole step loop -- 20-30 VGPRs, one or two wavefronts per SIMD.
      int k = kc, tid_c = (int)threadIdx.x;
      asm volatile("" : "+s"(k));
      asm volatile("" : "+v"(tid_c));              // likewise the thread number wherever a chunk's prologue / epilogue uses it:
                                                   // what is derived from it there is recomputed per chunk, not kept
      if (kc > 0) lds_barrier();                   // every wavefront is done with the previous chunk's window
      cur = ctx_of(k, ahead);
      // the chunk's first records are requested BEFORE the window is filled: they land while the dictionary -> gather
      // chain runs (the register stage is idle at a chunk boundary anyway)
      sa = setup(cur, 0);
      issue(cur, sa, regs_a);
      if (cur.windowed) fill_window(window0, cur.d0, cur.nd_all, tid_c, 64 * kH, std::integral_constant<int, 4>{});
      if (kc + 1 < n_chunks) ahead = request(k + 1, tid_c & 63);
      lds_barrier();                               // the window is complete
      auto run = [&](auto wtag) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {             // (zero here already: a round's end clears them -- this tells the compiler)
#pragma unroll
          for (int kk = 0; kk < KS; ++kk) ap[kk][f] = aw[kk][f] = 0.0f;
        }
        for (;;) {                                 // two register stages, alternating: nothing in flight is ever copied
          sb = advance(cur, sa);
          issue(cur, sb, regs_b);
          process(cur, sa, regs_a, sb.rho != sa.rho, wtag);
          if (sb.rho >= cur.rounds) break;
          sa = advance(cur, sb);
          issue(cur, sa, regs_a);
          process(cur, sb, regs_b, sa.rho != sb.rho, wtag);
          if (sa.rho >= cur.rounds) break;
        }
      };
      if (cur.windowed) run(std::true_type{}); else run(std::false_type{});
      int lane_f = lane;
      asm volatile("" : "+v"(lane_f));
      finish(cur, lane_f);
    }
  } else {
    // ---- LOADER: one flat loop over the steps of all chunks ---------------------------------------------------------------
    Ctx nxt;
    int k_chunk = 0;
    lds_barrier();                                  // B_init
    cur = make_ctx(lds_raw(0), window0);
    sa = setup(cur, 0);
    issue(cur, sa, regs_a);
    nxt = cur;
    sb = sa;
    // One phase: `s` (records in `R`) is the step to sum; its successor -- the next batch of the same chunk, or the first
    // batch of the NEXT chunk -- is requested first, into the other register stage.  Returns true after the last chunk.
    auto phase = [&](Step& s, rg_u32x4 (&R)[KPRE], Step& t, rg_u32x4 (&T)[KPRE]) -> bool {
      const bool same_chunk = s.left > KPRE || s.rho + 1 < cur.rounds;       // wave-uniform
      const bool more_chunks = k_chunk + 1 < n_chunks;
      if (same_chunk) {
        t = advance(cur, s);
        issue(cur, t, T);
        if (cur.windowed) process(cur, s, R, t.rho != s.rho, std::true_type{});
        else process(cur, s, R, t.rho != s.rho, std::false_type{});
        return false;
      }
      if (more_chunks) {                           // metadata k+1 was published during chunk k-1: visible since B_{k-1}
        nxt = make_ctx(lds_raw(k_chunk + 1), window0 + ((k_chunk + 1) & 1) * win_floats);
        t = setup(nxt, 0);
        issue(nxt, t, T);
      }
      if (cur.windowed) process(cur, s, R, true, std::true_type{});
      else process(cur, s, R, true, std::false_type{});
      finish(cur, lane);
      lds_barrier();                               // B_k: window k+1 is complete, window k may be refilled
      if (!more_chunks) return true;
      cur = nxt;
      ++k_chunk;
      return false;
    };
    for (;;) {
      if (phase(sa, regs_a, sb, regs_b)) break;
      if (phase(sb, regs_b, sa, regs_a)) break;
    }
  }

  if constexpr (PRODUCTS) {
    if (p_col_val && lane < cur.nrows) {           // the column's rows are the same on every level
      const long xy = cur.r0 - (long)cur.plane * a.n_xy + lane;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const size_t o = ((size_t)piece * NF + f) * a.n_xy + xy;
        p_col_val[o] = best[f].v;
        if (p_col_arg) p_col_arg[o] = best[f].idx;
      }
    }
  }
}

}  // namespace
#endif   // RG_EXPERIMENTS

namespace {

// (max, first arg) of a column from the partial results of its level pieces, merged in ascending level order: a later
// piece wins only when strictly greater (rg_products.hip: merge<true> with b.idx > a.idx)
__global__ __launch_bounds__(256) void columns_merge_kernel(const float* __restrict__ part_val, const int32_t* __restrict__ part_arg,
                                                            int pieces, long n_planes_xy, float* __restrict__ out_val,
                                                            int32_t* __restrict__ out_arg) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_planes_xy) return;
  float v = part_val[i];
  int idx = part_arg[i];
  for (int p = 1; p < pieces; ++p) {
    const float bv = part_val[(size_t)p * n_planes_xy + i];
    const int bi = part_arg[(size_t)p * n_planes_xy + i];
    if (bi >= 0 && (idx < 0 || bv > v)) { v = bv; idx = bi; }
  }
  out_val[i] = v;
  if (out_arg) out_arg[i] = idx;
}

#ifdef RG_EXPERIMENTS
template <typename IndT, int NF, bool LOADER>
int launch_columns(const ColumnsPointers& p, const ColumnsArgs& a, bool products, hipStream_t s) {
  constexpr int STRIDE = stride_for(NF);
  constexpr int WS = window_entry_floats<NF>();
  constexpr int kWindows = LOADER ? 2 : 1;
  constexpr int kFixed = (LOADER ? (int)sizeof(ChunkMeta) * kMetaRing : 0) + (RowwiseConfig<NF>::regs ? 0 : kH * 64 * NF * 8);
  ColumnsArgs b = a;
  // the window(s) (+ one sentinel entry each) next to the row sums [and the metadata ring], within the 64 KiB a launch
  // gets without opting in to more; a smaller window only sends more chunks down the per-pair path (same results)
  const long room = ((65536 - kFixed - 64) / kWindows / (4 * WS)) - 2;
  if (b.window_cap > room) b.window_cap = (int)(room < 0 ? 0 : room);
  const size_t win_floats = (size_t)(((b.window_cap + 1) * WS + 3) & ~3);
  const size_t lds_bytes = (size_t)kFixed + kWindows * win_floats * sizeof(float);
  const unsigned blocks = b.n_cols * (unsigned)b.pieces;
  const unsigned threads = LOADER ? kColThreads : 64 * kH;
  const IndT* ip = static_cast<const IndT*>(p.indptr);
  if (products)
    hipLaunchKernelGGL((csr_columns_kernel<IndT, NF, STRIDE, true, LOADER>), dim3(blocks), dim3(threads), lds_bytes, s, ip,
                       p.dict_ptr, p.dict, p.packed, p.rec, p.rec_ptr, p.order, p.out, p.planes, p.col_val, p.col_arg, b);
  else
    hipLaunchKernelGGL((csr_columns_kernel<IndT, NF, STRIDE, false, LOADER>), dim3(blocks), dim3(threads), lds_bytes, s, ip,
                       p.dict_ptr, p.dict, p.packed, p.rec, p.rec_ptr, p.order, p.out, p.planes, p.col_val, p.col_arg, b);
  return rg::check_launch("rg_csr_compact_apply_columns_f32");
}

template <typename IndT>
int launch_columns_nf(int nf, const ColumnsPointers& p, const ColumnsArgs& a, bool products, bool loader, hipStream_t s) {
  if (loader) {
    switch (nf) {
      case 1: return launch_columns<IndT, 1, true>(p, a, products, s);
      case 2: return launch_columns<IndT, 2, true>(p, a, products, s);
      case 3: return launch_columns<IndT, 3, true>(p, a, products, s);
      default: return launch_columns<IndT, 4, true>(p, a, products, s);
    }
  }
  switch (nf) {
    case 1: return launch_columns<IndT, 1, false>(p, a, products, s);
    case 2: return launch_columns<IndT, 2, false>(p, a, products, s);
    case 3: return launch_columns<IndT, 3, false>(p, a, products, s);
    default: return launch_columns<IndT, 4, false>(p, a, products, s);
  }
}

#endif   // RG_EXPERIMENTS

}  // namespace

extern "C" int64_t rg_csr_columns_workspace_bytes(int64_t lines_per_plane, int64_t line_len, int32_t n_fields,
                                                  int32_t z_pieces) {
  if (lines_per_plane <= 0 || line_len <= 0 || n_fields < 1 || n_fields > 4 || z_pieces < 1) return RG_EINVAL;
  if (z_pieces == 1) return 0;
  return (int64_t)z_pieces * n_fields * lines_per_plane * line_len * 8;      // partial (max, arg) planes
}

extern "C" int rg_csr_compact_apply_columns_f32(const void* indptr, int32_t indptr_is_i64, const void* records,
                                                const int64_t* rec_ptr, int32_t rec_order, uint32_t w_base,
                                                const int64_t* dict_ptr, const int32_t* dict, int64_t n_vox, int64_t n_pairs,
                                                int64_t line_len, int64_t lines_per_plane, const float* packed,
                                                int32_t n_fields, int32_t stride, int64_t n_gates, float fill_value,
                                                float* out, float* level_planes, int32_t keep_lo, int32_t n_keep,
                                                float* col_max, int32_t* col_arg, int32_t col_lo, int32_t col_hi,
                                                int32_t window_cap, int32_t z_pieces, const int32_t* order, void* workspace,
                                                int64_t workspace_bytes, int32_t lanes_hint, rg_stream_t stream) {
  RG_REQUIRE(rec_order == RG_REC_ORDER_SEGMENT || rec_order == RG_REC_ORDER_DISPATCH, RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: rec_order=%d is neither RG_REC_ORDER_SEGMENT nor RG_REC_ORDER_DISPATCH", rec_order);
  RG_REQUIRE(n_fields >= 1 && n_fields <= 4, RG_EUNSUPPORTED, "rg_csr_compact_apply_columns_f32: n_fields=%d not in 1..4",
             n_fields);
  RG_REQUIRE(stride == stride_for(n_fields), RG_EINVAL, "rg_csr_compact_apply_columns_f32: stride=%d, expected %d for %d fields",
             stride, stride_for(n_fields), n_fields);
  RG_REQUIRE(indptr && dict_ptr && rec_ptr, RG_EINVAL, "rg_csr_compact_apply_columns_f32: null indptr/dict_ptr/rec_ptr");
  RG_REQUIRE(out || level_planes || col_max, RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: nothing to produce (out, level_planes and col_max are all null)");
  RG_REQUIRE(n_vox >= 0 && n_pairs >= 0, RG_EINVAL, "rg_csr_compact_apply_columns_f32: negative size");
  RG_REQUIRE(n_pairs == 0 || (records && dict && packed && n_gates > 0), RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: pairs present but records/dict/packed/n_gates missing");
  RG_REQUIRE(packed && n_gates > 0, RG_EINVAL, "rg_csr_compact_apply_columns_f32: packed fields missing");
  RG_REQUIRE(n_gates <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_columns_f32: n_gates exceeds int32 gate indices");
  RG_REQUIRE(n_vox <= 0x3FFFFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_columns_f32: n_vox too large for one launch");
  RG_REQUIRE(window_cap >= 0 && window_cap <= RG_COMPACT_MAX_WINDOW, RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: window_cap %d outside 0..%d", window_cap, RG_COMPACT_MAX_WINDOW);
  RG_REQUIRE(rg::aligned16(records), RG_EALIGN, "rg_csr_compact_apply_columns_f32: records must be 16-byte aligned");
  RG_REQUIRE(rg::aligned16(packed), RG_EALIGN, "rg_csr_compact_apply_columns_f32: packed must be 16-byte aligned");
  RG_REQUIRE((w_base & 0x3FFFFFFu) == 0, RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: w_base=0x%08x must have its low 26 bits clear", w_base);
#ifdef RG_EXPERIMENTS      // lanes_hint + 1000: the loader-wavefront variant, + 2000: the prefetching walk (A/B builds only)
  int variant = 0;         // 0 = the column mode of the row-wise kernel (what ships)
  if (lanes_hint >= 1000) {
    variant = lanes_hint / 1000;
    lanes_hint %= 1000;
  }
#endif
  RG_REQUIRE(lanes_hint == 0 || (lanes_hint >= 1 && lanes_hint <= 64 && (lanes_hint & (lanes_hint - 1)) == 0) ||
                 (lanes_hint > 70 && lanes_hint <= 99), RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: lanes_hint must be 0, a power of two up to 64, or 71..99");
  if (n_vox == 0) return RG_OK;
  ChunkGrid cg;
  RG_REQUIRE(make_chunk_grid(n_vox, line_len, lines_per_plane, &cg), RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: n_vox=%ld is not planes x lines_per_plane=%ld x line_len=%ld", (long)n_vox,
             (long)lines_per_plane, (long)line_len);
  RG_REQUIRE(chunk_count(cg) <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_columns_f32: too many chunks for one launch");
  RG_REQUIRE(z_pieces >= 1 && z_pieces <= cg.n_planes, RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: z_pieces=%d outside 1..planes=%ld", z_pieces, (long)cg.n_planes);
  RG_REQUIRE(!level_planes || (keep_lo >= 0 && n_keep >= 1 && keep_lo + (long)n_keep <= cg.n_planes), RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: kept levels [%d, %d) outside the grid's %ld planes", keep_lo, keep_lo + n_keep,
             (long)cg.n_planes);
  RG_REQUIRE(!col_arg || col_max, RG_EINVAL, "rg_csr_compact_apply_columns_f32: col_arg needs col_max");
  RG_REQUIRE(!col_max || (col_lo >= 0 && col_lo <= col_hi && col_hi < cg.n_planes), RG_EINVAL,
             "rg_csr_compact_apply_columns_f32: column window [%d, %d] outside the grid's %ld planes", col_lo, col_hi,
             (long)cg.n_planes);
  const long n_xy = cg.lines_per_plane * cg.line_len;
  const long need_ws = (col_max && z_pieces > 1) ? (long)z_pieces * n_fields * n_xy * 8 : 0;
  RG_REQUIRE(need_ws == 0 || (workspace && workspace_bytes >= need_ws), RG_EWORKSPACE,
             "rg_csr_compact_apply_columns_f32: workspace of %ld bytes needed for %d level pieces (rg_csr_columns_workspace_bytes)",
             need_ws, z_pieces);
  const long n_cols = (long)cg.nyg * cg.nsx;
  RG_REQUIRE(n_cols * z_pieces <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_columns_f32: too many workgroups");
  float* part_val = nullptr;
  int32_t* part_arg = nullptr;
  if (col_max && z_pieces > 1) {
    part_val = static_cast<float*>(workspace);
    part_arg = reinterpret_cast<int32_t*>(part_val + (size_t)z_pieces * n_fields * n_xy);
  }
  hipStream_t s = (hipStream_t)stream;
  int st;
#ifdef RG_EXPERIMENTS
  if (variant != 0) {
    ColumnsPointers p;
    ColumnsArgs a;
    a.cg = cg;
    p.indptr = indptr;
    p.dict_ptr = dict_ptr;
    p.dict = dict;
    p.packed = packed;
    p.rec = static_cast<const rg_u32x4*>(records);
    p.rec_ptr = rec_ptr;
    p.order = order;
    p.out = out;
    p.planes = level_planes;
    p.col_val = part_val ? part_val : col_max;
    p.col_arg = part_val ? part_arg : col_arg;
    a.n_vox = n_vox;
    a.n_xy = n_xy;
    a.last_gate = (unsigned)(n_gates - 1);
    a.w_base = w_base;
    a.n_cols = (unsigned)n_cols;
    a.fill = fill_value;
    a.window_cap = window_cap;
    a.lanes_hint = lanes_hint;
    a.rec_order = rec_order;
    a.pieces = z_pieces;
    a.keep_lo = level_planes ? keep_lo : 0;
    a.n_keep = level_planes ? n_keep : 0;
    a.col_lo = col_lo;
    a.col_hi = col_hi;
    const bool products = level_planes || col_max;
    st = indptr_is_i64 ? launch_columns_nf<int64_t>(n_fields, p, a, products, variant == 1, s)
                       : launch_columns_nf<int32_t>(n_fields, p, a, products, variant == 1, s);
  } else
#endif
  {
    rgl::RowwiseColumns c;
    c.order = order;
    c.planes = level_planes;
    c.col_val = part_val ? part_val : col_max;
    c.col_arg = part_val ? part_arg : col_arg;
    c.n_xy = n_xy;
    c.n_cols = (unsigned)n_cols;
    c.pieces = z_pieces;
    c.keep_lo = level_planes ? keep_lo : 0;
    c.n_keep = level_planes ? n_keep : 0;
    c.col_lo = col_lo;
    c.col_hi = col_hi;
    st = rg_launch_rowwise_columns(n_fields, indptr_is_i64 != 0, window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates,
                                   fill_value, out, s, records, rec_ptr, w_base, rec_order, lanes_hint, c);
  }
  if (st != RG_OK) return st;
  if (part_val) {
    const long n = (long)n_fields * n_xy;
    hipLaunchKernelGGL(columns_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part_val, part_arg, z_pieces, n,
                       col_max, col_arg);
    return rg::check_launch("rg_csr_compact_apply_columns_f32 (merge)");
  }
  return RG_OK;
}
