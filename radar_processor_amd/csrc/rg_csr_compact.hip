// K1c  csr_apply over a COMPACT device copy of the CSR.  Two families of kernels share the chunk layout, the
// dictionaries and the LDS field window described here:
//   * the TILE kernels (rg_csr_compact_apply_f32; rg_csr_compact_apply_packed_f32 with tile = 384), first half of this
//     file: rg_csr_apply_f32's pipeline minus its gather stage -- the same results as rg_csr_apply_f32, bit for bit;
//   * the ROW-WISE kernel (rg_csr_compact_apply_packed_f32, tile = 0; second half): no LDS tile at all, the default for
//     passes of 1-4 fields -- the same results to float32 rounding, in an order of its own.
//
// The reference's CSR (radar_grid/geometry.py:46-52) stores a 32-bit gate index per pair, and K1 pays for it twice:
// 4 of the 8 streamed bytes per pair, and one global gather per pair (F values wide), which the texture-address path
// serves at about 22 cycles per 64-lane instruction wherever the gates lie.  Neighbouring voxels share
// almost all their gates, so this copy groups the rows of a small 2-D PATCH of the grid into a chunk, lists the chunk's
// DISTINCT gates once (`dict`) and stores a 16-bit position in that list per pair:
//
//   chunk (plane z, line group yg, segment sx) = the segments sx of the H consecutive grid lines y = yg*H .. yg*H+H-1
//   of plane z; a segment = up to 64 consecutive rows of one line, exactly the unit one wavefront of rg_csr_apply_f32
//   owns.  A chunk therefore covers H x 64 voxels (H = RG_COMPACT_LINES = 4: 256 rows), a patch instead of a 256 x 1
//   strip: 2-3x fewer distinct gates per chunk, hence a smaller LDS window, more resident workgroups and fewer
//   dictionary bytes (measured per configuration in DESIGN.md);
//   bytes per pair 8 -> 6 (+ 4 bytes per distinct gate per chunk, 0.1-0.4 bytes per pair);
//   one workgroup = one chunk = H wavefronts: it gathers the chunk's few hundred field entries (F values each, the
//   rg_pack_fields_f32 layout) into an LDS window once -- coalesced reads of `dict`, one global gather per DISTINCT
//   gate -- and every pair then reads its values from LDS.  The window holds `window_cap` entries (chosen per geometry
//   to cover all but a handful of chunks -- those next to the radar, where every ray converges); a chunk with more
//   distinct gates gathers per pair through its dictionary instead.
//
// Tile kernels: pair order, weights, tiles and the float32 arithmetic are those of rg_csr_apply_f32 (same segments, same
// tiles, same products, same dynamic row phase), so they agree with it exactly for every field count.  The compact copy
// is derived from the standard CSR on the device (grid_geometry.CompactCSR) and the standard arrays stay the
// interchange format.
//
// Roofline: HBM.  Bytes per launch = 6*P + 4*D + sizeof(indptr)*(V+1) + 8*(C+1) + F*(5*G + 4*V)  with D = total
// dictionary entries, C = chunks; with the packed records (see rg_csr_compact_pack) 16*R + 8*(S+1) replace 6*P.
#include "rg_compact_layout.hpp"

namespace {

// ABLATE (timing-only diagnostics, results wrong by construction): 1 = no row phase, 2 = no products and no row phase
// (the values still have to be looked up: they are summed into the output), 3 = neither products, row phase nor window
// PACKED: positions and weights come from 16-byte records of three pairs each (see rg_csr_compact_pack) instead of the
// 2-byte position and 4-byte weight arrays: 5.33 instead of 6 bytes per pair, one dwordx4 per lane and 192 pairs.
template <typename IndT, int NF, int STRIDE, int TILE, int ABLATE = 0, int AUX = 0, bool PACKED = false>
__global__ __launch_bounds__(64 * kH) void csr_compact_kernel(
    const IndT* __restrict__ indptr, const uint16_t* __restrict__ lidx, const float* __restrict__ wts,
    const int64_t* __restrict__ dict_ptr, const int32_t* __restrict__ dict, ChunkGrid cg,
    const float* __restrict__ packed, unsigned last_gate, float fill, int window_cap, long n_vox,
    float* __restrict__ out, const rg_u32x4* __restrict__ rec, const int64_t* __restrict__ rec_ptr, unsigned w_base,
    int rec_order) {
  static_assert(TILE % 64 == 0, "a wave handles 64 pairs per step");
  static_assert(!PACKED || TILE % 192 == 0, "a packed tile is whole wave-loads of 64 three-pair records");
  constexpr int IT = TILE / 64;
  // window_cap entries: the packed slots of a gate (STRIDE floats), except that a 3-field entry drops the padding slot
  extern __shared__ __attribute__((aligned(16))) float window[];
  __shared__ __attribute__((aligned(16))) float tile_all[kH][TILE * rg::tile_floats(NF, STRIDE)];
  __shared__ f32x2 rowacc_all[kH][64 * NF];
  static_assert((sizeof(tile_all) + sizeof(rowacc_all)) % 16 == 0,
                "the dynamic window starts where the static arrays end and is accessed 16 bytes wide");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tile = tile_all[wv];
  f32x2* rowacc = rowacc_all[wv];

  const unsigned chunk = block_chunk(cg, blockIdx.x);
  const long d0 = dict_ptr[chunk];
  // A chunk's dictionary holds at most 65536 gates (positions are 16 bits).  The one exception is a SPLIT chunk -- more
  // distinct gates than that, as around the radar itself on a dense scan --, which stores one dictionary per wavefront
  // behind a header of kH offsets; its entry count (header included) exceeds 65536, which is how it is recognised.
  const int nd_all = (int)(dict_ptr[chunk + 1] - d0);
  const bool split = nd_all > 65536;
  const bool windowed = nd_all <= window_cap;        // workgroup-uniform; the rare wider chunk gathers per pair
  const int w_lo = split ? dict[d0 + wv] : 0;
  const int w_hi = split ? (wv + 1 < kH ? dict[d0 + wv + 1] : nd_all) : nd_all;
  const int nd = w_hi - w_lo;                        // entries of the dictionary this wavefront's positions refer to
  const int nd_last = nd > 0 ? nd - 1 : 0;
  const int32_t* __restrict__ cdict = dict + d0 + w_lo;

  const Segment sg = chunk_segment(cg, chunk, wv);
  const int nrows = sg.nrows;                        // wave-uniform; a wave without rows still helps to fill the window
  const long r0 = sg.r0;
  const long seg_b = nrows ? (long)indptr[r0] : 0;
  const long seg_e = nrows ? (long)indptr[r0 + nrows] : 0;
  const int span = (int)(seg_e - seg_b);
  const int rs_o = nrows ? (int)((long)indptr[r0 + (lane < nrows ? lane : nrows)] - seg_b) : 0;
  const int re_o = nrows ? (int)((long)indptr[r0 + (lane + 1 < nrows ? lane + 1 : nrows)] - seg_b) : 0;
#pragma unroll
  for (int f = 0; f < NF; ++f) rowacc[lane * NF + f] = (f32x2)(0.0f);

  // Two register stages, loop unrolled by two, every load unconditional and range-checked against the segment's last
  // pair -- the same exact-wait-count pipeline as rg_csr_apply_f32, without a gather stage.
  struct StagePlain {
    int ci[IT];
    float cw[IT];
  };
  struct StagePacked {
    rg_u32x4 r[PACKED ? IT / 3 : 1];
  };
  using Stage = std::conditional_t<PACKED, StagePacked, StagePlain>;
  Stage st[2];
  const uint16_t* __restrict__ li = lidx + seg_b;
  const float* __restrict__ wi = wts + seg_b;
  const int lane2 = lane * 2, lane4 = lane * 4;
  long rec_b = 0, rec_n = 0;                 // this segment's records (PACKED)
  if constexpr (PACKED) {
    if (nrows) {
      const long slot = rec_order == RG_REC_ORDER_DISPATCH ? (long)blockIdx.x * kH + wv : sg.seg;
      rec_b = rec_ptr[slot];
      rec_n = rec_ptr[slot + 1] - rec_b;
    }
  }
  auto stream = [&](Stage& sgs, int t) {   // t wave-uniform: the resources live in SGPRs
    if constexpr (PACKED) {
      const long r_t = t / 3;               // tiles are multiples of 192 pairs = 64 records
      const rsrc_t rr = make_rsrc(rec + rec_b + r_t, (rec_n - r_t) * 16);
#pragma unroll
      for (int k = 0; k < IT / 3; ++k) sgs.r[k] = rg_buffer_load_v4u32(rr, lane * 16 + k * 1024, 0, AUX);
    } else {
      const rsrc_t ri = make_rsrc(li + t, ((long)span - t) * 2);
      const rsrc_t rw = make_rsrc(wi + t, ((long)span - t) * 4);
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        sgs.ci[it] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(ri, lane2 + it * 128, 0, AUX);
        sgs.cw[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, lane4 + it * 256, 0, AUX));
      }
    }
  };
  // pair slot `it` of a stage -> (position, weight) and its index in the tile.  Plain: pair it*64 + lane.  Packed: lane
  // holds record k = it / 3 of the tile's k-th wave-load, i.e. pairs k*192 + 3*lane + (it % 3); a record is
  // [w0:26 | p2[0:6]] [w1:26 | p2[6:12]] [w2:26 | p2[12:16]] [p0:16 | p1:16], w = float32 bits minus w_base.
  auto decode = [&](const Stage& sgs, int (&ci)[IT], float (&cw)[IT]) {
    if constexpr (PACKED) {
#pragma unroll
      for (int k = 0; k < IT / 3; ++k) {
        const rg_u32x4 q = sgs.r[k];
        cw[3 * k] = __builtin_bit_cast(float, (q.x & 0x3FFFFFFu) + w_base);
        cw[3 * k + 1] = __builtin_bit_cast(float, (q.y & 0x3FFFFFFu) + w_base);
        cw[3 * k + 2] = __builtin_bit_cast(float, (q.z & 0x3FFFFFFu) + w_base);
        ci[3 * k] = (int)(q.w & 0xFFFFu);
        ci[3 * k + 1] = (int)(q.w >> 16);
        ci[3 * k + 2] = (int)((q.x >> 26) | ((q.y >> 26) << 6) | ((q.z >> 26) << 12));
      }
    } else {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        ci[it] = sgs.ci[it];
        cw[it] = sgs.cw[it];
      }
    }
  };
  auto eidx = [&](int it) -> int { return PACKED ? (it / 3) * 192 + 3 * lane + (it % 3) : it * 64 + lane; };
  // the first two tiles are requested BEFORE the window is filled: the two latencies overlap
  stream(st[0], 0);
  stream(st[1], TILE);

  // ---- the chunk's field window: one gather per DISTINCT gate --------------------------------------------
  if (windowed && ABLATE != 4) {
    for (int i = threadIdx.x; i < nd_all; i += 64 * kH) {
      const unsigned g0 = (unsigned)cdict[i];
      const unsigned g = g0 < last_gate ? g0 : last_gate;   // clamp: never fault
      float v[STRIDE];
      rg::load_packed<STRIDE>(packed, g, v);
      if constexpr (STRIDE == 1) {
        window[i] = v[0];
      } else if constexpr (STRIDE == 2) {
        reinterpret_cast<f32x2*>(window)[i] = (f32x2){v[0], v[1]};
      } else if constexpr (NF == 3) {
        window[i * 3] = v[0]; window[i * 3 + 1] = v[1]; window[i * 3 + 2] = v[2];
      } else {
#pragma unroll
        for (int q = 0; q < STRIDE; q += 4)
          reinterpret_cast<f32x4*>(window)[(size_t)i * (STRIDE / 4) + q / 4] = (f32x4){v[q], v[q + 1], v[q + 2], v[q + 3]};
      }
    }
  }
  if constexpr (ABLATE != 4) __syncthreads();

  // The tile loop exists twice -- values from the LDS window, or (over-wide chunk) position -> gate -> value from memory
  // -- selected once per workgroup: a uniform branch INSIDE the unrolled loads made this compiler drop the register
  // copies of the last windowed read of a tile (3-field kernel, seen in the ISA), and the loop is leaner without it.
  auto run = [&](auto wtag) {
    constexpr bool kWindowed = decltype(wtag)::value;
    auto step = [&](int t, Stage& cur) {
      // ---- values of tile t ---------------------------------------------------------------------------------
      float val[IT][STRIDE];
      int ci[IT];
      float cw[IT];
      decode(cur, ci, cw);
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int pos = ci[it] < nd_last ? ci[it] : nd_last;
        if constexpr (kWindowed) {
          if constexpr (ABLATE >= 3) {
            val[it][0] = __builtin_bit_cast(float, pos);
          } else if constexpr (STRIDE == 1) {
            val[it][0] = window[pos];
          } else if constexpr (STRIDE == 2) {
            const f32x2 q = reinterpret_cast<const f32x2*>(window)[pos];
            val[it][0] = q.x; val[it][1] = q.y;
          } else if constexpr (NF == 3) {
            val[it][0] = window[pos * 3]; val[it][1] = window[pos * 3 + 1]; val[it][2] = window[pos * 3 + 2];
            val[it][3] = 0.0f;   // the padding slot is never looked at
          } else {
#pragma unroll
            for (int s = 0; s < STRIDE; s += 4) {
              const f32x4 q = reinterpret_cast<const f32x4*>(window)[(size_t)pos * (STRIDE / 4) + s / 4];
              val[it][s] = q.x; val[it][s + 1] = q.y; val[it][s + 2] = q.z; val[it][s + 3] = q.w;
            }
          }
        } else {
          const unsigned g0 = (unsigned)cdict[pos];
          const unsigned gc = g0 < last_gate ? g0 : last_gate;
          rg::load_packed<STRIDE>(packed, gc, val[it]);
        }
      }
      // ---- products of tile t -> LDS (layout and arithmetic: rg_row_phase.hpp) ----------------------------------
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        if constexpr (ABLATE >= 2) rowacc[lane * NF].x += val[it][0] * cw[it];
        else rg::store_products<NF, STRIDE>(tile, TILE, eidx(it), cw[it], val[it]);
      }
      stream(cur, t + 2 * TILE);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- dynamic row phase (shared with rg_csr_apply_f32: same lane split, same float32 adds) ----------------
      if constexpr (ABLATE == 0) rg::row_phase<NF, STRIDE, TILE>(tile, rowacc, t, rs_o, re_o, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    for (int t = 0; t < span;) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        step(t, st[u]);
        t += TILE;
        if (t >= span) break;
      }
    }
  };
  if (span > 0) {
    if (windowed) run(std::true_type{}); else run(std::false_type{});
  }

  if (lane < nrows) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const f32x2 s = rowacc[lane * NF + f];
      out[(size_t)f * n_vox + r0 + lane] = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
    }
  }
}


template <typename IndT, int NF, int TILE>
constexpr size_t static_lds() {
  return (size_t)kH * (TILE * rg::tile_floats(NF, stride_for(NF)) * 4 + 64 * NF * 8);
}

struct PackedStream {   // the packed form of positions + weights (rg_csr_compact_pack); null = the plain arrays
  const rg_u32x4* rec = nullptr;
  const int64_t* rec_ptr = nullptr;
  unsigned w_base = 0;
  int order = RG_REC_ORDER_SEGMENT;
};

template <typename IndT, int NF, int TILE, int ABLATE = 0, int AUX = 0, bool PACKED = false>
int launch_nf(int window_cap, const void* indptr, const uint16_t* lidx, const float* wts, const int64_t* dict_ptr,
              const int32_t* dict, const ChunkGrid& cg, long n_vox, const float* packed, long n_gates, float fill,
              float* out, hipStream_t s, PackedStream ps = PackedStream()) {
  constexpr int STRIDE = stride_for(NF);
  // static + dynamic LDS of one workgroup stay within the 64 KiB a launch gets without opting in to more; a window
  // smaller than the geometry asked for only sends more chunks down the per-pair path (same results)
  constexpr int WS = NF == 3 ? 3 : STRIDE;   // floats per window entry (see the kernel)
  const long room = (65536 - (long)static_lds<IndT, NF, TILE>() - 256) / (4 * WS);
  if (window_cap > room) window_cap = (int)(room < 0 ? 0 : room);
  hipLaunchKernelGGL((csr_compact_kernel<IndT, NF, STRIDE, TILE, ABLATE, AUX, PACKED>), dim3((unsigned)chunk_count(cg)),
                     dim3(64 * kH), ((size_t)window_cap * WS * sizeof(float) + 15) / 16 * 16, s,
                     static_cast<const IndT*>(indptr), lidx, wts, dict_ptr, dict, cg, packed, (unsigned)(n_gates - 1), fill,
                     window_cap, n_vox, out, ps.rec, ps.rec_ptr, ps.w_base, ps.order);
  return rg::check_launch("rg_csr_compact_apply_f32");
}

// Tiles: the defaults MUST be the tiles rg_csr_apply_f32 uses for the same field count (same partial sums).
template <typename IndT>
int launch(int nf, int tile, int window_cap, const void* indptr, const uint16_t* lidx, const float* wts,
           const int64_t* dict_ptr, const int32_t* dict, const ChunkGrid& cg, long n_vox, const float* packed, long n_gates,
           float fill, float* out, hipStream_t s) {
#define RG_K1C(NF_, TILE_) \
  launch_nf<IndT, NF_, TILE_>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s)
  switch (nf) {
    case 1:
      switch (tile) {
#ifdef RG_EXPERIMENTS   // timing-only ablations (tools/exp_nf1.py): results wrong by construction, never in the product library
        case 901: return launch_nf<IndT, 1, 384, 1>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 902: return launch_nf<IndT, 1, 384, 2>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 903: return launch_nf<IndT, 1, 384, 3>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 909: return launch_nf<IndT, 1, 384, 4>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 904: return launch_nf<IndT, 1, 384, 3, 2>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 905: return launch_nf<IndT, 1, 384, 0, 2>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 906: return launch_nf<IndT, 1, 384, 0, 1>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 907: return launch_nf<IndT, 1, 384, 0, 3>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
        case 908: return launch_nf<IndT, 1, 512, 3>(window_cap, indptr, lidx, wts, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s);
#endif
        case 128: return RG_K1C(1, 128);
        case 256: return RG_K1C(1, 256);
        case 512: return RG_K1C(1, 512);
        default: return RG_K1C(1, 384);
      }
    case 2:
      switch (tile) {
        case 128: return RG_K1C(2, 128);
        case 256: return RG_K1C(2, 256);
        case 512: return RG_K1C(2, 512);
        default: return RG_K1C(2, 384);
      }
    case 3:
      switch (tile) {
        case 128: return RG_K1C(3, 128);
        case 192: return RG_K1C(3, 192);
        case 256: return RG_K1C(3, 256);
        case 320: return RG_K1C(3, 320);
        default: return RG_K1C(3, 384);
      }
    case 4:
      switch (tile) {
        case 128: return RG_K1C(4, 128);
        case 256: return RG_K1C(4, 256);
        case 320: return RG_K1C(4, 320);
        default: return RG_K1C(4, 384);
      }
    case 5: return RG_K1C(5, 128);
    case 6: return RG_K1C(6, 128);
    case 7: return RG_K1C(7, 128);
    default: return RG_K1C(8, 128);
  }
#undef RG_K1C
}

}  // namespace

extern "C" int64_t rg_csr_compact_chunks(int64_t n_rows, int64_t line_len, int64_t lines_per_plane) {
  ChunkGrid cg;
  if (n_rows < 0 || !make_chunk_grid(n_rows, line_len, lines_per_plane, &cg)) return RG_EINVAL;
  return n_rows == 0 ? 0 : chunk_count(cg);
}

extern "C" int rg_csr_compact_apply_f32(const void* indptr, int32_t indptr_is_i64, const uint16_t* local_idx,
                                        const float* weights, const int64_t* dict_ptr, const int32_t* dict,
                                        int64_t n_vox, int64_t n_pairs, int64_t line_len, int64_t lines_per_plane,
                                        const float* packed, int32_t n_fields, int32_t stride, int64_t n_gates,
                                        float fill_value, float* out, int32_t window_cap, int32_t tile,
                                        rg_stream_t stream) {
  RG_REQUIRE(indptr && out && dict_ptr, RG_EINVAL, "rg_csr_compact_apply_f32: null indptr/dict_ptr/out");
  RG_REQUIRE(n_vox >= 0 && n_pairs >= 0, RG_EINVAL, "rg_csr_compact_apply_f32: negative size");
  RG_REQUIRE(n_fields >= 1 && n_fields <= RG_MAX_FIELDS, RG_EUNSUPPORTED,
             "rg_csr_compact_apply_f32: n_fields=%d not in 1..%d", n_fields, RG_MAX_FIELDS);
  RG_REQUIRE(stride == stride_for(n_fields), RG_EINVAL, "rg_csr_compact_apply_f32: stride=%d, expected %d for %d fields",
             stride, stride_for(n_fields), n_fields);
  RG_REQUIRE(n_pairs == 0 || (local_idx && weights && dict && packed && n_gates > 0), RG_EINVAL,
             "rg_csr_compact_apply_f32: pairs present but local_idx/weights/dict/packed/n_gates missing");
  RG_REQUIRE(n_gates <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_f32: n_gates exceeds int32 gate indices");
  RG_REQUIRE(n_vox <= 0x3FFFFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_f32: n_vox too large for one launch");
#ifdef RG_EXPERIMENTS
  const int32_t rot_override = tile / 1000;   // diagnostic: tile = 1000 * rotation + tile selects the block rotation
  tile %= 1000;
  const bool ablation = tile >= 901 && tile <= 909;
#else
  const int32_t rot_override = 0;
  const bool ablation = false;
#endif
  RG_REQUIRE(tile == 0 || tile == 128 || tile == 192 || tile == 256 || tile == 320 || tile == 384 || tile == 512 || ablation,
             RG_EINVAL, "rg_csr_compact_apply_f32: tile must be 0 (default), 128, 192, 256, 320, 384 or 512");
  RG_REQUIRE(window_cap >= 0 && window_cap <= RG_COMPACT_MAX_WINDOW, RG_EINVAL,
             "rg_csr_compact_apply_f32: window_cap %d outside 0..%d", window_cap, RG_COMPACT_MAX_WINDOW);
  RG_REQUIRE(rg::aligned16(packed), RG_EALIGN, "rg_csr_compact_apply_f32: packed must be 16-byte aligned");
  if (n_vox == 0) return RG_OK;
  ChunkGrid cg;
  RG_REQUIRE(make_chunk_grid(n_vox, line_len, lines_per_plane, &cg), RG_EINVAL,
             "rg_csr_compact_apply_f32: n_vox=%ld is not planes x lines_per_plane=%ld x line_len=%ld", (long)n_vox,
             (long)lines_per_plane, (long)line_len);
  RG_REQUIRE(chunk_count(cg) <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_f32: too many chunks for one launch");
  if (rot_override > 0) cg.rot_step = (unsigned)rot_override;
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    return launch<int64_t>(n_fields, tile, window_cap, indptr, local_idx, weights, dict_ptr, dict, cg, n_vox, packed,
                           n_gates, fill_value, out, s);
  return launch<int32_t>(n_fields, tile, window_cap, indptr, local_idx, weights, dict_ptr, dict, cg, n_vox, packed, n_gates,
                         fill_value, out, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Packed pair stream: positions AND weights of three consecutive pairs of a segment in one 16-byte record.
//   weight code = float32 bits of the weight minus w_base (= smallest exponent among the geometry's weights << 23); it
//   must fit 26 bits, i.e. all weights positive and within 8 binades -- Barnes weights span exp(-4)+1e-5 .. 1+1e-5, 7
//   binades; the host checks and falls back to the plain arrays otherwise.  Lossless: the kernel adds w_base back.
//   record = [w0:26 | p2 bits 0-5] [w1:26 | p2 bits 6-11] [w2:26 | p2 bits 12-15] [p0:16 | p1:16]
//   Every segment (one wavefront's rows) starts a new record; rec_ptr[seg] = its first record, segments numbered
//   line-major (line * ceil(line_len / 64) + sx).  5.33 bytes per pair instead of 6, and the kernel streams them with one
//   dwordx4 per lane and 192 pairs instead of six 2- and 4-byte loads.
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <typename IndT>
__global__ __launch_bounds__(256) void compact_pack_kernel(const IndT* __restrict__ indptr,
                                                            const uint16_t* __restrict__ lidx,
                                                            const float* __restrict__ wts, ChunkGrid cg, long n_slots,
                                                            int rec_order, const int64_t* __restrict__ rec_ptr,
                                                            unsigned w_base, rg_u32x4* __restrict__ rec,
                                                            int32_t* __restrict__ error_flag) {
  const int lane = threadIdx.x & 63;
  const long slot = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= n_slots) return;
  long r0;
  int nrows;
  if (rec_order == RG_REC_ORDER_DISPATCH) {      // slot = block * H + wavefront: the segment that wavefront reads
    const unsigned bid = (unsigned)(slot / kH);
    const Segment sg = chunk_segment(cg, block_chunk(cg, bid), (int)(slot - (long)bid * kH));
    r0 = sg.r0;
    nrows = sg.nrows;
  } else {                                       // slot = segment number, line-major
    const long line = slot / cg.nsx;
    const unsigned sx = (unsigned)(slot - line * cg.nsx);
    const unsigned x0 = sx * cg.seg_base + (sx < cg.seg_extra ? sx : cg.seg_extra);
    r0 = line * cg.line_len + x0;
    nrows = (int)(cg.seg_base + (sx < cg.seg_extra ? 1u : 0u));
  }
  const long p0 = nrows ? (long)indptr[r0] : 0, p1 = nrows ? (long)indptr[r0 + nrows] : 0;
  const long rb = rec_ptr[slot], rn = rec_ptr[slot + 1] - rb;
  if (lane == 0 && rn != (p1 - p0 + 2) / 3) atomicOr(error_flag, 1);
  if (lane == 0 && rn >= (1L << 27)) atomicOr(error_flag, 4);   // the apply kernels use 32-bit byte offsets per segment
  if (rn != (p1 - p0 + 2) / 3) return;           // never write outside the records rec_ptr gives this segment
  for (long r = lane; r < rn; r += 64) {
    unsigned code[3], pos[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const long p = p0 + 3 * r + j;
      code[j] = 0;
      pos[j] = 0;
      if (p < p1) {
        const unsigned bits = rg::f32_bits(wts[p]);
        code[j] = bits - w_base;
        if (bits < w_base || code[j] > 0x3FFFFFFu) atomicOr(error_flag, 2);   // not codable: the host checked, so never
        pos[j] = lidx[p];
      }
    }
    rg_u32x4 q;
    q.x = (code[0] & 0x3FFFFFFu) | ((pos[2] & 0x3Fu) << 26);
    q.y = (code[1] & 0x3FFFFFFu) | (((pos[2] >> 6) & 0x3Fu) << 26);
    q.z = (code[2] & 0x3FFFFFFu) | (((pos[2] >> 12) & 0xFu) << 26);
    q.w = pos[0] | (pos[1] << 16);
    rec[rb + r] = q;
  }
}

}  // namespace

extern "C" int rg_csr_compact_pack(const void* indptr, int32_t indptr_is_i64, const uint16_t* local_idx,
                                   const float* weights, int64_t n_rows, int64_t line_len, int64_t lines_per_plane,
                                   const int64_t* rec_ptr, int32_t rec_order, int64_t plane0, uint32_t w_base,
                                   void* records, int32_t* error_flag, rg_stream_t stream) {
  RG_REQUIRE(n_rows >= 0 && plane0 >= 0, RG_EINVAL, "rg_csr_compact_pack: negative size");
  RG_REQUIRE(rec_order == RG_REC_ORDER_SEGMENT || rec_order == RG_REC_ORDER_DISPATCH, RG_EINVAL,
             "rg_csr_compact_pack: rec_order=%d is neither RG_REC_ORDER_SEGMENT nor RG_REC_ORDER_DISPATCH", rec_order);
  if (n_rows == 0) return RG_OK;
  RG_REQUIRE(indptr && rec_ptr && error_flag, RG_EINVAL, "rg_csr_compact_pack: null pointer");
  RG_REQUIRE(rg::aligned16(records), RG_EALIGN, "rg_csr_compact_pack: records must be 16-byte aligned");
  ChunkGrid cg;
  RG_REQUIRE(make_chunk_grid(n_rows, line_len, lines_per_plane, &cg), RG_EINVAL,
             "rg_csr_compact_pack: n_rows=%ld is not planes x lines_per_plane=%ld x line_len=%ld", (long)n_rows,
             (long)lines_per_plane, (long)line_len);
  RG_REQUIRE(chunk_count(cg) <= 0x7FFFFFFFL / kH, RG_EUNSUPPORTED, "rg_csr_compact_pack: too many chunks for one launch");
  cg.grp0 = (unsigned)(((unsigned long)plane0 * cg.nyg) & 0xFFFFFFFFul);   // the rotation counts line groups mod 2^32
  const long n_slots = rec_order == RG_REC_ORDER_DISPATCH ? chunk_count(cg) * kH
                                                          : cg.n_planes * cg.lines_per_plane * (long)cg.nsx;
  const long blocks = (n_slots + 3) / 4;
  RG_REQUIRE(blocks <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_pack: too many segments for one launch");
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    hipLaunchKernelGGL(compact_pack_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                       static_cast<const int64_t*>(indptr), local_idx, weights, cg, n_slots, rec_order, rec_ptr, w_base,
                       static_cast<rg_u32x4*>(records), error_flag);
  else
    hipLaunchKernelGGL(compact_pack_kernel<int32_t>, dim3((unsigned)blocks), dim3(256), 0, s,
                       static_cast<const int32_t*>(indptr), local_idx, weights, cg, n_slots, rec_order, rec_ptr, w_base,
                       static_cast<rg_u32x4*>(records), error_flag);
  return rg::check_launch("rg_csr_compact_pack");
}

// ---------------------------------------------------------------------------------------------------------------
// Row-wise kernel over the packed stream (1-4 fields): the default of rg_csr_compact_apply_packed_f32.
// The tile kernel above moves every pair through LDS twice (the product side writes the tile, the row side reads it
// back): 8 bytes per pair each way for one field, 16 + 16 for three on top of the window gather, and from two fields on
// it is LDS- and latency-bound, not HBM-bound.  A packed record already holds three CONSECUTIVE
// pairs, so here the lanes of a row read the row's records straight from memory -- L = 2^k lanes per row, lane j takes
// records q0 + j, q0 + j + L, ... of the row's record range [rs / 3, ceil(re / 3)) -- and reduce them in registers: no
// tile, no transposition; LDS carries only the window gathers.  64 / L rows share a wave-load (L * 16 contiguous bytes
// each, neighbouring rows adjacent in memory); a record that straddles two rows is read by both (an L1 hit) and each
// takes its own pairs.  Pairs outside the lane's row are redirected to a sentinel window entry whose slots are all
// EXCLUDED, so the arithmetic needs no extra test.
// A STEP is one batch of KPRE record loads per lane; the loads of the next step (of the same rows, or of the next
// 64 / L rows) are always requested before the current step is summed -- two register stages, as in the tile kernel.
// Summation order (fixed by the geometry and the field count alone, so results are reproducible run to run, on any
// window size and on the per-pair path of an over-wide chunk): a lane's t-th record of a row (t = 0, 1, ...) sits in
// batch slot t mod KPRE and belongs to chain (t mod KPRE) mod 2; per lane and chain, the chain's records in ascending
// order and a record's pairs in order, one running (sum w*v, sum w) per field; chain 0 + chain 1; then the xor butterfly
// of rg_row_phase.hpp over the L lanes.
// L is chosen per segment from its mean row length (kTarget records per lane and row).  This is NOT the order of
// rg_csr_apply_f32: the two agree to float32 rounding, not bit for bit (the tile kernel over the same records, tile =
// 384, does).
// Per field count (measured on config 2 and the bench grid, profiles/r02_rowwise_sweep.json; three fields re-tuned in
// round 3 after the instruction diet of the loop: 3 records per step instead of 2, -2 %, profiles/r03_cfg3_sweep.json):
//   KPRE   records per lane and step;   kTarget  records per lane and row L aims for;
//   kNarrow  12-byte window entries for three fields (three 4-byte LDS reads per pair instead of one 16-byte read,
//            but a quarter less LDS per workgroup);
//   kRegs    the row sums travel to lane == row by shuffle and wait in registers instead of an LDS array;
//   one field: the window holds (value, 1) per gate, (0, 0) where it is excluded, and a pair contributes w * (v', m) --
//            the same float32 values as selecting on the sentinel, in packed multiply / add instructions.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kRowwiseChunksPerBlock = 1;   // consecutive chunks one workgroup takes (see the kernel: 1 measured best)

// DIAG (timing-only diagnostics of tools/exp_placement4.py, results wrong by construction; one field only): bit 0 = the
// window is not gathered (no dictionary / field reads), bit 1 = no output store, bit 2 / bit 3 = cache policy sc0 / nt on
// the record loads, bit 4 = no record loads at all (the stream is replaced by a constant)
// Workgroups per CU the compiler must leave room for (= wavefronts per SIMD: a workgroup is one wavefront per SIMD), per
// field count; 1 = no constraint.  -DRG_ROWWISE_WAVES1=.. / 3=..: A/B builds (tools/gpu_r03_ab_slots.sh occ).
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_WAVES1)
#undef RG_ROWWISE_WAVES1
#define RG_ROWWISE_WAVES1 1
#endif
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_WAVES3)
#undef RG_ROWWISE_WAVES3
#define RG_ROWWISE_WAVES3 1
#endif
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_WAVES8)
#undef RG_ROWWISE_WAVES8
#define RG_ROWWISE_WAVES8 1
#endif
#define RG_ROWWISE_BOUNDS __launch_bounds__(64 * kH, (NF == 1 ? RG_ROWWISE_WAVES1 : NF == 3 ? RG_ROWWISE_WAVES3 : NF >= 5 ? RG_ROWWISE_WAVES8 : 1))
// COLS (rg_csr_compact_apply_columns_f32, csrc/rg_csr_columns.hip): the chunks a workgroup takes one after the other are
// not consecutive blocks of the dispatch order but the LEVELS of one column of chunks -- the same (line group, segment)
// patch from plane z0 to z1 - 1 of its level piece -- so that lane == row sees the voxels of its (y, x) column in ascending
// level order and can keep the column maximum / first argmax in registers and store selected levels as planes; `out`
// may then be null (products only: the 3-D grid is never written).  Everything between a chunk's row pointers and its
// row sums is the same code: the same bits.
// REGS: where the row sums wait for lane == row -- -1 = the field count's default (RowwiseConfig<NF>::regs), 0 = the LDS
// array, 1 = registers.  Four fields: registers cost 99 VGPRs (4 wavefronts per SIMD), the LDS array 95 (5 wavefronts) and
// 8 KiB of LDS per workgroup -- the launcher picks the array wherever the LDS still admits five workgroups per CU.
template <typename IndT, int NF, int STRIDE, int DIAG = 0, bool COLS = false, int REGS = -1>
__global__ RG_ROWWISE_BOUNDS void csr_compact_rowwise_kernel(
    const IndT* __restrict__ indptr, const int64_t* __restrict__ dict_ptr, const int32_t* __restrict__ dict, ChunkGrid cg,
    const float* __restrict__ packed, unsigned last_gate, float fill, int window_cap, long n_vox, float* __restrict__ out,
    const rg_u32x4* __restrict__ rec, const int64_t* __restrict__ rec_ptr, unsigned w_base, int lanes_hint,
    int rec_order, unsigned n_chunks, int chunks_per_block, const RowwiseColumns cols) {
  static_assert(NF >= 1 && NF <= 8 && STRIDE == stride_for(NF), "passes of 1-8 fields");
  using Cfg = RowwiseConfig<NF>;
  constexpr int KPRE = Cfg::kpre;
  // Record prefetch (experiment builds only; measured SLOWER, EXPERIMENTS.md R4.11): touch loads -- one dword per 64 bytes, never
  // read -- of the segment's first kPrefetchHead bytes before the window fill and, when round rho begins, of the records of round
  // rho + kPrefetch - 1, so that the real loads would hit in L2.
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_PREFETCH)
  constexpr int kPrefetch = NF >= RG_ROWWISE_PREFETCH_MIN_NF ? RG_ROWWISE_PREFETCH : 0;
  constexpr int kPrefetchHead = RG_ROWWISE_PREFETCH_HEAD;
#else
  constexpr int kPrefetch = 0;
  constexpr int kPrefetchHead = 0;
#endif
  constexpr bool kByteMask = rowwise_bytemask<NF>();       // window entries = (v' ..., byte mask): rg_compact_layout.hpp
  constexpr bool kNarrow = Cfg::narrow && !kByteMask, kRegs = REGS < 0 ? Cfg::regs : REGS != 0;
  // Five fields and more (never the column mode): a row's sums do not travel to lane == row and wait there (2 * NF registers
  // for the whole segment) -- when a round ends the L lanes of a row, which all hold all its sums after the butterfly, SHARE
  // the fields: lane `sub` divides fields sub, sub + L, ... (a select tree over the bits of sub picks them) and parks the VALUES
  // in 2 KiB of LDS per wavefront; lane == row stores them as whole row runs when the segment ends (storing per round wrote 16-32-row
  // pieces: 1.37x the grid's bytes in partial lines, +1.4 %).  The same sums, the same division: the same bits.
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_SCATTER_MIN_NF)
  constexpr bool kScatter = NF >= RG_ROWWISE_SCATTER_MIN_NF && !COLS;
  constexpr int kFenceMinNF = RG_ROWWISE_FENCE_MIN_NF;
#else
  constexpr bool kScatter = NF >= 5 && !COLS;        // three / four fields: measured slower (stores of 16 rows x 4 fields)
  constexpr int kFenceMinNF = 3;
#endif
  // one field: the window holds (v', m) = (value, 1) of a gate, (0, 0) where it is excluded, so that a pair contributes
  // w * (v', m) -- the same float32 values as selecting on the EXCLUDED sentinel (w * 0 = +0, w * 1 = w) in two packed
  // instructions instead of a compare, two selects, a product and two adds
  constexpr bool kPremask = NF == 1;
  extern __shared__ __attribute__((aligned(16))) float window[];   // window_cap + 1 entries of rowwise_entry_words<NF>() words
  // byte masks of four fields and more: the mask words of the window_cap + 1 entries lie behind their value entries
  constexpr int kVW = rowwise_value_words<NF>(), kMW = rowwise_mask_words<NF>();
  // the mask byte of a usable field: 1 (read back with v_cvt_f32_ubyteN, one per field) or the fp8 (OCP e4m3) code of 1.0 (read
  // back two fields at a time with v_cvt_pk_f32_fp8)
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_MASK_UBYTE)
  constexpr bool kMaskFp8 = false;
#else
  constexpr bool kMaskFp8 = true;
#endif
  constexpr unsigned kMaskOne = kMaskFp8 ? 0x38u : 1u;
  (void)kMW;
  unsigned* const maskw = reinterpret_cast<unsigned*>(window + (size_t)(window_cap + 1) * kVW);
  __shared__ f32x2 rowacc_all[kRegs ? 1 : kH][kRegs ? 2 : 64 * NF];
  // kScatter with kStage: the finished values of a segment wait in LDS ([row][8 fields], 2 KiB per wavefront) so that lane == row
  // stores whole 248-byte row runs per field at the segment's end instead of 16-32-row pieces per round
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_NO_STAGE)
  constexpr bool kStage = false;
#else
  constexpr bool kStage = kScatter;
#endif
  __shared__ __attribute__((aligned(16))) float stage_all[kStage ? kH : 1][kStage ? 64 * 8 : 4];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x2* rowacc = rowacc_all[kRegs ? 0 : wv];
  float* const stage = stage_all[kStage ? wv : 0];
  // 26-bit weight mask in a VGPR the compiler cannot fold: (x & mask) | w_base is then ONE v_and_or_b32 (this ISA's VOP3
  // takes no literal, and a literal mask splits it into v_and + v_or)
  unsigned wmask = 0x3FFFFFFu;
  asm volatile("" : "+v"(wmask));

  // A workgroup takes chunks_per_block CONSECUTIVE blocks of the dispatch order, one after the other (default: 1).
  // Round-3 experiment: the kernel's only store costs 5-15 % of the launch (tools/exp_placement4.py: 7.4 ms without it,
  // 7.8-8.7 with it, depending on where records and grid lie in memory), and the first suspect was its acknowledgement
  // at the end of every workgroup's life.  Letting it overlap the next chunk's work changed nothing (2 / 4 / 8 / 32
  // chunks per workgroup: +0.1 ... +0.3 ms, the extra barrier): the cost is the memory system's, a trickle of writes
  // among the reads (tools/exp_placement5.py reproduces it with a bare read probe).  The loop stays as the knob it is.
  // COLS: this workgroup's column piece (item = piece * columns + column; columns rotated per line group like the blocks of
  // the dispatch order, so that consecutive workgroups -- consecutive XCDs -- do not pin a column of the grid to one XCD)
  unsigned col_yg = 0, col_sx = 0, col_piece = 0;
  int col_z0 = 0;
  ColumnBest best[COLS ? NF : 1];
  if constexpr (COLS) {
    const unsigned item = cols.order ? (unsigned)cols.order[blockIdx.x] : blockIdx.x;
    col_piece = item / cols.n_cols;
    const unsigned q = item - col_piece * cols.n_cols;
    col_yg = q / cg.nsx;
    const unsigned c = q - col_yg * cg.nsx;
    col_sx = c + (col_yg * cg.rot_step) % cg.nsx;
    col_sx = col_sx >= cg.nsx ? col_sx - cg.nsx : col_sx;
    col_z0 = (int)((long)col_piece * cg.n_planes / cols.pieces);
    chunks_per_block = (int)((long)(col_piece + 1) * cg.n_planes / cols.pieces) - col_z0;
#pragma unroll
    for (int f = 0; f < NF; ++f) { best[f].v = __builtin_nanf(""); best[f].idx = -1; }
  }
  long col_xy = 0;                                    // COLS: (y, x) of lane == row, the same on every level
  int col_nrows = 0;
  for (int cb = 0; cb < chunks_per_block; ++cb) {
  unsigned bid, chunk;
  if constexpr (COLS) {
    const unsigned grp = (unsigned)(col_z0 + cb) * cg.nyg + col_yg;
    chunk = grp * cg.nsx + col_sx;
    const unsigned shift = ((grp + cg.grp0) * cg.rot_step) % cg.nsx;      // the block whose rotated column is col_sx
    bid = grp * cg.nsx + (col_sx >= shift ? col_sx - shift : col_sx + cg.nsx - shift);
  } else {
    bid = blockIdx.x * (unsigned)chunks_per_block + (unsigned)cb;
    if (bid >= n_chunks) break;                       // workgroup-uniform
    chunk = block_chunk(cg, bid);
  }
  if (cb > 0) __syncthreads();                        // every wavefront is done with the previous chunk's window
  float mine_p[NF], mine_w[NF];                       // kRegs: lane == row
#pragma unroll
  for (int f = 0; f < NF; ++f) mine_p[f] = mine_w[f] = 0.0f;

  const long d0 = dict_ptr[chunk];
  const int nd_all = (int)(dict_ptr[chunk + 1] - d0);
  const bool split = nd_all > 65536;
  const bool windowed = nd_all <= window_cap;        // the window holds window_cap + 1 entries: the sentinel
  const int w_lo = split ? dict[d0 + wv] : 0;
  const int w_hi = split ? (wv + 1 < kH ? dict[d0 + wv + 1] : nd_all) : nd_all;
  const int nd = w_hi - w_lo;
  const int nd_last = nd > 0 ? nd - 1 : 0;
  const int32_t* __restrict__ cdict = dict + d0 + w_lo;

  const Segment sg = chunk_segment(cg, chunk, wv);
  const int nrows = sg.nrows;
  const long r0 = sg.r0;
  const long seg_b = nrows ? (long)indptr[r0] : 0;
  const long seg_e = nrows ? (long)indptr[r0 + nrows] : 0;
  const int span = (int)(seg_e - seg_b);
  const int rs_o = nrows ? (int)((long)indptr[r0 + (lane < nrows ? lane : nrows)] - seg_b) : 0;
  const int re_o = nrows ? (int)((long)indptr[r0 + (lane + 1 < nrows ? lane + 1 : nrows)] - seg_b) : 0;
  long rec_b = 0, rec_n = 0;
  if (nrows) {
    // dispatch order: the H segments of a workgroup's chunk are neighbours in the stream, and so are consecutive blocks
    const long slot = rec_order == RG_REC_ORDER_DISPATCH ? (long)bid * kH + wv : sg.seg;
    rec_b = rec_ptr[slot];
    rec_n = rec_ptr[slot + 1] - rec_b;
  }
  // DIAG & 2048 (timing-only): every segment reads its records 1/7 closer to the array's start, so that neighbours' streams
  // overlap by a seventh -- the same loads, 14 % fewer distinct bytes from HBM (what a 4.57-byte-per-pair record would stream)
  const rsrc_t rr = make_rsrc(rec + ((DIAG & 2048) ? rec_b * 6 / 7 : rec_b), rec_n * 16);
  constexpr int kOutOfRange = 0x7FFFFFF0;            // byte offset no segment reaches: the load returns zeros

  // ---- lanes per row ------------------------------------------------------------------------------------------
  int lgl;
  if (lanes_hint > 0 && lanes_hint <= 64) {
    lgl = 31 - __builtin_clz(lanes_hint);
  } else {
    const int target = lanes_hint > 70 ? lanes_hint - 70 : Cfg::target;   // records per lane and row to aim for
    const int mean_rec = nrows ? span / (3 * nrows) + 1 : 1;      // records a row touches, about
    const int need = (mean_rec + target - 1) / target;
    lgl = need <= 1 ? 0 : 32 - __builtin_clz(need - 1);
  }
  lgl = __builtin_amdgcn_readfirstlane(lgl > 6 ? 6 : lgl);
  const int nl = 1 << lgl, rpr = 64 >> lgl;          // lanes per row, rows per round
  const int sub = lane & (nl - 1), rgrp = lane >> lgl;
  const int rounds = (nrows + rpr - 1) >> (6 - lgl);
  // trips of a round = the most records any of its rows gives one lane; lane == row here, groups of rpr rows
  const unsigned q0_row = (unsigned)rs_o / 3u;
  const unsigned q1_row = re_o > rs_o ? ((unsigned)re_o + 2u) / 3u : q0_row;
  int trips_row = (int)((q1_row - q0_row + (unsigned)nl - 1u) >> lgl);
  for (int m = 1; m < rpr; m <<= 1) {
    const int o = __shfl_xor(trips_row, m, 64);
    trips_row = o > trips_row ? o : trips_row;
  }

  // ---- record prefetch (see run(): kPrefetch) -- the segment's first records are touched before the window fill -------
  unsigned pf_d0 = 0, pf_d1 = 0;
  if constexpr (kPrefetch > 0) {
    if (span > 0) {
      pf_d0 = rg_buffer_load_u32(rr, lane * 64, 0, 0);              // out-of-range pieces return 0 without a memory access
      if constexpr (kPrefetchHead > 4096) pf_d1 = rg_buffer_load_u32(rr, 4096 + lane * 64, 0, 0);
    }
  }

  // ---- the chunk's field window + the sentinel entry ----------------------------------------------------------
  // kFillBatch entries per thread at a time: their dictionary reads are issued back to back, then their field gathers, then
  // the LDS stores -- two memory latencies per batch.  (Round 2 walked the entries one by one: dictionary read, wait, gather,
  // wait, store -- 2 x 6 serialized latencies in front of the barrier on config 2's 1500-entry dictionaries; other
  // workgroups of the CU cover most of that, the batches are worth 2-5 % there.)  All loads are unconditional on clamped
  // indices so that nothing splits the batch.  Also tried around this prologue in round 3, both slower: issuing the first
  // step's record loads in front of the fill and holding them across it (+22 VGPRs, a wavefront of occupancy: 7-17 %
  // slower), and throw-away loads of the same addresses to warm the L2 meanwhile (+1.4-2.8 %).
  if (windowed) {
#if !defined(RG_EXPERIMENTS) || !defined(RG_FILL_BATCH)
#undef RG_FILL_BATCH
#define RG_FILL_BATCH 4                          // 1 / 2 / 8 measured: +2 % / +0.3 % / +0.5 % on config 2 (A/B builds)
#endif
    constexpr int kFillBatch = RG_FILL_BATCH;
    const int last_entry = nd_all > 0 ? nd_all - 1 : 0;
    const int32_t* __restrict__ cd = nd_all > 0 ? cdict : (const int32_t*)dict_ptr;   // never dereference an empty dictionary
    for (int i0 = threadIdx.x; i0 <= nd_all; i0 += 64 * kH * kFillBatch) {
      unsigned gate[kFillBatch];
#pragma unroll
      for (int u = 0; u < kFillBatch; ++u) {
        const int i = i0 + u * 64 * kH;
        gate[u] = (DIAG & 1) ? 0u : (unsigned)cd[i < last_entry ? i : last_entry];
      }
      float v[kFillBatch][STRIDE];
#pragma unroll
      for (int u = 0; u < kFillBatch; ++u) {
        if constexpr (DIAG & 1) {
#pragma unroll
          for (int s = 0; s < STRIDE; ++s) v[u][s] = 1.0f;
        } else {
          rg::load_packed<STRIDE>(packed, gate[u] < last_gate ? gate[u] : last_gate, v[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < kFillBatch; ++u) {
        // branch-free: a thread past the end stores the sentinel into the sentinel's entry once more (same bits from every
        // such thread), so that nothing conditional makes the compiler sink one of the batch's loads behind a wait
        const int i_raw = i0 + u * 64 * kH;
        const int i = i_raw < nd_all ? i_raw : nd_all;
        if (i_raw >= nd_all) {                   // the sentinel entry: every slot EXCLUDED
#pragma unroll
          for (int s = 0; s < STRIDE; ++s) v[u][s] = __builtin_bit_cast(float, RG_EXCLUDED_BITS);
        }
        if constexpr (kByteMask) {
          unsigned m[2] = {0u, 0u};
          float vv[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const bool good = rg::f32_bits(v[u][f]) != RG_EXCLUDED_BITS;
            vv[f] = good ? v[u][f] : 0.0f;
            m[f >> 2] |= good ? (kMaskOne << (8 * (f & 3))) : 0u;
          }
          if constexpr (NF == 3) {
            reinterpret_cast<f32x4*>(window)[i] = (f32x4){vv[0], vv[1], vv[2], __builtin_bit_cast(float, m[0])};
          } else if constexpr (NF == 4) {
            reinterpret_cast<f32x4*>(window)[i] = (f32x4){vv[0], vv[1], vv[2], vv[3]};
            maskw[i] = m[0];
          } else {
            reinterpret_cast<f32x4*>(window)[2 * i] = (f32x4){vv[0], vv[1], vv[2], vv[3]};
            reinterpret_cast<f32x4*>(window)[2 * i + 1] = (f32x4){vv[4], vv[5], vv[6], vv[7]};
            reinterpret_cast<uint2*>(maskw)[i] = make_uint2(m[0], m[1]);
          }
        } else if constexpr (kNarrow) {
          window[i * 3] = v[u][0]; window[i * 3 + 1] = v[u][1]; window[i * 3 + 2] = v[u][2];
        } else if constexpr (kPremask) {
          const bool good = rg::f32_bits(v[u][0]) != RG_EXCLUDED_BITS;
          reinterpret_cast<f32x2*>(window)[i] = good ? (f32x2){v[u][0], 1.0f} : (f32x2){0.0f, 0.0f};
        } else if constexpr (STRIDE == 1) {
          window[i] = v[u][0];
        } else if constexpr (STRIDE == 2) {
          reinterpret_cast<f32x2*>(window)[i] = (f32x2){v[u][0], v[u][1]};
        } else {
          reinterpret_cast<f32x4*>(window)[i] = (f32x4){v[u][0], v[u][1], v[u][2], v[u][3]};
        }
      }
    }
  }
  __syncthreads();

  // A STEP is one batch of KPRE record loads per lane: batch b of round rho.  A round whose rows need more than KPRE
  // trips simply takes several steps, so every load of the kernel is requested one step ahead whatever the row lengths.
  // Per-step lane state is kept in the form the loop consumes it, so that a record costs one add and one compare to
  // place: batch slot k of a step holds the lane's record number q + k * L.
  struct Step {
    int lo0;           // (first pair of the lane's row) - 3 * q: pair i of slot k is the row's iff 0 <= i - lo < len,
    unsigned len;      //   lo = lo0 - 3 * k * L; len = pairs of the row (0: no row)
    int rem;           // records of the row from q on: slot k holds one iff k * L < rem
    int off0;          // byte offset of record q in the segment's records
    int myrow, rho;
    int left;          // trips of the round still to do, this batch included (wave-uniform)
    bool live;
  };
  auto setup = [&](int rho) -> Step {
    Step r;
    r.rho = rho;
    r.myrow = rho * rpr + rgrp;
    r.live = r.myrow < nrows;
    // both shuffles unconditional: a lane that is dead in this round still has to SUPPLY its row bounds (a shuffle
    // under a divergent condition reads zeros from the lanes that skipped it)
    const int qs = __shfl(rs_o, r.myrow & 63, 64);
    const int qe_row = __shfl(re_o, r.myrow & 63, 64);
    const int qe = r.live ? qe_row : qs;
    const int q0 = (int)((unsigned)qs / 3u);
    const int q1 = qe > qs ? (int)(((unsigned)qe + 2u) / 3u) : q0;
    const int q = q0 + sub;
    r.lo0 = qs - 3 * q;
    r.len = (unsigned)(qe - qs);
    r.rem = q1 - q;
    r.off0 = q * 16;
    r.left = rho < rounds ? __builtin_amdgcn_readfirstlane(__shfl(trips_row, (rho * rpr) & 63, 64)) : 0;
    return r;
  };
  auto advance = [&](const Step& r) -> Step {      // the step after r (wave-uniform choice)
    if (r.left > KPRE) {
      Step n = r;
      n.lo0 -= 3 * (KPRE << lgl);
      n.rem -= KPRE << lgl;
      n.off0 += 16 * (KPRE << lgl);
      n.left -= KPRE;
      return n;
    }
    return setup(r.rho + 1);
  };
  auto issue = [&](const Step& r, rg_u32x4 (&regs)[KPRE]) {
#pragma unroll
    for (int k = 0; k < KPRE; ++k) {
      if constexpr (DIAG & 16) regs[k] = (rg_u32x4){0x3F00000u, 0x3F00000u, 0x3F00000u, (unsigned)lane};
      else regs[k] = rg_buffer_load_v4u32(rr, (k << lgl) < r.rem ? r.off0 + 16 * (k << lgl) : kOutOfRange, 0,
                                          (DIAG & 4) ? 1 : (DIAG & 8) ? 2 : 0);
    }
  };
  auto run = [&](auto wtag) {
    constexpr bool kWindowed = decltype(wtag)::value;
    // The running sums of the lane's row, across the round's steps: TWO chains -- batch slot k of every step adds into
    // chain k mod 2's (sum w*v, sum w) --, added up when the round ends.  Two independent chains half as long as round 2's
    // single one: worst relative error against the reference's ZDR fixtures 8.7e-6 -> 6.5e-6 at no cost (bench grid
    // 8.051 vs 8.050 ms, same process and arrays; <= 1.3 % for 2-4 fields).  One chain per slot (three) reaches 4.3e-6 --
    // exact sums would give 4.2e-6, the reference's own rounding -- but costs a wavefront of occupancy (75 -> 89 VGPRs for
    // one field): +2.1 % on the bench grid, +10 / +21 % for two / four fields (profiles/r03_slots_ab.json).
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_SLOTS)   // A/B builds only: 1 = round 2's one chain per lane
    constexpr int KS = RG_ROWWISE_SLOTS < KPRE ? RG_ROWWISE_SLOTS : KPRE;
#else
    constexpr int KS = 2;
#endif
    // Five fields and more: the weight sums keep ONE chain.  They add positive terms only, so their rounding is a few 1e-8
    // of the sum whatever the order; the products -- where mixed signs cancel and the order shows in the result -- keep two.
    // (Eight fields: 165 VGPRs with two chains each, 3 wavefronts per SIMD; 4 wavefronts need <= 128.)
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_WSLOTS8)
    constexpr int KSW = NF >= 5 ? (RG_ROWWISE_WSLOTS8 < KS ? RG_ROWWISE_WSLOTS8 : KS) : KS;
#else
    constexpr int KSW = NF >= 5 ? 1 : KS;
#endif
    // Byte-mask kernels keep the sums of field PAIRS in 64-bit register pairs (bp / bw: what v_pk_mul / v_pk_add / v_pk_fma
    // take, and what the per-pair asm fences can name without splitting the pairs); the others keep scalars the compiler pairs.
    constexpr int NP2 = (NF + 1) / 2;
    float ap[kByteMask ? 1 : KS][kByteMask ? 1 : NF], aw[kByteMask ? 1 : KSW][kByteMask ? 1 : NF];
    f32x2 bp[kByteMask ? KS : 1][kByteMask ? NP2 : 1], bw[kByteMask ? KSW : 1][kByteMask ? NP2 : 1];
#pragma unroll
    for (int k = 0; k < (kByteMask ? 1 : KS); ++k) {
#pragma unroll
      for (int f = 0; f < (kByteMask ? 1 : NF); ++f) ap[k][f] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < (kByteMask ? 1 : KSW); ++k) {
#pragma unroll
      for (int f = 0; f < (kByteMask ? 1 : NF); ++f) aw[k][f] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < (kByteMask ? KS : 1); ++k) {
#pragma unroll
      for (int j = 0; j < (kByteMask ? NP2 : 1); ++j) bp[k][j] = (f32x2)(0.0f);
    }
#pragma unroll
    for (int k = 0; k < (kByteMask ? KSW : 1); ++k) {
#pragma unroll
      for (int j = 0; j < (kByteMask ? NP2 : 1); ++j) bw[k][j] = (f32x2)(0.0f);
    }
    auto addp = [&](int k, int f, float x) {          // k, f compile-time after unrolling
      if constexpr (kByteMask) bp[k][f >> 1][f & 1] += x; else ap[k][f] += x;
    };
    auto addw = [&](int k, int f, float x) {
      if constexpr (kByteMask) bw[k][f >> 1][f & 1] += x; else aw[k][f] += x;
    };
    auto fence_sums = [&](int kp, int kw) {           // everything added so far is complete; no memory access moves across
      if constexpr (kByteMask) {
#pragma unroll
        for (int j = 0; j < NP2; ++j) asm volatile("" : "+v"(bp[kp][j]), "+v"(bw[kw][j]) : : "memory");
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(ap[kp][f]), "+v"(aw[kw][f]) : : "memory");
      }
    };
    auto consume = [&](const Step& r, const rg_u32x4& q4, int k) {     // k: slot of the step's batch (compile-time)
      // the record's pairs i = 0, 1, 2 belong to the lane's row iff lo <= i < lo + len (len = 0 for a lane without record)
      const int lo = r.lo0 - 3 * (k << lgl);
      const unsigned len = (k << lgl) < r.rem ? r.len : 0u;
      float w[3];
      int pos[3];
      // w_base has its low 26 bits clear (the entry point checks), so code | w_base == code + w_base
      w[0] = __builtin_bit_cast(float, (q4.x & wmask) | w_base);
      w[1] = __builtin_bit_cast(float, (q4.y & wmask) | w_base);
      w[2] = __builtin_bit_cast(float, (q4.z & wmask) | w_base);
      pos[0] = (int)(q4.w & 0xFFFFu);
      pos[1] = (int)(q4.w >> 16);
      unsigned p2b = q4.y >> 26, p2c = q4.z >> 26;
      asm volatile("" : "+v"(p2b), "+v"(p2c));     // keep the shifts apart: each OR then folds into a v_lshl_or_b32
      pos[2] = (int)((p2c << 12) | ((p2b << 6) | (q4.x >> 26)));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const bool mine = (unsigned)(i - lo) < len;
        // windowed: no clamp -- a position is 16 bits by construction, and an LDS read beyond the workgroup's allocation
        // returns zeros instead of faulting (a corrupt record cannot do worse than a wrong value)
        const int p = kWindowed ? pos[i] : (pos[i] < nd_last ? pos[i] : nd_last);
        float v[STRIDE];
        if constexpr (kWindowed) {
          const int e = mine ? p : nd_all;          // not this row's pair: the all-EXCLUDED sentinel entry
          __builtin_assume((unsigned)e <= 65536u);  // lets e * 12 be a 24-bit multiply-add instead of a 64-bit one
          if constexpr (kByteMask) {
            // (v', byte mask): sum w*v' by multiply + add (w * +0 = +0 where excluded, as the select + legacy multiply gave),
            // sum w*g by fma (exact product: the same float32 as adding w or +0)
            float vv[8];
            unsigned m[2] = {0u, 0u};
            if constexpr (NF <= 4) {
              const f32x4 x = reinterpret_cast<const f32x4*>(window)[e];
              vv[0] = x.x; vv[1] = x.y; vv[2] = x.z; vv[3] = x.w;
              m[0] = NF == 3 ? rg::f32_bits(x.w) : maskw[e];
            } else {
              if constexpr (DIAG & 1024) {      // timing-only: no window reads
                const float fe = __builtin_bit_cast(float, 0x3F800000u | (unsigned)e);
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] = fe;
                m[0] = m[1] = 0x38383838u;
              } else {
              const f32x4 x = reinterpret_cast<const f32x4*>(window)[2 * e], y = reinterpret_cast<const f32x4*>(window)[2 * e + 1];
              vv[0] = x.x; vv[1] = x.y; vv[2] = x.z; vv[3] = x.w; vv[4] = y.x; vv[5] = y.y; vv[6] = y.z; vv[7] = y.w;
              const uint2 mm = reinterpret_cast<const uint2*>(maskw)[e];
              m[0] = mm.x; m[1] = mm.y;
              }
            }
            using g2_t = decltype(__builtin_amdgcn_cvt_pk_f32_fp8(0, false));
            f32x2 g2[4];
            if constexpr (kMaskFp8) {      // one conversion per field PAIR: v_cvt_pk_f32_fp8 (OCP e4m3: 0x38 = 1.0, 0x00 = +0)
              const g2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)m[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)m[0], true);
              const g2_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)m[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)m[1], true);
              g2[0] = (f32x2){a[0], a[1]}; g2[1] = (f32x2){b[0], b[1]}; g2[2] = (f32x2){c[0], c[1]}; g2[3] = (f32x2){d[0], d[1]};
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                g2[j] = (f32x2){(float)((m[j >> 1] >> (16 * (j & 1))) & 0xFFu), (float)((m[j >> 1] >> (16 * (j & 1) + 8)) & 0xFFu)};
            }
            const f32x2 w2 = (f32x2){w[i], w[i]};
#pragma unroll
            for (int j = 0; j < NP2; ++j) {
              const f32x2 prod = w2 * (f32x2){vv[2 * j], vv[2 * j + 1]};       // float32 product, then the add (no contraction)
              bp[k % KS][j] += prod;
              bw[k % KSW][j] = __builtin_elementwise_fma(w2, g2[j], bw[k % KSW][j]);
            }
            if constexpr (NF >= kFenceMinNF) fence_sums(k % KS, k % KSW);   // one pair at a time: these sums are complete
            continue;                                                       // before the next pair's window reads are issued
          } else if constexpr (kNarrow) {
            v[0] = window[e * 3]; v[1] = window[e * 3 + 1]; v[2] = window[e * 3 + 2];
          } else if constexpr (kPremask) {
            const f32x2 term = (f32x2){w[i], w[i]} * reinterpret_cast<const f32x2*>(window)[e];
            ap[k % KS][0] += term.x;
            aw[k % KSW][0] += term.y;
            continue;
          } else if constexpr (STRIDE == 1) {
            v[0] = window[e];
          } else if constexpr (STRIDE == 2) {
            const f32x2 x = reinterpret_cast<const f32x2*>(window)[e];
            v[0] = x.x; v[1] = x.y;
          } else {
            const f32x4 x = reinterpret_cast<const f32x4*>(window)[e];
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
          }
        } else {
          const unsigned g0 = (unsigned)cdict[p];
          rg::load_packed<STRIDE>(packed, g0 < last_gate ? g0 : last_gate, v);
          if (!mine) {
#pragma unroll
            for (int s = 0; s < STRIDE; ++s) v[s] = __builtin_bit_cast(float, RG_EXCLUDED_BITS);
          }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {   // masked gate: contributes to neither sum (interpolate.py:78-79)
          const bool good = rg::f32_bits(v[f]) != RG_EXCLUDED_BITS;
          // ONE select per field and pair: the effective weight is w or +0, and v_mul_legacy_f32 makes 0 * sentinel = +0
          // where an IEEE multiply would make NaN (for a non-zero weight the two multiplies are the same operation, so
          // unmasked NaN / Inf data propagates exactly as before: same bits as good ? w * v : 0)
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_TWO_SELECTS)   // A/B builds only: round 2's form of the same arithmetic
          addp(k % KS, f, good ? w[i] * v[f] : 0.0f);
          addw(k % KSW, f, good ? w[i] : 0.0f);
#else
          const float wf = good ? w[i] : 0.0f;
          addp(k % KS, f, rg_fmul_legacy(wf, v[f]));
          addw(k % KSW, f, wf);
#endif
        }
        // the per-pair path of an over-wide chunk (rare): five fields and more take its pairs one at a time -- three 32-byte
        // gathers in flight per record would set the whole kernel's register count (167 instead of <= 128 for eight fields)
        if constexpr (NF >= kFenceMinNF) fence_sums(k % KS, k % KSW);
      }
    };
    // sums of step r's batch; `last`: the round ends here -> fold the row's lanes and hand the sums to the row
    auto process = [&](const Step& r, const rg_u32x4 (&regs)[KPRE], bool last) {
#pragma unroll
      for (int k = 0; k < KPRE; ++k) {
        if (k < r.left) consume(r, regs[k], k);   // wave-uniform
      }
      if (!last) return;
      float sv[2 * NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        float sp, sw;                                  // chains in ascending order
        if constexpr (kByteMask) {
          sp = bp[0][f >> 1][f & 1];
          sw = bw[0][f >> 1][f & 1];
#pragma unroll
          for (int k = 1; k < KS; ++k) sp += bp[k][f >> 1][f & 1];
#pragma unroll
          for (int k = 1; k < KSW; ++k) sw += bw[k][f >> 1][f & 1];
        } else {
          sp = ap[0][f];
          sw = aw[0][f];
          ap[0][f] = aw[0][f] = 0.0f;
#pragma unroll
          for (int k = 1; k < KS; ++k) {
            sp += ap[k][f];
            ap[k][f] = 0.0f;
          }
#pragma unroll
          for (int k = 1; k < KSW; ++k) {
            sw += aw[k][f];
            aw[k][f] = 0.0f;
          }
        }
        sv[2 * f] = sp;
        sv[2 * f + 1] = sw;
      }
      if constexpr (kByteMask) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
#pragma unroll
          for (int j = 0; j < NP2; ++j) bp[k][j] = (f32x2)(0.0f);
        }
#pragma unroll
        for (int k = 0; k < KSW; ++k) {
#pragma unroll
          for (int j = 0; j < NP2; ++j) bw[k][j] = (f32x2)(0.0f);
        }
      }
      rg::butterfly<2 * NF>(sv, nl);
      if constexpr (kScatter) {
        constexpr int NP = NF <= 2 ? 2 : NF <= 4 ? 4 : 8, LGP = NF <= 2 ? 1 : NF <= 4 ? 2 : 3;    // fields, padded to 2^LGP
        float p[NP], w[NP];
#pragma unroll
        for (int f = 0; f < NP; ++f) {
          p[f] = f < NF ? sv[2 * (f < NF ? f : 0)] : 0.0f;
          w[f] = f < NF ? sv[2 * (f < NF ? f : 0) + 1] : 0.0f;
        }
        // after s stages entry j holds field j * 2^s + (sub mod 2^s)
        const int stages = lgl < LGP ? lgl : LGP;                   // wave-uniform
#pragma unroll
        for (int st = 0; st < LGP; ++st) {
          if (st < stages) {
            const bool bit = ((sub >> st) & 1) != 0;
#pragma unroll
            for (int i = 0; i < (NP >> (st + 1)); ++i) {
              p[i] = bit ? p[2 * i + 1] : p[2 * i];
              w[i] = bit ? w[2 * i + 1] : w[2 * i];
            }
          }
        }
        const int lp = 1 << stages;                                 // fields a row's lanes share among themselves
        const int f0 = sub & (lp - 1);
        const bool owner = r.live && (sub >> stages) == 0;          // more lanes per row than (padded) fields: the first store
        float* dst = out + ((size_t)f0 * n_vox + r0 + r.myrow);
        const size_t step_f = (size_t)lp * n_vox;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (j < (NP >> stages)) {                                 // wave-uniform
            if constexpr (kStage) {
              if (owner && f0 + j * lp < NF) stage[r.myrow * 8 + f0 + j * lp] = w[j] > 0.0f ? p[j] / w[j] : fill;
            } else if constexpr (DIAG & 2) {      // timing-only: no output store
              if (owner && f0 + j * lp < NF && p[j] == 123.456f) dst[j * step_f] = w[j];
            } else {
              if (owner && f0 + j * lp < NF) dst[j * step_f] = w[j] > 0.0f ? p[j] / w[j] : fill;
            }
          }
        }
      } else if constexpr (kRegs) {      // every lane of a row holds the row's sums: lane == row fetches them
        const int first = r.myrow - rgrp;                   // the round's first row (wave-uniform)
        const bool take = lane >= first && lane < first + rpr;
        const int src = ((lane - first) << lgl) & 63;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const float gp = __shfl(sv[2 * f], src, 64), gw = __shfl(sv[2 * f + 1], src, 64);
          mine_p[f] = take ? gp : mine_p[f];
          mine_w[f] = take ? gw : mine_w[f];
        }
      } else if (r.live && sub == 0) {
#pragma unroll
        for (int f = 0; f < NF; ++f) rowacc[r.myrow * NF + f] = (f32x2){sv[2 * f], sv[2 * f + 1]};
      }
    };

    // touch the records of round rho (its rows are consecutive, so its records are one byte range; the first 4 KiB of it)
    auto prefetch_round = [&](int rho) {
      if (rho >= rounds) return;                                          // wave-uniform
      const int first = rho * rpr;
      const int last = (first + rpr < nrows ? first + rpr : nrows) - 1;
      const unsigned b = (unsigned)__builtin_amdgcn_readlane(rs_o, first), e = (unsigned)__builtin_amdgcn_readlane(re_o, last);
      const int ob = (int)(b / 3u) * 16, oe = (int)((e + 2u) / 3u) * 16;
      const int off = ob + lane * 64;
      asm volatile("" : : "v"(pf_d0));                                   // the previous touch has long returned
      pf_d0 = rg_buffer_load_u32(rr, off < oe ? off : kOutOfRange, 0, 0);
    };
    rg_u32x4 regs_a[KPRE], regs_b[KPRE];
    Step sa = setup(0), sb;
    issue(sa, regs_a);
    for (;;) {     // two register stages, alternating: nothing in flight is ever copied
      sb = advance(sa);
      issue(sb, regs_b);
      if constexpr (kPrefetch > 0) {
        if (sb.rho != sa.rho) prefetch_round(sb.rho + kPrefetch - 1);    // behind the real loads in the (in-order) queue
      }
      process(sa, regs_a, sb.rho != sa.rho);
      if (sb.rho >= rounds) break;
      sa = advance(sb);
      issue(sa, regs_a);
      if constexpr (kPrefetch > 0) {
        if (sa.rho != sb.rho) prefetch_round(sa.rho + kPrefetch - 1);
      }
      process(sb, regs_b, sa.rho != sb.rho);
      if (sa.rho >= rounds) break;
    }
  };
  if (span > 0) {
#if defined(RG_EXPERIMENTS) && defined(RG_ROWWISE_NO_FALLBACK)     // register-count probe only: over-wide chunks are skipped
    if (windowed) run(std::true_type{});
#else
    if (windowed) run(std::true_type{}); else run(std::false_type{});
#endif
  }
  if constexpr (kPrefetch > 0) asm volatile("" : : "v"(pf_d0), "v"(pf_d1));      // the touches end here
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  if constexpr (kStage) {
    if (span > 0 && lane < nrows) {                   // (the wave barrier above orders the rounds' LDS writes before these reads)
      const f32x4 lo = reinterpret_cast<const f32x4*>(stage)[2 * lane], hi = reinterpret_cast<const f32x4*>(stage)[2 * lane + 1];
      const float vals[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if constexpr (DIAG & 2) {
          if (vals[f] == 123.456f) out[(size_t)f * n_vox + r0 + lane] = vals[f];
        } else {
          out[(size_t)f * n_vox + r0 + lane] = vals[f];
        }
      }
    }
  }
  if (lane < nrows && !(kScatter && span > 0)) {     // kScatter: the rounds stored their rows; a segment without pairs has none
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      f32x2 s = (f32x2)(0.0f);
      if constexpr (kRegs) s = (f32x2){mine_p[f], mine_w[f]};
      else if (span > 0) s = rowacc[lane * NF + f];
      if constexpr (DIAG & 2) {
        if (s.x == 123.456f) out[(size_t)f * n_vox + r0 + lane] = s.y;       // practically never
      } else if constexpr (DIAG & 32) {      // dense, dispatch-ordered store: 1 KiB per workgroup, neighbours adjacent
        const long idx = ((long)bid * kH + wv) * 64 + lane;
        if (idx < n_vox) out[idx] = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
      } else if constexpr (DIAG & 128) {     // non-temporal store
        __builtin_nontemporal_store(s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill, &out[(size_t)f * n_vox + r0 + lane]);
      } else if constexpr (DIAG & 256) {     // write-through store (sc0 sc1)
        const float val = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
        float* dst = &out[(size_t)f * n_vox + r0 + lane];
        asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(dst), "v"(val) : "memory");
      } else if constexpr (DIAG & 512) {     // sc1 only
        const float val = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
        float* dst = &out[(size_t)f * n_vox + r0 + lane];
        asm volatile("global_store_dword %0, %1, off sc1" : : "v"(dst), "v"(val) : "memory");
      } else if constexpr (DIAG & 64) {      // grid layout, but only the 128-byte lines this wavefront owns entirely
        const long a = r0 + lane, lo = (r0 + 31) & ~31L, hi = (r0 + nrows) & ~31L;
        if (a >= lo && a < hi) out[a] = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
      } else if constexpr (COLS) {
        const float val = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
        const int z = col_z0 + cb;
        if (out) out[(size_t)f * n_vox + r0 + lane] = val;
        if (cols.planes && z >= cols.keep_lo && z < cols.keep_lo + cols.n_keep)
          cols.planes[((size_t)f * cols.n_keep + (z - cols.keep_lo)) * cols.n_xy + (r0 - (long)z * cols.n_xy) + lane] = val;
        if (cols.col_val && z >= cols.col_lo && z <= cols.col_hi) column_max_step(best[f], val, z);
      } else {
        out[(size_t)f * n_vox + r0 + lane] = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
      }
    }
  }
  if constexpr (COLS) {
    col_nrows = nrows;
    col_xy = r0 - (long)(col_z0 + cb) * cols.n_xy + lane;
    if constexpr (!kRegs) {      // the next level's rounds overwrite the row sums in LDS: this level's reads come first
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  }   // chunks of this workgroup
  if constexpr (COLS) {
    if (cols.col_val && lane < col_nrows) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const size_t o = ((size_t)col_piece * NF + f) * cols.n_xy + col_xy;
        cols.col_val[o] = best[f].v;
        if (cols.col_arg) cols.col_arg[o] = best[f].idx;
      }
    }
  }
}

// Four fields, row sums in LDS instead of registers (one wavefront per SIMD more): taken when the window leaves room for five
// workgroups per CU next to the 8 KiB array.
constexpr long kLdsPerCu = 160 * 1024;
inline bool rowwise_lds_rowsums(int nf, int window_cap) {
  return nf == 4 && ((long)(window_cap + 1) * 4 * rowwise_entry_words<4>() + (long)kH * 64 * 4 * 8 + 512) * 5 <= kLdsPerCu;
}

template <typename IndT, int NF, int DIAG = 0>
int launch_rowwise(int window_cap, const void* indptr, const int64_t* dict_ptr, const int32_t* dict, const ChunkGrid& cg,
                   long n_vox, const float* packed, long n_gates, float fill, float* out, hipStream_t s,
                   const PackedStream& ps, int lanes_hint, int chunks_per_block = 0) {
  constexpr int STRIDE = stride_for(NF);
  constexpr int WS = rowwise_entry_words<NF>();                                   // 4-byte words per window entry
  const bool lds_sums = DIAG == 0 && rowwise_lds_rowsums(NF, window_cap);
  const long kStatic = NF >= 5 ? (long)kH * 64 * 8 * 4 + 16                                     // the staged values of 5-8 fields
                               : (RowwiseConfig<NF>::regs && !lds_sums) ? 16 : (long)kH * 64 * NF * 8;   // the row-sum array, if any
  // one entry beyond window_cap: the sentinel; a smaller window only sends more chunks down the per-pair path
  const long room = (65536 - kStatic - 256) / (4 * WS) - 1;
  if (window_cap > room) window_cap = (int)room;
  const long n_chunks = chunk_count(cg);
  if (chunks_per_block <= 0) chunks_per_block = kRowwiseChunksPerBlock;      // an explicit request (tile = 2200 + n) is honoured
  const dim3 grid((unsigned)((n_chunks + chunks_per_block - 1) / chunks_per_block)), block(64 * kH);
  const size_t lds = ((size_t)(window_cap + 1) * WS * sizeof(float) + 15) / 16 * 16;
#define RG_ROWWISE_LAUNCH(REGS_)                                                                                              \
  hipLaunchKernelGGL((csr_compact_rowwise_kernel<IndT, NF, STRIDE, DIAG, false, REGS_>), grid, block, lds, s,                   \
                     static_cast<const IndT*>(indptr), dict_ptr, dict, cg, packed, (unsigned)(n_gates - 1), fill, window_cap, \
                     n_vox, out, ps.rec, ps.rec_ptr, ps.w_base, lanes_hint, ps.order, (unsigned)n_chunks, chunks_per_block,   \
                     RowwiseColumns())
  if constexpr (NF == 4 && DIAG == 0) {
    if (lds_sums) RG_ROWWISE_LAUNCH(0); else RG_ROWWISE_LAUNCH(-1);
  } else {
    RG_ROWWISE_LAUNCH(-1);
  }
#undef RG_ROWWISE_LAUNCH
  return rg::check_launch("rg_csr_compact_apply_packed_f32");
}

}  // namespace

// Column mode of the row-wise kernel: the launcher rg_csr_columns.hip calls (declared in rg_compact_layout.hpp).
template <typename IndT, int NF>
static int launch_rowwise_columns_t(int window_cap, const void* indptr, const int64_t* dict_ptr, const int32_t* dict,
                                    const ChunkGrid& cg, long n_vox, const float* packed, long n_gates, float fill, float* out,
                                    hipStream_t s, const void* rec, const int64_t* rec_ptr, unsigned w_base, int rec_order,
                                    int lanes_hint, const RowwiseColumns& cols) {
  constexpr int STRIDE = stride_for(NF);
  constexpr int WS = rowwise_entry_words<NF>();
  constexpr int kRegsCols = -1;      // (four fields with the row sums in LDS: 108 instead of 111 VGPRs, the same 4 wavefronts)
  constexpr long kStatic = RowwiseConfig<NF>::regs ? 16 : (long)kH * 64 * NF * 8;
  const long room = (65536 - kStatic - 256) / (4 * WS) - 1;
  if (window_cap > room) window_cap = (int)room;
  hipLaunchKernelGGL((csr_compact_rowwise_kernel<IndT, NF, STRIDE, 0, true, kRegsCols>), dim3(cols.n_cols * (unsigned)cols.pieces),
                     dim3(64 * kH), ((size_t)(window_cap + 1) * WS * sizeof(float) + 15) / 16 * 16, s,
                     static_cast<const IndT*>(indptr), dict_ptr, dict, cg, packed, (unsigned)(n_gates - 1), fill, window_cap, n_vox,
                     out, static_cast<const rg_u32x4*>(rec), rec_ptr, w_base, lanes_hint, rec_order,
                     (unsigned)chunk_count(cg), 1, cols);
  return rg::check_launch("rg_csr_compact_apply_columns_f32");
}

int rg_launch_rowwise_columns(int nf, bool i64, int window_cap, const void* indptr, const int64_t* dict_ptr, const int32_t* dict,
                              const ChunkGrid& cg, long n_vox, const float* packed, long n_gates, float fill, float* out,
                              hipStream_t s, const void* rec, const int64_t* rec_ptr, unsigned w_base, int rec_order,
                              int lanes_hint, const RowwiseColumns& cols) {
#define RG_COLS(IND_, NF_) \
  launch_rowwise_columns_t<IND_, NF_>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, rec, rec_ptr, \
                                      w_base, rec_order, lanes_hint, cols)
  if (i64) {
    switch (nf) {
      case 1: return RG_COLS(int64_t, 1);
      case 2: return RG_COLS(int64_t, 2);
      case 3: return RG_COLS(int64_t, 3);
      default: return RG_COLS(int64_t, 4);
    }
  }
  switch (nf) {
    case 1: return RG_COLS(int32_t, 1);
    case 2: return RG_COLS(int32_t, 2);
    case 3: return RG_COLS(int32_t, 3);
    default: return RG_COLS(int32_t, 4);
  }
#undef RG_COLS
}

namespace {

template <typename IndT>
int launch_rowwise_nf(int nf, int window_cap, const void* indptr, const int64_t* dict_ptr, const int32_t* dict,
                      const ChunkGrid& cg, long n_vox, const float* packed, long n_gates, float fill, float* out,
                      hipStream_t s, const PackedStream& ps, int lanes_hint) {
#define RG_ROW(NF_) \
  launch_rowwise<IndT, NF_>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, lanes_hint, cpb)
  int cpb = 0;
#ifdef RG_EXPERIMENTS
  if (lanes_hint >= 200 && lanes_hint <= 264) {                // tile = 2200 + n: n consecutive chunks per workgroup
    cpb = lanes_hint - 200;
    lanes_hint = 0;
  }
  if (nf == 8 && lanes_hint >= 100 && lanes_hint < 200) {      // eight fields, timing-only: 1 = no gather, 2 = no store, 16 = no
    switch (lanes_hint - 100) {                                //   record loads, 40 = no window reads, 42 = 40 + 2, 19 = 16 + 2 + 1
      case 1: return launch_rowwise<IndT, 8, 1>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      case 2: return launch_rowwise<IndT, 8, 2>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      case 16: return launch_rowwise<IndT, 8, 16>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      case 19: return launch_rowwise<IndT, 8, 19>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      case 40: return launch_rowwise<IndT, 8, 1024>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      case 42: return launch_rowwise<IndT, 8, 1026>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      case 43: return launch_rowwise<IndT, 8, 1027>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      case 59: return launch_rowwise<IndT, 8, 1043>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, 1);
      default: break;
    }
  }
  if (nf == 1 && lanes_hint >= 100 && lanes_hint < 200) {      // timing-only diagnostics (tile = 2100 + DIAG bits)
    const int cpb1 = 1;          // the diagnostics run one chunk per workgroup
#define RG_DIAG(D_) \
  case D_: return launch_rowwise<IndT, 1, D_>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, cpb1)
    switch (lanes_hint - 100) {
      RG_DIAG(1); RG_DIAG(2); RG_DIAG(3); RG_DIAG(4); RG_DIAG(8); RG_DIAG(16); RG_DIAG(17); RG_DIAG(19); RG_DIAG(32);
      RG_DIAG(64);
      case 70: return launch_rowwise<IndT, 1, 128>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, cpb1);
      case 71: return launch_rowwise<IndT, 1, 256>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, cpb1);
      case 72: return launch_rowwise<IndT, 1, 512>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, cpb1);
      case 73: return launch_rowwise<IndT, 1, 2048>(window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates, fill, out, s, ps, 0, cpb1);
      default: break;
    }
#undef RG_DIAG
  }
#endif
  switch (nf) {
    case 1: return RG_ROW(1);
    case 2: return RG_ROW(2);
    case 3: return RG_ROW(3);
    case 4: return RG_ROW(4);
    case 5: return RG_ROW(5);
    case 6: return RG_ROW(6);
    case 7: return RG_ROW(7);
    default: return RG_ROW(8);
  }
#undef RG_ROW
}

}  // namespace

// 1-4 fields over the packed stream.  tile = 0: the row-wise kernel (agrees with rg_csr_apply_f32 to float32 rounding);
// tile = 384 (one field: also 576 / 768): the tile kernel over the same records (agrees with it bit for bit);
// tile = 2000 + h: row-wise with a diagnostic lane split (h = 1..64: that many lanes per row; h = 70 + t: aim for t
// records per lane and row) -- a different split is a different order of the float32 adds.
extern "C" int rg_csr_compact_apply_packed_f32(const void* indptr, int32_t indptr_is_i64, const void* records,
                                               const int64_t* rec_ptr, int32_t rec_order, uint32_t w_base,
                                               const int64_t* dict_ptr,
                                               const int32_t* dict, int64_t n_vox, int64_t n_pairs, int64_t line_len,
                                               int64_t lines_per_plane, const float* packed, int32_t n_fields,
                                               int32_t stride, int64_t n_gates, float fill_value, float* out,
                                               int32_t window_cap, int32_t tile, rg_stream_t stream) {
  const bool rowwise = tile == 0 || tile >= 2000;
  const int lanes_hint = tile >= 2000 ? tile - 2000 : 0;
#ifdef RG_EXPERIMENTS   // 2100 + DIAG bits (timing-only, wrong results) and 2200 + chunks per workgroup: experiment builds only
  const bool experiment = (lanes_hint >= 100 && lanes_hint < 200) || (lanes_hint >= 201 && lanes_hint <= 264);
#else
  const bool experiment = false;
#endif
  RG_REQUIRE(tile == 0 || tile == 384 || ((tile == 576 || tile == 768) && n_fields == 1) ||
                 (tile >= 2000 && ((lanes_hint >= 1 && lanes_hint <= 64 && (lanes_hint & (lanes_hint - 1)) == 0) ||
                                   (lanes_hint > 70 && lanes_hint <= 99) || experiment)),
             RG_EINVAL,
             "rg_csr_compact_apply_packed_f32: tile must be 0 (row-wise kernel), 384 (tile kernel; one field: also 576 / "
             "768) or 2000 + lane split");
  RG_REQUIRE(rec_order == RG_REC_ORDER_SEGMENT || rec_order == RG_REC_ORDER_DISPATCH, RG_EINVAL,
             "rg_csr_compact_apply_packed_f32: rec_order=%d is neither RG_REC_ORDER_SEGMENT nor RG_REC_ORDER_DISPATCH", rec_order);
  RG_REQUIRE(n_fields >= 1 && n_fields <= (rowwise ? 8 : 4), RG_EUNSUPPORTED,
             "rg_csr_compact_apply_packed_f32: n_fields=%d not in 1..%d (the row-wise kernel takes 1-8 fields; the tile kernel "
             "over the records 1-4: 5-8 fields use 128-pair tiles, not a whole number of 64-record loads)", n_fields,
             rowwise ? 8 : 4);
  RG_REQUIRE(stride == stride_for(n_fields), RG_EINVAL, "rg_csr_compact_apply_packed_f32: stride=%d, expected %d for %d fields",
             stride, stride_for(n_fields), n_fields);
  RG_REQUIRE(indptr && out && dict_ptr && rec_ptr, RG_EINVAL, "rg_csr_compact_apply_packed_f32: null indptr/dict_ptr/rec_ptr/out");
  RG_REQUIRE(n_vox >= 0 && n_pairs >= 0, RG_EINVAL, "rg_csr_compact_apply_packed_f32: negative size");
  RG_REQUIRE(n_pairs == 0 || (records && dict && packed && n_gates > 0), RG_EINVAL,
             "rg_csr_compact_apply_packed_f32: pairs present but records/dict/packed/n_gates missing");
  RG_REQUIRE(n_gates <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_packed_f32: n_gates exceeds int32 gate indices");
  RG_REQUIRE(n_vox <= 0x3FFFFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_packed_f32: n_vox too large for one launch");
  RG_REQUIRE(window_cap >= 0 && window_cap <= RG_COMPACT_MAX_WINDOW, RG_EINVAL,
             "rg_csr_compact_apply_packed_f32: window_cap %d outside 0..%d", window_cap, RG_COMPACT_MAX_WINDOW);
  RG_REQUIRE(rg::aligned16(records), RG_EALIGN, "rg_csr_compact_apply_packed_f32: records must be 16-byte aligned");
  RG_REQUIRE((w_base & 0x3FFFFFFu) == 0, RG_EINVAL,
             "rg_csr_compact_apply_packed_f32: w_base=0x%08x must have its low 26 bits clear (exponent a multiple of 8)", w_base);
  if (n_vox == 0) return RG_OK;
  ChunkGrid cg;
  RG_REQUIRE(make_chunk_grid(n_vox, line_len, lines_per_plane, &cg), RG_EINVAL,
             "rg_csr_compact_apply_packed_f32: n_vox=%ld is not planes x lines_per_plane=%ld x line_len=%ld", (long)n_vox,
             (long)lines_per_plane, (long)line_len);
  RG_REQUIRE(chunk_count(cg) <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_packed_f32: too many chunks for one launch");
  PackedStream ps;
  ps.rec = static_cast<const rg_u32x4*>(records);
  ps.rec_ptr = rec_ptr;
  ps.w_base = w_base;
  ps.order = rec_order;
  hipStream_t s = (hipStream_t)stream;
  if (rowwise)
    return indptr_is_i64 ? launch_rowwise_nf<int64_t>(n_fields, window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates,
                                                      fill_value, out, s, ps, lanes_hint)
                         : launch_rowwise_nf<int32_t>(n_fields, window_cap, indptr, dict_ptr, dict, cg, n_vox, packed, n_gates,
                                                      fill_value, out, s, ps, lanes_hint);
#define RG_K1P(IND_, NF_)                                                                                              \
  launch_nf<IND_, NF_, 384, 0, 0, true>(window_cap, indptr, nullptr, nullptr, dict_ptr, dict, cg, n_vox, packed, n_gates, \
                                        fill_value, out, s, ps)
#define RG_K1PT(IND_, TILE_)                                                                                            \
  launch_nf<IND_, 1, TILE_, 0, 0, true>(window_cap, indptr, nullptr, nullptr, dict_ptr, dict, cg, n_vox, packed, n_gates, \
                                        fill_value, out, s, ps)
  if (tile == 576) return indptr_is_i64 ? RG_K1PT(int64_t, 576) : RG_K1PT(int32_t, 576);
  if (tile == 768) return indptr_is_i64 ? RG_K1PT(int64_t, 768) : RG_K1PT(int32_t, 768);
  if (indptr_is_i64) {
    switch (n_fields) {
      case 1: return RG_K1P(int64_t, 1);
      case 2: return RG_K1P(int64_t, 2);
      case 3: return RG_K1P(int64_t, 3);
      default: return RG_K1P(int64_t, 4);
    }
  }
  switch (n_fields) {
    case 1: return RG_K1P(int32_t, 1);
    case 2: return RG_K1P(int32_t, 2);
    case 3: return RG_K1P(int32_t, 3);
    default: return RG_K1P(int32_t, 4);
  }
#undef RG_K1P
#undef RG_K1PT
}

// ---------------------------------------------------------------------------------------------------------------
// Building the compact copy: the distinct gates of every chunk and each pair's position among them.
// One workgroup per chunk keeps an open-addressing hash set of gate indices in LDS.  Chunks whose dictionary would
// overload the table are processed in R = 2, 4, ... 32 rounds, round r taking the gates of one residue class of a
// second hash, so any chunk up to 65536 distinct gates is handled with 32 KiB of LDS.
//   count pass: distinct gates per chunk (and the rounds it needed)   -> rg_scan_counts_i64 gives dict_ptr
//   fill pass : same rounds; the lane that claims a slot gives the gate the next position and writes the dictionary
//               entry, a second sweep over the round's pairs looks every gate up and stores its 16-bit position.
// Positions depend on the insertion order (not reproducible run to run); the gridding result does not.
// `gate_idx` / `local_idx` are addressed by ABSOLUTE pair number (indptr values), so a caller that holds only a slab of
// the index array passes pointers shifted by the slab's first pair.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kSlots = 8192;          // hash slots per workgroup (32 KiB)
constexpr int kMaxLoad = 6144;        // distinct gates one round may insert
constexpr int kBuildThreads = 256;
constexpr int kMaxRounds = 32;
constexpr int kSplitFlag = 0x80;               // chunk_rounds bit: one dictionary per wavefront (see the apply kernel)
constexpr int kNotCompactable = 0x40000000;    // chunk_counts value: a single segment references > 65536 gates

__device__ __forceinline__ unsigned slot_hash(unsigned g) { return (g * 2654435761u) >> 19; }      // 13 bits
__device__ __forceinline__ unsigned round_hash(unsigned g) { return (g * 0x85EBCA6Bu) >> 27; }     // 5 bits

struct ChunkPairs {   // the (up to) H contiguous pair ranges of one chunk
  long p0[kH], p1[kH];
};

template <typename IndT>
__device__ __forceinline__ ChunkPairs chunk_pairs(const IndT* __restrict__ indptr, const ChunkGrid& cg, unsigned chunk) {
  ChunkPairs cp;
#pragma unroll
  for (int w = 0; w < kH; ++w) {
    const Segment s = chunk_segment(cg, chunk, w);
    cp.p0[w] = s.nrows ? (long)indptr[s.r0] : 0;
    cp.p1[w] = s.nrows ? (long)indptr[s.r0 + s.nrows] : 0;
  }
  return cp;
}

// Inserts the gates of residue class `r` (of `rounds`) among the chunk's pairs.  Returns false when the table overloads.
// With `ids`, the lane that claims a slot also gives the gate its position (base + order of arrival) and writes the
// dictionary entry: positions then follow the order in which the chunk's pairs first mention a gate, so the 64
// consecutive pairs of one gather mostly hold neighbouring positions (fewer LDS bank conflicts than any fixed order).
__device__ bool insert_round(const int32_t* __restrict__ gidx, const ChunkPairs& cp, int rounds, int r, int* table,
                             int* s_count, int* s_overflow, unsigned short* ids = nullptr, int base = 0,
                             int32_t* __restrict__ dict_out = nullptr, int room = 0) {
  for (int i = threadIdx.x; i < kSlots; i += kBuildThreads) table[i] = -1;
  if (threadIdx.x == 0) { *s_count = 0; *s_overflow = 0; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kH; ++w) {
    for (long p = cp.p0[w] + threadIdx.x; p < cp.p1[w]; p += kBuildThreads) {
      const int g = gidx[p];
      if (rounds > 1 && (int)(round_hash((unsigned)g) & (unsigned)(rounds - 1)) != r) continue;
      unsigned h = slot_hash((unsigned)g);
      for (int probe = 0; probe < kSlots; ++probe) {   // bounded: a full table ends the walk (overflow is flagged first)
        const int seen = *(volatile int*)&table[h];   // other lanes insert concurrently
        if (seen == g) break;
        if (seen == -1) {
          if (*(volatile int*)s_overflow) break;
          const int old = atomicCAS(&table[h], -1, g);
          if (old == -1) {
            const int order = atomicAdd(s_count, 1);
            if (order >= kMaxLoad) *(volatile int*)s_overflow = 1;
            if (ids && base + order < room) {   // never writes past the dictionary the count pass sized
              ids[h] = (unsigned short)(base + order);
              dict_out[base + order] = g;
            }
            break;
          }
          if (old == g) break;
        }
        h = (h + 1) & (kSlots - 1);
      }
    }
  }
  __syncthreads();
  return *s_overflow == 0;
}

template <typename IndT>
__global__ __launch_bounds__(kBuildThreads) void compact_count_kernel(const IndT* __restrict__ indptr,
                                                                      const int32_t* __restrict__ gidx, ChunkGrid cg,
                                                                      int32_t* __restrict__ chunk_counts,
                                                                      uint8_t* __restrict__ chunk_rounds) {
  __shared__ int table[kSlots];
  __shared__ int s_count, s_overflow;
  const unsigned chunk = blockIdx.x;
  const ChunkPairs cp = chunk_pairs(indptr, cg, chunk);
  // distinct gates among the given ranges, and the hashing rounds that took (-1: more than kMaxRounds can hold)
  auto count = [&](const ChunkPairs& ranges, int& rounds) {
    int total = 0;
    while (true) {
      total = 0;
      bool ok = true;
      for (int r = 0; r < rounds && ok; ++r) {
        ok = insert_round(gidx, ranges, rounds, r, table, &s_count, &s_overflow);
        total += s_count;
        __syncthreads();
      }
      if (ok) return total;
      rounds *= 2;
      if (rounds > kMaxRounds) { rounds = kMaxRounds; return -1; }
    }
  };
  int rounds = 1;
  int total = count(cp, rounds);
  int flag = 0;
  if (total < 0 || total > 65536) {
    // too rich for 16-bit positions into ONE dictionary: one dictionary per wavefront (segment) instead, behind a
    // header of kH offsets.  rounds = the most any of the wavefronts needs.
    flag = kSplitFlag;
    total = kH;
    int worst = 1;
    for (int w = 0; w < kH; ++w) {
      ChunkPairs one;
#pragma unroll
      for (int k = 0; k < kH; ++k) { one.p0[k] = 0; one.p1[k] = 0; }
      one.p0[0] = cp.p0[w];
      one.p1[0] = cp.p1[w];
      int rw = 1;
      const int tw = count(one, rw);
      if (tw < 0 || tw > 65536) { total = kNotCompactable; break; }
      total += tw;
      worst = rw > worst ? rw : worst;
    }
    rounds = worst;
  }
  if (threadIdx.x == 0) {
    chunk_counts[chunk] = total;
    chunk_rounds[chunk] = (uint8_t)(rounds | flag);
  }
}

template <typename IndT>
__global__ __launch_bounds__(kBuildThreads) void compact_fill_kernel(const IndT* __restrict__ indptr,
                                                                     const int32_t* __restrict__ gidx, ChunkGrid cg,
                                                                     const int64_t* __restrict__ dict_ptr,
                                                                     const uint8_t* __restrict__ chunk_rounds,
                                                                     int32_t* __restrict__ dict,
                                                                     uint16_t* __restrict__ local_idx,
                                                                     int32_t* __restrict__ error_flag) {
  __shared__ int table[kSlots];
  __shared__ unsigned short ids[kSlots];
  __shared__ int s_count, s_overflow;
  const unsigned chunk = blockIdx.x;
  const ChunkPairs cp = chunk_pairs(indptr, cg, chunk);
  const long d0 = dict_ptr[chunk];
  const int expect = (int)(dict_ptr[chunk + 1] - d0);
  const int rounds = chunk_rounds[chunk] & (kSplitFlag - 1);
  const bool split = (chunk_rounds[chunk] & kSplitFlag) != 0;
  // positions of the given ranges' pairs into a dictionary written at dict_out; returns its size
  auto fill = [&](const ChunkPairs& ranges, int32_t* __restrict__ dict_out, int room) {
    int base = 0;
    for (int r = 0; r < rounds; ++r) {
      // cannot overload when the inputs are those of the count pass; if they are not (gate_idx changed in between, a
      // wrong chunk_rounds) the walks below are bounded and the mismatch is reported through error_flag, never a hang
      insert_round(gidx, ranges, rounds, r, table, &s_count, &s_overflow, ids, base, dict_out, room);
      if (threadIdx.x == 0 && (s_overflow || base + s_count > room)) atomicOr(error_flag, 1);
#pragma unroll
      for (int w = 0; w < kH; ++w) {
        for (long p = ranges.p0[w] + threadIdx.x; p < ranges.p1[w]; p += kBuildThreads) {
          const int g = gidx[p];
          if (rounds > 1 && (int)(round_hash((unsigned)g) & (unsigned)(rounds - 1)) != r) continue;
          unsigned h = slot_hash((unsigned)g);
          int probe = 0;
          while (table[h] != g && probe < kSlots) { h = (h + 1) & (kSlots - 1); ++probe; }
          if (probe == kSlots) {          // the gate was never inserted: count and fill saw different inputs
            atomicOr(error_flag, 2);
            local_idx[p] = 0;
          } else {
            local_idx[p] = ids[h];
          }
        }
      }
      base += s_count;
      __syncthreads();
    }
    return base;
  };
  int base;
  if (!split) {
    base = fill(cp, dict + d0, expect < 65536 ? expect : 65536);
  } else {
    base = kH;                                       // header: offset of every wavefront's dictionary
    for (int w = 0; w < kH; ++w) {
      if (threadIdx.x == 0) dict[d0 + w] = base;
      ChunkPairs one;
#pragma unroll
      for (int k = 0; k < kH; ++k) { one.p0[k] = 0; one.p1[k] = 0; }
      one.p0[0] = cp.p0[w];
      one.p1[0] = cp.p1[w];
      const int left = expect - base;
      base += fill(one, dict + d0 + base, left < 65536 ? (left > 0 ? left : 0) : 65536);
    }
  }
  if (threadIdx.x == 0 && base != expect) atomicOr(error_flag, 4);
}

}  // namespace

extern "C" int rg_csr_compact_count(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, int64_t n_rows,
                                    int64_t line_len, int64_t lines_per_plane, int32_t* chunk_counts,
                                    uint8_t* chunk_rounds, rg_stream_t stream) {
  RG_REQUIRE(n_rows >= 0, RG_EINVAL, "rg_csr_compact_count: negative size");
  if (n_rows == 0) return RG_OK;
  RG_REQUIRE(indptr && chunk_counts && chunk_rounds, RG_EINVAL, "rg_csr_compact_count: null pointer");
  ChunkGrid cg;
  RG_REQUIRE(make_chunk_grid(n_rows, line_len, lines_per_plane, &cg), RG_EINVAL,
             "rg_csr_compact_count: n_rows=%ld is not planes x lines_per_plane=%ld x line_len=%ld", (long)n_rows,
             (long)lines_per_plane, (long)line_len);
  const long chunks = chunk_count(cg);
  RG_REQUIRE(chunks <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_count: too many chunks for one launch");
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    hipLaunchKernelGGL(compact_count_kernel<int64_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int64_t*>(indptr), gate_idx, cg, chunk_counts, chunk_rounds);
  else
    hipLaunchKernelGGL(compact_count_kernel<int32_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int32_t*>(indptr), gate_idx, cg, chunk_counts, chunk_rounds);
  return rg::check_launch("rg_csr_compact_count");
}

extern "C" int rg_csr_compact_fill(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, int64_t n_rows,
                                   int64_t line_len, int64_t lines_per_plane, const int64_t* dict_ptr,
                                   const uint8_t* chunk_rounds, int32_t* dict, uint16_t* local_idx, int32_t* error_flag,
                                   rg_stream_t stream) {
  RG_REQUIRE(n_rows >= 0, RG_EINVAL, "rg_csr_compact_fill: negative size");
  if (n_rows == 0) return RG_OK;
  RG_REQUIRE(indptr && dict_ptr && chunk_rounds && error_flag, RG_EINVAL, "rg_csr_compact_fill: null pointer");
  ChunkGrid cg;
  RG_REQUIRE(make_chunk_grid(n_rows, line_len, lines_per_plane, &cg), RG_EINVAL,
             "rg_csr_compact_fill: n_rows=%ld is not planes x lines_per_plane=%ld x line_len=%ld", (long)n_rows,
             (long)lines_per_plane, (long)line_len);
  const long chunks = chunk_count(cg);
  RG_REQUIRE(chunks <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_fill: too many chunks for one launch");
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    hipLaunchKernelGGL(compact_fill_kernel<int64_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int64_t*>(indptr), gate_idx, cg, dict_ptr, chunk_rounds, dict, local_idx,
                       error_flag);
  else
    hipLaunchKernelGGL(compact_fill_kernel<int32_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int32_t*>(indptr), gate_idx, cg, dict_ptr, chunk_rounds, dict, local_idx,
                       error_flag);
  return rg::check_launch("rg_csr_compact_fill");
}
