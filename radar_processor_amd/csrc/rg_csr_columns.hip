// K1p  rg_csr_compact_apply_columns_f32: the row-wise kernel of rg_csr_compact.hip with workgroups that are PERSISTENT over a
// COLUMN of chunks -- the same (line group, segment) patch through consecutive grid levels -- and an optional products
// epilogue (COLMAX / first-argmax in registers, selected levels stored as planes) that removes the 3-D grid's round trip
// through HBM.  radar_grid/interpolate.py:69-104 (the masked weighted mean), :137-140 (several fields, one pass);
// radar_grid/products.py:361-412 (CAPPI needs two levels), :462-490 (column maximum over a level window).
//
// Why: callers that keep 2-D products only (BASELINE configs 3 and 5) no longer write F x 4 x V bytes of grid and read them
// back -- on this part writes mixed into a streaming read cost ten times their stand-alone price.
//
// The kernel is the COLUMN MODE of the row-wise kernel itself (csr_compact_rowwise_kernel<..., COLS = true> in
// rg_csr_compact.hip): the same code from a chunk's row pointers to its row sums, one chunk after the other behind a barrier,
// with 4-12 more VGPRs for the running maxima.  This file holds the entry point and the merge of level pieces.
//
// Measured (round 4, profiles/r04_columns_variants_*.json; ms per pass, bench grid / config 2, same process and arrays):
// products only against row-wise kernel + separate COLMAX/argmax + CAPPI: one field 8.19 vs 8.07 / 1.16 vs 1.01, three
// fields 10.73 vs 10.52 / 1.41 vs 1.40, four fields 11.49 vs 12.03 / 1.53 vs 1.58 -- the store it saves (0.25-0.3 ms per
// pass here) is what walking columns costs (workgroups no longer sweep the grid as one front: neighbouring chunks stop
// sharing their gathers in L2), so it pays from four field-volumes per pass on, and always in memory (no F x 640 MB of grid).
// Two designs that ALSO tried to take the per-chunk chain of dependent loads (row pointers / offsets -> dictionary ->
// gathers -> barrier -> records) off the critical path were built, tested bit-identical and measured slower; they lived in
// this file at commit 54393d1 and are written up in EXPERIMENTS.md:
//   LOADER  a fifth wavefront fills a SECOND LDS window with chunk k+1 and publishes chunk k+2's metadata in an LDS ring while
//           four stream chunk k across the boundary: 1.6-2.6x slower (half the resident streaming wavefronts per CU);
//   WALK    four wavefronts, one window, chunk k+1's metadata held in registers a chunk ahead and its first records
//           requested before the window is filled: +7-11 % (12-36 more VGPRs cost a wavefront per SIMD).
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
