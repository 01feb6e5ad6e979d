// Layout of the compact CSR copy shared by the K1c kernels (rg_csr_compact.hip: tile and row-wise kernels;
// rg_csr_columns.hip: the column-persistent row-wise kernel): chunk grid, block -> chunk rotation, segments, buffer
// resources, per-field-count tuning of the row-wise kernels.  Everything here has internal linkage on purpose (each
// translation unit gets its own copy; nothing is exported).
#pragma once

#include <type_traits>

#include "rg_common.hpp"
#include "rg_row_phase.hpp"

// 16-byte buffer load by intrinsic name (this compiler's __builtin_amdgcn_raw_buffer_load_b128 returns the first dword
// in every element, see rg_csr_apply.hip); namespace scope: a name bound to an intrinsic must not have internal linkage.
using rg_u32x4 = unsigned __attribute__((ext_vector_type(4)));
__device__ rg_u32x4 rg_buffer_load_v4u32(__amdgpu_buffer_rsrc_t, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v4i32");

__device__ unsigned rg_buffer_load_u32(__amdgpu_buffer_rsrc_t, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.i32");

// v_mul_legacy_f32 by intrinsic name (this clang has no __builtin_amdgcn_fmul_legacy): 0 * x = +0 for EVERY x, NaN and
// infinity included; any other product is the IEEE one.
__device__ float rg_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");

// Types that cross translation units (the launcher of the row-wise kernel's column mode is called from
// rg_csr_columns.hip) live in a NAMED namespace: a function with a parameter of an anonymous-namespace type has no linkage.
namespace rgl {

// Where a chunk's wavefronts find their rows: the grid as planes x lines x rows (nz x ny x nx for a radar grid).
// Chunk arithmetic is 32-bit on purpose: every wavefront decodes its chunk number with two divisions, and a 64-bit
// division costs this ISA a few hundred instructions (measured: 7 % of the single-field kernel).
struct ChunkGrid {
  long line_len;            // rows per line (nx)
  long lines_per_plane;     // lines per plane (ny)
  long n_planes;            // planes (nz)
  unsigned nsx;             // segments per line = ceil(line_len / 64)
  unsigned nyg;             // line groups per plane = ceil(lines_per_plane / H)
  unsigned rot_step;        // columns the block -> chunk map rotates per line group (speed only)
  unsigned seg_base;        // a line's nsx segments are balanced: the first seg_extra hold seg_base + 1 rows, the
  unsigned seg_extra;       // others seg_base (<= 64 either way) -- rg_csr_apply_f32 cuts its lines the same way
  unsigned grp0;            // line groups in front of this grid when it is a slab of whole planes of a larger one
                            // (rg_csr_compact_pack's plane0 * nyg; 0 for the apply kernels): the rotation counts them
};

// ---- column mode of the row-wise kernel (rg_csr_compact_apply_columns_f32) ----
struct RowwiseColumns {
  const int32_t* order = nullptr;   // optional: workgroup -> piece * n_cols + column (heaviest first)
  float* planes = nullptr;          // [F][n_keep][n_xy] or null
  float* col_val = nullptr;         // [pieces][F][n_xy] (pieces == 1: the caller's plane) or null
  int32_t* col_arg = nullptr;       // same shape, or null
  long n_xy = 0;
  unsigned n_cols = 0;              // line groups per plane x segments per line
  int pieces = 1, keep_lo = 0, n_keep = 0, col_lo = 0, col_hi = 0;
};

}  // namespace rgl

namespace {

using rgl::ChunkGrid;
using rgl::RowwiseColumns;
using rg::f32x2;
using rg::f32x4;
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int kRsrcRaw32 = 0x00020000;   // gfx9 buffer resource word 3: DATA_FORMAT = 32, untyped access

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, long bytes) {   // `base` and `bytes` wave-uniform
  const unsigned nb = bytes >= 0xFFFFFFFFL ? 0xFFFFFFFFu : bytes <= 0 ? 0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)nb, kRsrcRaw32);
}

constexpr int kH = RG_COMPACT_LINES;   // grid lines (= wavefronts) per chunk

// Workgroups are dealt to the 8 XCDs round-robin by blockIdx.  With nsx segments per line a multiple of 8, chunk column
// sx would always land on XCD sx % 8 and each XCD would own one x-slab of the grid for the whole launch; rotating the
// columns by rot_step per line group makes every XCD see every column.  The chunk a workgroup takes is a bijection of
// blockIdx; with RG_REC_ORDER_DISPATCH the records are STORED in this order (see rg_csr_compact_pack), so the map is part
// of the layout: grid_geometry.CompactCSR.slot_of_segments restates it.
__device__ __forceinline__ unsigned block_chunk(const ChunkGrid& g, unsigned bid) {
  const unsigned grp = bid / g.nsx;
  const unsigned col = bid - grp * g.nsx;
  const unsigned rot = col + ((grp + g.grp0) * g.rot_step) % g.nsx;
  return grp * g.nsx + (rot >= g.nsx ? rot - g.nsx : rot);
}

__host__ __device__ inline long chunk_count(const ChunkGrid& g) { return g.n_planes * (long)g.nyg * (long)g.nsx; }

struct Segment {
  long r0;      // first row
  long seg;     // segment number, line-major: (plane * lines_per_plane + line) * nsx + sx
  int nrows;    // 0 for a wavefront past the last line of the plane
};

__device__ __forceinline__ Segment chunk_segment(const ChunkGrid& g, unsigned chunk, int w) {   // chunk < 2^31
  const unsigned grp = chunk / g.nsx;         // line group, counted through all planes
  const unsigned sx = chunk - grp * g.nsx;
  const unsigned plane = grp / g.nyg;
  const unsigned yg = grp - plane * g.nyg;
  const long y = (long)yg * kH + w;
  Segment s;
  if (y >= g.lines_per_plane) {
    s.r0 = 0;
    s.seg = 0;
    s.nrows = 0;
    return s;
  }
  s.seg = ((long)plane * g.lines_per_plane + y) * g.nsx + sx;
  const unsigned x0 = sx * g.seg_base + (sx < g.seg_extra ? sx : g.seg_extra);
  s.r0 = ((long)plane * g.lines_per_plane + y) * g.line_len + (long)x0;
  s.nrows = (int)(g.seg_base + (sx < g.seg_extra ? 1u : 0u));
  return s;
}


constexpr int stride_for(int nf) { return nf == 1 ? 1 : nf == 2 ? 2 : nf <= 4 ? 4 : 8; }

bool make_chunk_grid(int64_t n_rows, int64_t line_len, int64_t lines_per_plane, ChunkGrid* cg) {
  if (line_len <= 0) line_len = n_rows > 0 ? n_rows : 1;
  if (n_rows % line_len != 0) return false;
  const long n_lines = n_rows / line_len;
  if (lines_per_plane <= 0) lines_per_plane = n_lines > 0 ? n_lines : 1;
  if (n_lines % lines_per_plane != 0) return false;
  cg->line_len = line_len;
  cg->lines_per_plane = lines_per_plane;
  cg->n_planes = n_lines / lines_per_plane;
  const long nsx = (line_len + 63) / 64, nyg = (lines_per_plane + kH - 1) / kH;
  if (nsx > 0x7FFFFFFFL || nyg > 0x7FFFFFFFL) return false;
  cg->nsx = (unsigned)nsx;
  cg->nyg = (unsigned)nyg;
  cg->seg_base = (unsigned)(line_len / nsx);
  cg->seg_extra = (unsigned)(line_len % nsx);
  // measured on the bench grid (32 columns), ms per launch: 0 -> 14.2 (every XCD keeps its columns), 8 -> 10.5,
  // 1 -> 9.35, 2 -> 9.27, 3 -> 9.24, 5 -> 9.23, 7 -> 9.21, 9 -> 9.22, 11 -> 9.31, 17 -> 9.22
  cg->rot_step = RG_COMPACT_ROTATION;
  cg->grp0 = 0;
  return true;
}


// ---- per-field-count configuration of the row-wise kernels (see rg_csr_compact.hip for what the knobs mean) ----
template <int NF> struct RowwiseConfig;
template <> struct RowwiseConfig<1> { static constexpr int kpre = 3, target = 4; static constexpr bool narrow = false, regs = false; };
template <> struct RowwiseConfig<2> { static constexpr int kpre = 3, target = 4; static constexpr bool narrow = false, regs = false; };
// Tuning knobs of three fields: fixed in the product library; -DRG_EXPERIMENTS builds (tools/build_experiments.py) may
// override them with -DRG_ROWWISE_KPRE3=.. etc. for A/B measurements.
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_KPRE3)
#undef RG_ROWWISE_KPRE3
#define RG_ROWWISE_KPRE3 3
#endif
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_TARGET3)
#undef RG_ROWWISE_TARGET3
#define RG_ROWWISE_TARGET3 6
#endif
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_REGS3)
#undef RG_ROWWISE_REGS3
#define RG_ROWWISE_REGS3 true
#endif
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_NARROW3)
#undef RG_ROWWISE_NARROW3
#define RG_ROWWISE_NARROW3 true
#endif
template <> struct RowwiseConfig<3> { static constexpr int kpre = RG_ROWWISE_KPRE3, target = RG_ROWWISE_TARGET3; static constexpr bool narrow = RG_ROWWISE_NARROW3, regs = RG_ROWWISE_REGS3; };
template <> struct RowwiseConfig<4> { static constexpr int kpre = 3, target = 8; static constexpr bool narrow = false, regs = true; };
// Five to eight fields (one pass over the records for up to eight volumes of the same geometry: batch.VolumeBatch)
template <> struct RowwiseConfig<5> { static constexpr int kpre = 3, target = 8; static constexpr bool narrow = false, regs = true; };
template <> struct RowwiseConfig<6> { static constexpr int kpre = 3, target = 8; static constexpr bool narrow = false, regs = true; };
template <> struct RowwiseConfig<7> { static constexpr int kpre = 3, target = 8; static constexpr bool narrow = false, regs = true; };
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_KPRE8)
#undef RG_ROWWISE_KPRE8
#define RG_ROWWISE_KPRE8 3
#endif
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_TARGET8)
#undef RG_ROWWISE_TARGET8
#define RG_ROWWISE_TARGET8 12
#endif
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_REGS8)
#undef RG_ROWWISE_REGS8
#define RG_ROWWISE_REGS8 true
#endif
template <> struct RowwiseConfig<8> { static constexpr int kpre = RG_ROWWISE_KPRE8, target = RG_ROWWISE_TARGET8; static constexpr bool narrow = false, regs = RG_ROWWISE_REGS8; };

// Byte-mask window entries: the window holds v' = the value, or +0 where the gate is excluded, and one BYTE per field that is
// 1 / 0 = usable / excluded.  A pair then costs, per field, half a packed multiply and half a packed add (sum w*v': w * +0 = +0,
// what select + v_mul_legacy_f32 gave), a byte -> float conversion and half a packed fma (sum w*g: the product is exact, so the
// same float32 as adding w or +0) instead of compare + select + v_mul_legacy_f32 + two half packed adds -- the same bits from
// fewer instructions.  Three fields: the mask word is the entry's fourth slot (16-byte entries); four fields: a second array of
// words behind the 16-byte value entries; five to eight fields: 32-byte value entries and two mask words per entry behind them.
// Measured with the pairs of a record taken one at a time (the asm fences of the kernel; without them the wider entries
// cost a wavefront per SIMD and the gain): three / four fields -4 ... -5 % on the bench grid, -1 ... -4 % on config 2
// (19 % fewer VALU instructions: profiles/r04_bytemask_*.json); five fields and more have no other form.
// RG_ROWWISE_BYTEMASK = the smallest field count that uses it (experiment builds: 5 = the select + legacy-multiply form
// for three and four fields).
#if !defined(RG_EXPERIMENTS) || !defined(RG_ROWWISE_BYTEMASK)
#undef RG_ROWWISE_BYTEMASK
#define RG_ROWWISE_BYTEMASK 3
#endif
template <int NF> constexpr bool rowwise_bytemask() { return NF >= 5 || (NF >= 3 && NF >= RG_ROWWISE_BYTEMASK); }
// 4-byte words of LDS per window entry of the row-wise kernel: values, then masks
template <int NF> constexpr int rowwise_value_words() {
  return rowwise_bytemask<NF>() ? (NF <= 4 ? 4 : 8) : RowwiseConfig<NF>::narrow ? 3 : NF <= 2 ? 2 : 4;
}
template <int NF> constexpr int rowwise_mask_words() { return !rowwise_bytemask<NF>() || NF == 3 ? 0 : NF == 4 ? 1 : 2; }
template <int NF> constexpr int rowwise_entry_words() { return rowwise_value_words<NF>() + rowwise_mask_words<NF>(); }


// ---- column mode of the row-wise kernel (rg_csr_compact_apply_columns_f32) --------------------------------------------------
struct ColumnBest {
  float v;     // NaN = nothing seen yet
  int idx;     // -1 = nothing seen yet
};

// np.fmax.reduce in level order (rg_products.hip: step<true>): the first non-NaN level starts the reduction, a later one
// replaces it only when strictly greater; NaN never wins.
__device__ __forceinline__ void column_max_step(ColumnBest& acc, float v, int z) {
  if (acc.idx < 0) {
    if (!isnan(v)) { acc.v = v; acc.idx = z; }
  } else {
    const bool keep = acc.v >= v || isnan(v);
    if (!keep) { acc.v = v; acc.idx = z; }
  }
}

}  // namespace

// defined in rg_csr_compact.hip (next to the kernel), called by rg_csr_columns.hip
int rg_launch_rowwise_columns(int nf, bool i64, int window_cap, const void* indptr, const int64_t* dict_ptr, const int32_t* dict,
                              const rgl::ChunkGrid& cg, long n_vox, const float* packed, long n_gates, float fill, float* out,
                              hipStream_t s, const void* rec, const int64_t* rec_ptr, unsigned w_base, int rec_order,
                              int lanes_hint, const rgl::RowwiseColumns& cols);
