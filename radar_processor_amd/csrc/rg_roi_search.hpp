// Shared pieces of the ROI neighbour search (geometry builder and fused gridder).
// Include only from translation units compiled with -ffp-contract=off.
#pragma once

#include "rg_common.hpp"

namespace rg {
namespace roi {

struct Cells {
  double x0, y0, inv_cx, inv_cy, z_lo, z_hi;
  int ncx, ncy;
  int levels;      // > 1: per-level gate lists, the cells of level iz at iz * ncx * ncy
  int level0;      // ... and the level the call's zc[0] is
};

inline Cells to_cells(const rg_cellgrid* c) {
  Cells r;
  r.x0 = c->x0; r.y0 = c->y0; r.inv_cx = c->inv_cx; r.inv_cy = c->inv_cy; r.z_lo = c->z_lo; r.z_hi = c->z_hi;
  r.ncx = c->ncx; r.ncy = c->ncy;
  r.levels = c->levels > 1 ? c->levels : 0;
  r.level0 = c->levels > 1 ? c->level0 : 0;
  return r;
}

__device__ __forceinline__ double cell_coord(double g, double origin, double inv) { return floor((g - origin) * inv); }

__device__ __forceinline__ int cell_clamped(double g, double origin, double inv, int n) {
  const double t = cell_coord(g, origin, inv);
  return t < 0.0 ? 0 : (t >= (double)n ? n - 1 : (int)t);
}

struct SearchArgs {
  const rg_gate4* sorted;
  const int* cell_start;
  Cells c;
  const float* xc;
  const float* yc;
  const float* zc;
  int nz, ny, nx;
  long n_vox;
  double min_radius, beam_factor;
};

template <int W>
__device__ __forceinline__ float roi_weight(double d2, double r2) {
  if constexpr (W == RG_W_BARNES2) {
    return (float)(exp(-d2 / (r2 / 4.0)) + 1e-5);  // compute.py:83
  } else if constexpr (W == RG_W_CRESSMAN) {
    return (float)((r2 - d2) / (r2 + d2));         // compute.py:85
  } else {
    return 1.0f;                                    // compute.py:87 ('nearest' = uniform mean)
  }
}


inline int check_search_args(const char* fn, const rg_gate4* sorted, const int32_t* cell_start, const rg_cellgrid* cells,
                             const float* xc, const float* yc, const float* zc, int nz, int ny, int nx) {
  RG_REQUIRE(sorted && cell_start && cells && xc && yc && zc, RG_EINVAL, "%s: null pointer", fn);
  RG_REQUIRE(nz >= 1 && ny >= 1 && nx >= 1, RG_EINVAL, "%s: bad grid shape (%d,%d,%d)", fn, nz, ny, nx);
  RG_REQUIRE(cells->ncx >= 1 && cells->ncy >= 1 && (long)cells->ncx * cells->ncy < 0x7FFFFFFFL, RG_EINVAL,
             "%s: bad cell grid %dx%d", fn, cells->ncx, cells->ncy);
  RG_REQUIRE(rg::aligned16(sorted), RG_EALIGN, "%s: sorted_gates must be 16-byte aligned", fn);
  RG_REQUIRE(cells->levels <= 1 || (cells->level0 >= 0 && cells->level0 + nz <= cells->levels &&
                                    (long)cells->ncx * cells->ncy * cells->levels < 0x7FFFFFFFL), RG_EINVAL,
             "%s: per-level gate lists for %d levels, call covers levels %d .. %d (or too many cells)", fn, cells->levels,
             cells->level0, cells->level0 + nz - 1);
  return RG_OK;
}

inline SearchArgs make_args(const rg_gate4* sorted, const int32_t* cell_start, const rg_cellgrid* cells, const float* xc,
                            const float* yc, const float* zc, int nz, int ny, int nx, double min_radius,
                            double beam_factor) {
  SearchArgs a;
  a.sorted = sorted; a.cell_start = cell_start; a.c = to_cells(cells);
  a.xc = xc; a.yc = yc; a.zc = zc; a.nz = nz; a.ny = ny; a.nx = nx;
  a.n_vox = (long)nz * ny * nx;
  a.min_radius = min_radius; a.beam_factor = beam_factor;
  return a;
}

}  // namespace roi
}  // namespace rg
