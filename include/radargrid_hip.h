/*
 * radargrid_hip.h -- C ABI of libradargrid_hip.so (MI355X / gfx950 HIP kernels for the radar_grid hot path).
 *
 * The reference (jgmarti84/radar-processor) is pure Python/NumPy: it has no FFI, plugin or operator
 * interface for this path.  The drop-in boundary is therefore the Python function surface of
 * `radar_grid` (src/radar_grid/__init__.py:39-82), and this header is the native layer directly beneath
 * it.  Each entry point names the reference code it replaces (paths relative to /root/reference/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - the caller owns every buffer; no entry point allocates, frees or synchronises;
 *   - all work is enqueued on the caller's `hipStream_t` (passed as void*), so calls are graph-capturable;
 *   - return value: RG_OK (0) or a negative rg_status; rg_last_error() gives a thread-local message;
 *   - float32 device arrays read with 16-byte vector loads must be 16-byte aligned (RG_EALIGN otherwise);
 *   - no global mutable state: entry points are thread-safe per stream.
 *
 * Data layout in HBM
 *   gates      flat ray-major index g = (sweep*n_az + iaz)*n_gates + k   (radar_grid/utils.py:35-37)
 *   voxels     z-major flat index v = (iz*ny + iy)*nx + ix               (radar_grid/compute.py:188-190,257-258)
 *   CSR        indptr[V+1] (int32 or int64), gate_idx int32[P], weights float32[P]  (radar_grid/geometry.py:46-52)
 *   packed fields  float32 [G][stride], stride in {1,2,4,8}: slot f of gate g holds field f's value, or the
 *              EXCLUDED sentinel when the gate is masked/filtered for that field, so that the gather in
 *              rg_csr_apply_f32 needs ONE load per pair for all fields (mask folded into the value).
 */
#ifndef RADARGRID_HIP_H
#define RADARGRID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a signature in this header changes (an argument inserted, a meaning changed).  The Python
 * binding (radar_processor_amd/_native.py: ABI_VERSION) refuses a library whose rg_version() differs, so a stale git-ignored
 * .so is reported as such instead of being called with shifted arguments.  History: 100 rounds 1-2; 101 rec_order / plane0 of
 * rg_csr_compact_pack and rg_csr_compact_apply_packed_f32 (round 3); 102 rg_csr_compact_apply_columns_f32, diagnostic tile
 * codes refused by the product build (round 4); 103 the row-wise kernel of rg_csr_compact_apply_packed_f32 takes 1-8 fields
 * (no signature changed: a 102 library answers RG_EUNSUPPORTED for 5-8); 104 rg_cellgrid.levels + the per-level gate lists
 * (rg_geom_bin_levels_count / rg_geom_bin_gates_levels_f32). */
#define RG_VERSION 104
#define RG_MAX_FIELDS 8

typedef void* rg_stream_t; /* hipStream_t */

typedef enum rg_status {
  RG_OK = 0,
  RG_EINVAL = -1,       /* bad argument (null pointer, negative size, unknown enum) */
  RG_EALIGN = -2,       /* a vector-loaded buffer is not 16-byte aligned */
  RG_ELAUNCH = -3,      /* hipGetLastError() after a launch */
  RG_EWORKSPACE = -4,   /* workspace too small */
  RG_EUNSUPPORTED = -5, /* e.g. n_fields > RG_MAX_FIELDS */
  RG_ENODEVICE = -6     /* no HIP device visible */
} rg_status;

/* Quiet-NaN bit pattern that marks "gate excluded for this field" inside packed fields.  A data NaN that
 * is NOT masked keeps its own payload and propagates like in NumPy (interpolate.py:78-82). */
#define RG_EXCLUDED_BITS 0x7FD1CE5Du

int rg_version(void);
const char* rg_last_error(void);
/* number of visible HIP devices, or RG_ENODEVICE */
int rg_device_count(void);
/* measurement aid (no reference counterpart): streams `bytes` of `buffer` with 16-byte loads and stores nothing
 * (`sink` = one float the kernel never writes in practice); bench.py times it to report the read bandwidth this
 * GPU delivers to a pure streaming kernel next to the 8 TB/s spec peak (roofline.ceiling_measured). */
int rg_stream_read_probe(const void* buffer, int64_t bytes, float* sink, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a1  antenna -> Cartesian (PyART antenna_vectors_to_cartesian, call sites radar_grid/utils.py:35-37).
 * float64 math, float32 stores.  x,y,z are [n_rays][n_gates] ray-major.
 * ------------------------------------------------------------------------------------------------- */
int rg_antenna_to_cartesian_f32(const double* ranges_m, int32_t n_gates,
                                const double* az_deg, const double* el_deg, int32_t n_rays,
                                float* x, float* y, float* z, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a3  GateFilter predicates (radar_grid/filters.py:114-258): mask_inout[g] |= pred(data[g]).
 * NaN compares false, so threshold filters never exclude NaN gates (filters.py:134).
 * ------------------------------------------------------------------------------------------------- */
typedef enum rg_gate_op {
  RG_GATE_BELOW = 0,   /* data <  a            filters.py:134 */
  RG_GATE_ABOVE = 1,   /* data >  a            filters.py:157 */
  RG_GATE_BETWEEN = 2, /* a < data < b         filters.py:182 */
  RG_GATE_OUTSIDE = 3, /* data < a || data > b filters.py:207 */
  RG_GATE_EQUAL = 4,   /* |data - a| < b       filters.py:232 */
  RG_GATE_INVALID = 5  /* NaN or Inf           filters.py:257 */
} rg_gate_op;

int rg_gate_mask_f32(const float* data, int64_t n_gates, int32_t op, float a, float b,
                     uint8_t* mask_inout, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a2/a8 prologue  mask merge (radar_grid/interpolate.py:59-64) folded into the field values.
 *   packed[g*stride + f] = (masks_host[f] && masks_host[f][g]) || (shared_mask && shared_mask[g])
 *                          ? EXCLUDED : fields_host[f][g]          for f <  n_fields
 *                          = EXCLUDED                               for f >= n_fields (padding slots)
 * fields_host / masks_host are HOST arrays of n_fields DEVICE pointers (mask entries may be NULL).
 * ------------------------------------------------------------------------------------------------- */
int rg_pack_fields_f32(int32_t n_fields, const float* const* fields_host, const uint8_t* const* masks_host,
                       const uint8_t* shared_mask, int64_t n_gates, int32_t stride, float* packed,
                       rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a8/a9  K1 csr_apply: replaces radar_grid/interpolate.py:69-104 (apply_geometry) and the per-field loop
 * of :137-140 (apply_geometry_multi) with ONE pass over the CSR for all n_fields field-volumes.
 *   out[f*n_vox + v] = sum_j w_j*val_f(g_j) / sum_j w_j   over pairs j of row v whose gate is not EXCLUDED
 *                      for field f, if that weight sum is > 0; otherwise fill_value.
 * Arithmetic: float32 products and float32 sums (as the reference under NumPy >= 2), added in a fixed,
 * launch-independent order (per 64-row segment: tiles of pairs, a few interleaved partial sums per row, combined with
 * wavefront shuffles) -- bit-reproducible run to run, not NumPy's pairwise order; only the final division is done in
 * float64 and rounded to float32.  Worst relative deviation from the reference's grids on the golden fixtures is recorded
 * by tests/test_gpu_parity.py::test_reference_grid_relative_error (a few float32 ulps).
 * the gather goes through a range-checked buffer resource of n_gates * stride * 4 bytes (which must stay below
 * 4 GiB): a gate index outside [0, n_gates) reads as 0.0 and cannot fault the GPU.
 * indptr must be non-decreasing with indptr[0] = 0 and indptr[n_vox] = n_pairs.
 * line_len: rows per grid line (nx of an nz x ny x nx grid; n_vox must be a multiple of it; <= 0 = one line of n_vox
 * rows).  A wavefront owns a segment of up to 64 consecutive rows of ONE line (a line is cut into ceil(line_len / 64)
 * balanced segments); the value only changes which rows share a wavefront -- hence the order of the float32 adds --
 * never which pairs are summed.
 * ------------------------------------------------------------------------------------------------- */
int rg_csr_apply_f32(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, const float* weights,
                     int64_t n_vox, int64_t n_pairs, int64_t line_len,
                     const float* packed, int32_t n_fields, int32_t stride, int64_t n_gates,
                     float fill_value, float* out, rg_stream_t stream);
/* same kernel with the pipeline tile selected explicitly: variant = 0 (what rg_csr_apply_f32 runs) or 128, 192, 256, 320,
 * 384, 512 pairs per step (right answers, another order of the float32 adds; used by the bit-identity tests of the compact
 * kernels).  Anything else is RG_EINVAL in the product library; the tuning / timing-only variants of earlier rounds exist
 * only in -DRG_EXPERIMENTS builds (tools/build_experiments.py). */
int rg_csr_apply_f32_ex(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, const float* weights,
                        int64_t n_vox, int64_t n_pairs, int64_t line_len,
                        const float* packed, int32_t n_fields, int32_t stride, int64_t n_gates,
                        float fill_value, float* out, int32_t variant, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a11  K3 column reduce: radar_grid/products.py:488-490 (nanmax), :533-535 (nanmin), :578-580 (nanmean)
 * over levels z_lo..z_hi (inclusive, already clipped by the caller as products.py:484-485 does).
 * out_arg (optional, MAX/MIN only): first level attaining the extremum, -1 for an all-NaN column
 * (np.nanargmax order; build-defined).
 * ------------------------------------------------------------------------------------------------- */
typedef enum rg_column_op { RG_COL_MAX = 0, RG_COL_MIN = 1, RG_COL_MEAN = 2 } rg_column_op;

int rg_column_reduce_f32(const float* grid, int32_t nz, int64_t n_xy, int32_t z_lo, int32_t z_hi, int32_t op,
                         float* out, int32_t* out_arg, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a10  K4 CAPPI lerp: radar_grid/products.py:406-412.  out = fl32(fl32(w_lo*lo) + fl32(w_hi*hi)), the
 * float32 arithmetic NumPy >= 2 performs with weak Python-float weights; NaN in either level -> NaN.
 * The scalar control flow (products.py:361-404) stays on the host.
 * ------------------------------------------------------------------------------------------------- */
int rg_cappi_lerp_f32(const float* grid, int64_t n_xy, int32_t k_lo, float w_lo, float w_hi, float* out,
                      rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a12  constant-elevation PPI: radar_grid/products.py:168-314.  Per pixel the beam height of the elevation
 * (4/3-earth model of products.py:70-87 when earth_curvature != 0, flat earth :164-165 otherwise) selects the
 * altitude; linear != 0 -> float64 lerp between the bracketing levels (`out` is double[ny*nx], products.py:276-309),
 * linear == 0 -> nearest level (`out` is float[ny*nx], products.py:262-272).  The host passes the scalars NumPy
 * evaluates once: cos_clamped = max(cos(elev), 0.01), sin(elev), tan(elev), ke_re = ke * 6371000, ke_re_sq = ke_re**2.
 * xc / yc are the float32 linspace tables of products.py:232-233.
 * ------------------------------------------------------------------------------------------------- */
int rg_elevation_ppi_f32(const float* grid, const float* xc, const float* yc, int32_t nz, int32_t ny, int32_t nx,
                         double cos_clamped, double sin_elev, double tan_elev, double ke_re, double ke_re_sq,
                         double z_min, double z_max, double z_step, int32_t earth_curvature, int32_t linear,
                         void* out, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * a6/a7  geometry builder: replaces radar_grid/compute.py:18-103 (_process_single_level) and the merge of
 * :232-272.  Membership and weights are evaluated in float64 from the float32 inputs with FP contraction
 * disabled, i.e. the reference's arithmetic: valid gate (fl32(z - radar_altitude) <= toa), d2 < r2,
 * r = max(min_radius, sqrt(x^2+y^2+z^2)*beam_factor).
 *
 * Search structure (build-specific): gates are bucketed into a uniform (x,y) cell grid and stably sorted
 * by cell, so every cell row is one contiguous run of `sorted_gates`; a wavefront tests a voxel's runs
 * 64 gates at a time.
 * ------------------------------------------------------------------------------------------------- */
typedef struct rg_cellgrid {
  double x0, y0;         /* lower corner of cell (0,0) */
  double inv_cx, inv_cy; /* 1 / cell size */
  double z_lo, z_hi;     /* gates with z_rel outside [z_lo, z_hi] cannot reach any voxel and are dropped */
  int32_t ncx, ncy;
  int32_t levels;        /* 0 / 1: one list for all grid levels (cell_start[ncx*ncy + 1]); nz: one list PER LEVEL, holding only the
                          * gates that can reach that level (rg_geom_bin_gates_levels_f32; cell_start[nz*ncx*ncy + 1], the cells
                          * of level iz at iz*ncx*ncy ..) -- a third of the candidates per voxel block */
  int32_t level0;        /* per-level lists: the grid level the call's zc[0] is (a call may cover levels level0 .. level0 + nz - 1
                          * of the binned grid by passing zc + level0); 0 otherwise */
} rg_cellgrid;

typedef struct rg_gate4 { float x, y, z; int32_t index; } rg_gate4; /* 16 bytes: one dwordx4 per candidate */

/* RG_W_NEAREST is the reference's 'nearest' = uniform mean over the ROI (radar_grid/compute.py:86-87).
 * RG_W_CLOSEST (rg_roi_grid_f32 only) takes the value of the single closest non-excluded gate inside the ROI -- the
 * selection rule of PyART's map_gates_to_grid(weighting_function='nearest') that radar_processor/processor.py:152-163
 * uses behind the 3-D grid cache; PyART is not in the reference tree, so this mode is parity-unpinned. */
typedef enum rg_weighting { RG_W_BARNES2 = 0, RG_W_CRESSMAN = 1, RG_W_NEAREST = 2, RG_W_CLOSEST = 3 } rg_weighting;

/* bytes of scratch rg_geom_bin_gates_f32 needs for n_gates gates */
int64_t rg_geom_bin_workspace_bytes(int64_t n_gates, int32_t ncx, int32_t ncy);

/* sorted_gates[0 .. n_binned) in (cell, gate index) order (buffer of n_gates records);
 * cell_start[ncx*ncy + 1], with cell_start[ncx*ncy] = n_binned = number of gates that were kept */
int rg_geom_bin_gates_f32(const float* gate_x, const float* gate_y, const float* gate_z, int64_t n_gates,
                          float radar_altitude, float toa, const rg_cellgrid* cells_host,
                          rg_gate4* sorted_gates, int32_t* cell_start,
                          void* workspace, int64_t workspace_bytes, rg_stream_t stream);

/* Per-level lists: gate g is listed under level iz iff |z_rel(g) - zc[iz]| <= R_g, R_g = max(min_radius, beam_factor * |g| /
 * (1 - beam_factor)) (slightly inflated) -- a bound on the radius of influence of ANY voxel the gate can be a neighbour of
 * (compute.py:46-47: r_v = max(min_radius, |v| * beam_factor) and |v| <= |g| + r_v), so no neighbour is lost; needs
 * 0 <= beam_factor < 1.  rg_geom_bin_levels_count writes the number of (gate, level) entries to *total (device memory);
 * rg_geom_bin_gates_levels_f32 then fills sorted_gates[n_entries] in (level, cell, gate index) order and
 * cell_start[nz*ncx*ncy + 1] (cells->levels must equal nz).  zc: the float32 level coordinates (device). */
int rg_geom_bin_levels_count(const float* gate_x, const float* gate_y, const float* gate_z, int64_t n_gates,
                             float radar_altitude, float toa, const rg_cellgrid* cells_host, const float* zc, int32_t nz,
                             double min_radius, double beam_factor, int64_t* total, rg_stream_t stream);
int64_t rg_geom_bin_levels_workspace_bytes(int64_t n_gates, int64_t n_entries, int64_t n_cells_total);
int rg_geom_bin_gates_levels_f32(const float* gate_x, const float* gate_y, const float* gate_z, int64_t n_gates,
                                 float radar_altitude, float toa, const rg_cellgrid* cells_host, const float* zc, int32_t nz,
                                 double min_radius, double beam_factor, int64_t n_entries, rg_gate4* sorted_gates,
                                 int32_t* cell_start, void* workspace, int64_t workspace_bytes, rg_stream_t stream);

/* counts[v] = number of gates within voxel v's radius of influence; xc/yc/zc are the float32 linspace
 * coordinate tables of radar_grid/compute.py:184-186 (device pointers). */
int rg_geom_count_f32(const rg_gate4* sorted_gates, const int32_t* cell_start, const rg_cellgrid* cells_host,
                      const float* xc, const float* yc, const float* zc, int32_t nz, int32_t ny, int32_t nx,
                      double min_radius, double beam_factor, int32_t* counts, rg_stream_t stream);

/* indptr[0..n] = exclusive prefix sum of counts[0..n) widened to int64 (indptr[n] = total pairs) */
int64_t rg_scan_workspace_bytes(int64_t n);
int rg_scan_counts_i64(const int32_t* counts, int64_t n, int64_t* indptr, void* workspace,
                       int64_t workspace_bytes, rg_stream_t stream);

/* second pass: writes gate_idx / weights of every row at indptr[v].. in (cell row, gate index) order */
int rg_geom_fill_f32(const rg_gate4* sorted_gates, const int32_t* cell_start, const rg_cellgrid* cells_host,
                     const float* xc, const float* yc, const float* zc, int32_t nz, int32_t ny, int32_t nx,
                     double min_radius, double beam_factor, int32_t weighting, const int64_t* indptr,
                     int32_t* gate_idx, float* weights, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K2  fused on-the-fly gridding: the neighbour search of rg_geom_count/fill_f32 fused with the masked
 * weighted mean of rg_csr_apply_f32 -- radar_grid/compute.py:46-91 + radar_grid/interpolate.py:69-104 in
 * one kernel, no CSR in memory (for grids whose pair count makes the CSR pointless or impossible,
 * SURVEY.md F6).  Same neighbour sets as the builder, weights evaluated in float32 (|rel err| < 2e-6 vs the builder's
 * float64-exact ones), float32 sums per voxel lane; results differ from the CSR path by that and by summation order.  `packed` is the rg_pack_fields_f32 layout over the same gate numbering as the
 * gates that were binned.
 * ------------------------------------------------------------------------------------------------- */
int rg_roi_grid_f32(const rg_gate4* sorted_gates, const int32_t* cell_start, const rg_cellgrid* cells_host,
                    const float* xc, const float* yc, const float* zc, int32_t nz, int32_t ny, int32_t nx,
                    double min_radius, double beam_factor, int32_t weighting,
                    const float* packed, int32_t n_fields, int32_t stride, float fill_value, float* out,
                    rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * (f)3  processor-style collapse of the cached 3-D grid to the 2-D product plane:
 * radar_processor/processor.py:480-551 (collapse_grid_to_2d) and radar_processor/utils.py:336-387
 * (collapse_field_3d_to_2d).  'cappi' (nearest level, :530-533) and 'colmax' (:534-535) are
 * rg_column_reduce_f32 over one level / all levels; 'ppi' (:512-528) is the entry below: per pixel
 * r = sqrt(x^2 + y^2), z_target = r*sin_elev + r^2 / two_re (two_re = 2 * 8.49e6 m), level = first argmin of
 * |z_target - z[k]|, all in float64 as NumPy evaluates it; out[p] = grid[level][p].  The host passes
 * sin_elev = sin(deg2rad(elevation)).  x[nx], y[ny], z[nz] are float64 device tables.  out_level may be NULL.
 * A masked voxel is a NaN (the cache package is masked_invalid data).
 * ------------------------------------------------------------------------------------------------- */
int rg_collapse_ppi_f32(const float* grid, const double* x, const double* y, const double* z, int32_t nz, int32_t ny,
                        int32_t nx, double sin_elev, double two_re, float* out, int32_t* out_level,
                        rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * (f)3  threshold masks on a 2-D plane: the re-mask of collapse_grid_to_2d (processor.py:541-546:
 * masked_less_equal / masked_less against vmin) and _apply_filter_masks (processor.py:802-886: visual filters on
 * the plotted field, cross-field and QC filters on cached QC planes), all OR-ed in one pass.  A test drops pixel p
 * when  q < lo (RG_TEST_LO),  q <= lo (RG_TEST_LO | RG_TEST_LO_INCLUSIVE),  q > hi (RG_TEST_HI)  or q is +-inf / NaN
 * (RG_TEST_NONFINITE, the np.ma.masked_invalid of processor.py:541), with
 * q = plane[p] (or src[p] when plane is NULL), compared in float32 as NumPy compares a float32 array with a Python
 * float; NaN never compares true.  A pixel counts as already masked when src_mask[p] != 0 or, without a src_mask,
 * when src[p] is NaN.  out[p] = dropped or masked ? NaN : src[p]  (NULL to skip); out_mask[p] = dropped or masked
 * (NULL to skip).  `tests` is a HOST
 * array of at most RG_MAX_PLANE_TESTS entries whose plane pointers are device pointers.
 * ------------------------------------------------------------------------------------------------- */
#define RG_MAX_PLANE_TESTS 12
typedef enum rg_plane_test_flags {
  RG_TEST_LO = 1, RG_TEST_HI = 2, RG_TEST_LO_INCLUSIVE = 4, RG_TEST_NONFINITE = 8
} rg_plane_test_flags;
typedef struct rg_plane_test {
  const float* plane; /* NULL: test the source plane itself */
  float lo, hi;
  int32_t flags;
} rg_plane_test;

int rg_plane_filter_f32(const float* src, const uint8_t* src_mask, int64_t n, const rg_plane_test* tests,
                        int32_t n_tests, float* out, uint8_t* out_mask, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * (f)4  colormap -> RGBA uint8: radar_grid/geotiff.py:70-145 (apply_colormap_to_array), i.e. matplotlib's
 * Normalize(vmin, vmax, clip=True) followed by Colormap.__call__ and (rgba * 255).astype(uint8).
 *
 * rg_nan_minmax: over the pixels that are not no-data (no-data = equal to `fill` when has_fill, NaN otherwise)
 * out[0] = min and out[1] = max ignoring NaN, out[2] = how many such non-NaN pixels there are, out[3] = how many
 * pixels are not no-data (geotiff.py:111-125: np.nanmin / np.nanmax of valid_data and len(valid_data)); +inf / -inf /
 * 0 when there are none.  `data` is float32 or float64 (data_is_f64), `workspace` needs RG_MINMAX_WORKSPACE_BYTES,
 * out is double[4] on the device.
 *
 * rg_colormap_rgba: per pixel v = minimum(maximum(x, vmin), vmax); v -= vmin; v /= (vmax - vmin); v *= n_lut;
 * v == n_lut -> n_lut - 1; index = trunc(v), or n_lut (under) when v < 0, n_lut + 1 (over) when v >= n_lut,
 * n_lut + 2 (bad) when NaN; rgba = lut[index]; alpha = 0 where x is no-data (NaN, or == fill when has_fill; the
 * comparison with fill is made in the data's dtype).  The arithmetic runs in float64 for float32 and float64 data
 * alike: Normalize stores its limits as Python floats, so np.clip promotes a float32 array to float64 before the
 * in-place operations (matplotlib 3.10 / NumPy 2).  vmin == vmax selects entry 0 for every pixel (Normalize fills 0);
 * vmin > vmax is RG_EINVAL.  lut: uint8 [n_lut + 3][4] on the device = (colormap table * 255) truncated,
 * n_lut <= RG_MAX_LUT.  out: uint8 [n][4].
 * ------------------------------------------------------------------------------------------------- */
#define RG_MINMAX_WORKSPACE_BYTES 32768
#define RG_MAX_LUT 4093
int rg_nan_minmax(const void* data, int32_t data_is_f64, int64_t n, int32_t has_fill, double fill, void* workspace,
                  double* out, rg_stream_t stream);
int rg_colormap_rgba(const void* data, int32_t data_is_f64, int64_t n, double vmin, double vmax, int32_t has_fill,
                     double fill, const uint8_t* lut, int32_t n_lut, uint8_t* out, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * GridFilter (radar_grid/filters.py:609-780): value filters on a product plane after interpolation.
 * out[p] = hit(p) ? fill_value : src[p]  with  hit = (flags & RG_TEST_LO and v < lo) or (flags & RG_TEST_HI and v > hi)
 * or (flags & RG_TEST_NONFINITE and v is NaN / +-inf) or (mask != NULL and mask[p] != 0); comparisons in the
 * plane's dtype (float32 when data_is_f64 == 0, as NumPy compares a float32 array with a Python float), so a NaN
 * pixel is only ever replaced by the NONFINITE test or the mask.  src == out is allowed.
 * ------------------------------------------------------------------------------------------------- */
int rg_grid_filter(const void* src, int32_t data_is_f64, int64_t n, int32_t flags, double lo, double hi,
                   const uint8_t* mask, double fill_value, void* out, rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K1c  csr_apply over a compact device copy of the CSR -- same results as rg_csr_apply_f32 for every field count,
 * bit for bit, from 6 instead of 8 streamed bytes per pair and one gather per DISTINCT gate of a chunk instead of one
 * per pair.  (Its packed form, rg_csr_compact_apply_packed_f32 below, is the faster one where the weights allow it.)  The grid is planes x lines x rows (n_vox = n_planes * lines_per_plane * line_len; nz x ny x nx for a radar
 * grid).  A chunk is a 2-D patch: segment sx (one of the ceil(line_len / 64) balanced pieces of a line, the unit one
 * wavefront of rg_csr_apply_f32 owns) of the RG_COMPACT_LINES consecutive lines yg*RG_COMPACT_LINES.. of one plane; chunks are
 * numbered c = (plane * ceil(lines_per_plane / RG_COMPACT_LINES) + yg) * ceil(line_len / 64) + sx.  Chunk c lists its
 * distinct gate indices in dict[dict_ptr[c] .. dict_ptr[c+1]) (at most 65536 of them, any order) and pair p of one of
 * its rows stores local_idx[p] = position of its gate in that list.  A chunk with MORE distinct gates is stored split:
 * dict[dict_ptr[c] + w], w < RG_COMPACT_LINES, is the offset (from dict_ptr[c]) of a dictionary of its own for line w's
 * segment, and positions refer to that; such a chunk is recognised by dict_ptr[c+1] - dict_ptr[c] > 65536.
 * indptr and weights are those of the standard CSR
 * (radar_grid/geometry.py:46-52), which stays the interchange format; the compact arrays are derived from it on the
 * device (radar_processor_amd/grid_geometry.py: CompactCSR).  `packed` / n_fields / stride: as rg_csr_apply_f32.
 * window_cap: how many dictionary entries (of `stride` floats) a workgroup keeps in LDS (<= RG_COMPACT_MAX_WINDOW,
 * clamped to what fits next to the kernel's own LDS); chunks with a longer dictionary gather per pair from memory, so
 * any value is correct and the choice only affects speed.  tile: pairs per pipeline step, 0 = default (= the tile of
 * rg_csr_apply_f32 for the same field count; other values change the order of the float32 adds).  Timing-only
 * ablations (901-909, results wrong by construction) and the block-rotation override exist only in -DRG_EXPERIMENTS builds
 * (tools/build_experiments.py); the product library answers RG_EINVAL.
 * line_len <= 0 means one line of n_vox rows, lines_per_plane <= 0 one plane.
 * ------------------------------------------------------------------------------------------------- */
#ifndef RG_COMPACT_LINES
#define RG_COMPACT_LINES 4
#endif
#define RG_COMPACT_MAX_WINDOW 8192
/* Block -> chunk map of the apply kernels: block b = (line group grp, column col) takes chunk column
 * (col + (grp * RG_COMPACT_ROTATION) mod nsx) mod nsx of its line group (grp counted through all planes, 32-bit unsigned
 * arithmetic), so that every XCD sees every column of the grid.  Part of the record layout when the records are stored
 * in dispatch order (below). */
#define RG_COMPACT_ROTATION 5
int rg_csr_compact_apply_f32(const void* indptr, int32_t indptr_is_i64, const uint16_t* local_idx, const float* weights,
                             const int64_t* dict_ptr, const int32_t* dict, int64_t n_vox, int64_t n_pairs,
                             int64_t line_len, int64_t lines_per_plane, const float* packed, int32_t n_fields,
                             int32_t stride, int64_t n_gates, float fill_value, float* out, int32_t window_cap,
                             int32_t tile, rg_stream_t stream);

/* Packed pair stream of the compact copy: positions and weights of three consecutive pairs of a segment in one 16-byte
 * record -- 5.33 bytes per pair instead of 6, streamed with one 16-byte load per lane.
 *   record = [w0:26 | p2 bits 0-5] [w1:26 | p2 bits 6-11] [w2:26 | p2 bits 12-15] [p0:16 | p1:16]   (4 x uint32)
 *   weight code = float32 bits of the weight - w_base; w_base has its low 26 bits clear (an exponent that is a multiple of
 *   8, shifted left by 23), so adding it back is an OR; the caller guarantees that every code fits 26 bits (all weights
 *   positive, exponents w_base >> 23 .. (w_base >> 23) + 7), which makes the coding lossless.
 * A segment (the <= 64 rows one wavefront owns) holds ceil(pairs / 3) records, fewer than 2^27; where they lie is
 * rec_order's business -- rec_ptr[s] .. rec_ptr[s+1] are the records of the segment in SLOT s:
 *   RG_REC_ORDER_SEGMENT   s = seg = line * ceil(line_len / 64) + sx (line-major; rec_ptr has segments + 1 entries);
 *   RG_REC_ORDER_DISPATCH  s = b * RG_COMPACT_LINES + w for the segment wavefront w of block b reads (block -> chunk map:
 *                          RG_COMPACT_ROTATION above; rec_ptr has chunks * RG_COMPACT_LINES + 1 entries, slots of lines
 *                          past the end of a plane are empty).  The four segments of a workgroup are then neighbours in
 *                          the stream and consecutive workgroups read consecutive stretches of it: whatever the physical
 *                          placement of the array, the launch reads it as ONE moving front.
 * rec_ptr is built by the caller.  rg_csr_compact_pack fills `records` from local_idx + weights (error_flag: 1 = rec_ptr
 * inconsistent with indptr, 2 = a weight outside the code, 4 = a segment with 2^27 records or more); for a slab of whole
 * planes of a larger grid pass the slab's indptr / n_rows, rec_ptr + the slab's first slot and plane0 = its first plane.
 * rg_csr_compact_apply_packed_f32 grids 1-8 fused fields (row-wise kernel; the tile kernel over the records: 1-4) through
 * the records (interpolate.py:69-104 / :137-140, the same masked weighted mean as rg_csr_apply_f32: float32 products and sums,
 * float64 only in the final division):
 *   tile = 0    the ROW-WISE kernel: the lanes of a row read the row's records straight from memory and sum them in
 *               registers (no LDS tile), L = 2^k lanes per row chosen per segment from its mean row length.  The order
 *               of the float32 adds is fixed by the geometry and the field count alone (reproducible run to run, on any
 *               window_cap), but it is not the order of rg_csr_apply_f32: the two agree to float32 rounding (the bar of
 *               the parity tests: 1e-5 relative + 1e-5 * max|field|), not bit for bit.  This is the fast path:
 *               1.0-1.5 ms for 1-4 fields on BASELINE config 2 where the tile kernels need 1.1-3.3 ms.  Five to eight
 *               fields (stride 8; eight volumes of one geometry in ONE pass over the records: 15.5 ms on the bench grid
 *               against 2 x 10.5 for two passes of four) keep 40 bytes of LDS per window entry: worth it where the
 *               geometry's window is <= 768 entries (the Python layer decides: gridding.fields_per_pass).
 *   tile = 384  the TILE kernel of rg_csr_compact_apply_f32 over the same records: the results of rg_csr_apply_f32 for the
 *               same fields, bit for bit (576 / 768: single-field tuning variants of it).
 *   tile = 2000 + h   row-wise with a diagnostic lane split: h = 1, 2, 4 .. 64 lanes per row, or h = 71 .. 99 = 70 + t to
 *               aim for t records per lane and row (another split = another order of the adds; right answers).
 *   tile = 2100 .. 2199 (timing-only ablations: no store, no record loads ... -- WRONG results by construction) and
 *   2201 .. 2264 (several chunks per workgroup) are compiled only into -DRG_EXPERIMENTS builds (tools/build_experiments.py);
 *   the product library answers RG_EINVAL.
 * window_cap as for rg_csr_compact_apply_f32; the row-wise kernel keeps one entry more (an all-EXCLUDED sentinel). */
#define RG_REC_ORDER_SEGMENT 0
#define RG_REC_ORDER_DISPATCH 1
int rg_csr_compact_pack(const void* indptr, int32_t indptr_is_i64, const uint16_t* local_idx, const float* weights,
                        int64_t n_rows, int64_t line_len, int64_t lines_per_plane, const int64_t* rec_ptr,
                        int32_t rec_order, int64_t plane0, uint32_t w_base, void* records, int32_t* error_flag,
                        rg_stream_t stream);
int rg_csr_compact_apply_packed_f32(const void* indptr, int32_t indptr_is_i64, const void* records,
                                    const int64_t* rec_ptr, int32_t rec_order, uint32_t w_base, const int64_t* dict_ptr,
                                    const int32_t* dict, int64_t n_vox, int64_t n_pairs, int64_t line_len,
                                    int64_t lines_per_plane, const float* packed, int32_t n_fields, int32_t stride,
                                    int64_t n_gates, float fill_value, float* out, int32_t window_cap, int32_t tile,
                                    rg_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * K1p  the row-wise kernel in COLUMN MODE, with an optional products epilogue.
 * Replaces, in one pass over the packed records: radar_grid/interpolate.py:69-104 (apply_geometry) / :137-140
 * (apply_geometry_multi) and -- when the caller keeps 2-D products only -- the read-back of the 3-D grid by
 * radar_grid/products.py:462-490 (column_max over a level window) and :361-412 (CAPPI: the two bracketing levels).
 * A workgroup walks the chunks (plane z, line group yg, segment sx) of ONE (yg, sx) through the planes of its level piece,
 * each chunk exactly as rg_csr_compact_apply_packed_f32 (tile = 0) processes it, so lane == row sees its (y, x) column in
 * ascending level order.  Measured: it pays from four field-volumes per pass on (the store it saves is about what walking
 * columns instead of sweeping the grid costs) and always in memory -- no n_fields x 4 x n_vox bytes of grid.
 * Arguments up to fill_value: as rg_csr_compact_apply_packed_f32 (either record order is accepted).
 *   out           [n_fields][n_vox] 3-D grids, or NULL: the grids are not stored (products only);
 *   level_planes  [n_fields][n_keep][ny*nx] or NULL: planes keep_lo .. keep_lo + n_keep - 1 of every field's grid (what a
 *                 CAPPI blends with rg_cappi_lerp_f32, products.py:406-412, or returns as is, :378,386);
 *   col_max / col_arg  [n_fields][ny*nx] or NULL: NaN-ignoring maximum over planes col_lo .. col_hi and the first plane
 *                 attaining it (-1: all NaN) -- the contract of rg_column_reduce_f32(RG_COL_MAX), bit for bit, because lane
 *                 == (y, x) column sees its levels in ascending order;
 *   z_pieces      1 .. planes: a column is cut into this many level ranges, one workgroup each (more workgroups for
 *                 small grids; any value gives the same results).  With col_max and z_pieces > 1 the partial planes live
 *                 in `workspace` (rg_csr_columns_workspace_bytes) and are merged in ascending level order by a second
 *                 small kernel on the same stream;
 *   order         NULL, or a permutation of 0 .. z_pieces * ceil(ny / RG_COMPACT_LINES) * ceil(nx / 64) - 1: workgroup b
 *                 takes item order[b] = piece * columns + (yg * ceil(nx / 64) + c) -- the caller may list the heaviest
 *                 columns first (speed only);
 *   lanes_hint    0, or the diagnostic lane split of the row-wise kernel (1 .. 64 lanes per row, 70 + t).
 * The grids are the same BITS as rg_csr_compact_apply_packed_f32 (tile = 0) for the same lanes_hint (same order of the
 * float32 adds: geometry and field count alone fix it).  At least one of out / level_planes / col_max must be given.
 * ------------------------------------------------------------------------------------------------- */
int64_t rg_csr_columns_workspace_bytes(int64_t lines_per_plane, int64_t line_len, int32_t n_fields, int32_t z_pieces);
int rg_csr_compact_apply_columns_f32(const void* indptr, int32_t indptr_is_i64, const void* records,
                                     const int64_t* rec_ptr, int32_t rec_order, uint32_t w_base, const int64_t* dict_ptr,
                                     const int32_t* dict, int64_t n_vox, int64_t n_pairs, int64_t line_len,
                                     int64_t lines_per_plane, const float* packed, int32_t n_fields, int32_t stride,
                                     int64_t n_gates, float fill_value, float* out, float* level_planes, int32_t keep_lo,
                                     int32_t n_keep, float* col_max, int32_t* col_arg, int32_t col_lo, int32_t col_hi,
                                     int32_t window_cap, int32_t z_pieces, const int32_t* order, void* workspace,
                                     int64_t workspace_bytes, int32_t lanes_hint, rg_stream_t stream);

/* number of chunks of a grid of n_rows rows (negative rg_status when the sizes do not factor) */
int64_t rg_csr_compact_chunks(int64_t n_rows, int64_t line_len, int64_t lines_per_plane);

/* Building the compact copy from a standard CSR, or from a slab of whole planes of it: pass indptr + first row of the
 * slab, n_rows of the slab, and gate_idx / local_idx pointers such that element p is ABSOLUTE pair p (the row pointers
 * hold absolute pair positions).  rg_csr_compact_count: chunk_counts[c] = dictionary entries of chunk c (header
 * included for a split chunk; >= 0x40000000 = a single segment references more than 65536 gates: not compactable),
 * chunk_rounds[c] = hashing rounds the chunk needed and its split flag (opaque, handed to the fill pass).  The caller turns the counts into dict_ptr (rg_scan_counts_i64, chunk_counts needs one
 * spare entry) and calls rg_csr_compact_fill, which writes dict[dict_ptr[c] ..] and local_idx[p] for every pair.
 * Dictionary order is unspecified.  *error_flag (device int32, zeroed by the caller) becomes non-zero when the fill
 * pass meets inputs that differ from what the count pass saw (gate_idx modified in between, wrong chunk_rounds):
 * every table walk is bounded, so such a mismatch is reported instead of hanging the GPU. */
int rg_csr_compact_count(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, int64_t n_rows,
                         int64_t line_len, int64_t lines_per_plane, int32_t* chunk_counts, uint8_t* chunk_rounds,
                         rg_stream_t stream);
int rg_csr_compact_fill(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, int64_t n_rows,
                        int64_t line_len, int64_t lines_per_plane, const int64_t* dict_ptr,
                        const uint8_t* chunk_rounds, int32_t* dict, uint16_t* local_idx, int32_t* error_flag,
                        rg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RADARGRID_HIP_H */
