// Geometry builder: gate -> voxel neighbour search + weights, on the GPU.
// Replaces radar_grid/compute.py:18-103 (_process_single_level: per-level cKDTree + Python loop over
// voxels) and the CSR merge of compute.py:232-272.
//
// THIS TRANSLATION UNIT IS COMPILED WITH -ffp-contract=off: membership `d2 < r2` and the weights are the
// reference's float64 expressions evaluated operation by operation (compute.py:46-47,69-74,82-87); a fused
// multiply-add would round differently from NumPy and could flip a rim gate.
//
// Search structure: gates that pass the toa test are bucketed into a uniform (x,y) cell grid that covers
// the voxel grid grown by the largest ROI, then stably radix-sorted by cell id (rocPRIM).  All gates of
// one cell row [cy][cx0..cx1] are therefore ONE contiguous run of 16-byte records, and the runs a voxel
// has to test are a provable superset of its ROI ball: cell_coord() is monotone in the coordinate, so a
// gate with x - r <= gx <= x + r lies in a cell between cell(x - r) and cell(x + r).
//
// One such list serves all grid levels (rg_geom_bin_gates_f32), or -- the default of the Python layer -- every level gets its OWN
// list holding only the gates that can reach it (rg_geom_bin_gates_levels_f32, see LevelArgs below): the search kernels then
// stream a third of the candidates.
//
// This file holds the binning (bucket + stable radix sort + cell table) and the prefix sum of the row lengths; the
// count / fill passes over the voxels run on the voxel-blocked search kernel of rg_roi_grid.hip, shared with the
// fused gridder.  A row comes out in (cell row, sorted position) order -- deterministic, no atomics.
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "rg_common.hpp"
#include "rg_roi_search.hpp"

namespace {

using namespace rg::roi;

// ---- binning ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(rg::kBlock) void bin_keys_kernel(const float* __restrict__ gx, const float* __restrict__ gy,
                                                              const float* __restrict__ gz, long n, float radar_alt,
                                                              float toa, Cells c, unsigned* __restrict__ keys,
                                                              unsigned* __restrict__ vals) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // compute.py:182 -- float32 subtraction (weak Python scalar under NumPy >= 2); compute.py:193 -- z_rel <= toa
  const float z_rel = __fsub_rn(gz[i], radar_alt);
  const double x = (double)gx[i], y = (double)gy[i], z = (double)z_rel;
  const double tx = cell_coord(x, c.x0, c.inv_cx), ty = cell_coord(y, c.y0, c.inv_cy);
  // NaN coordinates fail every comparison and are dropped
  const bool keep = z_rel <= toa && z >= c.z_lo && z <= c.z_hi && tx >= 0.0 && tx < (double)c.ncx && ty >= 0.0 &&
                    ty < (double)c.ncy;
  keys[i] = keep ? (unsigned)((int)ty * c.ncx + (int)tx) : (unsigned)(c.ncx * c.ncy);
  vals[i] = (unsigned)i;
}

// cell_start[c] = first sorted position whose key >= c, c = 0..n_cells (cell_start[n_cells] = #binned gates)
__global__ __launch_bounds__(rg::kBlock) void cell_start_kernel(const unsigned* __restrict__ keys, long n,
                                                                unsigned n_cells, int* __restrict__ cell_start) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_cells) return;
  long lo = 0, hi = n;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (keys[mid] < c) lo = mid + 1; else hi = mid;
  }
  cell_start[c] = (int)lo;
}

__global__ __launch_bounds__(rg::kBlock) void gather_sorted_kernel(const unsigned* __restrict__ keys,
                                                                   const unsigned* __restrict__ vals, long n,
                                                                   unsigned n_cells, const float* __restrict__ gx,
                                                                   const float* __restrict__ gy,
                                                                   const float* __restrict__ gz, float radar_alt,
                                                                   rg_gate4* __restrict__ sorted) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || keys[j] >= n_cells) return;
  const unsigned g = vals[j];
  rg_gate4 r;
  r.x = gx[g];
  r.y = gy[g];
  r.z = __fsub_rn(gz[g], radar_alt);
  r.index = (int)g;
  sorted[j] = r;
}

// ---- per-level lists ---------------------------------------------------------------------------------
// Gate g can only be a neighbour of voxels whose radius of influence is below R_g = max(min_radius, bf * |g| / (1 - bf)):
// r_v = max(min_radius, bf * |v|) (compute.py:46-47) and |v| <= |g| + |g - v| < |g| + r_v, so r_v > min_radius implies
// r_v (1 - bf) < bf |g|.  A neighbour needs |z_g - z_v| <= |g - v| < r_v <= R_g: the gate is listed under the levels within
// R_g (inflated by 1e-6 + 1 mm against the rounding of this very computation) of its height, and under no other.
struct LevelArgs {
  const float* zc;
  int nz;
  double min_radius, k;      // k = bf / (1 - bf)
};

__device__ __forceinline__ bool gate_cell(const float* __restrict__ gx, const float* __restrict__ gy,
                                          const float* __restrict__ gz, long i, float radar_alt, float toa, const Cells& c,
                                          unsigned* cell, double* zrel, double* reach, const LevelArgs& la) {
  const float z_rel = __fsub_rn(gz[i], radar_alt);      // compute.py:182, as bin_keys_kernel
  const double x = (double)gx[i], y = (double)gy[i], z = (double)z_rel;
  const double tx = cell_coord(x, c.x0, c.inv_cx), ty = cell_coord(y, c.y0, c.inv_cy);
  const bool keep = z_rel <= toa && z >= c.z_lo && z <= c.z_hi && tx >= 0.0 && tx < (double)c.ncx && ty >= 0.0 &&
                    ty < (double)c.ncy;
  if (!keep) return false;
  *cell = (unsigned)((int)ty * c.ncx + (int)tx);
  *zrel = z;
  const double norm = sqrt(x * x + y * y + z * z);
  *reach = fmax(la.min_radius, norm * la.k) * (1.0 + 1e-6) + 1e-3;
  return true;
}

__global__ __launch_bounds__(rg::kBlock) void level_count_kernel(const float* __restrict__ gx, const float* __restrict__ gy,
                                                                 const float* __restrict__ gz, long n, float radar_alt,
                                                                 float toa, Cells c, LevelArgs la,
                                                                 unsigned* __restrict__ counts,
                                                                 unsigned long long* __restrict__ total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned cnt = 0;
  if (i < n) {
    unsigned cell;
    double z, reach;
    if (gate_cell(gx, gy, gz, i, radar_alt, toa, c, &cell, &z, &reach, la)) {
      for (int iz = 0; iz < la.nz; ++iz) cnt += fabs(z - (double)la.zc[iz]) <= reach ? 1u : 0u;
    }
    if (counts) counts[i] = cnt;
  }
  if (total) {      // one atomic per wavefront
    unsigned long long sum = cnt;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sum += __shfl_xor(sum, m, 64);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(total, sum);
  }
}

__global__ __launch_bounds__(rg::kBlock) void level_expand_kernel(const float* __restrict__ gx, const float* __restrict__ gy,
                                                                  const float* __restrict__ gz, long n, float radar_alt,
                                                                  float toa, Cells c, LevelArgs la,
                                                                  const unsigned* __restrict__ offsets,
                                                                  unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned cell;
  double z, reach;
  if (!gate_cell(gx, gy, gz, i, radar_alt, toa, c, &cell, &z, &reach, la)) return;
  unsigned pos = offsets[i];
  const unsigned per_level = (unsigned)c.ncx * (unsigned)c.ncy;
  for (int iz = 0; iz < la.nz; ++iz) {
    if (fabs(z - (double)la.zc[iz]) <= reach) {
      keys[pos] = (unsigned)iz * per_level + cell;
      vals[pos] = (unsigned)i;
      ++pos;
    }
  }
}

struct WidenI32 {
  __host__ __device__ long long operator()(int v) const { return (long long)v; }
};

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t sort_temp_bytes(size_t n, unsigned end_bit) {
  size_t bytes = 0;
  unsigned* nul = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, n, 0u, end_bit, (hipStream_t)0, false);
  return bytes;
}

unsigned key_bits(unsigned n_cells) {
  unsigned b = 1;
  while (b < 32 && (1ull << b) <= n_cells) ++b;
  return b;
}

}  // namespace

extern "C" int64_t rg_geom_bin_workspace_bytes(int64_t n_gates, int32_t ncx, int32_t ncy) {
  if (n_gates < 0 || ncx < 1 || ncy < 1) return RG_EINVAL;
  const size_t n = (size_t)n_gates;
  const size_t arr = round_up(n * sizeof(unsigned), 256);
  return (int64_t)(4 * arr + round_up(sort_temp_bytes(n, key_bits((unsigned)ncx * (unsigned)ncy)), 256) + 256);
}

extern "C" int rg_geom_bin_gates_f32(const float* gate_x, const float* gate_y, const float* gate_z, int64_t n_gates,
                                     float radar_altitude, float toa, const rg_cellgrid* cells_host,
                                     rg_gate4* sorted_gates, int32_t* cell_start, void* workspace,
                                     int64_t workspace_bytes, rg_stream_t stream) {
  RG_REQUIRE(cells_host && cell_start, RG_EINVAL, "rg_geom_bin_gates_f32: null pointer");
  RG_REQUIRE(n_gates >= 0 && n_gates <= 0x7FFFFFFFL, RG_EINVAL, "rg_geom_bin_gates_f32: n_gates out of range");
  RG_REQUIRE(n_gates == 0 || (gate_x && gate_y && gate_z && sorted_gates && workspace), RG_EINVAL,
             "rg_geom_bin_gates_f32: null pointer");
  RG_REQUIRE(cells_host->ncx >= 1 && cells_host->ncy >= 1 && (long)cells_host->ncx * cells_host->ncy < 0x7FFFFFFFL,
             RG_EINVAL, "rg_geom_bin_gates_f32: bad cell grid");
  RG_REQUIRE(rg::aligned16(sorted_gates), RG_EALIGN, "rg_geom_bin_gates_f32: sorted_gates must be 16-byte aligned");
  const int64_t need = rg_geom_bin_workspace_bytes(n_gates, cells_host->ncx, cells_host->ncy);
  RG_REQUIRE(workspace_bytes >= need, RG_EWORKSPACE, "rg_geom_bin_gates_f32: workspace %lld < %lld bytes",
             (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  const Cells c = to_cells(cells_host);
  const unsigned n_cells = (unsigned)c.ncx * (unsigned)c.ncy;
  const size_t n = (size_t)n_gates;
  const size_t arr = round_up(n * sizeof(unsigned), 256);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  unsigned* keys_in = reinterpret_cast<unsigned*>(base);
  unsigned* vals_in = reinterpret_cast<unsigned*>(base + arr);
  unsigned* keys_out = reinterpret_cast<unsigned*>(base + 2 * arr);
  unsigned* vals_out = reinterpret_cast<unsigned*>(base + 3 * arr);
  void* temp = base + 4 * arr;
  if (n > 0) {
    const dim3 grid((unsigned)((n + rg::kBlock - 1) / rg::kBlock)), block(rg::kBlock);
    hipLaunchKernelGGL(bin_keys_kernel, grid, block, 0, s, gate_x, gate_y, gate_z, (long)n, radar_altitude, toa, c,
                       keys_in, vals_in);
    const unsigned bits = key_bits(n_cells);
    size_t temp_bytes = sort_temp_bytes(n, bits);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, s, false);
    RG_REQUIRE(e == hipSuccess, RG_ELAUNCH, "rg_geom_bin_gates_f32: radix sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(gather_sorted_kernel, grid, block, 0, s, keys_out, vals_out, (long)n, n_cells, gate_x, gate_y,
                       gate_z, radar_altitude, sorted_gates);
  }
  hipLaunchKernelGGL(cell_start_kernel, dim3((n_cells + 1 + rg::kBlock - 1) / rg::kBlock), dim3(rg::kBlock), 0, s,
                     keys_out, (long)n, n_cells, cell_start);
  return rg::check_launch("rg_geom_bin_gates_f32");
}

namespace {

size_t scan_u32_temp_bytes(size_t n) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, 0u, n, rocprim::plus<unsigned>(),
                                (hipStream_t)0, false);
  return bytes;
}

int check_levels(const char* fn, const rg_cellgrid* cells, const float* zc, int nz, double min_radius, double beam_factor) {
  RG_REQUIRE(cells && zc, RG_EINVAL, "%s: null pointer", fn);
  RG_REQUIRE(nz >= 1 && cells->levels == nz && cells->level0 == 0, RG_EINVAL, "%s: cells->levels=%d, level0=%d, nz=%d", fn,
             cells->levels, cells->level0, nz);
  RG_REQUIRE(cells->ncx >= 1 && cells->ncy >= 1 && (long)cells->ncx * cells->ncy * nz < 0x7FFFFFFFL, RG_EINVAL,
             "%s: bad cell grid %dx%d x %d levels", fn, cells->ncx, cells->ncy, nz);
  RG_REQUIRE(min_radius >= 0.0 && beam_factor >= 0.0 && beam_factor < 0.5, RG_EUNSUPPORTED,
             "%s: per-level gate lists need 0 <= beam_factor < 0.5 (got %g); use the single list", fn, beam_factor);
  return RG_OK;
}

}  // namespace

extern "C" int rg_geom_bin_levels_count(const float* gate_x, const float* gate_y, const float* gate_z, int64_t n_gates,
                                        float radar_altitude, float toa, const rg_cellgrid* cells_host, const float* zc,
                                        int32_t nz, double min_radius, double beam_factor, int64_t* total,
                                        rg_stream_t stream) {
  const int rc = check_levels("rg_geom_bin_levels_count", cells_host, zc, nz, min_radius, beam_factor);
  if (rc != RG_OK) return rc;
  RG_REQUIRE(total && n_gates >= 0 && n_gates <= 0x7FFFFFFFL, RG_EINVAL, "rg_geom_bin_levels_count: bad arguments");
  RG_REQUIRE(n_gates == 0 || (gate_x && gate_y && gate_z), RG_EINVAL, "rg_geom_bin_levels_count: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(total, 0, sizeof(int64_t), s) != hipSuccess) {
    rg::set_error("rg_geom_bin_levels_count: memset failed");
    return RG_ELAUNCH;
  }
  if (n_gates > 0) {
    const LevelArgs la{zc, nz, min_radius, beam_factor / (1.0 - beam_factor)};
    hipLaunchKernelGGL(level_count_kernel, dim3((unsigned)((n_gates + rg::kBlock - 1) / rg::kBlock)), dim3(rg::kBlock), 0, s,
                       gate_x, gate_y, gate_z, (long)n_gates, radar_altitude, toa, to_cells(cells_host), la, (unsigned*)nullptr,
                       reinterpret_cast<unsigned long long*>(total));
  }
  return rg::check_launch("rg_geom_bin_levels_count");
}

extern "C" int64_t rg_geom_bin_levels_workspace_bytes(int64_t n_gates, int64_t n_entries, int64_t n_cells_total) {
  if (n_gates < 0 || n_entries < 0 || n_cells_total < 1 || n_entries > 0x7FFFFFFFL || n_cells_total >= 0x7FFFFFFFL) return RG_EINVAL;
  const size_t ne = (size_t)n_entries, ng = (size_t)n_gates + 1;
  const size_t arr = round_up(ne * sizeof(unsigned), 256), garr = round_up(ng * sizeof(unsigned), 256);
  const size_t temp = sort_temp_bytes(ne, key_bits((unsigned)n_cells_total)), temp2 = scan_u32_temp_bytes(ng);
  return (int64_t)(4 * arr + 2 * garr + round_up(temp > temp2 ? temp : temp2, 256) + 256);
}

extern "C" int rg_geom_bin_gates_levels_f32(const float* gate_x, const float* gate_y, const float* gate_z, int64_t n_gates,
                                            float radar_altitude, float toa, const rg_cellgrid* cells_host, const float* zc,
                                            int32_t nz, double min_radius, double beam_factor, int64_t n_entries,
                                            rg_gate4* sorted_gates, int32_t* cell_start, void* workspace,
                                            int64_t workspace_bytes, rg_stream_t stream) {
  const int rc = check_levels("rg_geom_bin_gates_levels_f32", cells_host, zc, nz, min_radius, beam_factor);
  if (rc != RG_OK) return rc;
  RG_REQUIRE(cell_start && n_gates >= 0 && n_gates <= 0x7FFFFFFFL && n_entries >= 0 && n_entries <= 0x7FFFFFFFL, RG_EINVAL,
             "rg_geom_bin_gates_levels_f32: bad arguments");
  RG_REQUIRE(n_entries == 0 || (gate_x && gate_y && gate_z && sorted_gates && workspace), RG_EINVAL,
             "rg_geom_bin_gates_levels_f32: null pointer");
  RG_REQUIRE(rg::aligned16(sorted_gates), RG_EALIGN, "rg_geom_bin_gates_levels_f32: sorted_gates must be 16-byte aligned");
  const Cells c = to_cells(cells_host);
  const unsigned n_cells = (unsigned)c.ncx * (unsigned)c.ncy * (unsigned)nz;
  const int64_t need = rg_geom_bin_levels_workspace_bytes(n_gates, n_entries, n_cells);
  RG_REQUIRE(need >= 0 && workspace_bytes >= need, RG_EWORKSPACE, "rg_geom_bin_gates_levels_f32: workspace %lld < %lld bytes",
             (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  const size_t ne = (size_t)n_entries, ng = (size_t)n_gates + 1;
  const size_t arr = round_up(ne * sizeof(unsigned), 256), garr = round_up(ng * sizeof(unsigned), 256);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  unsigned* keys_in = reinterpret_cast<unsigned*>(base);
  unsigned* vals_in = reinterpret_cast<unsigned*>(base + arr);
  unsigned* keys_out = reinterpret_cast<unsigned*>(base + 2 * arr);
  unsigned* vals_out = reinterpret_cast<unsigned*>(base + 3 * arr);
  unsigned* counts = reinterpret_cast<unsigned*>(base + 4 * arr);
  unsigned* offsets = reinterpret_cast<unsigned*>(base + 4 * arr + garr);
  void* temp = base + 4 * arr + 2 * garr;
  if (n_gates > 0 && ne > 0) {
    const LevelArgs la{zc, nz, min_radius, beam_factor / (1.0 - beam_factor)};
    const dim3 ggrid((unsigned)((ng + rg::kBlock - 1) / rg::kBlock)), block(rg::kBlock);
    hipError_t e = hipMemsetAsync(counts, 0, ng * sizeof(unsigned), s);
    RG_REQUIRE(e == hipSuccess, RG_ELAUNCH, "rg_geom_bin_gates_levels_f32: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(level_count_kernel, ggrid, block, 0, s, gate_x, gate_y, gate_z, (long)n_gates, radar_altitude, toa, c, la,
                       counts, (unsigned long long*)nullptr);
    size_t scan_bytes = scan_u32_temp_bytes(ng);
    e = rocprim::exclusive_scan(temp, scan_bytes, counts, offsets, 0u, ng, rocprim::plus<unsigned>(), s, false);
    RG_REQUIRE(e == hipSuccess, RG_ELAUNCH, "rg_geom_bin_gates_levels_f32: scan: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(level_expand_kernel, ggrid, block, 0, s, gate_x, gate_y, gate_z, (long)n_gates, radar_altitude, toa, c, la,
                       offsets, keys_in, vals_in);
    const unsigned bits = key_bits(n_cells);
    size_t temp_bytes = sort_temp_bytes(ne, bits);
    e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, ne, 0u, bits, s, false);
    RG_REQUIRE(e == hipSuccess, RG_ELAUNCH, "rg_geom_bin_gates_levels_f32: radix sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(gather_sorted_kernel, dim3((unsigned)((ne + rg::kBlock - 1) / rg::kBlock)), block, 0, s, keys_out, vals_out,
                       (long)ne, n_cells, gate_x, gate_y, gate_z, radar_altitude, sorted_gates);
  }
  hipLaunchKernelGGL(cell_start_kernel, dim3((n_cells + 1 + rg::kBlock - 1) / rg::kBlock), dim3(rg::kBlock), 0, s, keys_out,
                     (long)ne, n_cells, cell_start);
  return rg::check_launch("rg_geom_bin_gates_levels_f32");
}

extern "C" int64_t rg_scan_workspace_bytes(int64_t n) {
  if (n < 0) return RG_EINVAL;
  size_t bytes = 0;
  auto in = rocprim::make_transform_iterator((const int*)nullptr, WidenI32());
  (void)rocprim::exclusive_scan(nullptr, bytes, in, (long long*)nullptr, 0ll, (size_t)n + 1, rocprim::plus<long long>(),
                                (hipStream_t)0, false);
  return (int64_t)round_up(bytes, 256) + 256;
}

extern "C" int rg_scan_counts_i64(const int32_t* counts, int64_t n, int64_t* indptr, void* workspace,
                                  int64_t workspace_bytes, rg_stream_t stream) {
  // counts must hold n+1 readable entries (the last one is ignored by the exclusive scan's output[n] = total)
  RG_REQUIRE(counts && indptr && workspace, RG_EINVAL, "rg_scan_counts_i64: null pointer");
  RG_REQUIRE(n >= 0, RG_EINVAL, "rg_scan_counts_i64: negative size");
  RG_REQUIRE(workspace_bytes >= rg_scan_workspace_bytes(n), RG_EWORKSPACE, "rg_scan_counts_i64: workspace too small");
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  size_t bytes = (size_t)workspace_bytes - (size_t)(base - reinterpret_cast<char*>(workspace));
  auto in = rocprim::make_transform_iterator(counts, WidenI32());
  hipError_t e = rocprim::exclusive_scan(base, bytes, in, reinterpret_cast<long long*>(indptr), 0ll, (size_t)n + 1,
                                         rocprim::plus<long long>(), (hipStream_t)stream, false);
  RG_REQUIRE(e == hipSuccess, RG_ELAUNCH, "rg_scan_counts_i64: %s", hipGetErrorString(e));
  return RG_OK;
}
