// 2-D raster stage behind the 3-D grid cache:
//   collapse 'ppi'      radar_processor/processor.py:512-528, radar_processor/utils.py:366-375
//   threshold masks     radar_processor/processor.py:541-546 and :802-886
//   colormap -> RGBA    radar_grid/geotiff.py:70-145 (matplotlib Normalize + Colormap.__call__)
//
// Roofline: HBM, pure streaming; every kernel touches each pixel once (4 B in, 4 B out; the PPI gather reads one
// of nz levels per pixel and the z table from scalar cache).  These planes are at most a few MB, so the launches are
// latency-bound; they exist so that the 2-D tier can be filled without a host round trip.
#include "rg_common.hpp"

namespace {

// numpy's maximum / minimum loops: (a >= b || isnan(a)) ? a : b -- NaN in either operand propagates
template <typename C> __device__ __forceinline__ C np_maximum(C a, C b) { return (a >= b || a != a) ? a : b; }
template <typename C> __device__ __forceinline__ C np_minimum(C a, C b) { return (a <= b || a != a) ? a : b; }

__global__ __launch_bounds__(rg::kBlock) void collapse_ppi_kernel(const float* __restrict__ grid,
                                                                  const double* __restrict__ x,
                                                                  const double* __restrict__ y,
                                                                  const double* __restrict__ z, int nz, int ny, int nx,
                                                                  double sin_elev, double two_re,
                                                                  float* __restrict__ out, int* __restrict__ out_level) {
  const long n_xy = (long)ny * nx;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_xy) return;
  const int iy = (int)(p / nx), ix = (int)(p - (long)iy * nx);
  const double X = x[ix], Y = y[iy];
  const double r = sqrt(X * X + Y * Y);                       // processor.py:516
  const double zt = r * sin_elev + (r * r) / two_re;          // processor.py:520
  // np.argmin over |zt - z[k]|: first minimum, a NaN counts as the minimum (processor.py:523)
  double best = fabs(zt - z[0]);
  int k_best = 0;
  for (int k = 1; k < nz; ++k) {
    const double d = fabs(zt - z[k]);                          // z[k]: wave-uniform address -> scalar load
    if (d < best || (d != d && best == best)) { best = d; k_best = k; }
  }
  out[p] = grid[(size_t)k_best * n_xy + p];                    // processor.py:526-528
  if (out_level) out_level[p] = k_best;
}

struct PlaneTests {
  int n;
  rg_plane_test t[RG_MAX_PLANE_TESTS];
};

__global__ __launch_bounds__(rg::kBlock) void plane_filter_kernel(const float* __restrict__ src,
                                                                  const uint8_t* __restrict__ src_mask, long n,
                                                                  PlaneTests tests, float* __restrict__ out,
                                                                  uint8_t* __restrict__ out_mask) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float v = src[p];
  bool drop = false;
  for (int i = 0; i < tests.n; ++i) {                          // wave-uniform trip count and branches
    const rg_plane_test t = tests.t[i];
    const float q = t.plane ? t.plane[p] : v;
    if (t.flags & RG_TEST_LO) drop |= (t.flags & RG_TEST_LO_INCLUSIVE) ? q <= t.lo : q < t.lo;
    if (t.flags & RG_TEST_HI) drop |= q > t.hi;
    if (t.flags & RG_TEST_NONFINITE) drop |= !(fabsf(q) < __builtin_inff());
  }
  const bool masked = drop || (src_mask ? src_mask[p] != 0 : v != v);   // an explicit mask is authoritative
  if (out) out[p] = masked ? __builtin_nanf("") : v;
  if (out_mask) out_mask[p] = masked ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(rg::kBlock) void grid_filter_kernel(const T* __restrict__ src, long n, int flags, T lo, T hi,
                                                                 const uint8_t* __restrict__ mask, T fill,
                                                                 T* __restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const T v = src[p];
  bool hit = mask && mask[p];
  if (flags & RG_TEST_LO) hit |= v < lo;                                   // filters.py:657
  if (flags & RG_TEST_HI) hit |= v > hi;                                   // filters.py:686
  if (flags & RG_TEST_NONFINITE) hit |= !(fabs((double)v) < __builtin_inf());   // filters.py:746
  out[p] = hit ? fill : v;
}

// ---- min / max / count of the valid pixels ---------------------------------------------------------------
constexpr int kMinmaxBlocks = 1024;   // partials: 4 doubles per block -> 32 KiB of workspace

struct Mm {
  double lo, hi, cnt;   // of the pixels that are neither no-data nor NaN
  double kept;          // pixels that are not no-data (geotiff.py:117: len(valid_data))
};

__device__ __forceinline__ Mm mm_merge(Mm a, Mm b) {
  Mm r;
  r.lo = b.lo < a.lo ? b.lo : a.lo;
  r.hi = b.hi > a.hi ? b.hi : a.hi;
  r.cnt = a.cnt + b.cnt;
  r.kept = a.kept + b.kept;
  return r;
}

__device__ __forceinline__ Mm mm_block_reduce(Mm m) {
  __shared__ Mm part[rg::kBlock / rg::kWave];
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    Mm o;
    o.lo = __shfl_xor(m.lo, s, 64); o.hi = __shfl_xor(m.hi, s, 64); o.cnt = __shfl_xor(m.cnt, s, 64);
    o.kept = __shfl_xor(m.kept, s, 64);
    m = mm_merge(m, o);
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  Mm r = part[0];
#pragma unroll
  for (int w = 1; w < rg::kBlock / rg::kWave; ++w) r = mm_merge(r, part[w]);
  return r;
}

template <typename T>
__global__ __launch_bounds__(rg::kBlock) void minmax_partial_kernel(const T* __restrict__ data, long n, int has_fill,
                                                                    T fill, double* __restrict__ partial) {
  Mm m;
  m.lo = __builtin_inf(); m.hi = -__builtin_inf(); m.cnt = 0.0; m.kept = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const T v = data[i];
    const bool nodata = has_fill ? v == fill : v != v;         // geotiff.py:111-115
    m.kept += nodata ? 0.0 : 1.0;
    if (!nodata && v == v) {                                   // np.nanmin / np.nanmax skip NaN (:121,123)
      const double d = (double)v;
      m.lo = d < m.lo ? d : m.lo;
      m.hi = d > m.hi ? d : m.hi;
      m.cnt += 1.0;
    }
  }
  m = mm_block_reduce(m);
  if (threadIdx.x == 0) {
    double* q = partial + 4 * blockIdx.x;
    q[0] = m.lo; q[1] = m.hi; q[2] = m.cnt; q[3] = m.kept;
  }
}

__global__ __launch_bounds__(rg::kBlock) void minmax_final_kernel(const double* __restrict__ partial, int n_part,
                                                                  double* __restrict__ out) {
  Mm m;
  m.lo = __builtin_inf(); m.hi = -__builtin_inf(); m.cnt = 0.0; m.kept = 0.0;
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
    Mm o;
    o.lo = partial[4 * i]; o.hi = partial[4 * i + 1]; o.cnt = partial[4 * i + 2]; o.kept = partial[4 * i + 3];
    m = mm_merge(m, o);
  }
  m = mm_block_reduce(m);
  if (threadIdx.x == 0) { out[0] = m.lo; out[1] = m.hi; out[2] = m.cnt; out[3] = m.kept; }
}

// ---- colormap --------------------------------------------------------------------------------------------
// T: dtype of the data.  The normalisation itself is float64 whatever T is (see the header).
template <typename T>
__global__ __launch_bounds__(rg::kBlock) void colormap_kernel(const T* __restrict__ data, long n, double vmin,
                                                              double vmax, int flat, int has_fill, T fill,
                                                              const uint32_t* __restrict__ lut, int n_lut,
                                                              uint32_t* __restrict__ out) {
  extern __shared__ uint32_t lds_lut[];
  for (int i = threadIdx.x; i < n_lut + 3; i += blockDim.x) lds_lut[i] = lut[i];
  __syncthreads();
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const T x = data[p];
  const bool nodata = has_fill ? x == fill : x != x;           // geotiff.py:111-115
  int idx = 0;                                                 // vmin == vmax: Normalize fills 0 -> entry 0
  if (!flat) {
    const double nl = (double)n_lut;
    double v = np_minimum(np_maximum((double)x, vmin), vmax);  // np.clip(result.filled(vmax), vmin, vmax)
    v = v - vmin;                                              // resdat -= vmin
    v = v / (vmax - vmin);                                     // resdat /= (vmax - vmin)
    v = v * nl;                                                // xa *= self.N
    if (v == nl) v = nl - 1.0;                                 // xa[xa == N] = N - 1
    if (v != v) idx = n_lut + 2;                               // bad
    else if (v < 0.0) idx = n_lut;                             // under
    else if (v >= nl) idx = n_lut + 1;                         // over
    else idx = (int)v;                                         // astype(int): truncation
  }
  uint32_t rgba = lds_lut[idx];
  if (nodata) rgba &= 0x00FFFFFFu;                             // alpha byte (little endian: byte 3) -> 0
  out[p] = rgba;
}

template <typename T>
void launch_colormap(const void* data, long n, double vmin, double vmax, int flat, int has_fill, double fill,
                     const uint8_t* lut, int n_lut, uint8_t* out, hipStream_t s) {
  const unsigned blocks = (unsigned)((n + rg::kBlock - 1) / rg::kBlock);
  hipLaunchKernelGGL((colormap_kernel<T>), dim3(blocks), dim3(rg::kBlock), (size_t)(n_lut + 3) * 4, s,
                     static_cast<const T*>(data), n, vmin, vmax, flat, has_fill, (T)fill,
                     reinterpret_cast<const uint32_t*>(lut), n_lut, reinterpret_cast<uint32_t*>(out));
}

}  // namespace

extern "C" int rg_collapse_ppi_f32(const float* grid, const double* x, const double* y, const double* z, int32_t nz,
                                   int32_t ny, int32_t nx, double sin_elev, double two_re, float* out,
                                   int32_t* out_level, rg_stream_t stream) {
  RG_REQUIRE(nz >= 1 && ny >= 0 && nx >= 0, RG_EINVAL, "rg_collapse_ppi_f32: bad shape (%d, %d, %d)", nz, ny, nx);
  const long n_xy = (long)ny * nx;
  if (n_xy == 0) return RG_OK;
  RG_REQUIRE(grid && x && y && z && out, RG_EINVAL, "rg_collapse_ppi_f32: null pointer");
  RG_REQUIRE(two_re != 0.0, RG_EINVAL, "rg_collapse_ppi_f32: two_re must not be 0");
  const unsigned blocks = (unsigned)((n_xy + rg::kBlock - 1) / rg::kBlock);
  hipLaunchKernelGGL(collapse_ppi_kernel, dim3(blocks), dim3(rg::kBlock), 0, (hipStream_t)stream, grid, x, y, z, nz, ny,
                     nx, sin_elev, two_re, out, out_level);
  return rg::check_launch("rg_collapse_ppi_f32");
}

extern "C" int rg_plane_filter_f32(const float* src, const uint8_t* src_mask, int64_t n, const rg_plane_test* tests,
                                   int32_t n_tests, float* out, uint8_t* out_mask, rg_stream_t stream) {
  RG_REQUIRE(n >= 0, RG_EINVAL, "rg_plane_filter_f32: negative size");
  RG_REQUIRE(n_tests >= 0 && n_tests <= RG_MAX_PLANE_TESTS, RG_EINVAL, "rg_plane_filter_f32: %d tests (at most %d)",
             n_tests, RG_MAX_PLANE_TESTS);
  RG_REQUIRE(n_tests == 0 || tests, RG_EINVAL, "rg_plane_filter_f32: null tests");
  if (n == 0) return RG_OK;
  RG_REQUIRE(src && (out || out_mask), RG_EINVAL, "rg_plane_filter_f32: null pointer");
  PlaneTests pt;
  pt.n = n_tests;
  for (int i = 0; i < RG_MAX_PLANE_TESTS; ++i) {
    if (i < n_tests) {
      pt.t[i] = tests[i];
      RG_REQUIRE((pt.t[i].flags & ~(RG_TEST_LO | RG_TEST_HI | RG_TEST_LO_INCLUSIVE | RG_TEST_NONFINITE)) == 0, RG_EINVAL,
                 "rg_plane_filter_f32: test %d has unknown flags 0x%x", i, pt.t[i].flags);
    } else {
      pt.t[i].plane = nullptr; pt.t[i].lo = pt.t[i].hi = 0.0f; pt.t[i].flags = 0;
    }
  }
  const unsigned blocks = (unsigned)((n + rg::kBlock - 1) / rg::kBlock);
  hipLaunchKernelGGL(plane_filter_kernel, dim3(blocks), dim3(rg::kBlock), 0, (hipStream_t)stream, src, src_mask, (long)n,
                     pt, out, out_mask);
  return rg::check_launch("rg_plane_filter_f32");
}

extern "C" int rg_nan_minmax(const void* data, int32_t data_is_f64, int64_t n, int32_t has_fill, double fill,
                             void* workspace, double* out, rg_stream_t stream) {
  static_assert(kMinmaxBlocks * 4 * sizeof(double) <= RG_MINMAX_WORKSPACE_BYTES, "workspace too small");
  RG_REQUIRE(n >= 0, RG_EINVAL, "rg_nan_minmax: negative size");
  RG_REQUIRE(workspace && out && (data || n == 0), RG_EINVAL, "rg_nan_minmax: null pointer");
  long want = (n + rg::kBlock * 8 - 1) / (rg::kBlock * 8);
  const int blocks = (int)(want < 1 ? 1 : want > kMinmaxBlocks ? kMinmaxBlocks : want);
  double* partial = static_cast<double*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  if (data_is_f64)
    hipLaunchKernelGGL(minmax_partial_kernel<double>, dim3(blocks), dim3(rg::kBlock), 0, s,
                       static_cast<const double*>(data), (long)n, has_fill, fill, partial);
  else
    hipLaunchKernelGGL(minmax_partial_kernel<float>, dim3(blocks), dim3(rg::kBlock), 0, s,
                       static_cast<const float*>(data), (long)n, has_fill, (float)fill, partial);
  const int rc = rg::check_launch("rg_nan_minmax");
  if (rc != RG_OK) return rc;
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(rg::kBlock), 0, s, partial, blocks, out);
  return rg::check_launch("rg_nan_minmax");
}

extern "C" int rg_colormap_rgba(const void* data, int32_t data_is_f64, int64_t n, double vmin, double vmax,
                                int32_t has_fill, double fill, const uint8_t* lut, int32_t n_lut, uint8_t* out,
                                rg_stream_t stream) {
  RG_REQUIRE(n >= 0, RG_EINVAL, "rg_colormap_rgba: negative size");
  RG_REQUIRE(n_lut >= 1 && n_lut <= RG_MAX_LUT, RG_EINVAL, "rg_colormap_rgba: n_lut %d outside 1..%d", n_lut, RG_MAX_LUT);
  RG_REQUIRE(!(vmin > vmax), RG_EINVAL, "rg_colormap_rgba: minvalue must be less than or equal to maxvalue");
  if (n == 0) return RG_OK;
  RG_REQUIRE(data && lut && out, RG_EINVAL, "rg_colormap_rgba: null pointer");
  RG_REQUIRE((reinterpret_cast<uintptr_t>(lut) & 3u) == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0, RG_EALIGN,
             "rg_colormap_rgba: lut and out must be 4-byte aligned");
  const int flat = vmin == vmax;
  hipStream_t s = (hipStream_t)stream;
  if (data_is_f64)
    launch_colormap<double>(data, (long)n, vmin, vmax, flat, has_fill, fill, lut, n_lut, out, s);
  else
    launch_colormap<float>(data, (long)n, vmin, vmax, flat, has_fill, fill, lut, n_lut, out, s);
  return rg::check_launch("rg_colormap_rgba");
}

extern "C" int rg_grid_filter(const void* src, int32_t data_is_f64, int64_t n, int32_t flags, double lo, double hi,
                              const uint8_t* mask, double fill_value, void* out, rg_stream_t stream) {
  RG_REQUIRE(n >= 0, RG_EINVAL, "rg_grid_filter: negative size");
  RG_REQUIRE((flags & ~(RG_TEST_LO | RG_TEST_HI | RG_TEST_NONFINITE)) == 0, RG_EINVAL, "rg_grid_filter: unknown flags 0x%x",
             flags);
  if (n == 0) return RG_OK;
  RG_REQUIRE(src && out, RG_EINVAL, "rg_grid_filter: null pointer");
  const unsigned blocks = (unsigned)((n + rg::kBlock - 1) / rg::kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (data_is_f64)
    hipLaunchKernelGGL(grid_filter_kernel<double>, dim3(blocks), dim3(rg::kBlock), 0, s, static_cast<const double*>(src),
                       (long)n, flags, lo, hi, mask, fill_value, static_cast<double*>(out));
  else
    hipLaunchKernelGGL(grid_filter_kernel<float>, dim3(blocks), dim3(rg::kBlock), 0, s, static_cast<const float*>(src),
                       (long)n, flags, (float)lo, (float)hi, mask, (float)fill_value, static_cast<float*>(out));
  return rg::check_launch("rg_grid_filter");
}
