// K2  roi_grid -- fused on-the-fly gridding: neighbour search + weights + masked weighted mean in one kernel,
// without materialising a CSR.  Equivalent to radar_grid/compute.py:46-91 fused with
// radar_grid/interpolate.py:69-104.  Needed where the CSR is too large to keep (the 14x720x2000
// volume on a 40x2000x2000 grid is ~24 G pairs = 195 GB, 11x over the reference's int32 indptr).
//
// Bound: NOT HBM (compulsory traffic is only the sorted gates + packed fields + the output grid); the kernel
// is limited by float64 VALU issue for the membership test and by the L2-served gather of field values.
//
// Compiled with -ffp-contract=off: membership uses the reference's unfused float64 arithmetic, so the
// neighbour sets equal the CSR builder's and the result differs from csr_apply only by summation order.
//
// Mapping: as in the builder, one wavefront per voxel, 64 candidate gates per step; each lane keeps float64
// partial sums of (w*v, w) per field, combined at the end of the voxel by a wavefront shuffle reduction.
#include "rg_common.hpp"
#include "rg_roi_search.hpp"

namespace {

using namespace rg::roi;

template <int STRIDE>
__device__ __forceinline__ void load_packed(const float* __restrict__ p, unsigned g, float (&v)[STRIDE]) {
  if constexpr (STRIDE == 1) {
    v[0] = p[g];
  } else if constexpr (STRIDE == 2) {
    const float2 t = reinterpret_cast<const float2*>(p)[g];
    v[0] = t.x; v[1] = t.y;
  } else {
#pragma unroll
    for (int s = 0; s < STRIDE; s += 4) {
      const float4 t = reinterpret_cast<const float4*>(p)[(size_t)g * (STRIDE / 4) + s / 4];
      v[s] = t.x; v[s + 1] = t.y; v[s + 2] = t.z; v[s + 3] = t.w;
    }
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <int W, int NF, int STRIDE>
__global__ __launch_bounds__(rg::kBlock) void roi_grid_kernel(SearchArgs a, const float* __restrict__ packed, float fill,
                                                              float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wave = (long)blockIdx.x * (rg::kBlock / rg::kWave) + wv;
  const long vbeg = wave * kVoxPerWave;
  for (int t = 0; t < kVoxPerWave; ++t) {
    const long v = vbeg + t;
    if (v >= a.n_vox) break;  // wave-uniform
    const VoxelBox b = voxel_box(a, v);
    double acc_p[NF], acc_w[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { acc_p[f] = 0.0; acc_w[f] = 0.0; }
    for (int cy = b.cy0; cy <= b.cy1; ++cy) {
      const int s = a.cell_start[cy * a.c.ncx + b.cx0];
      const int e = a.cell_start[cy * a.c.ncx + b.cx1 + 1];
      for (int jb = s; jb < e; jb += 64) {
        const int j = jb + lane;
        if (j < e) {
          const rg_gate4 g = a.sorted[j];
          const double dx = (double)g.x - b.x, dy = (double)g.y - b.y, dz = (double)g.z - b.z;
          const double d2 = dx * dx + dy * dy + dz * dz;
          if (d2 < b.r2) {
            const float w = roi_weight<W>(d2, b.r2);
            float val[STRIDE];
            load_packed<STRIDE>(packed, (unsigned)g.index, val);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const bool ok = rg::f32_bits(val[f]) != RG_EXCLUDED_BITS;
              acc_p[f] += (double)(ok ? __fmul_rn(w, val[f]) : 0.0f);   // float32 product, as interpolate.py:82
              acc_w[f] += (double)(ok ? w : 0.0f);
            }
          }
        }
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const double p = wave_sum(acc_p[f]);
      const double w = wave_sum(acc_w[f]);
      if (lane == 0) out[(size_t)f * a.n_vox + v] = w > 0.0 ? (float)(p / w) : fill;
    }
  }
}

template <int W, int NF, int STRIDE>
int launch(const SearchArgs& a, const float* packed, float fill, float* out, hipStream_t s) {
  hipLaunchKernelGGL((roi_grid_kernel<W, NF, STRIDE>), search_grid(a.n_vox), dim3(rg::kBlock), 0, s, a, packed, fill, out);
  return rg::check_launch("rg_roi_grid_f32");
}

template <int W>
int dispatch(int nf, const SearchArgs& a, const float* packed, float fill, float* out, hipStream_t s) {
  switch (nf) {
    case 1: return launch<W, 1, 1>(a, packed, fill, out, s);
    case 2: return launch<W, 2, 2>(a, packed, fill, out, s);
    case 3: return launch<W, 3, 4>(a, packed, fill, out, s);
    case 4: return launch<W, 4, 4>(a, packed, fill, out, s);
    case 5: return launch<W, 5, 8>(a, packed, fill, out, s);
    case 6: return launch<W, 6, 8>(a, packed, fill, out, s);
    case 7: return launch<W, 7, 8>(a, packed, fill, out, s);
    default: return launch<W, 8, 8>(a, packed, fill, out, s);
  }
}

inline int stride_for(int nf) { return nf == 1 ? 1 : nf == 2 ? 2 : nf <= 4 ? 4 : 8; }

}  // namespace

extern "C" int rg_roi_grid_f32(const rg_gate4* sorted_gates, const int32_t* cell_start, const rg_cellgrid* cells_host,
                               const float* xc, const float* yc, const float* zc, int32_t nz, int32_t ny, int32_t nx,
                               double min_radius, double beam_factor, int32_t weighting, const float* packed,
                               int32_t n_fields, int32_t stride, float fill_value, float* out, rg_stream_t stream) {
  const int rc = check_search_args("rg_roi_grid_f32", sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx);
  if (rc != RG_OK) return rc;
  RG_REQUIRE(packed && out, RG_EINVAL, "rg_roi_grid_f32: null pointer");
  RG_REQUIRE(weighting >= RG_W_BARNES2 && weighting <= RG_W_NEAREST, RG_EINVAL, "rg_roi_grid_f32: unknown weighting %d",
             weighting);
  RG_REQUIRE(n_fields >= 1 && n_fields <= RG_MAX_FIELDS, RG_EUNSUPPORTED, "rg_roi_grid_f32: n_fields=%d not in 1..%d",
             n_fields, RG_MAX_FIELDS);
  RG_REQUIRE(stride == stride_for(n_fields), RG_EINVAL, "rg_roi_grid_f32: stride=%d, expected %d for %d fields", stride,
             stride_for(n_fields), n_fields);
  RG_REQUIRE(rg::aligned16(packed), RG_EALIGN, "rg_roi_grid_f32: packed must be 16-byte aligned");
  const SearchArgs a = make_args(sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx, min_radius, beam_factor);
  hipStream_t s = (hipStream_t)stream;
  switch (weighting) {
    case RG_W_BARNES2: return dispatch<RG_W_BARNES2>(n_fields, a, packed, fill_value, out, s);
    case RG_W_CRESSMAN: return dispatch<RG_W_CRESSMAN>(n_fields, a, packed, fill_value, out, s);
    default: return dispatch<RG_W_NEAREST>(n_fields, a, packed, fill_value, out, s);
  }
}
