// K3  column reduce (COLMAX / COLMIN / COLMEAN, optional arg index)  -- radar_grid/products.py:420-580
// K4  CAPPI linear interpolation between two levels                  -- radar_grid/products.py:406-412
//
// Roofline: HBM, pure streaming.  Algorithmic bytes: K3 4*(z_hi-z_lo+1)*Vxy read + 4*Vxy (+4*Vxy arg)
// written; K4 8*Vxy read + 4*Vxy written.
//
// K3 mapping: a lane owns VEC consecutive (y,x) columns (VEC = 4 -> one dwordx4 per level, 1 KiB per
// wave-instruction).  For grids too small to fill 256 CUs with one lane per column group, the level range
// is additionally split over ZS = 4 lane groups of the same wavefront (lanes l, l+16, l+32, l+48 share a
// column group) and combined with two wavefront shuffles; max/min with first-index ties are associative,
// so the split is bit-exact.  The mean keeps NumPy's sequential float32 order (ZS = 1).
#include "rg_common.hpp"

namespace {

struct Best {
  float v;   // NaN = nothing seen yet
  int idx;   // -1 = nothing seen yet
};

// np.fmax / np.fmin reduction step in level order: keep acc when (acc >= v || isnan(v)) (numpy's
// @TYPE@_fmax loop), so the first of equal values -- including the sign of a zero -- wins.
template <bool IS_MAX>
__device__ __forceinline__ void step(Best& acc, float v, int z) {
  const bool keep = (IS_MAX ? acc.v >= v : acc.v <= v) || isnan(v);
  if (!keep) { acc.v = v; acc.idx = z; }
}

// combine two partial results; `b` covers levels that may interleave with `a`: lower index wins ties
template <bool IS_MAX>
__device__ __forceinline__ Best merge(Best a, Best b) {
  if (b.idx < 0) return a;
  if (a.idx < 0) return b;
  const bool b_better = IS_MAX ? b.v > a.v : b.v < a.v;
  const bool tie = b.v == a.v;
  if (b_better || (tie && b.idx < a.idx)) return b;
  return a;
}

template <int OP, int VEC, int ZS>
__global__ __launch_bounds__(rg::kBlock) void column_kernel(const float* __restrict__ grid, long n_xy, int z_lo,
                                                            int z_hi, float* __restrict__ out,
                                                            int* __restrict__ out_arg) {
  static_assert(ZS == 1 || ZS == 4, "");
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long cg;   // column group
  int zs;    // which quarter of the level range this lane covers
  if constexpr (ZS == 1) {
    cg = t; zs = 0;
  } else {
    const int lane = threadIdx.x & 63;
    cg = (t >> 6) * 16 + (lane & 15);
    zs = lane >> 4;
  }
  const long c0 = cg * VEC;
  const bool live = c0 < n_xy;

  if constexpr (OP == RG_COL_MEAN) {
    float sum[VEC];
    int cnt[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { sum[k] = 0.0f; cnt[k] = 0; }
    if (live) {
      for (int z = z_lo; z <= z_hi; ++z) {
        float v[VEC];
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(grid + (size_t)z * n_xy + c0);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
          v[0] = grid[(size_t)z * n_xy + c0];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const bool nan = isnan(v[k]);
          sum[k] = __fadd_rn(sum[k], nan ? 0.0f : v[k]);   // np.nanmean: NaN -> 0, float32 adds in level order
          cnt[k] += nan ? 0 : 1;
        }
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) out[c0 + k] = (float)((double)sum[k] / (double)cnt[k]);  // 0/0 -> NaN
    }
    return;
  } else {
    constexpr bool IS_MAX = OP == RG_COL_MAX;
    Best acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { acc[k].v = __builtin_nanf(""); acc[k].idx = -1; }
    if (live) {
      for (int z = z_lo + zs; z <= z_hi; z += ZS) {
        float v[VEC];
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4*>(grid + (size_t)z * n_xy + c0);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
          v[0] = grid[(size_t)z * n_xy + c0];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          if (acc[k].idx < 0) {               // first level of this lane: fmax.reduce starts from it
            if (!isnan(v[k])) { acc[k].v = v[k]; acc[k].idx = z; }
          } else {
            step<IS_MAX>(acc[k], v[k], z);
          }
        }
      }
    }
    if constexpr (ZS == 4) {
      // wavefront-shuffle reduction over the 4 level quarters (lanes l, l^16, l^32, l^48)
#pragma unroll
      for (int m = 16; m <= 32; m <<= 1) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          Best o;
          o.v = __shfl_xor(acc[k].v, m, 64);
          o.idx = __shfl_xor(acc[k].idx, m, 64);
          acc[k] = merge<IS_MAX>(acc[k], o);
        }
      }
      if (zs != 0) return;
    }
    if (live) {
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(out + c0) = make_float4(acc[0].v, acc[1].v, acc[2].v, acc[3].v);
        if (out_arg) *reinterpret_cast<int4*>(out_arg + c0) = make_int4(acc[0].idx, acc[1].idx, acc[2].idx, acc[3].idx);
      } else {
        out[c0] = acc[0].v;
        if (out_arg) out_arg[c0] = acc[0].idx;
      }
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(rg::kBlock) void cappi_lerp_kernel(const float* __restrict__ lo, const float* __restrict__ hi,
                                                                long n_xy, float w_lo, float w_hi,
                                                                float* __restrict__ out) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (i >= n_xy) return;
  if constexpr (VEC == 4) {
    const float4 a = *reinterpret_cast<const float4*>(lo + i);
    const float4 b = *reinterpret_cast<const float4*>(hi + i);
    float4 r;
    // two float32 products and one float32 sum, no fused multiply-add (NumPy evaluates them separately)
    r.x = __fadd_rn(__fmul_rn(w_lo, a.x), __fmul_rn(w_hi, b.x));
    r.y = __fadd_rn(__fmul_rn(w_lo, a.y), __fmul_rn(w_hi, b.y));
    r.z = __fadd_rn(__fmul_rn(w_lo, a.z), __fmul_rn(w_hi, b.z));
    r.w = __fadd_rn(__fmul_rn(w_lo, a.w), __fmul_rn(w_hi, b.w));
    *reinterpret_cast<float4*>(out + i) = r;
  } else {
    out[i] = __fadd_rn(__fmul_rn(w_lo, lo[i]), __fmul_rn(w_hi, hi[i]));
  }
}

// ---------------------------------------------------------------------------------------------------
// constant-elevation PPI (radar_grid/products.py:168-314): per pixel, the beam height of the requested elevation
// (4/3-earth model products.py:70-87, or flat earth :164-165) picks the altitude, then the column is sampled by
// linear interpolation between the bracketing levels (float64, products.py:276-309) or at the nearest level
// (float32, products.py:262-272).  Every operation is an IEEE basic operation in NumPy's order and dtype
// (float32 for the horizontal distance, float64 afterwards; scalars such as cos/sin of the elevation are
// evaluated by the host exactly as NumPy does), so the result is bit-identical to the reference.
// ---------------------------------------------------------------------------------------------------
struct PpiArgs {
  double cos_c;    // np.maximum(np.cos(elev), 0.01)
  double sin_e;    // np.sin(elev)
  double tan_e;    // np.tan(elev)                (flat earth)
  double ke_re;    // ke * EARTH_RADIUS
  double ke_re2;   // ke_re ** 2
  double z_min, z_max, z_step;
  int curved, nz, ny, nx;
};

__device__ __forceinline__ double ppi_target_z(const PpiArgs& a, float x, float y) {
  // products.py:240 -- float32: xx**2 + yy**2, sqrt
  const float hd = sqrtf(x * x + y * y);
  if (a.curved) {
    const double sr = (double)hd / a.cos_c;                                    // products.py:80
    const double t = ((2.0 * sr) * a.ke_re) * a.sin_e;                         // products.py:84, left to right
    return (sqrt((sr * sr + a.ke_re2) + t) - a.ke_re) + 0.0;                   // products.py:83-87, radar_altitude = 0.0
  }
  return (double)hd * a.tan_e + 0.0;                                           // products.py:165
}

template <bool LINEAR>
__global__ __launch_bounds__(rg::kBlock) void elevation_ppi_kernel(const float* __restrict__ grid,
                                                                   const float* __restrict__ xc,
                                                                   const float* __restrict__ yc, PpiArgs a,
                                                                   double* __restrict__ out64, float* __restrict__ out32) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_xy = (long)a.ny * a.nx;
  if (i >= n_xy) return;
  const int ix = (int)(i % a.nx), iy = (int)(i / a.nx);
  const double tz = ppi_target_z(a, xc[ix], yc[iy]);
  const double zf = (tz - a.z_min) / a.z_step;                                 // products.py:263,279
  if constexpr (LINEAR) {
    const double fl = floor(zf);
    const long lo = (long)fl, hi = lo + 1;                                     // products.py:280-281
    const double w_hi = zf - (double)lo, w_lo = 1.0 - w_hi;                    // products.py:284-285
    const long lo_s = lo < 0 ? 0 : (lo > a.nz - 1 ? a.nz - 1 : lo);            // products.py:293-294
    const long hi_s = hi < 0 ? 0 : (hi > a.nz - 1 ? a.nz - 1 : hi);
    const double v_lo = (double)grid[lo_s * n_xy + i], v_hi = (double)grid[hi_s * n_xy + i];
    double r = w_lo * v_lo + w_hi * v_hi;                                      // products.py:303
    if (tz < a.z_min || tz > a.z_max) r = __builtin_nan("");                  // products.py:306,309
    out64[i] = r;
  } else {
    const double rn = rint(zf);                                                // np.round: half to even
    const long k = (long)rn;
    const bool valid = k >= 0 && k < a.nz;                                     // products.py:266
    const long ks = k < 0 ? 0 : (k > a.nz - 1 ? a.nz - 1 : k);
    out32[i] = valid ? grid[ks * n_xy + i] : __builtin_nanf("");              // products.py:271-272
  }
}

template <int OP>
int launch_column(const float* grid, long n_xy, int z_lo, int z_hi, float* out, int* out_arg, hipStream_t s) {
  const bool vec = (n_xy % 4 == 0) && rg::aligned16(grid) && rg::aligned16(out) && (!out_arg || rg::aligned16(out_arg));
  const long groups = vec ? n_xy / 4 : n_xy;
  // split the level range over lanes when one lane per column group would leave most CUs idle
  const bool split = OP != RG_COL_MEAN && groups < 256L * 1024 && (z_hi - z_lo + 1) >= 8;
  if (split) {
    const long waves = (groups + 15) / 16;
    const dim3 g((unsigned)((waves + 3) / 4)), b(rg::kBlock);
    if constexpr (OP != RG_COL_MEAN) {
      if (vec) hipLaunchKernelGGL((column_kernel<OP, 4, 4>), g, b, 0, s, grid, n_xy, z_lo, z_hi, out, out_arg);
      else hipLaunchKernelGGL((column_kernel<OP, 1, 4>), g, b, 0, s, grid, n_xy, z_lo, z_hi, out, out_arg);
    }
  } else {
    const dim3 g((unsigned)((groups + rg::kBlock - 1) / rg::kBlock)), b(rg::kBlock);
    if (vec) hipLaunchKernelGGL((column_kernel<OP, 4, 1>), g, b, 0, s, grid, n_xy, z_lo, z_hi, out, out_arg);
    else hipLaunchKernelGGL((column_kernel<OP, 1, 1>), g, b, 0, s, grid, n_xy, z_lo, z_hi, out, out_arg);
  }
  return rg::check_launch("rg_column_reduce_f32");
}

}  // namespace

extern "C" int rg_column_reduce_f32(const float* grid, int32_t nz, int64_t n_xy, int32_t z_lo, int32_t z_hi,
                                    int32_t op, float* out, int32_t* out_arg, rg_stream_t stream) {
  RG_REQUIRE(grid && out, RG_EINVAL, "rg_column_reduce_f32: null pointer");
  RG_REQUIRE(nz >= 1 && n_xy >= 0, RG_EINVAL, "rg_column_reduce_f32: bad shape nz=%d n_xy=%lld", nz, (long long)n_xy);
  RG_REQUIRE(z_lo >= 0 && z_hi < nz, RG_EINVAL, "rg_column_reduce_f32: level window [%d,%d] outside 0..%d", z_lo, z_hi, nz - 1);
  RG_REQUIRE(op >= RG_COL_MAX && op <= RG_COL_MEAN, RG_EINVAL, "rg_column_reduce_f32: unknown op %d", op);
  RG_REQUIRE(!(op == RG_COL_MEAN && out_arg), RG_EINVAL, "rg_column_reduce_f32: no arg index for the mean");
  if (n_xy == 0) return RG_OK;
  hipStream_t s = (hipStream_t)stream;
  // an empty window (z_lo > z_hi) yields all-NaN / -1, like np.nanmax over an empty... (callers clip first)
  switch (op) {
    case RG_COL_MAX: return launch_column<RG_COL_MAX>(grid, n_xy, z_lo, z_hi, out, out_arg, s);
    case RG_COL_MIN: return launch_column<RG_COL_MIN>(grid, n_xy, z_lo, z_hi, out, out_arg, s);
    default: return launch_column<RG_COL_MEAN>(grid, n_xy, z_lo, z_hi, out, nullptr, s);
  }
}

extern "C" int rg_cappi_lerp_f32(const float* grid, int64_t n_xy, int32_t k_lo, float w_lo, float w_hi, float* out,
                                 rg_stream_t stream) {
  RG_REQUIRE(grid && out, RG_EINVAL, "rg_cappi_lerp_f32: null pointer");
  RG_REQUIRE(n_xy >= 0 && k_lo >= 0, RG_EINVAL, "rg_cappi_lerp_f32: bad shape");
  if (n_xy == 0) return RG_OK;
  const float* lo = grid + (size_t)k_lo * n_xy;
  const float* hi = lo + n_xy;
  const bool vec = (n_xy % 4 == 0) && rg::aligned16(lo) && rg::aligned16(out);
  hipStream_t s = (hipStream_t)stream;
  if (vec) {
    const long n4 = n_xy / 4;
    hipLaunchKernelGGL(cappi_lerp_kernel<4>, dim3((unsigned)((n4 + rg::kBlock - 1) / rg::kBlock)), dim3(rg::kBlock), 0, s,
                       lo, hi, (long)n_xy, w_lo, w_hi, out);
  } else {
    hipLaunchKernelGGL(cappi_lerp_kernel<1>, dim3((unsigned)((n_xy + rg::kBlock - 1) / rg::kBlock)), dim3(rg::kBlock), 0, s,
                       lo, hi, (long)n_xy, w_lo, w_hi, out);
  }
  return rg::check_launch("rg_cappi_lerp_f32");
}

extern "C" int rg_elevation_ppi_f32(const float* grid, const float* xc, const float* yc, int32_t nz, int32_t ny, int32_t nx,
                                    double cos_clamped, double sin_elev, double tan_elev, double ke_re, double ke_re_sq,
                                    double z_min, double z_max, double z_step, int32_t earth_curvature, int32_t linear,
                                    void* out, rg_stream_t stream) {
  RG_REQUIRE(grid && xc && yc && out, RG_EINVAL, "rg_elevation_ppi_f32: null pointer");
  RG_REQUIRE(nz >= 1 && ny >= 1 && nx >= 1, RG_EINVAL, "rg_elevation_ppi_f32: bad shape (%d,%d,%d)", nz, ny, nx);
  RG_REQUIRE(z_step != 0.0, RG_EINVAL, "rg_elevation_ppi_f32: z_step is zero");
  PpiArgs a;
  a.cos_c = cos_clamped; a.sin_e = sin_elev; a.tan_e = tan_elev; a.ke_re = ke_re; a.ke_re2 = ke_re_sq;
  a.z_min = z_min; a.z_max = z_max; a.z_step = z_step; a.curved = earth_curvature != 0; a.nz = nz; a.ny = ny; a.nx = nx;
  const long n_xy = (long)ny * nx;
  const dim3 g((unsigned)((n_xy + rg::kBlock - 1) / rg::kBlock)), b(rg::kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (linear)
    hipLaunchKernelGGL(elevation_ppi_kernel<true>, g, b, 0, s, grid, xc, yc, a, static_cast<double*>(out), (float*)nullptr);
  else
    hipLaunchKernelGGL(elevation_ppi_kernel<false>, g, b, 0, s, grid, xc, yc, a, (double*)nullptr, static_cast<float*>(out));
  return rg::check_launch("rg_elevation_ppi_f32");
}
