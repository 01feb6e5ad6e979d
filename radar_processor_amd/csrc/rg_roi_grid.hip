// K2  roi_grid -- fused on-the-fly gridding: neighbour search + weights + masked weighted mean in one kernel,
// without materialising a CSR.  Equivalent to radar_grid/compute.py:46-91 fused with
// radar_grid/interpolate.py:69-104.  Needed where the CSR is too large to keep (the 14x720x2000
// volume on a 40x2000x2000 grid is ~24 G pairs = 195 GB, 11x over the reference's int32 indptr).
//
// Bound: NOT HBM (compulsory traffic is only the sorted gates + packed fields + the output grid); the kernel is
// limited by VALU issue for the candidate test and by the L1/L2-served loads of gate records.
//
// Structure (one wavefront = up to 64 consecutive voxels of ONE grid row, processed in groups of VB = 4):
//   * lane l first computes voxel l's search box in the reference's float64 arithmetic (compute.py:46-47,57) --
//     64 boxes for the price of one; the group loop then broadcasts them with readlane (SGPRs);
//   * voxel blocking: neighbouring voxels (240 m apart, ROI >= 250 m) share almost all candidates, so every gate
//     record is loaded ONCE per group and tested against all VB voxels (y and z are common to the row, only x
//     differs): 4x fewer loads, steps and gathers than one voxel at a time.  The kernel was latency-bound on the
//     dependent chain cell_start -> gate record; the chain is broken by loading the bounds of all cell rows of
//     the group with one vector load and by prefetching the next step's records before testing the current;
//   * candidate test: 64 gates per step, one dwordx4 record each, float32 distance against a slightly INFLATED
//     radius (r2 * (1 + 2e-6), rounded up): a conservative pre-filter that can only admit extra candidates;
//   * survivors are compacted with ballot + mbcnt into a per-wave LDS ring together with a VB-bit mask of the
//     voxels they may belong to; whenever 64 are queued, and at the end of the group, they are processed on
//     DENSE lanes: exact float64 d2 per voxel and the reference's strict `d2 < r2` (compute.py:69-74) -- so the
//     neighbour set equals the CSR builder's --, the weight (float32 exp; |rel err| < 1e-6), ONE gather from the
//     packed fields shared by the VB voxels, masked accumulation;
//   * per voxel a wavefront shuffle reduction of the lane partials; lane t keeps voxel t's result and the wave
//     finishes with one coalesced 256-byte store per field.
//
// Compiled with -ffp-contract=off like every TU (the exact test must not be fused); the pre-filter uses explicit
// fmaf, its error is covered by the inflation.
#include "rg_common.hpp"
#include "rg_roi_search.hpp"

namespace {

using namespace rg::roi;

constexpr int kVoxPerWaveK2 = 64;
constexpr int kRing = 128;  // queue slots per wave (power of two, >= 2 * 64)

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

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// float32 weight from the exact float64 d2 / r2 (compute.py:82-87); relative error < 1e-6
template <int W>
__device__ __forceinline__ float weight_f32(double d2, double r2, float inv_r2q) {
  if constexpr (W == RG_W_BARNES2) {
    return __expf(-((float)d2 * inv_r2q)) + 1e-5f;
  } else if constexpr (W == RG_W_CRESSMAN) {
    return (float)(r2 - d2) / (float)(r2 + d2);
  } else {
    return 1.0f;
  }
}

constexpr int kMaskShift = 28;  // queue entries carry the voxel mask in the top bits of the gate index

template <int W, int NF, int STRIDE, int VB>
__global__ __launch_bounds__(rg::kBlock) void roi_grid_kernel(SearchArgs a, const float* __restrict__ packed, float fill,
                                                              float* __restrict__ out) {
  static_assert(VB >= 1 && VB <= 4, "voxel mask has 4 bits");
  __shared__ rg_gate4 ring_all[rg::kBlock / rg::kWave][kRing];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  rg_gate4* ring = ring_all[wv];
  const int wpr = (a.nx + kVoxPerWaveK2 - 1) / kVoxPerWaveK2;  // waves per grid row
  const long wave = (long)blockIdx.x * (rg::kBlock / rg::kWave) + wv;
  const long grow = wave / wpr;                                  // grid row = iz * ny + iy
  if (grow >= (long)a.nz * a.ny) return;                         // wave-uniform
  const int ix0 = (int)(wave - grow * wpr) * kVoxPerWaveK2;
  const int n_here = a.nx - ix0 < kVoxPerWaveK2 ? a.nx - ix0 : kVoxPerWaveK2;
  const int iy = (int)(grow % a.ny), iz = (int)(grow / a.ny);
  const double y = (double)a.yc[iy], z = (double)a.zc[iz];       // common to the whole wave
  const float yf = (float)y, zf = (float)z;                      // grid coordinates ARE float32 values: exact
  const long vbeg = grow * a.nx + ix0;

  // ---- 64 search boxes at once: lane l <-> voxel ix0 + l ----------------------------------------------
  double bx, br2;
  int bcx0, bcx1, bcy0, bcy1;
  {
    bx = (double)a.xc[ix0 + (lane < n_here ? lane : 0)];
    const double dist = sqrt(bx * bx + y * y + z * z);           // compute.py:46
    const double r = fmax(a.min_radius, dist * a.beam_factor);   // compute.py:47
    br2 = r * r;                                                 // compute.py:57
    bcx0 = cell_clamped(bx - r, a.c.x0, a.c.inv_cx, a.c.ncx);
    bcx1 = cell_clamped(bx + r, a.c.x0, a.c.inv_cx, a.c.ncx);
    bcy0 = cell_clamped(y - r, a.c.y0, a.c.inv_cy, a.c.ncy);
    bcy1 = cell_clamped(y + r, a.c.y0, a.c.inv_cy, a.c.ncy);
  }

  float my_res[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) my_res[f] = fill;

  for (int g0 = 0; g0 < n_here; g0 += VB) {
    const int nv = n_here - g0 < VB ? n_here - g0 : VB;  // wave-uniform
    // per-voxel constants of the group (SGPRs) and the union of the VB search boxes
    double xk[VB], r2k[VB];
    float xfk[VB], r2hik[VB], iqk[VB];
    int cx0 = 0x7FFFFFFF, cx1 = -1, cy0 = 0x7FFFFFFF, cy1 = -1;
#pragma unroll
    for (int k = 0; k < VB; ++k) {
      const int src = g0 + (k < nv ? k : 0);
      xk[k] = readlane_f64(bx, src);
      r2k[k] = readlane_f64(br2, src);
      xfk[k] = (float)xk[k];
      // inflated float32 radius: fl32 distance error < 4e-7 relative, r2 -> float rounding 6e-8
      r2hik[k] = k < nv ? (float)(r2k[k] * (1.0 + 2e-6)) * (1.0f + 2.4e-7f) : -1.0f;
      iqk[k] = (float)(4.0 / r2k[k]);
      const int a0 = __builtin_amdgcn_readlane(bcx0, src), a1 = __builtin_amdgcn_readlane(bcx1, src);
      const int b0 = __builtin_amdgcn_readlane(bcy0, src), b1 = __builtin_amdgcn_readlane(bcy1, src);
      cx0 = a0 < cx0 ? a0 : cx0; cx1 = a1 > cx1 ? a1 : cx1;
      cy0 = b0 < cy0 ? b0 : cy0; cy1 = b1 > cy1 ? b1 : cy1;
    }

    float acc_p[VB][NF], acc_w[VB][NF];
#pragma unroll
    for (int k = 0; k < VB; ++k)
#pragma unroll
      for (int f = 0; f < NF; ++f) { acc_p[k][f] = 0.0f; acc_w[k][f] = 0.0f; }
    int head = 0, tail = 0;  // ring positions (wave-uniform, monotone)

    auto process = [&](int n) {  // n <= 64 queued candidates on dense lanes
      if (lane < n) {
        const rg_gate4 g = ring[(head + lane) & (kRing - 1)];
        const unsigned mask = (unsigned)g.index >> kMaskShift;
        const unsigned gate = (unsigned)g.index & ((1u << kMaskShift) - 1);
        const double gx = (double)g.x;
        const double dy = (double)g.y - y, dz = (double)g.z - z;   // compute.py:70-71
        const double dy2 = dy * dy, dz2 = dz * dz;
        float w[VB];
        bool any = false;
#pragma unroll
        for (int k = 0; k < VB; ++k) {
          const double dx = gx - xk[k];                            // compute.py:69
          const double d2 = dx * dx + dy2 + dz2;                   // compute.py:72, same association
          const bool in = ((mask >> k) & 1u) != 0 && d2 < r2k[k];  // compute.py:74
          w[k] = in ? weight_f32<W>(d2, r2k[k], iqk[k]) : 0.0f;
          any = any || in;
        }
        if (any) {
          float val[STRIDE];
          load_packed<STRIDE>(packed, gate, val);
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const bool ok = rg::f32_bits(val[f]) != RG_EXCLUDED_BITS;
            const float v = ok ? val[f] : 0.0f;
#pragma unroll
            for (int k = 0; k < VB; ++k) {
              // w[k] > 0 <=> the gate is a neighbour of voxel k; an unmasked NaN/Inf value must only reach those
              acc_p[k][f] += w[k] > 0.0f ? w[k] * v : 0.0f;         // float32 product, as interpolate.py:82
              acc_w[k][f] += ok ? w[k] : 0.0f;
            }
          }
        }
      }
      head += n;
    };

    const int nrows = cy1 - cy0 + 1;
    for (int rb = 0; rb < nrows; rb += 64) {
      // bounds of up to 64 cell rows with one vector load each (lane <-> cell row)
      int rs_l = 0, re_l = 0;
      if (rb + lane < nrows) {
        const int base = (cy0 + rb + lane) * a.c.ncx;
        rs_l = a.cell_start[base + cx0];
        re_l = a.cell_start[base + cx1 + 1];
      }
      const int nr = nrows - rb < 64 ? nrows - rb : 64;
      int row = -1, jb = 0, je = 0;
      auto advance = [&]() -> bool {  // next 64-candidate step; all state wave-uniform
        jb += 64;
        while (jb >= je) {
          if (++row >= nr) return false;
          jb = __builtin_amdgcn_readlane(rs_l, row);
          je = __builtin_amdgcn_readlane(re_l, row);
        }
        return true;
      };
      bool have = advance();
      rg_gate4 gn;
      gn.x = gn.y = gn.z = 0.0f; gn.index = 0;
      bool vn = false;
      if (have) { vn = jb + lane < je; if (vn) gn = a.sorted[jb + lane]; }
      while (have) {
        const rg_gate4 g = gn;
        const bool valid = vn;
        have = advance();
        if (have) { vn = jb + lane < je; if (vn) gn = a.sorted[jb + lane]; }  // prefetch the next step
        // float32 pre-filter against all VB voxels of the group (y, z shared)
        const float dy = g.y - yf, dz = g.z - zf;
        const float dyz2 = __builtin_fmaf(dz, dz, dy * dy);
        unsigned mask = 0;
#pragma unroll
        for (int k = 0; k < VB; ++k) {
          const float dx = g.x - xfk[k];
          mask |= (__builtin_fmaf(dx, dx, dyz2) <= r2hik[k] ? 1u : 0u) << k;
        }
        const bool pre = valid && mask != 0;
        const unsigned long long m = __ballot(pre);
        if (pre) {
          const int pos = tail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          rg_gate4 q = g;
          q.index = (int)((unsigned)g.index | (mask << kMaskShift));
          ring[pos & (kRing - 1)] = q;
        }
        tail += __popcll(m);
        if (tail - head >= 64) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          process(64);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    process(tail - head);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

#pragma unroll
    for (int k = 0; k < VB; ++k) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const float p = wave_sum_f32(acc_p[k][f]);
        const float w = wave_sum_f32(acc_w[k][f]);
        const float r = w > 0.0f ? (float)((double)p / (double)w) : fill;
        if (k < nv && lane == g0 + k) my_res[f] = r;
      }
    }
  }

  if (lane < n_here) {
#pragma unroll
    for (int f = 0; f < NF; ++f) out[(size_t)f * a.n_vox + vbeg + lane] = my_res[f];
  }
}

inline dim3 k2_grid(const SearchArgs& a) {
  const long wpr = (a.nx + kVoxPerWaveK2 - 1) / kVoxPerWaveK2;
  const long waves = wpr * a.ny * a.nz;
  return dim3((unsigned)((waves + 3) / 4));
}

template <int W, int NF, int STRIDE>
int launch(const SearchArgs& a, const float* packed, float fill, float* out, hipStream_t s) {
  constexpr int VB = NF <= 4 ? 4 : 2;   // accumulator registers: 2 * VB * NF
  hipLaunchKernelGGL((roi_grid_kernel<W, NF, STRIDE, VB>), k2_grid(a), dim3(rg::kBlock), 0, s, a, packed, fill, out);
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
  RG_REQUIRE((long)nx * ny * nz > 0 && (long)((nx + 63) / 64) * ny * nz < 0x3FFFFFFFFL, RG_EUNSUPPORTED,
             "rg_roi_grid_f32: grid too large for one launch");
  const SearchArgs a = make_args(sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx, min_radius, beam_factor);
  hipStream_t s = (hipStream_t)stream;
  switch (weighting) {
    case RG_W_BARNES2: return dispatch<RG_W_BARNES2>(n_fields, a, packed, fill_value, out, s);
    case RG_W_CRESSMAN: return dispatch<RG_W_CRESSMAN>(n_fields, a, packed, fill_value, out, s);
    default: return dispatch<RG_W_NEAREST>(n_fields, a, packed, fill_value, out, s);
  }
}
