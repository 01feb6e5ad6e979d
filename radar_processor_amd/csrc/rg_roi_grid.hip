// K2  roi_grid -- fused on-the-fly gridding: neighbour search + weights + masked weighted mean in one kernel,
// without materialising a CSR.  Equivalent to radar_grid/compute.py:46-91 fused with
// radar_grid/interpolate.py:69-104.  Needed where the CSR is too large to keep (the 14x720x2000
// volume on a 40x2000x2000 grid is ~24 G pairs = 195 GB, 11x over the reference's int32 indptr).
//
// Bound: NOT HBM (compulsory traffic is only the sorted gates + packed fields + the output grid); the kernel is
// limited by VALU issue, mostly in the dense stage (about 21 VALU instructions per 64 (record, voxel) tests).
//
// Structure (one wavefront = a 16 x 4 patch of one grid level, processed as 4 blocks of 4 x 4 = 16 voxels; the candidates come
// from the level's OWN cell-sorted gate list when the search structure keeps one per level -- rg_geom_bin_gates_levels_f32):
//   * voxel blocking: neighbouring voxels (240 m apart, ROI >= 250 m) share almost all candidates, so every gate
//     record is loaded ONCE per 16-voxel block.  The chain cell_start -> gate record that made the first version
//     latency-bound is broken by loading the bounds of all cell rows of the block with one vector load and by
//     prefetching the next step's records before testing the current ones;
//   * candidate stage (lanes = 64 candidates per step, one dwordx4 record each): a float32 lower bound of the
//     distance to the NEAREST voxel of the block (clamped x- and y-distances to the block's extent; z is common)
//     against the block's largest radius, inflated by 2e-6 -- a conservative pre-filter, ~17 VALU per 64
//     candidates regardless of the block size; survivors are compacted with ballot + mbcnt into a per-wave LDS ring;
//   * dense stage (lanes = 16 voxels x 4 queued records per step): lane k owns voxel k of the block -- its
//     constants live in its registers, its sums never leave it -- and tests 4 records per step (LDS broadcast
//     reads).  float32 d2 decides membership whenever it is clear of the rim by 2e-6; inside that band the lane
//     falls back to the reference's exact float64 `d2 < r2` (compute.py:69-74), so the neighbour set equals the
//     CSR builder's.  Weight in float32 (|rel err| < 2e-6).  The packed field slots of a gate are fetched once,
//     when the gate is queued, and parked in a second LDS ring one candidate step later; the dense stage reads
//     records and values from LDS one step ahead of its arithmetic;
//   * per block two wavefront shuffles fold the 4 record slots; 16 lanes store 16 consecutive voxels per field.
//
// Compiled with -ffp-contract=off like every TU (the exact test must not be fused); the float32 tests use explicit
// fmaf, their error is covered by the 2e-6 band.
#include "rg_common.hpp"
#include "rg_roi_search.hpp"

namespace {

using namespace rg::roi;

// queue slots per wave (power of two): <= 63 waiting + 64 new records; with the value ring + the 64 records of the
// previous step whose field values are still in flight.
constexpr int kRingBuild = 128, kRingGrid = 256;

using rg::load_packed;

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}


// float32 weight from the float32 d2 (compute.py:82-87); relative error < 2e-6
// inv_r2q: Barnes -- MINUS log2(e) * 4 / r2, so that exp(-d2 / (r2 / 4)) is one multiply and one v_exp_f32 (= 2^x)
template <int W>
__device__ __forceinline__ float weight_from_f32(float d2f, float r2f, float inv_r2q) {
  if constexpr (W == RG_W_BARNES2) {
    return __builtin_amdgcn_exp2f(d2f * inv_r2q) + 1e-5f;
  } else if constexpr (W == RG_W_CRESSMAN) {
    return (r2f - d2f) / (r2f + d2f);
  } else {
    return 1.0f;
  }
}

// What the dense stage does with a (record, voxel) hit:
//   kGridMode   accumulate the masked weighted mean (rg_roi_grid_f32)
//   kCountMode  count it                            (rg_geom_count_f32: row lengths of the CSR)
//   kFillMode   append (gate index, float64-exact weight) to the voxel's CSR row (rg_geom_fill_f32)
// Count and fill classify hits with the same code, so the second pass writes exactly what the first one counted; a
// voxel's hits arrive in (cell row, sorted position) order, the row order the CSR has always had.
constexpr int kGridMode = 0, kCountMode = 1, kFillMode = 2;
#if !defined(RG_EXPERIMENTS) || !defined(RG_K2_BX)
#undef RG_K2_BX
#define RG_K2_BX 4            // block shape BX x (16 / BX) voxels; experiment builds: -DRG_K2_BX=8 / 16
#endif

// BX x BY = 16 voxels per block (BY rows of BX consecutive voxels); a wavefront walks 4 blocks side by side in x.
template <int MODE, int W, int NF, int STRIDE, int BX>
__global__ __launch_bounds__(rg::kBlock) void roi_block_kernel(SearchArgs a, const float* __restrict__ packed, float fill,
                                                               float* __restrict__ out, int* __restrict__ counts,
                                                               const long long* __restrict__ indptr,
                                                               int* __restrict__ gidx, float* __restrict__ wts) {
  constexpr int kVB = 16;             // voxels per block
  constexpr int BY = kVB / BX;
  constexpr int PX = 4 * BX;          // patch of one wavefront: PX x BY voxels of one level
  constexpr int kSlots = 64 / kVB;    // queued records tested per dense step
  static_assert(kSlots == 4, "the builder's slot masks assume 4 records per dense step");
  constexpr int kLgBX = BX == 16 ? 4 : BX == 8 ? 3 : 2;
  constexpr bool GRID = MODE == kGridMode;
  // Value ring (grid mode, 1-2 field slots): the packed field slots of every queued gate are fetched ONCE, when the
  // gate is queued (a queued gate hits up to 16 voxels over several dense steps; gathering per hit also puts a memory
  // round trip into every dense step).  With 4 or 8 slots the ring would cost 4-8 KiB more LDS per wavefront and an
  // unconditional 16-32 byte LDS read per lane and step -- measured 35 % slower than gathering per hit, which stays.
  constexpr bool VRING = GRID && STRIDE <= 2;
  // one field, weighted modes: the value takes the place of the gate index in the queued record (the index is needed by the
  // builder and by the closest-gate tie-break only), so the dense stage reads ONE 16-byte LDS entry per record
  constexpr bool VPACK = VRING && STRIDE == 1 && W != RG_W_CLOSEST;
  constexpr int kRing = VRING ? kRingGrid : kRingBuild;
  __shared__ rg_gate4 ring_all[rg::kBlock / rg::kWave][kRing];
  __shared__ float ringv_all[rg::kBlock / rg::kWave][(VRING && !VPACK) ? kRing * STRIDE : 1];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  rg_gate4* ring = ring_all[wv];
  float* ringv = ringv_all[wv];
  // 32-bit on purpose: a 64-bit division costs this ISA a few hundred instructions, and every wavefront does three
  // (the launcher guarantees that the patch count fits)
  const unsigned wpx = (unsigned)((a.nx + PX - 1) / PX), wpy = (unsigned)((a.ny + BY - 1) / BY);   // patches per level
  const unsigned wave = blockIdx.x * (unsigned)(rg::kBlock / rg::kWave) + (unsigned)wv;
  const unsigned iz_l = wave / (wpx * wpy);
  if (iz_l >= (unsigned)a.nz) return;                            // wave-uniform
  const unsigned rem = wave - iz_l * (wpx * wpy);
  const int iz = (int)iz_l;
  const unsigned py = rem / wpx;
  const int iy0 = (int)py * BY, ix0 = (int)(rem - py * wpx) * PX;
  const int nvy = a.ny - iy0 < BY ? a.ny - iy0 : BY;
  const double z = (double)a.zc[iz];                             // common to the whole wave
  const int lvl_off = a.c.levels > 1 ? (a.c.level0 + iz) * (a.c.ncx * a.c.ncy) : 0;   // per-level gate lists: this level's cells
  const float zf = (float)z;                                     // grid coordinates ARE float32 values: exact
  const int vl = lane & (kVB - 1);                               // voxel of the block this lane owns ...
  const int bxl = vl & (BX - 1), byl = vl >> kLgBX;              // ... at (bxl, byl) inside the block
  const int slot = lane >> 4;                                    // which of the 4 records of a dense step
  const float ya = a.yc[iy0], yb = a.yc[iy0 + nvy - 1];
  const float ylo = fminf(ya, yb), yhi = fmaxf(ya, yb);

  for (int b0 = 0; b0 < PX && ix0 + b0 < a.nx; b0 += BX) {
    const int nvx = a.nx - ix0 - b0 < BX ? a.nx - ix0 - b0 : BX;  // wave-uniform
    const bool vlive = bxl < nvx && byl < nvy;
    // ---- this lane's voxel: the reference's float64 ROI (compute.py:46-47,57) and its float32 bounds --------
    const double x = (double)a.xc[ix0 + b0 + (vlive ? bxl : 0)];
    const double y = (double)a.yc[iy0 + (vlive ? byl : 0)];
    const double dist = sqrt(x * x + y * y + z * z);
    const double r = fmax(a.min_radius, dist * a.beam_factor);
    const double r2 = r * r;
    const float xf = (float)x, yf = (float)y, r2f = (float)r2;
    // float32 d2 carries < 4e-7 relative error: outside [r2_lo, r2_hi] the float32 comparison is already exact
    const float r2_hi = vlive ? (float)(r2 * (1.0 + 2e-6)) * (1.0f + 2.4e-7f) : -1.0f;
    const float r2_lo = vlive ? (float)(r2 * (1.0 - 2e-6)) * (1.0f - 2.4e-7f) : -1.0f;  // dead lanes never hit
    const float inv_r2q = (float)(-1.4426950408889634 * 4.0 / r2);      // see weight_from_f32
    // ---- block-wide (wave-uniform) quantities ------------------------------------------------------------
    double rmax = vlive ? r : 0.0;
#pragma unroll
    for (int m = 1; m < kVB; m <<= 1) rmax = fmax(rmax, __shfl_xor(rmax, m, 64));
    rmax = readlane_f64(rmax, 0);
    const float xa = a.xc[ix0 + b0], xb = a.xc[ix0 + b0 + nvx - 1];
    const float xlo = fminf(xa, xb), xhi = fmaxf(xa, xb);
    const float r2max_hi = (float)(rmax * rmax * (1.0 + 2e-6)) * (1.0f + 2.4e-7f);
    const int cx0 = __builtin_amdgcn_readfirstlane(cell_clamped((double)xlo - rmax, a.c.x0, a.c.inv_cx, a.c.ncx));
    const int cx1 = __builtin_amdgcn_readfirstlane(cell_clamped((double)xhi + rmax, a.c.x0, a.c.inv_cx, a.c.ncx));
    const int cy0 = __builtin_amdgcn_readfirstlane(cell_clamped((double)ylo - rmax, a.c.y0, a.c.inv_cy, a.c.ncy));
    const int cy1 = __builtin_amdgcn_readfirstlane(cell_clamped((double)yhi + rmax, a.c.y0, a.c.inv_cy, a.c.ncy));

    // weighted modes: acc_p = sum w*v, acc_w = sum w.  Closest-gate mode: acc_p = value of the closest gate so far,
    // acc_w = its float32 d2 (+inf = none), best_idx = its gate index (ties go to the lower index).
    constexpr bool CLOSEST = W == RG_W_CLOSEST;
    float acc_p[NF], acc_w[NF];
    int best_idx[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { acc_p[f] = 0.0f; acc_w[f] = CLOSEST ? __builtin_inff() : 0.0f; best_idx[f] = 0x7FFFFFFF; }
    int head = 0, tail = 0;  // ring positions (wave-uniform, monotone)
    int ready = 0;           // records [head, ready) are complete (grid mode: their values have been parked)
    // builder modes: hits of this lane's voxel so far (identical in the voxel's 4 slot lanes) and its row base
    int cursor = 0;
    long long row_base = 0;
    if constexpr (MODE == kFillMode) {
      if (vlive) row_base = indptr[((size_t)iz * a.ny + (iy0 + byl)) * a.nx + (ix0 + b0 + bxl)];
    }
    const unsigned long long vox_lanes = 0x0001000100010001ull << vl;            // the 4 slot lanes of voxel vl
    const unsigned long long lower_slots = vox_lanes & ((1ull << (16 * slot)) - 1ull);

    auto dense = [&](int n) {  // test n queued records against the block's 16 voxels, 4 records per step
#if defined(RG_EXPERIMENTS) && defined(RG_K2_ABL) && RG_K2_ABL == 1     // timing-only (tools/exp_k2_breakdown.py): the dense stage
      head += n;                                                        // does nothing -- what is left is the candidate side
      return;
#endif
      // value-ring variant: the next step's record and field slots are read from LDS before this step's arithmetic,
      // so the LDS latency overlaps it; slots past n hold stale but addressable ring entries and are ignored
      rg_gate4 g_nx;
      g_nx.x = g_nx.y = g_nx.z = 0.0f; g_nx.index = 0;
      float val_nx[STRIDE];
#pragma unroll
      for (int f = 0; f < STRIDE; ++f) val_nx[f] = 0.0f;
      if constexpr (VRING) {
        g_nx = ring[(head + slot) & (kRing - 1)];
        if constexpr (VPACK) val_nx[0] = __builtin_bit_cast(float, g_nx.index);
        else load_packed<STRIDE>(ringv, (unsigned)((head + slot) & (kRing - 1)), val_nx);
      }
      for (int e0 = 0; e0 < n; e0 += kSlots) {
        const int e = e0 + slot;
        bool in = false;
        rg_gate4 g = g_nx;
        float d2f = 0.0f;
        float val[STRIDE];
#pragma unroll
        for (int f = 0; f < STRIDE; ++f) val[f] = val_nx[f];
        if constexpr (VRING) {
          if (e0 + kSlots < n) {
            g_nx = ring[(head + e + kSlots) & (kRing - 1)];
            if constexpr (VPACK) val_nx[0] = __builtin_bit_cast(float, g_nx.index);
            else load_packed<STRIDE>(ringv, (unsigned)((head + e + kSlots) & (kRing - 1)), val_nx);
          }
          const float dx = g.x - xf, dy = g.y - yf, dz = g.z - zf;
          d2f = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          in = e < n && d2f <= r2_lo;
          if (e < n && !in && d2f <= r2_hi) {  // within 2e-6 of the rim: the reference's float64 arithmetic decides
            const double ex = (double)g.x - x, ey = (double)g.y - y, ez = (double)g.z - z;  // compute.py:69-71
            in = ex * ex + ey * ey + ez * ez < r2;                                          // compute.py:72,74
          }
        } else if (e < n) {
          g = ring[(head + e) & (kRing - 1)];
          const float dx = g.x - xf, dy = g.y - yf, dz = g.z - zf;
          d2f = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          in = d2f <= r2_lo;
          if (!in && d2f <= r2_hi) {  // within 2e-6 of the rim: the reference's float64 arithmetic decides
            const double ex = (double)g.x - x, ey = (double)g.y - y, ez = (double)g.z - z;  // compute.py:69-71
            in = ex * ex + ey * ey + ez * ez < r2;                                          // compute.py:72,74
          }
        }
        if constexpr (MODE != kGridMode) {
          const unsigned long long hits = __ballot(in);   // executed by every lane of the wave
          if constexpr (MODE == kFillMode) {
            if (in) {
              const double ex = (double)g.x - x, ey = (double)g.y - y, ez = (double)g.z - z;
              const double d2 = ex * ex + ey * ey + ez * ez;                                // compute.py:72
              const long long pos = row_base + cursor + __popcll(hits & lower_slots);
              gidx[pos] = g.index;
              wts[pos] = roi_weight<W>(d2, r2);                                             // compute.py:82-87
            }
          }
          cursor += __popcll(hits & vox_lanes);
        } else {
          if (in) {
            const float w = weight_from_f32<W>(d2f, r2f, inv_r2q);
            if constexpr (!VRING) load_packed<STRIDE>(packed, (unsigned)g.index, val);   // one gather per hit
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const bool ok = rg::f32_bits(val[f]) != RG_EXCLUDED_BITS;
              if constexpr (CLOSEST) {
                const bool better = ok && (d2f < acc_w[f] || (d2f == acc_w[f] && g.index < best_idx[f]));
                acc_p[f] = better ? val[f] : acc_p[f];
                acc_w[f] = better ? d2f : acc_w[f];
                best_idx[f] = better ? g.index : best_idx[f];
              } else {
                acc_p[f] += ok ? w * val[f] : 0.0f;  // float32 product, as interpolate.py:82
                acc_w[f] += ok ? w : 0.0f;
              }
            }
          }
        }
      }
      head += n;
    };

    // grid mode: the survivor this lane queued in the previous candidate step and its values, still in flight
    bool pend = false;
    int pend_pos = 0;
    float pend_val[STRIDE];
#pragma unroll
    for (int f = 0; f < STRIDE; ++f) pend_val[f] = 0.0f;
    auto flush_pending = [&]() {
      if (pend) {
        if constexpr (VPACK) {
          ring[pend_pos].index = __builtin_bit_cast(int, pend_val[0]);
        } else {
#pragma unroll
          for (int f = 0; f < STRIDE; ++f) ringv[pend_pos * STRIDE + f] = pend_val[f];
        }
      }
      pend = false;
      ready = tail;
    };

    const int nrows = cy1 - cy0 + 1;
    for (int rb = 0; rb < nrows; rb += 64) {
      // bounds of up to 64 cell rows with one vector load each (lane <-> cell row)
      int rs_l = 0, re_l = 0;
      if (rb + lane < nrows) {
        const int base = lvl_off + (cy0 + rb + lane) * a.c.ncx;
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
        if constexpr (VRING) flush_pending();   // the values requested one step ago have had that step to arrive
        // lower bound of the distance to the nearest voxel of the block vs the block's largest (inflated) radius
        const float dz = g.z - zf;
        const float dxb = fmaxf(fmaxf(xlo - g.x, g.x - xhi), 0.0f);
        const float dyb = fmaxf(fmaxf(ylo - g.y, g.y - yhi), 0.0f);
        const float d2min = __builtin_fmaf(dxb, dxb, __builtin_fmaf(dyb, dyb, dz * dz));
        const bool pre = valid && d2min <= r2max_hi;
        const unsigned long long m = __ballot(pre);
        if (pre) {
          const int pos = tail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          ring[pos & (kRing - 1)] = g;
          if constexpr (VRING) {  // request the gate's field slots now, park them in the ring one step later
            pend = true;
            pend_pos = pos & (kRing - 1);
            load_packed<STRIDE>(packed, (unsigned)g.index, pend_val);
          }
        }
        tail += __popcll(m);
        if constexpr (!VRING) ready = tail;
        if (ready - head >= 64) {   // only records whose values are in the ring (without a value ring: ready == tail)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          dense(64);
        }
      }
    }
    if constexpr (VRING) flush_pending();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    dense(tail - head);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if constexpr (MODE == kCountMode) {
      if (slot == 0 && vlive) counts[((size_t)iz * a.ny + (iy0 + byl)) * a.nx + (ix0 + b0 + bxl)] = cursor;
      continue;
    } else if constexpr (MODE == kFillMode) {
      continue;
    }
    // ---- fold the 4 record slots; lanes 0..15 hold the block's 16 voxels ----------------------------------
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float p = acc_p[f], w = acc_w[f];
      if constexpr (CLOSEST) {
        int bi = best_idx[f];
#pragma unroll
        for (int m = kVB; m < 64; m <<= 1) {
          const float op = __shfl_xor(p, m, 64), ow = __shfl_xor(w, m, 64);
          const int oi = __shfl_xor(bi, m, 64);
          const bool take = ow < w || (ow == w && oi < bi);
          p = take ? op : p; w = take ? ow : w; bi = take ? oi : bi;
        }
      } else {
#pragma unroll
        for (int m = kVB; m < 64; m <<= 1) { p += __shfl_xor(p, m, 64); w += __shfl_xor(w, m, 64); }
      }
      if (slot == 0 && vlive) {
        const size_t v = ((size_t)iz * a.ny + (iy0 + byl)) * a.nx + (ix0 + b0 + bxl);
        if constexpr (CLOSEST) out[(size_t)f * a.n_vox + v] = w < __builtin_inff() ? p : fill;
        else out[(size_t)f * a.n_vox + v] = w > 0.0f ? (float)((double)p / (double)w) : fill;
      }
    }
  }
}

template <int BX>
inline dim3 k2_grid(const SearchArgs& a) {
  constexpr int BY = 16 / BX, PX = 4 * BX;
  const long waves = (long)((a.nx + PX - 1) / PX) * ((a.ny + BY - 1) / BY) * a.nz;
  return dim3((unsigned)((waves + 3) / 4));
}

template <int W, int NF, int STRIDE>
int launch(const SearchArgs& a, const float* packed, float fill, float* out, hipStream_t s) {
  // block shape: with one gate list for all levels 8 x 2 measured best (16.0 ms on the bench grid against 16.5 for 4 x 4 and 19.4
  // for 16 x 1); with the per-level lists, whose candidate stage is a third as long, the more compact 4 x 4 block (fewer survivors
  // per block) wins: 11.1 against 11.6 ms
  hipLaunchKernelGGL((roi_block_kernel<kGridMode, W, NF, STRIDE, RG_K2_BX>), k2_grid<RG_K2_BX>(a), dim3(rg::kBlock), 0, s, a, packed, fill,
                     out, (int*)nullptr, (const long long*)nullptr, (int*)nullptr, (float*)nullptr);
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
  RG_REQUIRE(weighting >= RG_W_BARNES2 && weighting <= RG_W_CLOSEST, RG_EINVAL, "rg_roi_grid_f32: unknown weighting %d",
             weighting);
  RG_REQUIRE(n_fields >= 1 && n_fields <= RG_MAX_FIELDS, RG_EUNSUPPORTED, "rg_roi_grid_f32: n_fields=%d not in 1..%d",
             n_fields, RG_MAX_FIELDS);
  RG_REQUIRE(stride == stride_for(n_fields), RG_EINVAL, "rg_roi_grid_f32: stride=%d, expected %d for %d fields", stride,
             stride_for(n_fields), n_fields);
  RG_REQUIRE(rg::aligned16(packed), RG_EALIGN, "rg_roi_grid_f32: packed must be 16-byte aligned");
  RG_REQUIRE((long)nx * ny * nz > 0 &&
                 (long)((nx + 4 * RG_K2_BX - 1) / (4 * RG_K2_BX)) * ((ny + 16 / RG_K2_BX - 1) / (16 / RG_K2_BX)) * nz < 0xFFFFFFF0L,
             RG_EUNSUPPORTED,
             "rg_roi_grid_f32: grid too large for one launch");
  const SearchArgs a = make_args(sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx, min_radius, beam_factor);
  hipStream_t s = (hipStream_t)stream;
  switch (weighting) {
    case RG_W_BARNES2: return dispatch<RG_W_BARNES2>(n_fields, a, packed, fill_value, out, s);
    case RG_W_CRESSMAN: return dispatch<RG_W_CRESSMAN>(n_fields, a, packed, fill_value, out, s);
    case RG_W_CLOSEST: return dispatch<RG_W_CLOSEST>(n_fields, a, packed, fill_value, out, s);
    default: return dispatch<RG_W_NEAREST>(n_fields, a, packed, fill_value, out, s);
  }
}

// ---------------------------------------------------------------------------------------------------
// a6/a7 builder passes on the same block kernel: radar_grid/compute.py:54-91 for every voxel, and the CSR merge of
// compute.py:232-272 replaced by count -> prefix sum (rg_scan_counts_i64, rg_geometry.hip) -> fill.
// ---------------------------------------------------------------------------------------------------
extern "C" int rg_geom_count_f32(const rg_gate4* sorted_gates, const int32_t* cell_start, const rg_cellgrid* cells_host,
                                 const float* xc, const float* yc, const float* zc, int32_t nz, int32_t ny, int32_t nx,
                                 double min_radius, double beam_factor, int32_t* counts, rg_stream_t stream) {
  const int rc = check_search_args("rg_geom_count_f32", sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx);
  if (rc != RG_OK) return rc;
  RG_REQUIRE(counts, RG_EINVAL, "rg_geom_count_f32: null counts");
  const SearchArgs a = make_args(sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx, min_radius, beam_factor);
  hipLaunchKernelGGL((roi_block_kernel<kCountMode, RG_W_NEAREST, 1, 1, RG_K2_BX>), k2_grid<RG_K2_BX>(a), dim3(rg::kBlock), 0,
                     (hipStream_t)stream, a, (const float*)nullptr, 0.0f, (float*)nullptr, counts,
                     (const long long*)nullptr, (int*)nullptr, (float*)nullptr);
  return rg::check_launch("rg_geom_count_f32");
}

extern "C" int rg_geom_fill_f32(const rg_gate4* sorted_gates, const int32_t* cell_start, const rg_cellgrid* cells_host,
                                const float* xc, const float* yc, const float* zc, int32_t nz, int32_t ny, int32_t nx,
                                double min_radius, double beam_factor, int32_t weighting, const int64_t* indptr,
                                int32_t* gate_idx, float* weights, rg_stream_t stream) {
  const int rc = check_search_args("rg_geom_fill_f32", sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx);
  if (rc != RG_OK) return rc;
  RG_REQUIRE(indptr && gate_idx && weights, RG_EINVAL, "rg_geom_fill_f32: null pointer");
  RG_REQUIRE(weighting >= RG_W_BARNES2 && weighting <= RG_W_NEAREST, RG_EINVAL, "rg_geom_fill_f32: unknown weighting %d",
             weighting);
  const SearchArgs a = make_args(sorted_gates, cell_start, cells_host, xc, yc, zc, nz, ny, nx, min_radius, beam_factor);
  const dim3 grid = k2_grid<RG_K2_BX>(a), block(rg::kBlock);
  hipStream_t s = (hipStream_t)stream;
  const long long* ip = reinterpret_cast<const long long*>(indptr);
#define RG_FILL(W_)                                                                                                     \
  hipLaunchKernelGGL((roi_block_kernel<kFillMode, W_, 1, 1, RG_K2_BX>), grid, block, 0, s, a, (const float*)nullptr, 0.0f,     \
                     (float*)nullptr, (int*)nullptr, ip, gate_idx, weights)
  switch (weighting) {
    case RG_W_BARNES2: RG_FILL(RG_W_BARNES2); break;
    case RG_W_CRESSMAN: RG_FILL(RG_W_CRESSMAN); break;
    default: RG_FILL(RG_W_NEAREST); break;
  }
#undef RG_FILL
  return rg::check_launch("rg_geom_fill_f32");
}
