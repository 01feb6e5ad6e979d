// The tile -> row reduction shared by rg_csr_apply_f32 (K1) and rg_csr_compact_apply_f32 (K1c).
//
// Both kernels stream a segment's pairs in tiles of TILE pairs, turn every pair into masked float32 products and park
// them in the wavefront's private LDS tile (pair order); this header is the other half: the rows that touch the tile
// share the 64 lanes, every lane sums a strided subset of its row's pairs, the lanes of a row are folded with a
// butterfly and the row's running sums live in a small LDS array.  Keeping it in ONE place is what makes the two
// kernels agree bit for bit: same lane split, same order of float32 additions.
//
// LDS image of a tile (per wavefront):
//   one field   f32x2 tile[TILE]                       (w*v, w) per pair, 0 for an excluded gate
//   F > 1       float  P[TILE][STRIDE], W[TILE][STRIDE] w*v_f and the masked weight per field slot, 0 where excluded or
//                                                      padding -- two arrays, so a pair's entry is one 8 / 16 / 32-byte
//                                                      vector access at a lane stride of the same size (conflict-free)
// Lane split: L = 2^lg lanes per row, the largest power of two <= 64 / (rows touching the tile); lane k of a row sums
// pairs k, k+L, k+2L, ... of the row's part of the tile with two independent accumulators per value (elements k, k+2L,
// ... and k+L, k+3L, ...), ALL F fields at once ("whole-entry" lanes: one vector read feeds 2F additions, where one
// lane per field slot would spend a read and a loop trip per single addition pair).
// Fold: xor-butterfly over the L lanes.  Steps 1..8 run on the DPP cross-lane path (quad_perm / row_half_mirror /
// row_mirror: after the earlier steps all lanes of a quad / half row hold identical bits, so a mirror delivers exactly
// the xor partner's value); 16 and 32 -- at most two rows in a tile -- go through ds_bpermute.
#pragma once

#include "rg_common.hpp"

namespace rg {

using f32x2 = float __attribute__((ext_vector_type(2)));
using f32x4 = float __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Butterfly over the 2^lg lanes of a row, N values at once: v[i] += v[i] of lane (lane ^ m) for m = 1, 2, ... 2^(lg-1).
// One uniform branch per step for all N values.  quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror (lane i <- lane 7-i
// of its half row), row_mirror (lane i <- lane 15-i of its row), then ds_bpermute for 16 and 32.
template <int N>
__device__ __forceinline__ void butterfly(float (&v)[N], int nsub) {
  if (nsub > 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f32<0xB1>(v[i]);
  }
  if (nsub > 2) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f32<0x4E>(v[i]);
  }
  if (nsub > 4) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f32<0x141>(v[i]);
  }
  if (nsub > 8) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f32<0x140>(v[i]);
  }
  if (nsub > 16) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __shfl_xor(v[i], 16, 64);
  }
  if (nsub > 32) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __shfl_xor(v[i], 32, 64);
  }
}

constexpr int tile_floats(int nf, int stride) { return nf == 1 ? 2 : 2 * stride; }   // LDS floats per pair

// ---- product side: one pair -> its LDS entry --------------------------------------------------------------------
template <int NF, int STRIDE>
__device__ __forceinline__ void store_products(float* __restrict__ tile, int tile_pairs, int e, float w,
                                               const float (&v)[STRIDE]) {
  if constexpr (NF == 1) {
    const bool ok = f32_bits(v[0]) != RG_EXCLUDED_BITS;
    f32x2 p;
    p.x = ok ? w * v[0] : 0.0f;
    p.y = ok ? w : 0.0f;
    reinterpret_cast<f32x2*>(tile)[e] = p;
  } else {
    float p[STRIDE], m[STRIDE];
#pragma unroll
    for (int f = 0; f < STRIDE; ++f) {   // padding slots hold the sentinel -> (0, 0)
      const bool ok = f < NF && f32_bits(v[f]) != RG_EXCLUDED_BITS;
      p[f] = ok ? w * v[f] : 0.0f;
      m[f] = ok ? w : 0.0f;
    }
    float* __restrict__ pp = tile + (size_t)e * STRIDE;
    float* __restrict__ ww = tile + (size_t)tile_pairs * STRIDE + (size_t)e * STRIDE;
    if constexpr (STRIDE == 2) {
      *reinterpret_cast<f32x2*>(pp) = (f32x2){p[0], p[1]};
      *reinterpret_cast<f32x2*>(ww) = (f32x2){m[0], m[1]};
    } else {
#pragma unroll
      for (int q = 0; q < STRIDE; q += 4) {
        *reinterpret_cast<f32x4*>(pp + q) = (f32x4){p[q], p[q + 1], p[q + 2], p[q + 3]};
        *reinterpret_cast<f32x4*>(ww + q) = (f32x4){m[q], m[q + 1], m[q + 2], m[q + 3]};
      }
    }
  }
}

// ---- row side ---------------------------------------------------------------------------------------------------
// rs_o / re_o: this lane's row = pairs [rs_o, re_o) of the segment (lane == row); t = first pair of the tile.
// rowacc: f32x2[64 * STRIDE], entry (row, f) = running (sum w*v, sum w).
template <int NF, int STRIDE, int TILE>
__device__ __forceinline__ void row_phase(const float* __restrict__ tile, f32x2* __restrict__ rowacc, int t, int rs_o,
                                          int re_o, int lane) {
  const unsigned long long act = __ballot(re_o > rs_o && re_o > t && rs_o < t + TILE);
  if (act == 0) return;  // wave-uniform
  const int ra = __builtin_ctzll(act), rb = 63 - __builtin_clzll(act);
  const int nact = rb - ra + 1;
  const int lg = 31 - __builtin_clz(64 / nact);    // lanes per row = 2^lg <= 64 / rows
  const int rpr = 64 >> lg;                        // rows per round
  const int sub = lane & ((1 << lg) - 1), nsub = 1 << lg;
  for (int rbase = ra; rbase <= rb; rbase += rpr) {
    const int myrow = rbase + (lane >> lg);
    const bool live = myrow <= rb;
    const int qs = __shfl(rs_o, myrow & 63, 64);
    const int qe = __shfl(re_o, myrow & 63, 64);
    const int a = (qs > t ? qs : t) - t;
    const int b = live ? (qe < t + TILE ? qe : t + TILE) - t : a;
    if constexpr (NF == 1) {
      const f32x2* __restrict__ tl = reinterpret_cast<const f32x2*>(tile);
      f32x2 part0 = (f32x2)(0.0f), part1 = (f32x2)(0.0f);
      int j = a + sub;
      for (; j + nsub < b; j += 2 * nsub) {  // two elements per trip, two independent partial sums
        part0 += tl[j];
        part1 += tl[j + nsub];
      }
      if (j < b) part0 += tl[j];
      const f32x2 sum = part0 + part1;
      float sv[2] = {sum.x, sum.y};
      butterfly<2>(sv, nsub);
      if (live && sub == 0) rowacc[myrow] += (f32x2){sv[0], sv[1]};  // one owner per row: plain read-modify-write
    } else {
      const float* __restrict__ pp = tile;
      const float* __restrict__ ww = tile + (size_t)TILE * STRIDE;
      float p0[NF], w0[NF], p1[NF], w1[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) p0[f] = w0[f] = p1[f] = w1[f] = 0.0f;
      auto add = [&](float (&ap)[NF], float (&aw)[NF], int j) {
        float ep[STRIDE], ew[STRIDE];
        if constexpr (STRIDE == 2) {
          const f32x2 x = *reinterpret_cast<const f32x2*>(pp + (size_t)j * 2);
          const f32x2 y = *reinterpret_cast<const f32x2*>(ww + (size_t)j * 2);
          ep[0] = x.x; ep[1] = x.y; ew[0] = y.x; ew[1] = y.y;
        } else {
#pragma unroll
          for (int q = 0; q < STRIDE; q += 4) {
            if (q < NF) {   // whole vectors of padding slots are not even read
              const f32x4 x = *reinterpret_cast<const f32x4*>(pp + (size_t)j * STRIDE + q);
              const f32x4 y = *reinterpret_cast<const f32x4*>(ww + (size_t)j * STRIDE + q);
              ep[q] = x.x; ep[q + 1] = x.y; ep[q + 2] = x.z; ep[q + 3] = x.w;
              ew[q] = y.x; ew[q + 1] = y.y; ew[q + 2] = y.z; ew[q + 3] = y.w;
            }
          }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          ap[f] += ep[f];
          aw[f] += ew[f];
        }
      };
      int j = a + sub;
      for (; j + nsub < b; j += 2 * nsub) {
        add(p0, w0, j);
        add(p1, w1, j + nsub);
      }
      if (j < b) add(p0, w0, j);
      float sv[2 * NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        sv[2 * f] = p0[f] + p1[f];
        sv[2 * f + 1] = w0[f] + w1[f];
      }
      butterfly<2 * NF>(sv, nsub);
      if (live && sub == 0) {
#pragma unroll
        for (int f = 0; f < NF; ++f) rowacc[myrow * STRIDE + f] += (f32x2){sv[2 * f], sv[2 * f + 1]};
      }
    }
  }
}

}  // namespace rg
