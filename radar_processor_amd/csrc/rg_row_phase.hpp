// The tile -> row reduction shared by rg_csr_apply_f32 (K1) and rg_csr_compact_apply_f32 (K1c).
//
// Both kernels stream a segment's pairs in tiles of TILE pairs, turn every pair into masked float32 products and park
// them in the wavefront's private LDS tile (pair order); this header is the other half: the rows that touch the tile
// share the 64 lanes, every lane sums a strided subset of its row's pairs, the lanes of a row are folded with a
// butterfly and the row's running sums live in a small LDS array.  Keeping it in ONE place is what makes the two
// kernels agree bit for bit: same lane split, same order of float32 additions.
//
// LDS image of a tile (per wavefront):
//   one field   f32x2 tile[TILE]                   (w*v, w) per pair, 0 for an excluded gate
//   F > 1       float V[TILE][STRIDE], W[TILE]     the pair's raw field slots (sentinel = excluded) and its weight; for 3
//                                                  fields the weight rides in the padding slot: one 16-byte entry.  A
//                                                  pair's entry is one 8 / 16 / 32-byte vector access at a lane stride
//                                                  of the same size (conflict-free); the row-phase lanes do the mask
//                                                  test and the product
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

// LDS floats per pair: (w*v, w) for one field; the raw field slots + the weight for several (the weight rides in the
// padding slot of a 3-field entry, otherwise in its own array behind the values)
constexpr int tile_floats(int nf, int stride) { return nf == 1 ? 2 : nf == 3 ? 4 : stride + 1; }

// ---- product side: one pair -> its LDS entry --------------------------------------------------------------------
// One field: the masked products (w*v, w).  Several fields: the pair's RAW packed slots (EXCLUDED sentinel = masked)
// and its weight -- 12 / 16 / 20 / 36 bytes instead of the 16 / 32 / 32 / 64 of F (w*v, w) pairs.  LDS stores are the
// slow direction of this part (about 70 B/clk/CU against 256 for loads), so the mask test and the product are done by
// the row-phase lanes, which are fully occupied anyway, and the tile carries the fewest bytes that define them.
template <int NF, int STRIDE>
__device__ __forceinline__ void store_products(float* __restrict__ tile, int tile_pairs, int e, float w,
                                               const float (&v)[STRIDE]) {
  if constexpr (NF == 1) {
    const bool ok = f32_bits(v[0]) != RG_EXCLUDED_BITS;
    f32x2 p;
    p.x = ok ? w * v[0] : 0.0f;
    p.y = ok ? w : 0.0f;
    reinterpret_cast<f32x2*>(tile)[e] = p;
  } else if constexpr (NF == 3) {
    reinterpret_cast<f32x4*>(tile)[e] = (f32x4){v[0], v[1], v[2], w};
  } else {
    float* __restrict__ vv = tile + (size_t)e * STRIDE;
    if constexpr (STRIDE == 2) {
      *reinterpret_cast<f32x2*>(vv) = (f32x2){v[0], v[1]};
    } else {
#pragma unroll
      for (int q = 0; q < STRIDE; q += 4) *reinterpret_cast<f32x4*>(vv + q) = (f32x4){v[q], v[q + 1], v[q + 2], v[q + 3]};
    }
    tile[(size_t)tile_pairs * STRIDE + e] = w;
  }
}

// ---- row side ---------------------------------------------------------------------------------------------------
// rs_o / re_o: this lane's row = pairs [rs_o, re_o) of the segment (lane == row); t = first pair of the tile.
// rowacc: f32x2[64 * NF], entry (row, f) = running (sum w*v, sum w).
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
      float p0[NF], w0[NF], p1[NF], w1[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) p0[f] = w0[f] = p1[f] = w1[f] = 0.0f;
      auto add = [&](float (&ap)[NF], float (&aw)[NF], int j) {
        float ev[STRIDE], w;
        if constexpr (NF == 3) {
          const f32x4 x = reinterpret_cast<const f32x4*>(tile)[j];
          ev[0] = x.x; ev[1] = x.y; ev[2] = x.z; w = x.w;
        } else {
          const float* __restrict__ vv = tile + (size_t)j * STRIDE;
          if constexpr (STRIDE == 2) {
            const f32x2 x = *reinterpret_cast<const f32x2*>(vv);
            ev[0] = x.x; ev[1] = x.y;
          } else {
#pragma unroll
            for (int q = 0; q < STRIDE; q += 4) {
              if (q < NF) {   // whole vectors of padding slots are not even read
                const f32x4 x = *reinterpret_cast<const f32x4*>(vv + q);
                ev[q] = x.x; ev[q + 1] = x.y; ev[q + 2] = x.z; ev[q + 3] = x.w;
              }
            }
          }
          w = tile[(size_t)TILE * STRIDE + j];
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {   // masked gate: contributes to neither sum (interpolate.py:78-79)
          const bool ok = f32_bits(ev[f]) != RG_EXCLUDED_BITS;
          ap[f] += ok ? w * ev[f] : 0.0f;
          aw[f] += ok ? w : 0.0f;
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
        for (int f = 0; f < NF; ++f) rowacc[myrow * NF + f] += (f32x2){sv[2 * f], sv[2 * f + 1]};
      }
    }
  }
}

}  // namespace rg
