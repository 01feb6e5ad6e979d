// K1c  csr_apply over a COMPACT device copy of the CSR (same results as rg_csr_apply_f32, bit for bit).
//
// The reference's CSR (radar_grid/geometry.py:46-52) stores a 32-bit gate index per pair, and K1 pays for it twice:
// 4 of the 8 streamed bytes per pair, and one global gather per pair, which the texture-address path serves at about
// 22 cycles per 64-lane instruction wherever the gates lie.  Neighbouring voxels share almost all
// their gates, so this copy groups RG_COMPACT_ROWS = 256 consecutive rows into a chunk, lists the chunk's DISTINCT
// gates once (`dict`, ascending gate index) and stores a 16-bit position in that list per pair:
//
//   bytes per pair 8 -> 6 (+ 4 bytes per distinct gate per chunk, ~0.7 bytes per pair on the bench geometry);
//   one workgroup = one chunk: it gathers the chunk's ~1000 field values into an LDS window once (coalesced reads of
//   `dict`, one gather per DISTINCT gate) and every pair then reads its value from LDS.  The window holds
//   `window_cap` values (chosen per geometry to cover all but a handful of chunks -- those next to the radar, where
//   every ray converges); a chunk with more distinct gates gathers per pair through its dictionary instead.
//
// Pair order, weights and the float32 arithmetic are those of rg_csr_apply_f32 (same tiles, same products, same
// dynamic row phase), so the two kernels agree exactly; the compact copy is derived from the standard CSR on the
// device (gridding.CompactCsr) and the standard arrays stay the interchange format.
//
// Roofline: HBM.  Bytes per launch = 6*P + 4*D + sizeof(indptr)*(V+1) + 8*(C+1) + F*(5*G + 4*V)  with D = total
// dictionary entries, C = chunks.
#include <type_traits>

#include "rg_common.hpp"

namespace {

using f32x2 = float __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int kRsrcRaw32 = 0x00020000;   // gfx9 buffer resource word 3: DATA_FORMAT = 32, untyped access

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, long bytes) {   // `base` and `bytes` wave-uniform
  const unsigned nb = bytes >= 0xFFFFFFFFL ? 0xFFFFFFFFu : bytes <= 0 ? 0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)nb, kRsrcRaw32);
}

constexpr int kWaves = RG_COMPACT_ROWS / 64;   // 4 wavefronts of 64 rows per chunk

template <typename IndT, int TILE>
__global__ __launch_bounds__(64 * kWaves) void csr_compact_kernel(
    const IndT* __restrict__ indptr, const uint16_t* __restrict__ lidx, const float* __restrict__ wts,
    const int64_t* __restrict__ dict_ptr, const int32_t* __restrict__ dict, long n_vox,
    const float* __restrict__ packed, unsigned last_gate, float fill, int window_cap, float* __restrict__ out) {
  static_assert(TILE % 64 == 0, "a wave handles 64 pairs per step");
  constexpr int IT = TILE / 64;
  extern __shared__ float window[];                  // field values of the chunk's distinct gates (window_cap entries)
  __shared__ f32x2 tile_all[kWaves][TILE];
  __shared__ f32x2 rowacc_all[kWaves][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x2* tile = tile_all[wv];
  f32x2* rowacc = rowacc_all[wv];

  const long chunk = blockIdx.x;
  const long d0 = dict_ptr[chunk];
  const int nd = (int)(dict_ptr[chunk + 1] - d0);    // <= 65536: positions are 16 bits
  const bool windowed = nd <= window_cap;            // workgroup-uniform; the rare wider chunk gathers per pair
  const int nd_last = nd > 0 ? nd - 1 : 0;
  const int32_t* __restrict__ cdict = dict + d0;

  const long r0 = chunk * RG_COMPACT_ROWS + (long)wv * 64;
  const bool alive = r0 < n_vox;                     // wave-uniform; a dead wave still helps to fill the window
  const long row = r0 + lane;
  const long seg_b = alive ? (long)indptr[r0] : 0;
  const long seg_e = alive ? (long)indptr[r0 + 64 < n_vox ? r0 + 64 : n_vox] : 0;
  const int span = (int)(seg_e - seg_b);
  const int rs_o = alive ? (int)((long)indptr[row < n_vox ? row : n_vox] - seg_b) : 0;
  const int re_o = alive ? (int)((long)indptr[row + 1 < n_vox ? row + 1 : n_vox] - seg_b) : 0;
  rowacc[lane] = (f32x2)(0.0f);

  // Two register stages, loop unrolled by two, every load unconditional and range-checked against the chunk's last
  // pair -- the same exact-wait-count pipeline as rg_csr_apply_f32, without a gather stage.
  struct Stage {
    int ci[IT];
    float cw[IT];
  };
  Stage st[2];
  const uint16_t* __restrict__ li = lidx + seg_b;
  const float* __restrict__ wi = wts + seg_b;
  const int lane2 = lane * 2, lane4 = lane * 4;
  auto stream = [&](Stage& sg, int t) {   // t wave-uniform: the resources live in SGPRs
    const rsrc_t ri = make_rsrc(li + t, ((long)span - t) * 2);
    const rsrc_t rw = make_rsrc(wi + t, ((long)span - t) * 4);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      sg.ci[it] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(ri, lane2 + it * 128, 0, 0);
      sg.cw[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, lane4 + it * 256, 0, 0));
    }
  };
  // the first two tiles are requested BEFORE the window is filled: the two latencies overlap
  stream(st[0], 0);
  stream(st[1], TILE);

  // ---- the chunk's field window: one gather per DISTINCT gate --------------------------------------------
  if (windowed) {
    for (int i = threadIdx.x; i < nd; i += 64 * kWaves) {
      const unsigned g = (unsigned)dict[d0 + i];
      window[i] = packed[g < last_gate ? g : last_gate];   // clamp: never fault
    }
  }
  __syncthreads();

  if (span > 0) {
    auto step = [&](int t, Stage& cur) {
      // ---- products of tile t -> LDS (values come from the window) -----------------------------------
      float val[IT];
      if (windowed) {
#pragma unroll
        for (int it = 0; it < IT; ++it) val[it] = window[cur.ci[it] < nd_last ? cur.ci[it] : nd_last];
      } else {   // chunk with more distinct gates than the window holds: position -> gate -> value, from memory
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const unsigned g = (unsigned)cdict[cur.ci[it] < nd_last ? cur.ci[it] : nd_last];
          val[it] = packed[g < last_gate ? g : last_gate];
        }
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const bool ok = rg::f32_bits(val[it]) != RG_EXCLUDED_BITS;
        f32x2 e;
        e.x = ok ? cur.cw[it] * val[it] : 0.0f;
        e.y = ok ? cur.cw[it] : 0.0f;
        tile[it * 64 + lane] = e;
      }
      stream(cur, t + 2 * TILE);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- dynamic row phase (identical to rg_csr_apply_f32 with one field slot) ---------------------------
      const unsigned long long act = __ballot(re_o > rs_o && re_o > t && rs_o < t + TILE);
      if (act != 0) {  // wave-uniform
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
          f32x2 part0 = (f32x2)(0.0f), part1 = (f32x2)(0.0f);
          int j = a + sub;
          for (; j + nsub < b; j += 2 * nsub) {  // two elements per trip, two independent partial sums
            part0 += tile[j];
            part1 += tile[j + nsub];
          }
          if (j < b) part0 += tile[j];
          f32x2 sum = part0 + part1;
          for (int m = 1; m < (1 << lg); m <<= 1) {
            sum.x += __shfl_xor(sum.x, m, 64);
            sum.y += __shfl_xor(sum.y, m, 64);
          }
          if (live && sub == 0) rowacc[myrow] += sum;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    for (int t = 0; t < span;) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        step(t, st[u]);
        t += TILE;
        if (t >= span) break;
      }
    }
  }

  if (row < n_vox) {
    const f32x2 s = rowacc[lane];
    out[row] = s.y > 0.0f ? (float)((double)s.x / (double)s.y) : fill;
  }
}

template <typename IndT>
int launch(int tile, int window_cap, const void* indptr, const uint16_t* lidx, const float* wts, const int64_t* dict_ptr,
           const int32_t* dict, long n_vox, const float* packed, long n_gates, float fill, float* out, hipStream_t s) {
  const long chunks = (n_vox + RG_COMPACT_ROWS - 1) / RG_COMPACT_ROWS;
#define RG_K1C(TILE_)                                                                                                 \
  hipLaunchKernelGGL((csr_compact_kernel<IndT, TILE_>), dim3((unsigned)chunks), dim3(64 * kWaves),                    \
                     (size_t)window_cap * sizeof(float), s, static_cast<const IndT*>(indptr), lidx, wts, dict_ptr, dict, \
                     n_vox, packed, (unsigned)(n_gates - 1), fill, window_cap, out)
  switch (tile) {   // the default must be the tile of rg_csr_apply_f32's single-field kernel: same partial sums
    case 256: RG_K1C(256); break;
    case 512: RG_K1C(512); break;
    default: RG_K1C(384); break;
  }
#undef RG_K1C
  return rg::check_launch("rg_csr_compact_apply_f32");
}

}  // namespace

extern "C" int rg_csr_compact_apply_f32(const void* indptr, int32_t indptr_is_i64, const uint16_t* local_idx,
                                        const float* weights, const int64_t* dict_ptr, const int32_t* dict,
                                        int64_t n_vox, int64_t n_pairs, const float* packed, int64_t n_gates,
                                        float fill_value, float* out, int32_t window_cap, int32_t tile,
                                        rg_stream_t stream) {
  RG_REQUIRE(indptr && out && dict_ptr, RG_EINVAL, "rg_csr_compact_apply_f32: null indptr/dict_ptr/out");
  RG_REQUIRE(n_vox >= 0 && n_pairs >= 0, RG_EINVAL, "rg_csr_compact_apply_f32: negative size");
  RG_REQUIRE(n_pairs == 0 || (local_idx && weights && dict && packed && n_gates > 0), RG_EINVAL,
             "rg_csr_compact_apply_f32: pairs present but local_idx/weights/dict/packed/n_gates missing");
  RG_REQUIRE(n_gates <= 0x7FFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_f32: n_gates exceeds int32 gate indices");
  RG_REQUIRE(n_vox <= 0x3FFFFFFFFFL, RG_EUNSUPPORTED, "rg_csr_compact_apply_f32: n_vox too large for one launch");
  RG_REQUIRE(tile == 0 || tile == 256 || tile == 384 || tile == 512, RG_EINVAL,
             "rg_csr_compact_apply_f32: tile must be 0 (default), 256, 384 or 512");
  RG_REQUIRE(window_cap >= 0 && window_cap <= RG_COMPACT_MAX_WINDOW, RG_EINVAL,
             "rg_csr_compact_apply_f32: window_cap %d outside 0..%d", window_cap, RG_COMPACT_MAX_WINDOW);
  if (n_vox == 0) return RG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    return launch<int64_t>(tile, window_cap, indptr, local_idx, weights, dict_ptr, dict, n_vox, packed, n_gates,
                           fill_value, out, s);
  return launch<int32_t>(tile, window_cap, indptr, local_idx, weights, dict_ptr, dict, n_vox, packed, n_gates, fill_value,
                         out, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Building the compact copy: the distinct gates of every chunk and each pair's position among them.
// One workgroup per chunk keeps an open-addressing hash set of gate indices in LDS.  Chunks whose dictionary would
// overload the table are processed in R = 2, 4, ... 32 rounds, round r taking the gates of one residue class of a
// second hash, so any chunk up to 65536 distinct gates is handled with 32 KiB of LDS.
//   count pass: distinct gates per chunk (and the rounds it needed)   -> rg_scan_counts_i64 gives dict_ptr
//   fill pass : same rounds; the lane that claims a slot gives the gate the next position and writes the dictionary
//               entry, a second sweep over the round's pairs looks every gate up and stores its 16-bit position.
// Positions depend on the insertion order (not reproducible run to run); the gridding result does not.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kSlots = 8192;          // hash slots per workgroup (32 KiB)
constexpr int kMaxLoad = 6144;        // distinct gates one round may insert
constexpr int kBuildThreads = 256;
constexpr int kMaxRounds = 32;

__device__ __forceinline__ unsigned slot_hash(unsigned g) { return (g * 2654435761u) >> 19; }      // 13 bits
__device__ __forceinline__ unsigned round_hash(unsigned g) { return (g * 0x85EBCA6Bu) >> 27; }     // 5 bits

// Inserts the gates of residue class `r` (of `rounds`) among pairs [p0, p1).  Returns false when the table overloads.
// With `ids`, the lane that claims a slot also gives the gate its position (base + order of arrival) and writes the
// dictionary entry: positions then follow the order in which the chunk's pairs first mention a gate, so the 64
// consecutive pairs of one gather mostly hold neighbouring positions (fewer LDS bank conflicts than any fixed order).
template <typename IndT>
__device__ bool insert_round(const int32_t* __restrict__ gidx, long p0, long p1, int rounds, int r, int* table,
                             int* s_count, int* s_overflow, unsigned short* ids = nullptr, int base = 0,
                             int32_t* __restrict__ dict_out = nullptr) {
  for (int i = threadIdx.x; i < kSlots; i += kBuildThreads) table[i] = -1;
  if (threadIdx.x == 0) { *s_count = 0; *s_overflow = 0; }
  __syncthreads();
  for (long p = p0 + threadIdx.x; p < p1; p += kBuildThreads) {
    const int g = gidx[p];
    if (rounds > 1 && (int)(round_hash((unsigned)g) & (unsigned)(rounds - 1)) != r) continue;
    unsigned h = slot_hash((unsigned)g);
    while (true) {
      const int seen = *(volatile int*)&table[h];   // other lanes insert concurrently
      if (seen == g) break;
      if (seen == -1) {
        if (*(volatile int*)s_overflow) break;
        const int old = atomicCAS(&table[h], -1, g);
        if (old == -1) {
          const int order = atomicAdd(s_count, 1);
          if (order >= kMaxLoad) *(volatile int*)s_overflow = 1;
          if (ids) {
            ids[h] = (unsigned short)(base + order);
            dict_out[base + order] = g;
          }
          break;
        }
        if (old == g) break;
      }
      h = (h + 1) & (kSlots - 1);
    }
  }
  __syncthreads();
  return *s_overflow == 0;
}

template <typename IndT>
__global__ __launch_bounds__(kBuildThreads) void compact_count_kernel(const IndT* __restrict__ indptr,
                                                                      const int32_t* __restrict__ gidx, long n_rows,
                                                                      int32_t* __restrict__ chunk_counts,
                                                                      uint8_t* __restrict__ chunk_rounds) {
  __shared__ int table[kSlots];
  __shared__ int s_count, s_overflow;
  const long chunk = blockIdx.x;
  const long r0 = chunk * RG_COMPACT_ROWS;
  const long r1 = r0 + RG_COMPACT_ROWS < n_rows ? r0 + RG_COMPACT_ROWS : n_rows;
  const long p0 = (long)indptr[r0], p1 = (long)indptr[r1];
  int rounds = 1, total = 0;
  while (true) {
    total = 0;
    bool ok = true;
    for (int r = 0; r < rounds && ok; ++r) {
      ok = insert_round<IndT>(gidx, p0, p1, rounds, r, table, &s_count, &s_overflow);
      total += s_count;
      __syncthreads();
    }
    if (ok) break;
    rounds *= 2;
    if (rounds > kMaxRounds) { total = 65537; rounds = kMaxRounds; break; }   // not compactable
  }
  if (threadIdx.x == 0) {
    chunk_counts[chunk] = total;
    chunk_rounds[chunk] = (uint8_t)rounds;
  }
}

template <typename IndT>
__global__ __launch_bounds__(kBuildThreads) void compact_fill_kernel(const IndT* __restrict__ indptr,
                                                                     const int32_t* __restrict__ gidx, long n_rows,
                                                                     const int64_t* __restrict__ dict_ptr,
                                                                     const uint8_t* __restrict__ chunk_rounds,
                                                                     int32_t* __restrict__ dict,
                                                                     uint16_t* __restrict__ local_idx) {
  __shared__ int table[kSlots];
  __shared__ unsigned short ids[kSlots];
  __shared__ int s_count, s_overflow;
  const long chunk = blockIdx.x;
  const long r0 = chunk * RG_COMPACT_ROWS;
  const long r1 = r0 + RG_COMPACT_ROWS < n_rows ? r0 + RG_COMPACT_ROWS : n_rows;
  const long p0 = (long)indptr[r0], p1 = (long)indptr[r1];
  const long d0 = dict_ptr[chunk];
  const int rounds = chunk_rounds[chunk];
  int base = 0;
  for (int r = 0; r < rounds; ++r) {
    // cannot overload: the count pass sized the rounds
    insert_round<IndT>(gidx, p0, p1, rounds, r, table, &s_count, &s_overflow, ids, base, dict + d0);
    for (long p = p0 + threadIdx.x; p < p1; p += kBuildThreads) {
      const int g = gidx[p];
      if (rounds > 1 && (int)(round_hash((unsigned)g) & (unsigned)(rounds - 1)) != r) continue;
      unsigned h = slot_hash((unsigned)g);
      while (table[h] != g) h = (h + 1) & (kSlots - 1);
      local_idx[p] = ids[h];
    }
    base += s_count;
    __syncthreads();
  }
}

}  // namespace

extern "C" int rg_csr_compact_count(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, int64_t n_rows,
                                    int32_t* chunk_counts, uint8_t* chunk_rounds, rg_stream_t stream) {
  RG_REQUIRE(n_rows >= 0, RG_EINVAL, "rg_csr_compact_count: negative size");
  if (n_rows == 0) return RG_OK;
  RG_REQUIRE(indptr && chunk_counts && chunk_rounds, RG_EINVAL, "rg_csr_compact_count: null pointer");
  const long chunks = (n_rows + RG_COMPACT_ROWS - 1) / RG_COMPACT_ROWS;
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    hipLaunchKernelGGL(compact_count_kernel<int64_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int64_t*>(indptr), gate_idx, (long)n_rows, chunk_counts, chunk_rounds);
  else
    hipLaunchKernelGGL(compact_count_kernel<int32_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int32_t*>(indptr), gate_idx, (long)n_rows, chunk_counts, chunk_rounds);
  return rg::check_launch("rg_csr_compact_count");
}

extern "C" int rg_csr_compact_fill(const void* indptr, int32_t indptr_is_i64, const int32_t* gate_idx, int64_t n_rows,
                                   const int64_t* dict_ptr, const uint8_t* chunk_rounds, int32_t* dict,
                                   uint16_t* local_idx, rg_stream_t stream) {
  RG_REQUIRE(n_rows >= 0, RG_EINVAL, "rg_csr_compact_fill: negative size");
  if (n_rows == 0) return RG_OK;
  RG_REQUIRE(indptr && dict_ptr && chunk_rounds, RG_EINVAL, "rg_csr_compact_fill: null pointer");
  const long chunks = (n_rows + RG_COMPACT_ROWS - 1) / RG_COMPACT_ROWS;
  hipStream_t s = (hipStream_t)stream;
  if (indptr_is_i64)
    hipLaunchKernelGGL(compact_fill_kernel<int64_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int64_t*>(indptr), gate_idx, (long)n_rows, dict_ptr, chunk_rounds, dict, local_idx);
  else
    hipLaunchKernelGGL(compact_fill_kernel<int32_t>, dim3((unsigned)chunks), dim3(kBuildThreads), 0, s,
                       static_cast<const int32_t*>(indptr), gate_idx, (long)n_rows, dict_ptr, chunk_rounds, dict, local_idx);
  return rg::check_launch("rg_csr_compact_fill");
}
