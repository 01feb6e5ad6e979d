// Library plumbing (version, thread-local error string, device count) and the small gate-side kernels:
//   a1 antenna -> Cartesian, a3 GateFilter predicates, a2/a8-prologue mask folding ("pack fields").
#include <stdarg.h>
#include <string.h>

#include "rg_common.hpp"

namespace rg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rg

extern "C" int rg_version(void) { return RG_VERSION; }
extern "C" const char* rg_last_error(void) { return rg::g_err; }
extern "C" int rg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    rg::set_error("no HIP device visible");
    return RG_ENODEVICE;
  }
  return n;
}

namespace {

// ---------------------------------------------------------------------------------------------------
// a1: 4/3-earth antenna -> Cartesian.  One thread per gate, float64 math, float32 coalesced stores.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(rg::kBlock) void antenna_kernel(const double* __restrict__ ranges, int n_gates,
                                                             const double* __restrict__ az_deg,
                                                             const double* __restrict__ el_deg, long n_total,
                                                             float* __restrict__ x, float* __restrict__ y,
                                                             float* __restrict__ z) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  const long ray = i / n_gates;
  const int k = (int)(i - ray * n_gates);
  constexpr double kDeg = 3.14159265358979323846 / 180.0;
  constexpr double R = 6371000.0 * (4.0 / 3.0);
  const double r = ranges[k];
  const double az = az_deg[ray] * kDeg, el = el_deg[ray] * kDeg;
  double se, ce, sa, ca;
  sincos(el, &se, &ce);
  sincos(az, &sa, &ca);
  const double zz = sqrt(r * r + R * R + 2.0 * r * R * se) - R;
  const double s = R * asin(r * ce / (R + zz));
  x[i] = (float)(s * sa);
  y[i] = (float)(s * ca);
  z[i] = (float)zz;
}

// ---------------------------------------------------------------------------------------------------
// a3: GateFilter predicates OR-ed into a uint8 mask.  4 gates per thread (float4 in, uchar4 in/out).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool gate_pred(int op, float d, float a, float b) {
  switch (op) {
    case RG_GATE_BELOW: return d < a;
    case RG_GATE_ABOVE: return d > a;
    case RG_GATE_BETWEEN: return d > a && d < b;
    case RG_GATE_OUTSIDE: return d < a || d > b;
    case RG_GATE_EQUAL: return fabsf(d - a) < b;
    default: return isnan(d) || isinf(d);
  }
}

__global__ __launch_bounds__(rg::kBlock) void gate_mask_kernel(const float* __restrict__ data, long n, int op,
                                                               float a, float b, uint8_t* __restrict__ mask) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 4 <= n) {
    const float4 d = *reinterpret_cast<const float4*>(data + i4);
    uchar4 m = *reinterpret_cast<uchar4*>(mask + i4);
    m.x |= gate_pred(op, d.x, a, b);
    m.y |= gate_pred(op, d.y, a, b);
    m.z |= gate_pred(op, d.z, a, b);
    m.w |= gate_pred(op, d.w, a, b);
    *reinterpret_cast<uchar4*>(mask + i4) = m;
  } else {
    for (long i = i4; i < n; ++i) mask[i] |= (uint8_t)gate_pred(op, data[i], a, b);
  }
}

// ---------------------------------------------------------------------------------------------------
// a2 / a8 prologue: fold masks into the values and interleave the fields gate-major.
// ---------------------------------------------------------------------------------------------------
struct PackArgs {
  const float* field[RG_MAX_FIELDS];
  const uint8_t* mask[RG_MAX_FIELDS];
  const uint8_t* shared;
};

template <int STRIDE>
__global__ __launch_bounds__(rg::kBlock) void pack_kernel(PackArgs a, int n_fields, long n, float* __restrict__ out) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const bool sh = a.shared != nullptr && a.shared[g] != 0;
  float v[STRIDE];
#pragma unroll
  for (int s = 0; s < STRIDE; ++s) {
    float val = rg::bits_f32(RG_EXCLUDED_BITS);
    if (s < n_fields) {
      const bool ex = sh || (a.mask[s] != nullptr && a.mask[s][g] != 0);
      if (!ex) {
        val = a.field[s][g];
        // an unmasked data NaN must keep propagating, never alias the sentinel
        if (rg::f32_bits(val) == RG_EXCLUDED_BITS) val = rg::bits_f32(0x7FC00000u);
      }
    }
    v[s] = val;
  }
  float* o = out + g * STRIDE;
  if constexpr (STRIDE == 1) {
    o[0] = v[0];
  } else if constexpr (STRIDE == 2) {
    *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int s = 0; s < STRIDE; s += 4) *reinterpret_cast<float4*>(o + s) = make_float4(v[s], v[s + 1], v[s + 2], v[s + 3]);
  }
}

inline unsigned blocks_for(long n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

}  // namespace

extern "C" int rg_antenna_to_cartesian_f32(const double* ranges_m, int32_t n_gates, const double* az_deg,
                                           const double* el_deg, int32_t n_rays, float* x, float* y, float* z,
                                           rg_stream_t stream) {
  RG_REQUIRE(ranges_m && az_deg && el_deg && x && y && z, RG_EINVAL, "rg_antenna_to_cartesian_f32: null pointer");
  RG_REQUIRE(n_gates >= 0 && n_rays >= 0, RG_EINVAL, "rg_antenna_to_cartesian_f32: negative size");
  const long n = (long)n_gates * n_rays;
  if (n == 0) return RG_OK;
  hipLaunchKernelGGL(antenna_kernel, dim3(blocks_for(n, rg::kBlock)), dim3(rg::kBlock), 0, (hipStream_t)stream,
                     ranges_m, n_gates, az_deg, el_deg, n, x, y, z);
  return rg::check_launch("rg_antenna_to_cartesian_f32");
}

extern "C" int rg_gate_mask_f32(const float* data, int64_t n_gates, int32_t op, float a, float b,
                                uint8_t* mask_inout, rg_stream_t stream) {
  RG_REQUIRE(data && mask_inout, RG_EINVAL, "rg_gate_mask_f32: null pointer");
  RG_REQUIRE(n_gates >= 0, RG_EINVAL, "rg_gate_mask_f32: negative size");
  RG_REQUIRE(op >= RG_GATE_BELOW && op <= RG_GATE_INVALID, RG_EINVAL, "rg_gate_mask_f32: unknown op %d", op);
  RG_REQUIRE(rg::aligned16(data) && (reinterpret_cast<uintptr_t>(mask_inout) & 3u) == 0, RG_EALIGN,
             "rg_gate_mask_f32: data must be 16-byte and mask 4-byte aligned");
  if (n_gates == 0) return RG_OK;
  hipLaunchKernelGGL(gate_mask_kernel, dim3(blocks_for((n_gates + 3) / 4, rg::kBlock)), dim3(rg::kBlock), 0,
                     (hipStream_t)stream, data, (long)n_gates, op, a, b, mask_inout);
  return rg::check_launch("rg_gate_mask_f32");
}

extern "C" int rg_pack_fields_f32(int32_t n_fields, const float* const* fields_host,
                                  const uint8_t* const* masks_host, const uint8_t* shared_mask, int64_t n_gates,
                                  int32_t stride, float* packed, rg_stream_t stream) {
  RG_REQUIRE(fields_host && (packed || n_gates == 0), RG_EINVAL, "rg_pack_fields_f32: null pointer");
  RG_REQUIRE(n_fields >= 1 && n_fields <= RG_MAX_FIELDS, RG_EUNSUPPORTED, "rg_pack_fields_f32: n_fields=%d not in 1..%d",
             n_fields, RG_MAX_FIELDS);
  RG_REQUIRE((stride == 1 || stride == 2 || stride == 4 || stride == 8) && stride >= n_fields, RG_EINVAL,
             "rg_pack_fields_f32: stride=%d must be 1,2,4,8 and >= n_fields=%d", stride, n_fields);
  RG_REQUIRE(n_gates >= 0, RG_EINVAL, "rg_pack_fields_f32: negative size");
  RG_REQUIRE(rg::aligned16(packed), RG_EALIGN, "rg_pack_fields_f32: packed must be 16-byte aligned");
  if (n_gates == 0) return RG_OK;  // nothing to pack (zero-size buffers may legitimately be null)
  PackArgs a;
  memset(&a, 0, sizeof(a));
  for (int f = 0; f < n_fields; ++f) {
    RG_REQUIRE(fields_host[f] != nullptr, RG_EINVAL, "rg_pack_fields_f32: field %d is null", f);
    a.field[f] = fields_host[f];
    a.mask[f] = masks_host ? masks_host[f] : nullptr;
  }
  a.shared = shared_mask;
  const dim3 grid(blocks_for(n_gates, rg::kBlock)), block(rg::kBlock);
  hipStream_t s = (hipStream_t)stream;
  switch (stride) {
    case 1: hipLaunchKernelGGL(pack_kernel<1>, grid, block, 0, s, a, n_fields, (long)n_gates, packed); break;
    case 2: hipLaunchKernelGGL(pack_kernel<2>, grid, block, 0, s, a, n_fields, (long)n_gates, packed); break;
    case 4: hipLaunchKernelGGL(pack_kernel<4>, grid, block, 0, s, a, n_fields, (long)n_gates, packed); break;
    default: hipLaunchKernelGGL(pack_kernel<8>, grid, block, 0, s, a, n_fields, (long)n_gates, packed); break;
  }
  return rg::check_launch("rg_pack_fields_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// Measurement aid (bench.py's roofline.ceiling_measured): what read bandwidth does this GPU deliver to a kernel that
// only streams a large array?  One dwordx4 per lane, workgroup b reads bytes [4096 b, 4096 b + 4096): the dispatcher
// hands the array out front to back in 4 KiB pieces.  No loop, no stores.  (Round 3: this shape reads 6.7-6.8 TB/s where
// the grid-stride loop of rounds 1-2 -- 8192 workgroups, four loads in flight per lane -- read 6.1-6.4 on the same
// boxes; tools/exp_placement3.py.  A ceiling should be the best known pattern.)
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void stream_read_kernel(const float4* __restrict__ a, long n16,
                                                          float* __restrict__ sink) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.0f;
  if (i < n16) {
    const float4 v = a[i];
    acc = v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;   // practically never true: keeps the loads alive without a store stream
}
}  // namespace

extern "C" int rg_stream_read_probe(const void* buffer, int64_t bytes, float* sink, rg_stream_t stream) {
  RG_REQUIRE(buffer && sink && bytes >= 16, RG_EINVAL, "rg_stream_read_probe: null buffer/sink or fewer than 16 bytes");
  RG_REQUIRE(rg::aligned16(buffer), RG_EALIGN, "rg_stream_read_probe: buffer must be 16-byte aligned");
  // a dispatch holds at most 2^32 - 1 work-items: buffers beyond 32 GiB take several launches back to back
  const long n16 = bytes / 16;
  const long kPerLaunch = 1L << 31;                      // 16-byte elements (= work-items) per launch
  for (long first = 0; first < n16; first += kPerLaunch) {
    const long n = n16 - first < kPerLaunch ? n16 - first : kPerLaunch;
    hipLaunchKernelGGL(stream_read_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const float4*>(buffer) + first, n, sink);
  }
  return rg::check_launch("rg_stream_read_probe");
}
