// Internal helpers shared by the HIP translation units of libsdeo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace sdeo {

// error convention: every C entry point returns 0 on success; the message of the last failure
// is kept per thread and returned by sdeo_last_error() (replaces the reference's CUASSERT /
// PLUGIN_FAIL macros, Engine.py:38-44, plugin/common/checkMacrosPlugin.cpp).
void set_error(const std::string& msg);
int fail(const char* fmt, ...);

#define SDEO_CHECK(cond, ...)                       \
  do {                                              \
    if (!(cond)) return ::sdeo::fail(__VA_ARGS__);  \
  } while (0)

#define SDEO_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      return ::sdeo::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// "done once" flag of a per-kernel, per-DEVICE setting (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is recorded per device): a
// process that drives a second GPU, or two host threads racing on first use, must not skip the call for a device that never
// received it.  One atomic bit per device ordinal.
struct DeviceOnce {
  unsigned long long bits = 0;
  bool need() const {               // true until mark() has been called on the current device
    int dev = 0;
    (void)hipGetDevice(&dev);
    return !((__atomic_load_n(&bits, __ATOMIC_ACQUIRE) >> (dev & 63)) & 1ull);
  }
  void mark() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    __atomic_fetch_or(&bits, 1ull << (dev & 63), __ATOMIC_RELEASE);
  }
};

// Kernel-argument warm-up.  hipcc fetches kernel arguments lazily: one s_load next to each first use, each followed by
// s_waitcnt lgkmcnt(0).  The arguments of a launch are cold (nothing has read that kernarg block before), so every FIRST touch of
// one of its 64-byte lines is a miss to memory, and a prologue that walks a 300-byte parameter block field by field pays several
// of those misses one after the other (measured: host-side kernarg placement alone moves the step from 6.07 to 7.06 ms, i.e. the
// fetch chain is on the critical path of every launch).  One dword of every line, requested back to back before anything else,
// turns the chain into ONE miss time; the lazy loads that follow hit the scalar cache.  BYTES = a LOWER bound of the size of the
// explicit arguments: nothing at or beyond BYTES is read (the kernarg block may end at the end of a mapped page).
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
  typedef __attribute__((address_space(4))) const unsigned* karg_p;
  karg_p kp = (karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int LINES = (BYTES + 63) / 64;                      // offsets 0, 64, ...: all < BYTES
  static_assert(BYTES >= 4 && (LINES - 1) * 64 + 4 <= BYTES, "kernarg_warm reads inside the explicit arguments only");
  unsigned v[LINES];
#pragma unroll
  for (int i = 0; i < LINES; ++i) v[i] = kp[i * 16];
#pragma unroll
  for (int i = 0; i < LINES; ++i) asm volatile("" ::"s"(v[i]));
}

// Sum over the 64 lanes of a wave, result in EVERY lane, without an LDS round trip per step (`__shfl_xor` compiles to ds_bpermute +
// s_waitcnt, ~130 clocks each, six in a dependent chain): quad_perm DPP for the partners at distance 1 and 2, row rotations by 4 and 8
// inside each row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows.  Fixed order, so results are reproducible;
// the order differs from the xor butterfly, so the last bits may differ from a __shfl_xor reduction of the same values.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);                       // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);                       // quad_perm [2,3,0,1]
  v += dpp_move<0x124>(v);                      // row_ror:4
  v += dpp_move<0x128>(v);                      // row_ror:8 -> every lane holds its row's sum
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));     // rows (0,1) and (2,3) exchanged
  v = a + b;
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));     // halves exchanged
  return a + b;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: far inside the fp16 output rounding
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// exact GELU (F.gelu default, `attention.py:56`)
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of the result): branch-free, ~20 issue
// slots against the ~50 of the device library's erff, which made the GEGLU epilogue VALU-bound (10.5 M evaluations per launch)
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(fmaf(-poly * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f)); }
// CLIP's "quick_gelu": x * sigmoid(1.702 x)
__device__ __forceinline__ float quick_gelu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v)); }

}  // namespace sdeo
