// Shared device/host helpers for the gfx950 kernels of libaurora_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/aurora_hip.h"

namespace aurora {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define AURORA_CHECK_ARG(cond, ...)          \
  do {                                       \
    if (!(cond)) {                           \
      ::aurora::set_error(__VA_ARGS__);      \
      return AURORA_E_ARG;                   \
    }                                        \
  } while (0)

int check_launch(const char* what);

// Compute units of the current device (cached per device ordinal).
int device_cus();
// Index of the current device, for per-device one-time setup (function attributes).
int current_device();

// ---- bf16 <-> fp32 (raw uint16 storage, round-to-nearest-even) ---------------------------------
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): two values per instruction.
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_hw{lo, hi}, bf16x2_hw));
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct elem;
template <> struct elem<float> {
  static constexpr int dtype = AURORA_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
  static constexpr int dtype = AURORA_BF16;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector of raw dwords: the unit of every global <-> LDS <-> register move here.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Two-term fp16 split of a pair of fp32 values (gemm.hip, "second variant"): h = fp16(a), l = fp16(a - h), packed two to a
// register.  Shared by every kernel that produces the fp16-pair layout, so that a split is the same wherever it happens.
// (The remainder could come from one mixed-precision FMA per value -- v_fma_mixlo/hi_f16: widen, subtract and round in
// one instruction, 3 operations per pair instead of 5.  Measured: no gain in the in-phase kernel, -15 % in the ping-pong
// one; the mix instructions do not issue at the packed conversions' rate.)
__device__ __forceinline__ void split_pair_f16(float a0, float a1, uint32_t& h, uint32_t& l) {
  const f32x2_t v = {a0, a1};
  const f16x2_t hh = __builtin_convertvector(v, f16x2_t);             // v_cvt_pk_f16_f32 (round to nearest even)
  const f32x2_t r = v - __builtin_convertvector(hh, f32x2_t);         // exact
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
}

// Load `n` (4 or 8) consecutive elements as floats.
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  u32x4 a = *reinterpret_cast<const u32x4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                       pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  u32x2 a = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

// ---- wavefront (64 lanes) reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Exact-erf GELU to ~4e-7 absolute (Abramowitz & Stegun 7.1.26, |erf error| <= 1.5e-7) in ~17 VALU
// ops instead of erff's ~60: Phi(x) = 1 - q (x >= 0) or q (x < 0) with q = poly(t) * exp(-x^2/2) / 2,
// t = 1/(1 + p|x|/sqrt2) -- written without the 1 - erf cancellation.  Used for the fp32 results of the
// operand-splitting GEMMs (packed form below); the native-fp32 GEMM keeps erff, bf16 results use gelu_sig.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float q = 0.5f * poly * t * __expf(-z * z);
  return x * (x >= 0.f ? 1.0f - q : q);
}
// GELU through the logistic form  x * sigma(2 q(x)),  q an odd quintic fitted (minimax over |x| <= 9, clamped beyond)
// to logit(Phi(x)) / 2: |error| <= 2.6e-5 absolute in 9 VALU issue slots (two of them transcendental) against 17 for
// the erf form above -- a hundredth of a bf16 ulp at |y| = 1, used for bf16 results only (the fc1 epilogue of every
// block runs this on 2 x 10^10 values per step).  Coefficients carry the factor -2 log2(e) of the exp2 argument.
__device__ __forceinline__ float gelu_sig(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -9.0f, 9.0f);
  const float x2 = xc * xc;
  float t = fmaf(x2, 1.0142631e-3f, -1.0677572e-1f);     // -2 log2(e) * (a5, a3)
  t = fmaf(t, x2, -2.3011213f);                           // -2 log2(e) * a1
  const float e = __builtin_amdgcn_exp2f(xc * t);          // exp(-2 q(x))
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// Two values at a time: the polynomial part runs on the packed fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32:
// two lanes' worth of fp32 per issue slot).  A VALU instruction costs a wave ~4 cycles of its SIMD's issue, a 256 x 256
// tile has 256 values per lane, and nothing else runs on the CU during a GEMM epilogue -- the activation is 8 us of a
// 22 us stage-0 fc1 tile in its scalar form.
__device__ __forceinline__ f32x2_hw gelu_sig2(f32x2_hw x) {
  const f32x2_hw xc = {__builtin_amdgcn_fmed3f(x.x, -9.0f, 9.0f), __builtin_amdgcn_fmed3f(x.y, -9.0f, 9.0f)};
  const f32x2_hw x2 = xc * xc;
  f32x2_hw t = __builtin_elementwise_fma(x2, f32x2_hw{1.0142631e-3f, 1.0142631e-3f}, f32x2_hw{-1.0677572e-1f, -1.0677572e-1f});
  t = __builtin_elementwise_fma(t, x2, f32x2_hw{-2.3011213f, -2.3011213f});
  const f32x2_hw u = xc * t;
  const f32x2_hw d = f32x2_hw{__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)} + f32x2_hw{1.0f, 1.0f};
  return x * f32x2_hw{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// Packed form of gelu_erf_fast (|error| <= 7.5e-8 |x|, i.e. a few fp32 ulps): the fp32 linears that run on the matrix
// pipe by operand splitting use it for their fp32 results -- erff costs ~60 VALU issue slots per value, and the
// decoder's fc1 alone applies GELU to 1.7 x 10^9 values per step.
__device__ __forceinline__ f32x2_hw gelu_erf_fast2(f32x2_hw x) {
  const f32x2_hw z = f32x2_hw{fabsf(x.x), fabsf(x.y)} * f32x2_hw{0.70710678118654752440f, 0.70710678118654752440f};
  const f32x2_hw den = __builtin_elementwise_fma(z, f32x2_hw{0.3275911f, 0.3275911f}, f32x2_hw{1.0f, 1.0f});
  const f32x2_hw t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  f32x2_hw poly = __builtin_elementwise_fma(t, f32x2_hw{1.061405429f, 1.061405429f}, f32x2_hw{-1.453152027f, -1.453152027f});
  poly = __builtin_elementwise_fma(poly, t, f32x2_hw{1.421413741f, 1.421413741f});
  poly = __builtin_elementwise_fma(poly, t, f32x2_hw{-0.284496736f, -0.284496736f});
  poly = __builtin_elementwise_fma(poly, t, f32x2_hw{0.254829592f, 0.254829592f});
  const f32x2_hw a = z * z * f32x2_hw{-1.4426950408889634f, -1.4426950408889634f};   // -z^2 log2(e)
  const f32x2_hw q = poly * t * f32x2_hw{0.5f, 0.5f} * f32x2_hw{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  return x * f32x2_hw{x.x >= 0.f ? 1.0f - q.x : q.x, x.y >= 0.f ? 1.0f - q.y : q.y};
}
template <typename T> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<uint16_t>(float x) { return gelu_sig(x); }   // every bf16 epilogue: same bits whatever the tile shape

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace aurora
