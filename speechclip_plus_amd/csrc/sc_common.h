// Shared device helpers for the gfx950 kernels (wave64, MFMA bf16, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "speechclip_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SC_WAVE 64

__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// erf-GELU (fairseq nn.GELU() / torch "gelu"):  x * Phi(x),  Phi(x) = 0.5 (1 + erf(x / sqrt 2)).
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off level): one v_rcp, one v_exp and a
// degree-5 Horner chain instead of libm erff's ~40 instructions - the GELU epilogue of FC1 / the conv stack and the
// conv-0 kernel are VALU-bound on it.  The lower tail is formed without cancellation (Phi(x<0) = 0.5 poly e).
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)   // identical rounding in every kernel that inlines it
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float half_tail = 0.5f * poly * e;                 // = Phi(-|x|)
    const float phi = x >= 0.f ? 1.0f - half_tail : half_tail;
    return x * phi;
}

// two elements at a time: the polynomial and the scalings become packed fp32 ops (v_pk_fma_f32 / v_pk_mul_f32, two
// lanes' worth of work per issue slot); rcp / exp2 stay scalar (transcendental unit).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
#pragma clang fp contract(off)
    f32x2 ax;
    ax.x = fabsf(x.x); ax.y = fabsf(x.y);
    const f32x2 z = ax * 0.70710678118654752f;
    const f32x2 den = __builtin_elementwise_fma(z, f32x2{0.3275911f, 0.3275911f}, f32x2{1.0f, 1.0f});
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
    f32x2 poly = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, f32x2{1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, f32x2{-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, f32x2{0.254829592f, 0.254829592f});
    poly = poly * t;
    const f32x2 ex = z * z * -1.4426950408889634f;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(ex.x); e.y = __builtin_amdgcn_exp2f(ex.y);
    const f32x2 half_tail = poly * e * 0.5f;
    f32x2 phi;
    phi.x = x.x >= 0.f ? 1.0f - half_tail.x : half_tail.x;
    phi.y = x.y >= 0.f ? 1.0f - half_tail.y : half_tail.y;
    return x * phi;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// host side
void sc_set_error(const char* fmt, ...);
int sc_num_cus();   // compute units of the current device (immutable cache; 256 on MI355X)
#define SC_CHECK(cond, ...)                 \
    do {                                    \
        if (!(cond)) {                      \
            sc_set_error(__VA_ARGS__);      \
            return -1;                      \
        }                                   \
    } while (0)
#define SC_LAUNCH_CHECK()                                                                        \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) {                                                                  \
            sc_set_error("%s:%d kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return -2;                                                                           \
        }                                                                                        \
    } while (0)
