// Shared device helpers for the gfx950 kernels (wave64, MFMA bf16, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#include "speechclip_hip.h"

// One-time, thread-safe raise of a kernel's dynamic-LDS limit (first launch of an instantiation may come from any host thread:
// std::call_once instead of an unguarded static flag).  Returns the result of the one attempt on every call.
struct sc_lds_attr_once {
    std::once_flag flag;
    hipError_t err = hipSuccess;
};
template <class Kernel>
static inline hipError_t sc_set_max_lds_once(sc_lds_attr_once& st, Kernel kernel, int bytes) {
    std::call_once(st.flag, [&] { st.err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    return st.err;
}

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
// Round 6: the pair goes through a 2-vector conversion = ONE v_cvt_pk_bf16_f32 lo, hi.  The former (uint32_t)f2bf(lo) | f2bf(hi) << 16
// compiled to two single-input conversions + v_lshlrev + v_or_b32_sdwa - four VALU instructions per pair in every VALU-bound epilogue
// (GEMM tiles, conv layer 0, LayerNorm rows).  Same rounding (nearest-even), same bits.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// erf-GELU (fairseq nn.GELU() / torch "gelu"):  x * Phi(x),  Phi(x) = 0.5 (1 + erf(x / sqrt 2)).
// erfc by Abramowitz-Stegun 7.1.28: erfc(z) = (1 + a1 z + ... + a6 z^6)^-16, |err| <= 3e-7 (7e-7 measured in fp32 incl.
// the four squarings) - ONE transcendental (v_rcp) and packed-fp32 FMAs instead of libm erff's ~40 instructions or the
// rcp + exp of 7.1.26: the GELU epilogues of FC1 / the conv stack and the conv-0 kernel are VALU-bound on it.
// gelu(x) = 0.5 x (1 + sign(x) (1 - erfc|z|)) = (h + |h|) - |h| erfc(|z|),  h = x / 2: no select, no cancellation in the
// lower tail.  Overflow of the 16th power gives rcp(inf) = 0 = the right limit.
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)   // identical rounding in every kernel that inlines it
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752f;
    float p = fmaf(0.0000430638f, z, 0.0002765672f);
    p = fmaf(p, z, 0.0001520143f);
    p = fmaf(p, z, 0.0092705272f);
    p = fmaf(p, z, 0.0422820123f);
    p = fmaf(p, z, 0.0705230784f);
    p = fmaf(p, z, 1.0f);
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    const float r = __builtin_amdgcn_rcpf(p);                // erfc(|z|)
    const float h = x * 0.5f, ah = ax * 0.5f;
    return fmaf(-ah, r, h + ah);
}

// Phi(x) + x phi(x) with the erfc of gelu_erf (A&S 7.1.28, one v_rcp) and one v_exp
__device__ __forceinline__ float gelu_grad_as(float x) {
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752f;
    float p = fmaf(0.0000430638f, z, 0.0002765672f);
    p = fmaf(p, z, 0.0001520143f);
    p = fmaf(p, z, 0.0092705272f);
    p = fmaf(p, z, 0.0422820123f);
    p = fmaf(p, z, 0.0705230784f);
    p = fmaf(p, z, 1.0f);
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    const float he = 0.5f * __builtin_amdgcn_rcpf(p);                 // erfc(|z|) / 2
    const float cdf = x >= 0.f ? 1.f - he : he;
    return fmaf(x * 0.3989422804014327f, __expf(-0.5f * x * x), cdf);
}

#ifndef SC_GELU_TERMS
#define SC_GELU_TERMS 5          // 3 / 4: the shorter fits measured on the way (A/B builds)
#endif
#if SC_GELU_TERMS == 5
#define SC_GK0 -2.3020437690e+00f
#define SC_GK1 -1.0522998852e-01f
#define SC_GK2 3.6205193708e-04f
#define SC_GK3 8.7909655076e-05f
#define SC_GK4 -3.2104365226e-06f
#elif SC_GELU_TERMS == 4
#define SC_GK0 -2.3016456068e+00f
#define SC_GK1 -1.0598370216e-01f
#define SC_GK2 7.3575809569e-04f
#define SC_GK3 2.4884072148e-05f
#endif
// scalar twin of gelu_bf2 (below): the same operations in the same order => the same bits (the bf16-output sites' erf-GELU)
__device__ __forceinline__ float gelu_bf(float x) {
#ifdef SC_GELU_EXACT
    return gelu_erf(x);
#else
#pragma clang fp contract(off)
#if SC_GELU_TERMS == 5
#ifdef SC_GELU_CLAMP
    const float t = fminf(x * x, 36.f);
#else
    const float t = x * x;                      // round 6: no clamp, see gelu_bf2
#endif
    float p = fmaf(SC_GK4, t, SC_GK3);
    p = fmaf(p, t, SC_GK2);
    p = fmaf(p, t, SC_GK1);
    p = fmaf(p, t, SC_GK0);
#elif SC_GELU_TERMS == 4
    const float t = fminf(x * x, 36.f);
    float p = fmaf(SC_GK3, t, SC_GK2);
    p = fmaf(p, t, SC_GK1);
    p = fmaf(p, t, SC_GK0);
#else
    const float t = fminf(x * x, 64.f);
    float p = fmaf(0.0010142630198970437f, t, -0.10677572339773178f);
    p = fmaf(p, t, -2.301121234893799f);
#endif
    const float w = p * x;
    const float d = __builtin_amdgcn_exp2f(w) + 1.0f;
    return x * __builtin_amdgcn_rcpf(d);
#endif
}

// activation codes of sc_act_bf16 / sc_gemm_args.act: 1 = erf-GELU (fairseq FFN), 2 = QuickGELU (CLIP MLP)
__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == 1) return gelu_bf(u);
    const float sg = 1.f / (1.f + __expf(-1.702f * u));            // QuickGELU (CLIP): u * sigmoid(1.702 u)
    return u * sg;
}
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == 1) return gelu_grad_as(u);                            // Phi(u) + u phi(u): ONE definition for sc_act_bf16 and both GEMM
                                                                     // tile families (round 4: libm erff cost 40 instructions per element)
    const float sg = 1.f / (1.f + __expf(-1.702f * u));
    return sg * (1.f + 1.702f * u * (1.f - sg));
}

// two elements at a time: the polynomial, squarings and scalings become packed fp32 ops (v_pk_fma_f32 / v_pk_mul_f32,
// two lanes' worth of work per issue slot); only the rcp stays scalar (transcendental unit).  Same operation order as
// gelu_erf => bitwise identical results.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
#pragma clang fp contract(off)
    f32x2 ax;
    ax.x = fabsf(x.x); ax.y = fabsf(x.y);
    const f32x2 z = ax * 0.70710678118654752f;
    f32x2 p = __builtin_elementwise_fma(f32x2{0.0000430638f, 0.0000430638f}, z, f32x2{0.0002765672f, 0.0002765672f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0001520143f, 0.0001520143f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0092705272f, 0.0092705272f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0422820123f, 0.0422820123f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0705230784f, 0.0705230784f});
    p = __builtin_elementwise_fma(p, z, f32x2{1.0f, 1.0f});
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(p.x); r.y = __builtin_amdgcn_rcpf(p.y);
    const f32x2 h = x * 0.5f, ah = ax * 0.5f;
    return __builtin_elementwise_fma(-ah, r, h + ah);
}

// The erf-GELU of the bf16-OUTPUT sites (GEMM epilogues of FC1 / the conv stack, conv layer 0, pos_conv, LayerNorm + GELU rows), round 5:
//     gelu(x) ~ x * sigmoid(x p(t)),  t = min(x^2, 36),  p = c0 + c1 t + c2 t^2 + c3 t^3 + c4 t^4
//     c = (1.595655148, 7.293986985e-2, -2.509552794e-4, -6.093432956e-5, 2.225305024e-6)
// - the logistic approximation of the normal CDF carried to five odd terms, coefficients fitted (iteratively re-weighted least squares
// on logit(Phi(x)) / x, then minimax on |x Phi(x) - approx| over [-12, 12]: tools/fit_gelu.py) - in fp32: |error| <= 3.1e-6 everywhere,
// relative error <= 2.0e-5 for x >= -1, 4.8e-4 down to -3; a bf16 store rounds by up to 3.9e-3 relative, the reference's fp16 store by
// 4.9e-4.  Why: those epilogues are VALU-bound on the activation, and this form is 9 packed ops + 2 v_exp + 2 v_rcp
// per PAIR of elements against 15 packed ops + 2 v_rcp for the 3e-7-accurate A&S 7.1.28 form above, which every fp32 site keeps (the head's
// row tail, the fp32 debug mode, gelu_grad_as).  The three-term fit (|error| 2.5e-5) was 0.3 % faster still and moved a train step's loss by
// 6e-4 - a SYSTEMATIC error through 19 activation sites, unlike rounding noise; with five terms the loss equals the A&S build's to five
// digits (1.33628).  The clamp keeps the odd polynomial monotone; beyond it the exponent keeps growing linearly: exp2 -> 0 / inf,
// rcp -> 1 / 0, i.e. gelu -> x / -0 exactly.  SC_GELU_EXACT (A/B builds): the A&S form everywhere.
__device__ __forceinline__ f32x2 gelu_bf2(f32x2 x) {
#ifdef SC_GELU_EXACT
    return gelu_erf2(x);
#else
#pragma clang fp contract(off)
    f32x2 t = x * x;
#if SC_GELU_TERMS >= 4
    // Round 6: the five-term form runs WITHOUT the clamp t = min(x^2, 36) of round 5 (two v_min_f32 of the 15 VALU instructions a pair
    // costs).  The clamp was there to keep the polynomial monotone beyond the fitted range; it is monotone anyway: p(t) < 0 for all
    // t >= 0 and p'(t) <= -0.33 for t >= 36 (the quartic term dominates), so beyond |x| = 6 the exponent x p(t) keeps growing in
    // magnitude with the right sign: exp2 -> 0 / inf, rcp -> 1 / 0, gelu -> x / -0, no NaN (inf - inf cannot occur: every Horner step
    // past overflow is (-inf) * (+inf) + c).  For x^2 <= 36 nothing changes (bit-identical); beyond, both forms sit within 6e-9 of the
    // exact value.  -DSC_GELU_CLAMP restores the clamp (A/B); the four-term fit keeps it.
#if SC_GELU_TERMS != 5 || defined(SC_GELU_CLAMP)
    t.x = fminf(t.x, 36.f); t.y = fminf(t.y, 36.f);
#endif
#if SC_GELU_TERMS == 5
    f32x2 p = __builtin_elementwise_fma(f32x2{SC_GK4, SC_GK4}, t, f32x2{SC_GK3, SC_GK3});
    p = __builtin_elementwise_fma(p, t, f32x2{SC_GK2, SC_GK2});
#else
    f32x2 p = __builtin_elementwise_fma(f32x2{SC_GK3, SC_GK3}, t, f32x2{SC_GK2, SC_GK2});
#endif
    p = __builtin_elementwise_fma(p, t, f32x2{SC_GK1, SC_GK1});
    p = __builtin_elementwise_fma(p, t, f32x2{SC_GK0, SC_GK0});
#else
    t.x = fminf(t.x, 64.f); t.y = fminf(t.y, 64.f);
    // coefficients times -log2(e): the exponent of 2 of exp(-x (c1 + c3 t + c5 t^2))
    f32x2 p = __builtin_elementwise_fma(f32x2{0.0010142630198970437f, 0.0010142630198970437f}, t, f32x2{-0.10677572339773178f, -0.10677572339773178f});
    p = __builtin_elementwise_fma(p, t, f32x2{-2.301121234893799f, -2.301121234893799f});
#endif
    const f32x2 w = p * x;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(w.x); e.y = __builtin_amdgcn_exp2f(w.y);
    const f32x2 d = e + 1.0f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    return x * r;
#endif
}

// gelu_grad_as on two elements: the polynomial, squarings and products as packed fp32 ops, rcp / exp per element.  Same operation order
// as gelu_grad_as => the same bits.
__device__ __forceinline__ f32x2 gelu_grad_as2(f32x2 x) {
    f32x2 ax;
    ax.x = fabsf(x.x); ax.y = fabsf(x.y);
    const f32x2 z = ax * 0.70710678118654752f;
    f32x2 p = __builtin_elementwise_fma(f32x2{0.0000430638f, 0.0000430638f}, z, f32x2{0.0002765672f, 0.0002765672f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0001520143f, 0.0001520143f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0092705272f, 0.0092705272f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0422820123f, 0.0422820123f});
    p = __builtin_elementwise_fma(p, z, f32x2{0.0705230784f, 0.0705230784f});
    p = __builtin_elementwise_fma(p, z, f32x2{1.0f, 1.0f});
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    f32x2 he;
    he.x = 0.5f * __builtin_amdgcn_rcpf(p.x); he.y = 0.5f * __builtin_amdgcn_rcpf(p.y);
    f32x2 cdf;
    cdf.x = x.x >= 0.f ? 1.f - he.x : he.x;
    cdf.y = x.y >= 0.f ? 1.f - he.y : he.y;
    const f32x2 xx = (x * -0.5f) * x;
    f32x2 ex;
    ex.x = __expf(xx.x); ex.y = __expf(xx.y);
    return __builtin_elementwise_fma(x * 0.3989422804014327f, ex, cdf);
}

// Stateless dropout masks (train mode): lowbias32 integer hash of (element-pair index ^ seed); the low / high 16 bits decide
// the even / odd element of the pair: keep iff bits >= thr16 = round(p * 65536).  The same function is evaluated on the host
// (tests/, numpy uint32) to reconstruct a mask exactly.
__device__ __forceinline__ uint32_t sc_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// keep flags of the 8 consecutive elements idx .. idx + 7 (idx even) as a bit mask
__device__ __forceinline__ uint32_t sc_keep8(uint32_t idx, uint32_t seed, uint32_t thr16) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t h = sc_hash32(((idx >> 1) + i) ^ seed);
        m |= ((h & 0xffffu) >= thr16 ? 1u : 0u) << (2 * i);
        m |= ((h >> 16) >= thr16 ? 1u : 0u) << (2 * i + 1);
    }
    return m;
}

// Attention-PROBABILITY dropout (round 6; the other dropout sites keep the 16-bit pair fields above): ONE hash word per FOUR consecutive
// keys of a query.  Element e = (probability row) * pitch + key; word = sc_hash32((e >> 2) ^ seed); position i = e & 3 of the quad reads
// BYTE (i & 1) * 2 + (i >> 1) of the word (0, 2, 1, 3: the forward kernel tests the even bytes of a word as one packed pair and the odd
// bytes as the next); keep iff byte >= thr8 = round(p * 256).  The rate actually applied is thr8 / 256 (p = 0.1 -> 26 / 256 = 0.1016)
// and the kept probabilities are scaled by 256 / (256 - thr8), so the expectation is exact.  Why: the hash was ~40 % of the train-mode
// attention kernel's VALU work with one word per two probabilities.  Host twin: tests/test_gpu_kernels.py _keep_mask8.
__device__ __forceinline__ uint32_t sc_drop8_thr(float p) { return (uint32_t)(p * 256.f + 0.5f); }
__device__ __forceinline__ float sc_drop8_scale(uint32_t thr8) { return 256.f / (float)(256u - thr8); }
__device__ __forceinline__ uint32_t sc_drop8_shift(uint32_t pos) { return 8u * ((pos & 1u) * 2u + (pos >> 1)); }      // bit offset of position pos
__device__ __forceinline__ bool sc_drop8_keep(uint32_t word, uint32_t pos, uint32_t thr8) { return ((word >> sc_drop8_shift(pos)) & 0xffu) >= thr8; }

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

// XOR swizzle (in 8-byte units) of a transposed [64 rows][64 x bf16] LDS tile with 128-byte rows, read as MFMA fragments with
// ds_read_b64 and staged with ds_write_b64 (csrc/attention.hip explains the three terms)
__device__ __forceinline__ int sc_tr_swizzle(int row) { return ((row >> 1) & 15) ^ (row & 1) ^ ((row >> 5) & 1); }

// host side
void sc_set_error(const char* fmt, ...);
int sc_num_cus();   // compute units of the current device (immutable cache; 256 on MI355X)
int sc_option(int key);   // tuning switches (sc_set_option): 1 = plain stores on residual tiles of the 256- / 128-row GEMMs
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
