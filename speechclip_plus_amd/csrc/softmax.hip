// Row softmax of the GEMM-based attention core (speechclip_plus_amd/mha_block.py: the attention block of the cascaded+/hybrid+
// branches, avssl/module/kw_modules/TransformerModels.py:101-126 -> nn.MultiheadAttention), forward and backward.
//
//   forward   P  = softmax(scale * scores | key mask)            scores fp32 [rows, n] (a batched GEMM's output), P bf16
//             Pd = keep . P / (1 - p)                             train mode only; keep = the stateless hash mask of sc_common.h
//   backward  dP' = keep . dP / (1 - p) ;  dS = scale * P (dP' - sum_k P dP')      dP fp32, dS bf16
//
// One wave per row, the row lives in registers (n <= 1024: up to 4 float4 per lane), fp32 statistics.  row -> utterance by
// rows_per_batch (= heads * queries); mask[b, k] != 0 marks a padded key (its probability is exactly 0).
// HBM-bound: forward reads 4 n and writes 2 n (4 n with dropout) bytes per row, backward reads 6 n and writes 2 n.
#include "sc_common.h"

namespace {

constexpr int MAXN = 1024;    // 64 lanes x 4 chunks x 4 values

template <int NV>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ mask,
                                                          uint16_t* __restrict__ P, uint16_t* __restrict__ Pd, int64_t rows, int n,
                                                          int rows_per_batch, float scale_log2e, uint32_t thr, float drop_scale,
                                                          uint32_t seed, float* __restrict__ P32) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < rows; row += nwaves) {
        const float* s = scores + row * n;
        const uint8_t* mk = mask + (row / rows_per_batch) * n;
        float v[NV][4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < n) {
                const float4 a = *(const float4*)(s + c);
                const uint32_t m4 = *(const uint32_t*)(mk + c);
                v[i][0] = (m4 & 0xffu) ? -INFINITY : a.x * scale_log2e;
                v[i][1] = (m4 & 0xff00u) ? -INFINITY : a.y * scale_log2e;
                v[i][2] = (m4 & 0xff0000u) ? -INFINITY : a.z * scale_log2e;
                v[i][3] = (m4 & 0xff000000u) ? -INFINITY : a.w * scale_log2e;
            } else {
                v[i][0] = v[i][1] = v[i][2] = v[i][3] = -INFINITY;
            }
            mx = fmaxf(fmaxf(mx, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
        }
        mx = wave_max(mx);
        if (mx == -INFINITY) mx = 0.f;               // fully masked row: all probabilities 0 (never read downstream)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = exp2f(v[i][e] - mx);
                sum += v[i][e];
            }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c >= n) continue;
            float p[4] = {v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv};
            if (P32) {                                   // fp32 debug mode: unrounded probabilities, nothing else
                *(float4*)(P32 + row * n + c) = float4{p[0], p[1], p[2], p[3]};
                continue;
            }
            uint2 o;
            o.x = pack2bf(p[0], p[1]);
            o.y = pack2bf(p[2], p[3]);
            *(uint2*)(P + row * n + c) = o;
            if (Pd) {
                const uint32_t idx = (uint32_t)(row * n + c);
                const uint32_t h0 = sc_hash32((idx >> 1) ^ seed), h1 = sc_hash32(((idx >> 1) + 1) ^ seed);
                p[0] = (h0 & 0xffffu) >= thr ? p[0] * drop_scale : 0.f;
                p[1] = (h0 >> 16) >= thr ? p[1] * drop_scale : 0.f;
                p[2] = (h1 & 0xffffu) >= thr ? p[2] * drop_scale : 0.f;
                p[3] = (h1 >> 16) >= thr ? p[3] * drop_scale : 0.f;
                o.x = pack2bf(p[0], p[1]);
                o.y = pack2bf(p[2], p[3]);
                *(uint2*)(Pd + row * n + c) = o;
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, const uint16_t* __restrict__ P,
                                                          uint16_t* __restrict__ dS, int64_t rows, int n, float scale, uint32_t thr,
                                                          float drop_scale, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < rows; row += nwaves) {
        float g[NV][4], p[NV][4];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < n) {
                const float4 a = *(const float4*)(dP + row * n + c);
                const uint2 b = *(const uint2*)(P + row * n + c);
                g[i][0] = a.x; g[i][1] = a.y; g[i][2] = a.z; g[i][3] = a.w;
                p[i][0] = bflo(b.x); p[i][1] = bfhi(b.x); p[i][2] = bflo(b.y); p[i][3] = bfhi(b.y);
                if (thr) {
                    const uint32_t idx = (uint32_t)(row * n + c);
                    const uint32_t h0 = sc_hash32((idx >> 1) ^ seed), h1 = sc_hash32(((idx >> 1) + 1) ^ seed);
                    g[i][0] = (h0 & 0xffffu) >= thr ? g[i][0] * drop_scale : 0.f;
                    g[i][1] = (h0 >> 16) >= thr ? g[i][1] * drop_scale : 0.f;
                    g[i][2] = (h1 & 0xffffu) >= thr ? g[i][2] * drop_scale : 0.f;
                    g[i][3] = (h1 >> 16) >= thr ? g[i][3] * drop_scale : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[i][e] = p[i][e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dot = fmaf(p[i][e], g[i][e], dot);
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c >= n) continue;
            uint2 o;
            o.x = pack2bf(scale * p[i][0] * (g[i][0] - dot), scale * p[i][1] * (g[i][1] - dot));
            o.y = pack2bf(scale * p[i][2] * (g[i][2] - dot), scale * p[i][3] * (g[i][3] - dot));
            *(uint2*)(dS + row * n + c) = o;
        }
    }
}

int grid_for(int64_t rows) { return (int)((rows + 3) / 4 < 16384 ? (rows + 3) / 4 : 16384); }

}  // namespace

extern "C" int sc_softmax_fwd(const float* scores, const uint8_t* key_mask, sc_bf16* P, sc_bf16* Pd, int64_t rows, int32_t n,
                              int32_t rows_per_batch, float scale, float drop_p, uint32_t drop_seed, void* stream) {
    SC_CHECK(scores && key_mask && P && rows > 0 && rows_per_batch > 0, "sc_softmax_fwd: bad args");
    SC_CHECK(n > 0 && n % 4 == 0 && n <= MAXN, "sc_softmax_fwd: n=%d must be a multiple of 4, <= %d", n, MAXN);
    SC_CHECK(drop_p >= 0.f && drop_p < 1.f && ((drop_p > 0.f) == (Pd != nullptr)), "sc_softmax_fwd: Pd is given iff drop_p > 0");
    SC_CHECK(drop_p == 0.f || rows * n < ((int64_t)1 << 32), "sc_softmax_fwd: dropout needs rows * n < 2^32");
    const uint32_t thr = (uint32_t)(drop_p * 65536.f + 0.5f);
    const float ds = 1.f / (1.f - drop_p), sl = scale * 1.4426950408889634f;
    const int nv = (n / 4 + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
#define SC_SMF(N) hipLaunchKernelGGL((softmax_fwd_kernel<N>), dim3(grid_for(rows)), dim3(256), 0, s, scores, key_mask, P, Pd, rows, n, \
                                     rows_per_batch, sl, thr, ds, drop_seed, (float*)nullptr)
    if (nv <= 1) SC_SMF(1); else if (nv <= 2) SC_SMF(2); else SC_SMF(4);
#undef SC_SMF
    SC_LAUNCH_CHECK();
    return 0;
}

// fp32 debug mode of the encoder (speechclip_plus_amd/debug_fp32.py): the same row kernel with unrounded fp32 probabilities
extern "C" int sc_softmax_fwd_f32(const float* scores, const uint8_t* key_mask, float* P, int64_t rows, int32_t n, int32_t rows_per_batch,
                                  float scale, void* stream) {
    SC_CHECK(scores && key_mask && P && rows > 0 && rows_per_batch > 0, "sc_softmax_fwd_f32: bad args");
    SC_CHECK(n > 0 && n % 4 == 0 && n <= MAXN && ((uintptr_t)P % 16) == 0, "sc_softmax_fwd_f32: n=%d must be a multiple of 4, <= %d", n, MAXN);
    const float sl = scale * 1.4426950408889634f;
    const int nv = (n / 4 + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
#define SC_SMF(N) hipLaunchKernelGGL((softmax_fwd_kernel<N>), dim3(grid_for(rows)), dim3(256), 0, s, scores, key_mask, (uint16_t*)nullptr, \
                                     (uint16_t*)nullptr, rows, n, rows_per_batch, sl, 0u, 1.f, 0u, P)
    if (nv <= 1) SC_SMF(1); else if (nv <= 2) SC_SMF(2); else SC_SMF(4);
#undef SC_SMF
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_softmax_bwd(const float* dP, const sc_bf16* P, sc_bf16* dS, int64_t rows, int32_t n, float scale, float drop_p,
                              uint32_t drop_seed, void* stream) {
    SC_CHECK(dP && P && dS && rows > 0, "sc_softmax_bwd: bad args");
    SC_CHECK(n > 0 && n % 4 == 0 && n <= MAXN, "sc_softmax_bwd: n=%d must be a multiple of 4, <= %d", n, MAXN);
    SC_CHECK(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rows * n < ((int64_t)1 << 32)), "sc_softmax_bwd: drop_p / size");
    const uint32_t thr = (uint32_t)(drop_p * 65536.f + 0.5f);
    const float ds = 1.f / (1.f - drop_p);
    const int nv = (n / 4 + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
#define SC_SMB(N) hipLaunchKernelGGL((softmax_bwd_kernel<N>), dim3(grid_for(rows)), dim3(256), 0, s, dP, P, dS, rows, n, scale, thr, ds, \
                                     drop_seed)
    if (nv <= 1) SC_SMB(1); else if (nv <= 2) SC_SMB(2); else SC_SMB(4);
#undef SC_SMB
    SC_LAUNCH_CHECK();
    return 0;
}
