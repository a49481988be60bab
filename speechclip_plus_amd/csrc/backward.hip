// Row kernels of the backward pass (HBM-bound): LayerNorm backward over bf16 activations, activation forward / backward
// (erf-GELU of fairseq's FFN, QuickGELU of CLIP's MLP).  Used by the frozen CLIP text tower's input gradient
// (avssl/module/clip_official.py:222-279 under autograd) and by the HuBERT layer backward (audio_encoder.trainable).
#include "sc_common.h"

namespace {

// dx = rstd (g - mean(g) - xhat mean(g xhat)) (+ dres),  g = dy gamma,  statistics recomputed from the saved LN input x.
// Persistent over rows (row = wave_global + k * n_waves) so that each lane can also accumulate its columns' dgamma / dbeta
// partials in registers; partial[workgroup][D] (the four waves added through LDS) is reduced by the caller (sc_colsum_f32):
// fixed order, no atomics.
// EXT (the differentiated HuBERT layers): the row that leaves is also what the NEXT products of the backward read - its dropped copy
// (the residual branch's F.dropout mask, regenerated: dx_drop, the operand of the branch's input- and weight-gradient GEMMs) and the
// column sums of that copy (the branch's bias gradient) leave from the same pass: no dropout launch, no column-sum pass over it.
template <int NCH, bool EXT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                            const uint16_t* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ gamma, const uint16_t* __restrict__ dres,
                                                            int64_t lddres, uint16_t* __restrict__ dx, int64_t lddx, int64_t rows,
                                                            int D, float eps, float* __restrict__ dg_part,
                                                            float* __restrict__ db_part, uint16_t* __restrict__ dx_drop, int64_t lddd,
                                                            uint32_t drop_thr, float drop_scale, uint32_t drop_seed,
                                                            float* __restrict__ ds_part) {
    const int lane = threadIdx.x & 63;
    const int64_t wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int nchunks = D >> 2;
    float ag[NCH][4], ab[NCH][4], gm[NCH][4], as[EXT ? NCH : 1][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ag[i][j] = 0.f;
            ab[i][j] = 0.f;
            if (EXT) as[i][j] = 0.f;
            gm[i][j] = ch < nchunks ? gamma[ch * 4 + j] : 0.f;
        }
    }
    for (int64_t row = wg; row < rows; row += nw) {
        const uint16_t* xr = x + row * ldx;
        const uint16_t* dr = dy + row * lddy;
        float xv[NCH][4], dv[NCH][4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunks) {
                const uint2 u = *(const uint2*)(xr + ch * 4), w = *(const uint2*)(dr + ch * 4);
                xv[i][0] = bflo(u.x); xv[i][1] = bfhi(u.x); xv[i][2] = bflo(u.y); xv[i][3] = bfhi(u.y);
                dv[i][0] = bflo(w.x); dv[i][1] = bfhi(w.x); dv[i][2] = bflo(w.y); dv[i][3] = bfhi(w.y);
                sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[i][j] = 0.f; dv[i][j] = 0.f; }
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (lane + i * 64 < nchunks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = xv[i][j] - mean;
                    sq += d * d;
                }
            }
        const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (lane + i * 64 < nchunks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[i][j] - mean) * rstd;
                    const float g = dv[i][j] * gm[i][j];
                    xv[i][j] = xh;                       // keep xhat
                    s1 += g;
                    s2 += g * xh;
                    ag[i][j] += dv[i][j] * xh;
                    ab[i][j] += dv[i][j];
                }
            }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunks) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (dv[i][j] * gm[i][j] - s1 - xv[i][j] * s2);
                if (dres) {
                    const uint2 r = *(const uint2*)(dres + row * lddres + ch * 4);
                    o[0] += bflo(r.x); o[1] += bfhi(r.x); o[2] += bflo(r.y); o[3] += bfhi(r.y);
                }
                uint2 w;
                w.x = pack2bf(o[0], o[1]);
                w.y = pack2bf(o[2], o[3]);
                *(uint2*)(dx + row * lddx + ch * 4) = w;
                if (EXT) {
                    // the stored bf16 values, dropped with the mask of element row * D + column (sc_keep8: 8 flags per hash word)
                    float d[4] = {bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y)};
                    if (drop_thr) {
                        const uint32_t idx = (uint32_t)row * (uint32_t)D + (uint32_t)(ch * 4);
                        const uint32_t keep = sc_keep8(idx & ~7u, drop_seed, drop_thr) >> (idx & 4u);
#pragma unroll
                        for (int j = 0; j < 4; ++j) d[j] = (keep >> j) & 1u ? d[j] * drop_scale : 0.f;
                    }
                    uint2 wd;
                    wd.x = pack2bf(d[0], d[1]);
                    wd.y = pack2bf(d[2], d[3]);
                    if (dx_drop) *(uint2*)(dx_drop + row * lddd + ch * 4) = wd;
                    as[i][0] += bflo(wd.x); as[i][1] += bfhi(wd.x); as[i][2] += bflo(wd.y); as[i][3] += bfhi(wd.y);
                }
            }
        }
    }
    if (dg_part) {                                   // the four waves' column partials are added through LDS: one row per workgroup
        __shared__ float red[EXT ? 3 : 2][4][NCH * 256];
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[0][w][(i * 64 + lane) * 4 + j] = ag[i][j];
                red[1][w][(i * 64 + lane) * 4 + j] = ab[i][j];
                if (EXT) red[2][w][(i * 64 + lane) * 4 + j] = as[i][j];
            }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            dg_part[(int64_t)blockIdx.x * D + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
            db_part[(int64_t)blockIdx.x * D + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
            if (EXT && ds_part) ds_part[(int64_t)blockIdx.x * D + c] = (red[2][0][c] + red[2][1][c]) + (red[2][2][c] + red[2][3][c]);
        }
    }
}

// out[b R + t] = (prev ? prev[b R + t] : 0) + bf16(w[0] * dX[b, t + row_off]) for t < T, prev (or 0) for the padding rows: the gradient
// that reaches hidden state n of a differentiated encoder = what came down from layer n + 1 plus its share of the weighted sum's
// gradient (fp32 [B, R, D], frames at row offset row_off) - one pass instead of slice + scale + cast + add.  Rounding as the
// element-wise formulation had it: the share is rounded to bf16, the sum again.
__global__ __launch_bounds__(256) void wsum_share_kernel(const float* __restrict__ dX, const float* __restrict__ w,
                                                         const uint16_t* __restrict__ prev, uint16_t* __restrict__ out, int B, int R, int T,
                                                         int D, int row_off) {
    const int64_t per_row = D >> 3, total = (int64_t)B * R * per_row;
    const float wn = w[0];
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t row = q / per_row;
        const int c = (int)(q - row * per_row) * 8;
        const int t = (int)(row % R);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t < T && t + row_off < R) {
            const float* gp = dX + (row + row_off) * D + c;
            const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
            uint4 r;                                                     // the share, rounded to bf16 first
            r.x = pack2bf(g0[0] * wn, g0[1] * wn); r.y = pack2bf(g0[2] * wn, g0[3] * wn);
            r.z = pack2bf(g1[0] * wn, g1[1] * wn); r.w = pack2bf(g1[2] * wn, g1[3] * wn);
            v[0] = bflo(r.x); v[1] = bfhi(r.x); v[2] = bflo(r.y); v[3] = bfhi(r.y);
            v[4] = bflo(r.z); v[5] = bfhi(r.z); v[6] = bflo(r.w); v[7] = bfhi(r.w);
        }
        if (prev) {
            const uint4 p = *(const uint4*)(prev + row * D + c);
            v[0] += bflo(p.x); v[1] += bfhi(p.x); v[2] += bflo(p.y); v[3] += bfhi(p.y);
            v[4] += bflo(p.z); v[5] += bfhi(p.z); v[6] += bflo(p.w); v[7] += bfhi(p.w);
        }
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        *(uint4*)(out + row * D + c) = o;
    }
}

// act_fwd / act_grad: sc_common.h (shared with the GEMM epilogue that fuses them, gemm_bf16.hip aux_mode)
// out = act(u)  (df == nullptr)   or   out = df * act'(u);   8 bf16 per thread
__global__ void act_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ df, uint16_t* __restrict__ out, int64_t n8,
                           int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const uint4 a = ((const uint4*)u)[i];
    float v[8] = {bflo(a.x), bfhi(a.x), bflo(a.y), bfhi(a.y), bflo(a.z), bfhi(a.z), bflo(a.w), bfhi(a.w)};
    if (df) {
        const uint4 d = ((const uint4*)df)[i];
        const float g[8] = {bflo(d.x), bfhi(d.x), bflo(d.y), bfhi(d.y), bflo(d.z), bfhi(d.z), bflo(d.w), bfhi(d.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g[j] * act_grad(v[j], act);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j], act);
    }
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    ((uint4*)out)[i] = o;
}

// out = dropout(x): keep with probability 1 - p, scale by 1 / (1 - p); the mask of element i is that of sc_gemm_args (sc_keep8);
// rows of ld elements, D of them used.  8 bf16 per thread.
__global__ void dropout_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out, int64_t ldo, int64_t rows,
                               int D, uint32_t thr, float scale, uint32_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, per_row = D >> 3;
    if (i >= rows * per_row) return;
    const int64_t row = i / per_row;
    const int c = (int)(i - row * per_row) * 8;
    const uint4 a = *(const uint4*)(x + row * ldx + c);
    float v[8] = {bflo(a.x), bfhi(a.x), bflo(a.y), bfhi(a.y), bflo(a.z), bfhi(a.z), bflo(a.w), bfhi(a.w)};
    const uint32_t keep = sc_keep8((uint32_t)(row * D + c), seed, thr);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (keep >> e) & 1u ? v[e] * scale : 0.f;
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    *(uint4*)(out + row * ldo + c) = o;
}

// out[i] = keep(i) ? 1 / (1 - p) : 0  (the multiplier F.dropout applies), same stateless hash as everywhere else: one launch in place of
// rand + compare + cast + scale
__global__ void dropout_mult_kernel(float* __restrict__ out, int64_t n8, uint32_t thr, float scale, uint32_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const uint32_t keep = sc_keep8((uint32_t)(i * 8), seed, thr);
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = (keep >> e) & 1u ? scale : 0.f;
        b[e] = (keep >> (4 + e)) & 1u ? scale : 0.f;
    }
    *(f32x4*)(out + i * 8) = a;
    *(f32x4*)(out + i * 8 + 4) = b;
}

}  // namespace

extern "C" int sc_dropout_mult_f32(float* out, int64_t n, float p, uint32_t seed, void* stream) {
    SC_CHECK(out && n > 0 && n % 8 == 0 && n < ((int64_t)1 << 32) && ((uintptr_t)out % 16) == 0, "sc_dropout_mult_f32: n=%lld must be a positive multiple of 8 below 2^32, out 16-byte aligned", (long long)n);
    SC_CHECK(p >= 0.f && p < 1.f, "sc_dropout_mult_f32: p=%f", (double)p);
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(dropout_mult_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n8,
                       (uint32_t)(p * 65536.f + 0.5f), 1.f / (1.f - p), seed);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_dropout_bf16(const sc_bf16* x, int64_t ldx, sc_bf16* out, int64_t ldo, int64_t rows, int32_t D, float p, uint32_t seed,
                               void* stream) {
    SC_CHECK(x && out && rows > 0 && D > 0 && D % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "sc_dropout_bf16: bad args");
    SC_CHECK(p >= 0.f && p < 1.f && rows * D < ((int64_t)1 << 32), "sc_dropout_bf16: p=%f, rows*D must be < 2^32", (double)p);
    const int64_t n8 = rows * (D / 8);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, rows, D,
                       (uint32_t)(p * 65536.f + 0.5f), 1.f / (1.f - p), seed);
    SC_LAUNCH_CHECK();
    return 0;
}

static int layernorm_bwd_launch(const uint16_t* x, int64_t ldx, const uint16_t* dy, int64_t lddy, const float* gamma, const uint16_t* dres,
                                int64_t lddres, uint16_t* dx, int64_t lddx, int64_t rows, int32_t D, float eps, float* dgamma_partial,
                                float* dbeta_partial, int32_t n_partial, bool ext, uint16_t* dx_drop, int64_t lddd, float drop_p,
                                uint32_t drop_seed, float* dsum_partial, void* stream) {
    SC_CHECK(x && dy && gamma && dx, "sc_layernorm_bwd_bf16: null pointer");
    SC_CHECK(rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, "sc_layernorm_bwd_bf16: D=%d must be a multiple of 4, <= 1024", D);
    SC_CHECK(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && (dres == nullptr || lddres % 4 == 0), "sc_layernorm_bwd_bf16: leading dims");
    SC_CHECK((dgamma_partial == nullptr) == (dbeta_partial == nullptr), "sc_layernorm_bwd_bf16: both partial buffers or none");
    if (ext) {
        SC_CHECK(drop_p >= 0.f && drop_p < 1.f && (dx_drop == nullptr || lddd % 4 == 0) && (dsum_partial == nullptr || dgamma_partial != nullptr) &&
                     (drop_p == 0.f || (rows * (int64_t)D < ((int64_t)1 << 32) && D % 8 == 0)),
                 "sc_layernorm_bwd_drop_bf16: p=%f, column sums need the parameter partial buffers, dropout needs rows * D < 2^32 and D %% 8 == 0",
                 (double)drop_p);
    }
    // with partial sums the caller's buffers fix the workgroup count (n_partial rows of D floats each)
    int grid;
    if (dgamma_partial) {
        SC_CHECK(n_partial >= 1, "sc_layernorm_bwd_bf16: n_partial must be positive");
        grid = n_partial;
    } else {
        grid = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    }
    hipStream_t s = (hipStream_t)stream;
    const int nch = (D / 4 + 63) / 64;
    const uint32_t thr = ext ? (uint32_t)(drop_p * 65536.f + 0.5f) : 0u;
    const float scale = ext && drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#define SC_LNB(N, E) hipLaunchKernelGGL((layernorm_bwd_kernel<N, E>), dim3(grid), dim3(256), 0, s, x, ldx, dy, lddy, gamma, dres, lddres, dx, \
                                        lddx, rows, D, eps, dgamma_partial, dbeta_partial, dx_drop, lddd, thr, scale, drop_seed, dsum_partial)
    if (ext) {
        switch (nch) {
            case 1: SC_LNB(1, true); break;
            case 2: SC_LNB(2, true); break;
            case 3: SC_LNB(3, true); break;
            default: SC_LNB(4, true); break;
        }
    } else {
        switch (nch) {
            case 1: SC_LNB(1, false); break;
            case 2: SC_LNB(2, false); break;
            case 3: SC_LNB(3, false); break;
            default: SC_LNB(4, false); break;
        }
    }
#undef SC_LNB
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_layernorm_bwd_bf16(const sc_bf16* x, int64_t ldx, const sc_bf16* dy, int64_t lddy, const float* gamma,
                                     const sc_bf16* dres, int64_t lddres, sc_bf16* dx, int64_t lddx, int64_t rows, int32_t D,
                                     float eps, float* dgamma_partial, float* dbeta_partial, int32_t n_partial, void* stream) {
    return layernorm_bwd_launch(x, ldx, dy, lddy, gamma, dres, lddres, dx, lddx, rows, D, eps, dgamma_partial, dbeta_partial, n_partial, false,
                                nullptr, 0, 0.f, 0u, nullptr, stream);
}

extern "C" int sc_layernorm_bwd_drop_bf16(const sc_bf16* x, int64_t ldx, const sc_bf16* dy, int64_t lddy, const float* gamma,
                                          const sc_bf16* dres, int64_t lddres, sc_bf16* dx, int64_t lddx, int64_t rows, int32_t D,
                                          float eps, float* dgamma_partial, float* dbeta_partial, int32_t n_partial, sc_bf16* dx_drop,
                                          int64_t lddd, float drop_p, uint32_t drop_seed, float* dsum_partial, void* stream) {
    return layernorm_bwd_launch(x, ldx, dy, lddy, gamma, dres, lddres, dx, lddx, rows, D, eps, dgamma_partial, dbeta_partial, n_partial, true,
                                dx_drop, lddd, drop_p, drop_seed, dsum_partial, stream);
}

extern "C" int sc_wsum_share_bf16(const float* dX, const float* w, const sc_bf16* prev, sc_bf16* out, int32_t B, int32_t R, int32_t T, int32_t D,
                                  int32_t row_off, void* stream) {
    SC_CHECK(dX && w && out && B > 0 && R > 0 && T >= 0 && T <= R && D > 0 && D % 8 == 0 && row_off >= 0, "sc_wsum_share_bf16: bad arguments");
    SC_CHECK(((uintptr_t)dX % 16) == 0 && ((uintptr_t)out % 16) == 0 && (prev == nullptr || ((uintptr_t)prev % 16) == 0), "sc_wsum_share_bf16: alignment");
    const int64_t total = (int64_t)B * R * (D >> 3);
    hipLaunchKernelGGL(wsum_share_kernel, dim3((unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0, (hipStream_t)stream, dX, w,
                       prev, out, B, R, T, D, row_off);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_act_bf16(const sc_bf16* u, const sc_bf16* df, sc_bf16* out, int64_t n, int32_t act, void* stream) {
    SC_CHECK(u && out && n > 0 && n % 8 == 0, "sc_act_bf16: n must be a positive multiple of 8");
    SC_CHECK(act == 1 || act == 2, "sc_act_bf16: act=%d (1 erf-GELU, 2 QuickGELU)", act);
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, df, out, n8, act);
    SC_LAUNCH_CHECK();
    return 0;
}

namespace {

// yT[c, r] = x[r, c]   ([rows, cols] bf16 row-major, ldx >= cols  ->  [cols, ldy >= rows]); 64 x 64 tiles through LDS.
// Feeds the weight-gradient GEMMs: dW[n, k] = sum_rows dY[row, n] X[row, k] = (dY^T) . (X^T)^T with the row index as the
// contraction (K) dimension of sc_gemm_bf16.
// colpart (optional): colpart[row block][c] = sum of the block's 64 rows of column c in fp32 - the bias gradient's first stage
// comes for free with the pass that already reads dY.
__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y,
                                                        int64_t ldy, int rows, int cols, float* __restrict__ colpart, int64_t sx = 0,
                                                        int64_t sy = 0) {
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
    x += blockIdx.z * sx;                          // batch (blockIdx.z): matrix z starts sx / sy elements after matrix z - 1
    y += blockIdx.z * sy;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 256, row = id >> 3, ch = id & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + row < rows && c0 + ch * 8 < cols) v = *(const uint4*)(x + (int64_t)(r0 + row) * ldx + c0 + ch * 8);
        uint16_t* d = &tile[row][ch * 8];
        d[0] = v.x & 0xffff; d[1] = v.x >> 16; d[2] = v.y & 0xffff; d[3] = v.y >> 16;
        d[4] = v.z & 0xffff; d[5] = v.z >> 16; d[6] = v.w & 0xffff; d[7] = v.w >> 16;
    }
    __syncthreads();
    if (colpart && tid < 64 && c0 + tid < cols) {
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s += bf2f(tile[r][tid]);       // rows past the end were loaded as zeros
        colpart[(int64_t)blockIdx.x * cols + c0 + tid] = s;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 256, c = id >> 3, ch = id & 7;
        if (c0 + c < cols && r0 + ch * 8 < rows) {
            uint4 o;
            o.x = tile[ch * 8 + 0][c] | ((uint32_t)tile[ch * 8 + 1][c] << 16);
            o.y = tile[ch * 8 + 2][c] | ((uint32_t)tile[ch * 8 + 3][c] << 16);
            o.z = tile[ch * 8 + 4][c] | ((uint32_t)tile[ch * 8 + 5][c] << 16);
            o.w = tile[ch * 8 + 6][c] | ((uint32_t)tile[ch * 8 + 7][c] << 16);
            *(uint4*)(y + (int64_t)(c0 + c) * ldy + r0 + ch * 8) = o;
        }
    }
}

// The bf16 working copies of a trainable fp32 weight in one pass: y = bf16(x) [rows, cols] and / or yT = bf16(x)^T [cols, rows]
// (both GEMM operand layouts of a projection: forward x W^T reads W, the input gradient dy W reads W^T).  Was _to_copy + clone per copy.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy,
                                                             uint16_t* __restrict__ yT, int64_t ldyT, int rows, int cols) {
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + i * 256, row = id >> 4, ch = id & 15;           // 16 chunks of 4 floats per tile row
        uint2 o = make_uint2(0, 0);
        if (r0 + row < rows && c0 + ch * 4 < cols) {
            const f32x4 v = *(const f32x4*)(x + (int64_t)(r0 + row) * ldx + c0 + ch * 4);
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            if (y) *(uint2*)(y + (int64_t)(r0 + row) * ldy + c0 + ch * 4) = o;
        }
        uint16_t* d = &tile[row][ch * 4];
        d[0] = o.x & 0xffff; d[1] = o.x >> 16; d[2] = o.y & 0xffff; d[3] = o.y >> 16;
    }
    if (!yT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + i * 256, c = id >> 4, ch = id & 15;
        if (c0 + c < cols && r0 + ch * 4 < rows) {
            uint2 o;
            o.x = tile[ch * 4 + 0][c] | ((uint32_t)tile[ch * 4 + 1][c] << 16);
            o.y = tile[ch * 4 + 2][c] | ((uint32_t)tile[ch * 4 + 3][c] << 16);
            *(uint2*)(yT + (int64_t)(c0 + c) * ldyT + r0 + ch * 4) = o;
        }
    }
}

// partial[blk][c] = sum over the block's rows of x[row, c]  (bf16 -> fp32; bias gradients).  A thread owns 8 consecutive columns
// (one 16-byte load per row); with fewer than 256 column chunks the block walks 256 / chunks rows at a time and adds its row lanes
// through LDS in fixed order at the end.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const uint16_t* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                          float* __restrict__ partial) {
    __shared__ float red[256 * 8];
    const int chunks = cols >> 3;
    const int cpb = chunks < 256 ? chunks : 256;            // column chunks per block
    const int rpi = 256 / cpb;                              // rows per iteration
    const int tc = threadIdx.x % cpb, tr = threadIdx.x / cpb;
    const int c8 = blockIdx.x * cpb + tc;
    const bool live = tr < rpi && c8 < chunks;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live)
        for (int64_t r = r0 + tr; r < r1; r += rpi) {
            const uint4 u = *(const uint4*)(x + r * ldx + c8 * 8);
            s[0] += bflo(u.x); s[1] += bfhi(u.x); s[2] += bflo(u.y); s[3] += bfhi(u.y);
            s[4] += bflo(u.z); s[5] += bfhi(u.z); s[6] += bflo(u.w); s[7] += bfhi(u.w);
        }
    if (rpi > 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
        __syncthreads();
        if (live && tr == 0)
            for (int j = 1; j < rpi; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += red[(j * cpb + tc) * 8 + e];
    }
    if (live && tr == 0) {
        float* o = partial + (int64_t)blockIdx.y * cols + c8 * 8;
        *(f32x4*)o = f32x4{s[0], s[1], s[2], s[3]};
        *(f32x4*)(o + 4) = f32x4{s[4], s[5], s[6], s[7]};
    }
}

// Input gradient of a channels-last Conv1d(k = 3, stride 2) that ran as a strided-row GEMM (window m = rows 2m, 2m + 1, 2m + 2 of the
// input): dcols [M, 3 C] = dy . W holds the gradient of every window; the input row r collects the windows that contain it -
//   dx[2m]     = dcols[m][0 : C] + dcols[m - 1][2C : 3C]        dx[2m + 1] = dcols[m][C : 2C]
// one pass (was: three strided torch copies / adds over GB-sized tensors, 20 ms of the fully-trainable step).
// u != NULL (round 4): the layer below ends in an activation whose pre-activation u [2M, C] was kept - the gradient leaves already
// multiplied by act'(u), i.e. dx <- bf16(dx) * act'(u), the same values as this kernel followed by sc_act_bf16(u, dx) (two passes over
// GB-sized tensors less per conv layer of the fully trainable encoder).
__global__ __launch_bounds__(256) void conv_overlap_add_kernel(const uint16_t* __restrict__ dcols, uint16_t* __restrict__ dx, int64_t M, int C,
                                                               const uint16_t* __restrict__ u, int act) {
    const int64_t chunks = C >> 3;                      // 8 bf16 per thread
    const int64_t total = M * chunks;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = q / chunks;
        const int c = (int)(q - m * chunks) * 8;
        const uint16_t* row = dcols + m * 3 * C;
        const uint4 t0 = *(const uint4*)(row + c), t1 = *(const uint4*)(row + C + c);
        uint4 e = t0;
        if (m > 0) {
            const uint4 p2 = *(const uint4*)(row - 3 * (int64_t)C + 2 * C + c);
            e.x = pack2bf(bflo(t0.x) + bflo(p2.x), bfhi(t0.x) + bfhi(p2.x));
            e.y = pack2bf(bflo(t0.y) + bflo(p2.y), bfhi(t0.y) + bfhi(p2.y));
            e.z = pack2bf(bflo(t0.z) + bflo(p2.z), bfhi(t0.z) + bfhi(p2.z));
            e.w = pack2bf(bflo(t0.w) + bflo(p2.w), bfhi(t0.w) + bfhi(p2.w));
        }
        uint4 o1 = t1;
        if (u) {
            const uint4 u0 = *(const uint4*)(u + (2 * m) * C + c), u1 = *(const uint4*)(u + (2 * m + 1) * C + c);
            e.x = pack2bf(bflo(e.x) * act_grad(bflo(u0.x), act), bfhi(e.x) * act_grad(bfhi(u0.x), act));
            e.y = pack2bf(bflo(e.y) * act_grad(bflo(u0.y), act), bfhi(e.y) * act_grad(bfhi(u0.y), act));
            e.z = pack2bf(bflo(e.z) * act_grad(bflo(u0.z), act), bfhi(e.z) * act_grad(bfhi(u0.z), act));
            e.w = pack2bf(bflo(e.w) * act_grad(bflo(u0.w), act), bfhi(e.w) * act_grad(bfhi(u0.w), act));
            o1.x = pack2bf(bflo(t1.x) * act_grad(bflo(u1.x), act), bfhi(t1.x) * act_grad(bfhi(u1.x), act));
            o1.y = pack2bf(bflo(t1.y) * act_grad(bflo(u1.y), act), bfhi(t1.y) * act_grad(bfhi(u1.y), act));
            o1.z = pack2bf(bflo(t1.z) * act_grad(bflo(u1.z), act), bfhi(t1.z) * act_grad(bfhi(u1.z), act));
            o1.w = pack2bf(bflo(t1.w) * act_grad(bflo(u1.w), act), bfhi(t1.w) * act_grad(bfhi(u1.w), act));
        }
        *(uint4*)(dx + (2 * m) * C + c) = e;
        *(uint4*)(dx + (2 * m + 1) * C + c) = o1;
    }
}

}  // namespace

extern "C" int sc_transpose_bf16(const sc_bf16* x, int64_t ldx, sc_bf16* y, int64_t ldy, int32_t rows, int32_t cols,
                                 float* colsum_partial, void* stream) {
    SC_CHECK(x && y && rows > 0 && cols > 0, "sc_transpose_bf16: bad args");
    SC_CHECK(rows % 8 == 0 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "sc_transpose_bf16: rows, cols and leading dims must be multiples of 8");
    hipLaunchKernelGGL(transpose_kernel, dim3((rows + 63) / 64, (cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols,
                       colsum_partial);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cast_transpose_f32_bf16(const float* x, int64_t ldx, sc_bf16* y, int64_t ldy, sc_bf16* yT, int64_t ldyT, int32_t rows,
                                          int32_t cols, void* stream) {
    SC_CHECK(x && (y || yT) && rows > 0 && cols > 0, "sc_cast_transpose_f32_bf16: bad args");
    SC_CHECK(rows % 4 == 0 && cols % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && (!y || (ldy % 4 == 0 && ((uintptr_t)y % 8) == 0)) &&
                 (!yT || (ldyT % 4 == 0 && ((uintptr_t)yT % 8) == 0)),
             "sc_cast_transpose_f32_bf16: rows, cols and leading dims must be multiples of 4, x 16-byte and y / yT 8-byte aligned");
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((rows + 63) / 64, (cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (uint16_t*)y, ldy, (uint16_t*)yT, ldyT, rows, cols);
    SC_LAUNCH_CHECK();
    return 0;
}

// nbatch equally shaped matrices at uniform strides in one launch (the transposed bf16 working copies of a trainable encoder's
// per-layer weights: 4 launches per step instead of 48)
extern "C" int sc_transpose_batched_bf16(const sc_bf16* x, int64_t ldx, int64_t sx, sc_bf16* y, int64_t ldy, int64_t sy, int32_t rows,
                                         int32_t cols, int32_t nbatch, void* stream) {
    SC_CHECK(x && y && rows > 0 && cols > 0 && nbatch > 0 && nbatch <= 65535, "sc_transpose_batched_bf16: bad args");
    SC_CHECK(rows % 8 == 0 && cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && sx % 8 == 0 && sy % 8 == 0 && ((uintptr_t)x % 16) == 0 &&
                 ((uintptr_t)y % 16) == 0, "sc_transpose_batched_bf16: rows, cols, leading dims and strides must be multiples of 8, operands 16-byte aligned");
    hipLaunchKernelGGL(transpose_kernel, dim3((rows + 63) / 64, (cols + 63) / 64, nbatch), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows,
                       cols, (float*)nullptr, sx, sy);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_colsum_bf16(const sc_bf16* x, int64_t ldx, int64_t rows, int32_t cols, float* partial, int32_t nblk, void* stream) {
    SC_CHECK(x && partial && rows > 0 && cols > 0 && cols % 8 == 0 && nblk > 0 && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 &&
             ((uintptr_t)partial % 16) == 0, "sc_colsum_bf16: cols / ldx multiples of 8, 16-byte aligned operands");
    const int chunks = cols / 8, cpb = chunks < 256 ? chunks : 256;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((chunks + cpb - 1) / cpb, nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols, partial);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv_overlap_add_bf16(const sc_bf16* dcols, sc_bf16* dx, int64_t M, int32_t C, void* stream) {
    SC_CHECK(dcols && dx && M > 0 && C > 0 && C % 8 == 0 && ((uintptr_t)dcols % 16) == 0 && ((uintptr_t)dx % 16) == 0, "sc_conv_overlap_add_bf16: bad args");
    const int64_t total = M * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(conv_overlap_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dcols, dx, M, C, (const uint16_t*)nullptr, 0);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv_overlap_add_act_bf16(const sc_bf16* dcols, const sc_bf16* u, sc_bf16* dx, int64_t M, int32_t C, int32_t act, void* stream) {
    SC_CHECK(dcols && u && dx && M > 0 && C > 0 && C % 8 == 0 && ((uintptr_t)dcols % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)u % 16) == 0,
             "sc_conv_overlap_add_act_bf16: bad args");
    SC_CHECK(act == 1 || act == 2, "sc_conv_overlap_add_act_bf16: act=%d (1 erf-GELU, 2 QuickGELU)", act);
    const int64_t total = M * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(conv_overlap_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dcols, dx, M, C, u, act);
    SC_LAUNCH_CHECK();
    return 0;
}
