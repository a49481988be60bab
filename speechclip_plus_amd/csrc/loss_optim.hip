// fp32 kernels of the contrastive loss and the optimiser step.
//   sc_sgemm_f32     generic strided fp32 GEMM (LDS-tiled 64x64x16, 4x4 register blocking)
//   sc_infonce_lse   masked row / column log-sum-exp of the (Bg x Bg) logits + loss scalar
//   sc_infonce_grad  dL/dlogits
//   sc_sumsq_f32 / sc_adam_f32   global-norm clip + Adam on a flat parameter buffer
// The loss is explicitly fp32 in the reference (avssl/model/kwClip.py:1012,1024); at Bg <= 512 these kernels
// are launch/latency bound, so they favour simple, deterministic reductions over peak rate.
#include "sc_common.h"

namespace {

// ------------------------------------------------------------------------------------------ sgemm
constexpr int TS = 64, TK = 16;

__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, int64_t sai, int64_t sak,
                                                    const float* __restrict__ Bm, int64_t sbj, int64_t sbk,
                                                    float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                    float alpha, const float* __restrict__ bias) {
    __shared__ float As[TK][TS + 4];
    __shared__ float Bs[TK][TS + 4];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
    const int ty = tid >> 4, tx = tid & 15;    // 16 x 16 threads, 4 x 4 outputs each
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    // loader mapping: make the unit-stride dimension the fast thread index
    const bool a_kfast = (sak == 1), b_kfast = (sbk == 1);
    for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = tid + e * 256;            // 1024 elements per tile
            int ii, kk;
            if (a_kfast) { kk = id & 15; ii = id >> 4; } else { ii = id & 63; kk = id >> 6; }
            const int gi = i0 + ii, gk = k0 + kk;
            As[kk][ii] = (gi < M && gk < K) ? A[gi * sai + gk * sak] : 0.f;
            int jj, kb;
            if (b_kfast) { kb = id & 15; jj = id >> 4; } else { jj = id & 63; kb = id >> 6; }
            const int gj = j0 + jj, gkb = k0 + kb;
            Bs[kb][jj] = (gj < N && gkb < K) ? Bm[gj * sbj + gkb * sbk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[kk][ty * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = Bs[kk][tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(av[a], bv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int gi = i0 + ty * 4 + a;
        if (gi >= M) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = j0 + tx * 4 + b;
            if (gj < N) C[gi * ldc + gj] = alpha * acc[a][b] + (bias ? bias[gj] : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------ InfoNCE
// block i handles row i and column i.  neg[i,j] = (ids[i] != ids[j]) || i == j
__global__ __launch_bounds__(256) void infonce_lse_kernel(const float* __restrict__ logits,
                                                          const int64_t* __restrict__ ids, int Bg,
                                                          float* __restrict__ lse_row, float* __restrict__ lse_col,
                                                          float* __restrict__ terms) {
    __shared__ float red[2][4];
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t idi = ids ? ids[i] : (int64_t)i;
    float mr = -INFINITY, mc = -INFINITY;
    for (int j = threadIdx.x; j < Bg; j += 256) {
        const bool neg = (j == i) || (ids ? ids[j] != idi : true);
        if (neg) {
            mr = fmaxf(mr, logits[(int64_t)i * Bg + j]);
            mc = fmaxf(mc, logits[(int64_t)j * Bg + i]);
        }
    }
    mr = wave_max(mr);
    mc = wave_max(mc);
    if (lane == 0) { red[0][wave] = mr; red[1][wave] = mc; }
    __syncthreads();
    mr = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    mc = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    __syncthreads();
    float sr = 0.f, sc = 0.f;
    for (int j = threadIdx.x; j < Bg; j += 256) {
        const bool neg = (j == i) || (ids ? ids[j] != idi : true);
        if (neg) {
            sr += __expf(logits[(int64_t)i * Bg + j] - mr);
            sc += __expf(logits[(int64_t)j * Bg + i] - mc);
        }
    }
    sr = wave_sum(sr);
    sc = wave_sum(sc);
    if (lane == 0) { red[0][wave] = sr; red[1][wave] = sc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lr = mr + logf((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        const float lc = mc + logf((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
        lse_row[i] = lr;
        lse_col[i] = lc;
        terms[i] = -2.f * logits[(int64_t)i * Bg + i] + lr + lc;
    }
}

__global__ __launch_bounds__(256) void reduce_mean_kernel(const float* __restrict__ terms, int n, float scale,
                                                          float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += terms[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// G[i,j] = gscale/(2Bg) * (neg * (exp(l - lse_row_i) + exp(l - lse_col_j)) - 2 [i == j]);
// dlogit_dot[i] = sum_j G[i,j] * logits[i,j]   (for d/d inv_temp = sum / inv_temp)
__global__ __launch_bounds__(256) void infonce_grad_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ ids,
                                                           const float* __restrict__ lse_row,
                                                           const float* __restrict__ lse_col, int Bg,
                                                           const float* __restrict__ gscale, float* __restrict__ G,
                                                           float* __restrict__ dlogit_dot) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    const int64_t idi = ids ? ids[i] : (int64_t)i;
    const float lr = lse_row[i];
    const float gs = gscale[0] / (2.f * (float)Bg);
    float dot = 0.f;
    for (int j = threadIdx.x; j < Bg; j += 256) {
        const float l = logits[(int64_t)i * Bg + j];
        const bool neg = (j == i) || (ids ? ids[j] != idi : true);
        float g = 0.f;
        if (neg) g = __expf(l - lr) + __expf(l - lse_col[j]);
        if (j == i) g -= 2.f;
        g *= gs;
        G[(int64_t)i * Bg + j] = g;
        dot += g * l;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) dlogit_dot[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------ optimiser
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float beta1, float beta2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, const float* __restrict__ gn_partial, int nblk,
                                                   float max_norm) {
    float clip = 1.f;
    if (gn_partial && max_norm > 0.f) {
        float tot = 0.f;
        for (int i = 0; i < nblk; ++i) tot += gn_partial[i];      // fixed order: deterministic
        const float norm = sqrtf(tot);
        clip = fminf(1.f, max_norm / (norm + 1e-6f));             // torch.nn.utils.clip_grad_norm_
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * clip;
        const float pi = p[i];
        gi = fmaf(wd, pi, gi);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

}  // namespace

extern "C" int sc_sgemm_f32(const float* A, int64_t sai, int64_t sak, const float* Bm, int64_t sbj, int64_t sbk,
                            float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, const float* bias,
                            void* stream) {
    SC_CHECK(A && Bm && C, "sc_sgemm_f32: null pointer");
    SC_CHECK(M > 0 && N > 0 && K > 0, "sc_sgemm_f32: bad shape");
    dim3 grid((N + TS - 1) / TS, (M + TS - 1) / TS);
    hipLaunchKernelGGL(sgemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, sai, sak, Bm, sbj, sbk, C, ldc, M, N, K, alpha, bias);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_infonce_lse(const float* logits, const int64_t* ids, int32_t Bg, float* lse_row, float* lse_col,
                              float* loss, void* stream) {
    SC_CHECK(logits && lse_row && lse_col && loss, "sc_infonce_lse: null pointer");
    SC_CHECK(Bg > 0, "sc_infonce_lse: Bg");
    // loss[0] = scalar, loss[1 .. Bg] = per-sample terms (caller provides Bg + 1 floats)
    hipLaunchKernelGGL(infonce_lse_kernel, dim3(Bg), dim3(256), 0, (hipStream_t)stream, logits, ids, Bg, lse_row, lse_col, loss + 1);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss + 1, Bg, 0.5f / (float)Bg, loss);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_infonce_grad(const float* logits, const int64_t* ids, const float* lse_row, const float* lse_col,
                               int32_t Bg, const float* gscale, float* G, float* dlogit_dot, void* stream) {
    SC_CHECK(logits && lse_row && lse_col && gscale && G && dlogit_dot, "sc_infonce_grad: null pointer");
    hipLaunchKernelGGL(infonce_grad_kernel, dim3(Bg), dim3(256), 0, (hipStream_t)stream, logits, ids, lse_row, lse_col, Bg, gscale, G, dlogit_dot);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_sumsq_f32(const float* x, int64_t n, float* partial, int32_t nblk, void* stream) {
    SC_CHECK(x && partial && nblk > 0 && n > 0, "sc_sumsq_f32: bad args");
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, n, partial);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int32_t step, const float* gnorm_sq_partial, int32_t nblk,
                           float max_norm, void* stream) {
    SC_CHECK(p && g && m && v && n > 0 && step >= 1, "sc_adam_f32: bad args");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, gnorm_sq_partial, nblk, max_norm);
    SC_LAUNCH_CHECK();
    return 0;
}
