// fp32 kernels of the one-row-per-utterance tail of the parallel branch head (B x D matrices, B = per-GPU batch):
// batched strided SGEMM with accumulate, row LayerNorm forward / backward, erf-GELU forward / backward, column sums,
// per-head query masking.  Everything here is latency-bound (B <= a few hundred rows): the point of owning these is
// launch count and determinism (fixed summation order, no float atomics), not bandwidth.
//
// Mirrors nn.TransformerEncoderLayer (post-LN, GELU) + final LayerNorm + Linear as instantiated by
// avssl/module/kw_modules/TransformerModels.py:48-97 and consumed at avssl/model/kw_branches.py:266-280.
#include <algorithm>

#include "sc_common.h"

namespace {

// ---------------------------------------------------------------------------------------------- batched sgemm
// C[z][i, j] = alpha * sum_k A[z][i, k] * B[z][j, k] (+ bias[z][j]) + beta * C[z][i, j]
constexpr int TS = 64, TK = 16;

__global__ __launch_bounds__(256) void sgemm_ex_kernel(const float* __restrict__ A, int64_t sai, int64_t sak, int64_t saz,
                                                       const float* __restrict__ Bm, int64_t sbj, int64_t sbk, int64_t sbz,
                                                       float* __restrict__ C, int64_t ldc, int64_t scz, int M, int N, int K,
                                                       float alpha, float beta, const float* __restrict__ bias,
                                                       int64_t sbiasz, int S, int Kc, float* __restrict__ P) {
    __shared__ __attribute__((aligned(16))) float As[TK][TS + 4];
    __shared__ __attribute__((aligned(16))) float Bs[TK][TS + 4];
    // split-K: blockIdx.z = z * S + ks ; slice ks reduces k in [ks Kc, min(K, (ks + 1) Kc)) into the workspace P[ks][z][M][N]
    const int z = blockIdx.z / S, ks = blockIdx.z - z * S;
    A += z * saz + (int64_t)ks * Kc * sak;
    Bm += z * sbz + (int64_t)ks * Kc * sbk;
    C += z * scz;
    if (bias) bias += z * sbiasz;
    if (S > 1) K = min(Kc, K - ks * Kc);
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
    const int ty = tid >> 4, tx = tid & 15;    // 16 x 16 threads, 4 x 4 outputs each
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const bool a_kfast = (sak == 1), b_kfast = (sbk == 1);   // make the unit-stride dimension the fast thread index
    // one 16-byte load per thread and operand when the unit-stride dimension allows it (aligned base, stride and extent multiples
    // of 4): k-fast -> 4 consecutive k of one row, otherwise 4 consecutive rows of one k
    const bool a_vec = (K % 4 == 0) && (((uintptr_t)A % 16) == 0) &&
                       (a_kfast ? (sai % 4 == 0) : (sai == 1 && sak % 4 == 0 && M % 4 == 0));
    const bool b_vec = (K % 4 == 0) && (((uintptr_t)Bm % 16) == 0) &&
                       (b_kfast ? (sbj % 4 == 0) : (sbj == 1 && sbk % 4 == 0 && N % 4 == 0));
    for (int k0 = 0; k0 < K; k0 += TK) {
        if (a_vec) {
            if (a_kfast) {
                const int kk = (tid & 3) * 4, ii = tid >> 2, gi = i0 + ii, gk = k0 + kk;
                const f32x4 v = (gi < M && gk < K) ? *(const f32x4*)(A + gi * sai + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
                As[kk][ii] = v[0]; As[kk + 1][ii] = v[1]; As[kk + 2][ii] = v[2]; As[kk + 3][ii] = v[3];
            } else {
                const int ii = (tid & 15) * 4, kk = tid >> 4, gi = i0 + ii, gk = k0 + kk;
                const f32x4 v = (gi < M && gk < K) ? *(const f32x4*)(A + gi + gk * sak) : f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4*)&As[kk][ii] = v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int id = tid + e * 256;
                int ii, kk;
                if (a_kfast) { kk = id & 15; ii = id >> 4; } else { ii = id & 63; kk = id >> 6; }
                const int gi = i0 + ii, gk = k0 + kk;
                As[kk][ii] = (gi < M && gk < K) ? A[gi * sai + gk * sak] : 0.f;
            }
        }
        if (b_vec) {
            if (b_kfast) {
                const int kb = (tid & 3) * 4, jj = tid >> 2, gj = j0 + jj, gkb = k0 + kb;
                const f32x4 v = (gj < N && gkb < K) ? *(const f32x4*)(Bm + gj * sbj + gkb) : f32x4{0.f, 0.f, 0.f, 0.f};
                Bs[kb][jj] = v[0]; Bs[kb + 1][jj] = v[1]; Bs[kb + 2][jj] = v[2]; Bs[kb + 3][jj] = v[3];
            } else {
                const int jj = (tid & 15) * 4, kb = tid >> 4, gj = j0 + jj, gkb = k0 + kb;
                const f32x4 v = (gj < N && gkb < K) ? *(const f32x4*)(Bm + gj + gkb * sbk) : f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4*)&Bs[kb][jj] = v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int id = tid + e * 256;
                int jj, kb;
                if (b_kfast) { kb = id & 15; jj = id >> 4; } else { jj = id & 63; kb = id >> 6; }
                const int gj = j0 + jj, gkb = k0 + kb;
                Bs[kb][jj] = (gj < N && gkb < K) ? Bm[gj * sbj + gkb * sbk] : 0.f;
            }
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
            if (gj >= N) continue;
            if (S > 1) {
                P[(((int64_t)ks * gridDim.z / S + z) * M + gi) * N + gj] = acc[a][b];
                continue;
            }
            float v = alpha * acc[a][b] + (bias ? bias[gj] : 0.f);
            if (beta != 0.f) v += beta * C[gi * ldc + gj];
            C[gi * ldc + gj] = v;
        }
    }
}

// C[z][i, j] = alpha * sum_s P[s][z][i][j] (+ bias[z][j]) + beta * C[z][i, j]   (slices added in order: deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ P, int S, int nbatch, int M, int N, float* __restrict__ C,
                                     int64_t ldc, int64_t scz, float alpha, float beta, const float* __restrict__ bias,
                                     int64_t sbiasz) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, per = (int64_t)M * N, total = per * nbatch;
    if (e >= total) return;
    const int z = (int)(e / per);
    const int64_t r = e - z * per;
    const int i = (int)(r / N), j = (int)(r - (int64_t)i * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += P[(int64_t)s * total + e];
    v = alpha * v + (bias ? bias[z * sbiasz + j] : 0.f);
    float* c = C + z * scz + (int64_t)i * ldc + j;
    if (beta != 0.f) v += beta * *c;
    *c = v;
}

// ---------------------------------------------------------------------------------------------- row LayerNorm
// y = LN(x + res) * gamma + beta ; xhat, rstd kept for the backward.  res row stride 0 = one row broadcast (the CLS
// token is the same residual for every utterance).  One wave per row.
__global__ __launch_bounds__(256) void rowln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        int64_t res_stride, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ xhat, float* __restrict__ rstd, int rows, int D,
                                                        float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * D;
    const float* rr = res ? res + row * res_stride : nullptr;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s += xr[j] + (rr ? rr[j] : 0.f);
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int j = lane; j < D; j += 64) {
        const float d = xr[j] + (rr ? rr[j] : 0.f) - mean;
        v += d * d;
    }
    const float rs = rsqrtf(wave_sum(v) / (float)D + eps);
    if (lane == 0) rstd[row] = rs;
    for (int j = lane; j < D; j += 64) {
        const float h = (xr[j] + (rr ? rr[j] : 0.f) - mean) * rs;
        xhat[(int64_t)row * D + j] = h;
        y[(int64_t)row * D + j] = h * gamma[j] + beta[j];
    }
}

// blocks [0, ceil(rows / 4)): dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma  (one wave per row)
// blocks after: dgamma[j] += sum_rows dy xhat ; dbeta[j] += sum_rows dy   (one thread per column, rows in order)
__global__ __launch_bounds__(256) void rowln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                        const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                        float* __restrict__ dx, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int rows, int D, int row_blocks) {
    if ((int)blockIdx.x < row_blocks) {
        const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= rows) return;
        const float* dr = dy + (int64_t)row * D;
        const float* hr = xhat + (int64_t)row * D;
        float s1 = 0.f, s2 = 0.f;
        for (int j = lane; j < D; j += 64) {
            const float g = dr[j] * gamma[j];
            s1 += g;
            s2 += g * hr[j];
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        const float rs = rstd[row];
        for (int j = lane; j < D; j += 64) dx[(int64_t)row * D + j] = rs * (dr[j] * gamma[j] - s1 - hr[j] * s2);
    } else {
        const int j = (blockIdx.x - row_blocks) * 256 + threadIdx.x;
        if (j >= D) return;
        float sg = 0.f, sb = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float d = dy[(int64_t)r * D + j];
            sg += d * xhat[(int64_t)r * D + j];
            sb += d;
        }
        dgamma[j] += sg;
        dbeta[j] += sb;
    }
}

// ---------------------------------------------------------------------------------------------- GELU (libm erf: exact form)
__global__ void gelu_fwd_kernel(const float* __restrict__ u, float* __restrict__ f, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float x = u[i];
        f[i] = 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ df, float* __restrict__ du, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float x = u[i];
        const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
        du[i] = df[i] * (cdf + x * pdf);
    }
}

// out[j] = beta * out[j] + alpha * sum_i x[i * ld + j]      (bias gradients, split-K slices, LayerNorm parameter partials)
// Many rows: block = CW columns x (1024 / CW) row groups; group g adds rows g, g + G, ... in order (four independent chains, joined in
// order), the group sums are added in order.  CW = 16 for tall, narrow inputs (the [1024, 768] LayerNorm partials of a differentiated
// layer: 48 workgroups x 16 rows per thread instead of 12 x 64 - 20 -> 6 us), 64 otherwise.
template <int CW>
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, int64_t ld, int rows, int cols,
                                                      float* __restrict__ out, float alpha, float beta) {
    constexpr int G = 1024 / CW;
    __shared__ float part[G][CW];
    const int c = threadIdx.x % CW, g = threadIdx.x / CW;
    const int64_t j = (int64_t)blockIdx.x * CW + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < cols) {
        int r = g;
        for (; r + 3 * G < rows; r += 4 * G) {
            s0 += x[(int64_t)r * ld + j];
            s1 += x[(int64_t)(r + G) * ld + j];
            s2 += x[(int64_t)(r + 2 * G) * ld + j];
            s3 += x[(int64_t)(r + 3 * G) * ld + j];
        }
        for (; r < rows; r += G) s0 += x[(int64_t)r * ld + j];
    }
    part[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && j < cols) {
        float t = part[0][c];
#pragma unroll 8
        for (int k = 1; k < G; ++k) t += part[k][c];
        out[j] = (beta != 0.f ? beta * out[j] : 0.f) + alpha * t;
    }
}
// Few rows (split-K slices over millions of columns): one thread per 4 consecutive columns, rows added in order.
__global__ __launch_bounds__(256) void colsum_few_rows_kernel(const float* __restrict__ x, int64_t ld, int rows, int64_t cols4,
                                                              float* __restrict__ out, float alpha, float beta) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= cols4) return;
    f32x4 s = *(const f32x4*)(x + j * 4);
    for (int r = 1; r < rows; ++r) {
        const f32x4 v = *(const f32x4*)(x + (int64_t)r * ld + j * 4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    f32x4 o = f32x4{alpha * s[0], alpha * s[1], alpha * s[2], alpha * s[3]};
    if (beta != 0.f) {
        const f32x4 p = *(const f32x4*)(out + j * 4);
        o[0] += beta * p[0]; o[1] += beta * p[1]; o[2] += beta * p[2]; o[3] += beta * p[3];
    }
    *(f32x4*)(out + j * 4) = o;
}

// dir 0: Qm[h, j] = q[j] if j / dh == h else 0  ([H, D] from [D]);  dir 1: q[j] = Qm[j / dh, j]
__global__ void headmask_kernel(float* __restrict__ q, float* __restrict__ Qm, int H, int D, int dh, int dir) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    if (dir == 0) {
        const float v = q[j];
        for (int h = 0; h < H; ++h) Qm[(int64_t)h * D + j] = (j / dh == h) ? v : 0.f;
    } else {
        q[j] = Qm[(int64_t)(j / dh) * D + j];
    }
}

}  // namespace

extern "C" int sc_sgemm_f32_ex(const float* A, int64_t sai, int64_t sak, int64_t saz, const float* Bm, int64_t sbj,
                               int64_t sbk, int64_t sbz, float* C, int64_t ldc, int64_t scz, int32_t M, int32_t N, int32_t K,
                               int32_t nbatch, float alpha, float beta, const float* bias, int64_t sbiasz, float* workspace,
                               int64_t workspace_floats, void* stream) {
    SC_CHECK(A && Bm && C, "sc_sgemm_f32_ex: null pointer");
    SC_CHECK(M > 0 && N > 0 && K > 0 && nbatch > 0, "sc_sgemm_f32_ex: bad shape M=%d N=%d K=%d batch=%d", M, N, K, nbatch);
    // few-row products (M = per-GPU batch) give only a handful of 64 x 64 tiles with a long serial K loop: split K over
    // enough workgroups to cover the chip, partial sums through the caller's workspace, slices reduced in order
    const int tiles = ((N + TS - 1) / TS) * ((M + TS - 1) / TS) * nbatch;
    int S = 1;
    if (workspace && tiles < 128 && K >= 256) {
        S = std::min((sc_num_cus() + tiles - 1) / tiles, K / 64);
        const int64_t per = (int64_t)M * N * nbatch;
        S = (int)std::min<int64_t>(S, workspace_floats / per);
        if (S < 2) S = 1;
    }
    int Kc = K;
    if (S > 1) {
        Kc = ((K + S - 1) / S + TK - 1) / TK * TK;
        S = (K + Kc - 1) / Kc;
    }
    dim3 grid((N + TS - 1) / TS, (M + TS - 1) / TS, nbatch * S);
    hipLaunchKernelGGL(sgemm_ex_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, sai, sak, saz, Bm, sbj, sbk, sbz, C, ldc, scz,
                       M, N, K, alpha, beta, bias, sbiasz, S, Kc, workspace);
    SC_LAUNCH_CHECK();
    if (S > 1) {
        const int64_t total = (int64_t)M * N * nbatch;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace,
                           S, nbatch, M, N, C, ldc, scz, alpha, beta, bias, sbiasz);
        SC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sc_rowln_f32_fwd(const float* x, const float* res, int64_t res_stride, const float* gamma, const float* beta,
                                float* y, float* xhat, float* rstd, int32_t rows, int32_t D, float eps, void* stream) {
    SC_CHECK(x && gamma && beta && y && xhat && rstd, "sc_rowln_f32_fwd: null pointer");
    SC_CHECK(rows > 0 && D > 0, "sc_rowln_f32_fwd: bad shape");
    hipLaunchKernelGGL(rowln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, res, res_stride, gamma, beta,
                       y, xhat, rstd, rows, D, eps);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rowln_f32_bwd(const float* dy, const float* xhat, const float* gamma, const float* rstd, float* dx,
                                float* dgamma_acc, float* dbeta_acc, int32_t rows, int32_t D, void* stream) {
    SC_CHECK(dy && xhat && gamma && rstd && dx && dgamma_acc && dbeta_acc, "sc_rowln_f32_bwd: null pointer");
    SC_CHECK(rows > 0 && D > 0, "sc_rowln_f32_bwd: bad shape");
    const int row_blocks = (rows + 3) / 4, col_blocks = (D + 255) / 256;
    hipLaunchKernelGGL(rowln_bwd_kernel, dim3(row_blocks + col_blocks), dim3(256), 0, (hipStream_t)stream, dy, xhat, gamma, rstd,
                       dx, dgamma_acc, dbeta_acc, rows, D, row_blocks);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_gelu_f32(const float* u, const float* df, float* out, int64_t n, void* stream) {
    SC_CHECK(u && out && n > 0, "sc_gelu_f32: bad args");
    const int grid = (int)((n + 255) / 256);
    if (df) hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u, df, out, n);
    else hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u, out, n);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_colsum_f32(const float* x, int64_t ld, int32_t rows, int32_t cols, float* out, float alpha, float beta,
                             void* stream) {
    SC_CHECK(x && out && rows > 0 && cols > 0, "sc_colsum_f32: bad args");
    // (up to 32 rows: the 28 split-K slices of a 768 x 768 weight gradient - 16-byte loads, 26.7 -> ~12 us)
    if (rows <= 32 && cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0) {
        const int64_t cols4 = cols / 4;
        hipLaunchKernelGGL(colsum_few_rows_kernel, dim3((unsigned)((cols4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, rows,
                           cols4, out, alpha, beta);
    } else {
        if (rows >= 128 && cols <= 8192)
            hipLaunchKernelGGL(colsum_kernel<16>, dim3((cols + 15) / 16), dim3(1024), 0, (hipStream_t)stream, x, ld, rows, cols, out, alpha, beta);
        else
            hipLaunchKernelGGL(colsum_kernel<64>, dim3((cols + 63) / 64), dim3(1024), 0, (hipStream_t)stream, x, ld, rows, cols, out, alpha, beta);
    }
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_headmask_f32(float* q, float* Qm, int32_t H, int32_t D, int32_t dh, int32_t dir, void* stream) {
    SC_CHECK(q && Qm && H > 0 && D > 0 && dh > 0 && H * dh == D, "sc_headmask_f32: bad args");
    hipLaunchKernelGGL(headmask_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, Qm, H, D, dh, dir);
    SC_LAUNCH_CHECK();
    return 0;
}
