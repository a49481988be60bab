// HBM-bound row kernels: LayerNorm, weighted sum over hidden states (fwd/bwd), pos_conv input regroup.
// One wave per row, 8-byte (4 x bf16) vector accesses, wavefront reductions (no LDS).
#include "sc_common.h"

namespace {

// ---------------------------------------------------------------------------------------- LayerNorm
template <int NCH>   // 4-element chunks per lane: D = 256 * NCH at most
__global__ __launch_bounds__(256) void layernorm_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, uint16_t* __restrict__ y,
                                                        int64_t ldy, int64_t rows, int D, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = D >> 2;
    float v[NCH][4];
    float sum = 0.f;
    const uint16_t* xr = x + row * ldx;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
            const uint2 u = *(const uint2*)(xr + ch * 4);
            v[i][0] = bflo(u.x); v[i][1] = bfhi(u.x); v[i][2] = bflo(u.y); v[i][3] = bfhi(u.y);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    uint16_t* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
            const f32x4 g = *(const f32x4*)(gamma + ch * 4);
            const f32x4 bt = *(const f32x4*)(beta + ch * 4);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + bt[j];
            if (act == 1) {
                const f32x2 g0 = gelu_bf2(f32x2{o[0], o[1]}), g1 = gelu_bf2(f32x2{o[2], o[3]});
                o[0] = g0.x; o[1] = g0.y; o[2] = g1.x; o[3] = g1.y;
            }
            uint2 w;
            w.x = pack2bf(o[0], o[1]);
            w.y = pack2bf(o[2], o[3]);
            *(uint2*)(yr + ch * 4) = w;
        }
    }
}

// Half a wave per row, 16-byte accesses (8 x bf16 per lane per chunk): half the load / store instructions of the kernel above
// and one shuffle step less per reduction.  D % 8 == 0, D <= 1024, rows 16-byte aligned.  Same arithmetic order per element
// (fp32 two-pass mean / variance); the reduction tree differs, so results match the kernel above to fp32 round-off, not bitwise.
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int NCH8>   // 8-element chunks per lane: D <= 256 * NCH8
__global__ __launch_bounds__(256) void layernorm16_kernel(const uint16_t* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, uint16_t* __restrict__ y, int64_t ldy,
                                                          int64_t rows, int D, float eps, int act) {
    const int l = threadIdx.x & 31;
    int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = row < rows;
    if (!live) row = rows - 1;                    // keep the half-wave in the shuffles; its stores are skipped
    const int nchunks = D >> 3;
    float v[NCH8][8];
    float sum = 0.f;
    const uint16_t* xr = x + row * ldx;
#pragma unroll
    for (int i = 0; i < NCH8; ++i) {
        const int ch = l + i * 32;
        if (ch < nchunks) {
            const uint4 u = *(const uint4*)(xr + ch * 8);
            v[i][0] = bflo(u.x); v[i][1] = bfhi(u.x); v[i][2] = bflo(u.y); v[i][3] = bfhi(u.y);
            v[i][4] = bflo(u.z); v[i][5] = bfhi(u.z); v[i][6] = bflo(u.w); v[i][7] = bfhi(u.w);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = half_wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH8; ++i)
        if (l + i * 32 < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    const float rstd = rsqrtf(half_wave_sum(sq) / (float)D + eps);
    if (!live) return;
    uint16_t* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < NCH8; ++i) {
        const int ch = l + i * 32;
        if (ch < nchunks) {
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 g = *(const f32x4*)(gamma + ch * 8 + 4 * h);
                const f32x4 bt = *(const f32x4*)(beta + ch * 8 + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[4 * h + j] = (v[i][4 * h + j] - mean) * rstd * g[j] + bt[j];
            }
            if (act == 1) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x2 gg = gelu_bf2(f32x2{o[j], o[j + 1]});
                    o[j] = gg.x; o[j + 1] = gg.y;
                }
            }
            uint4 w;
            w.x = pack2bf(o[0], o[1]); w.y = pack2bf(o[2], o[3]); w.z = pack2bf(o[4], o[5]); w.w = pack2bf(o[6], o[7]);
            *(uint4*)(yr + ch * 8) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------- weighted sum
// out[b, t + row_off, :] = sum_n w[n] * h[n, b, t, :]   (t + row_off < R)
__global__ __launch_bounds__(256) void wsum_fwd_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w,
                                                       int NL, uint16_t* __restrict__ out, int B, int R, int D,
                                                       int row_off) {
    const int64_t chunks_per_row = D >> 3;
    const int64_t total = (int64_t)B * R * chunks_per_row;
    const int64_t plane = (int64_t)B * R * D;
    float wl[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) wl[n] = n < NL ? w[n] : 0.f;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = q / chunks_per_row;
        const int cc = (int)(q % chunks_per_row);
        const int t = (int)(row % R);
        if (t + row_off >= R) continue;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint16_t* src = h + row * D + cc * 8;
#pragma unroll 4
        for (int n = 0; n < NL; ++n) {
            const uint4 u = *(const uint4*)(src + n * plane);
            const float wn = wl[n];
            acc[0] += wn * bflo(u.x); acc[1] += wn * bfhi(u.x); acc[2] += wn * bflo(u.y); acc[3] += wn * bfhi(u.y);
            acc[4] += wn * bflo(u.z); acc[5] += wn * bfhi(u.z); acc[6] += wn * bflo(u.w); acc[7] += wn * bfhi(u.w);
        }
        uint4 o;
        o.x = pack2bf(acc[0], acc[1]); o.y = pack2bf(acc[2], acc[3]);
        o.z = pack2bf(acc[4], acc[5]); o.w = pack2bf(acc[6], acc[7]);
        *(uint4*)(out + (row + row_off) * D + cc * 8) = o;
    }
}

// 8 consecutive gradient elements as fp32: the gradient arrives either as fp32 (autograd's default) or as bf16 rows (the attention
// block of the cascaded+/hybrid+ branches hands its input gradient over as it leaves the GEMM: no activation-sized cast in between)
template <typename GT>
__device__ __forceinline__ void load_g8(const GT* gp, f32x4& g0, f32x4& g1) {
    if constexpr (sizeof(GT) == 4) {
        g0 = *(const f32x4*)gp;
        g1 = *(const f32x4*)(gp + 4);
    } else {
        const uint4 u = *(const uint4*)gp;
        g0 = f32x4{bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y)};
        g1 = f32x4{bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
    }
}

// dw_partial[blk, n] = sum over this block's elements of g[b, t + row_off, d] * (h[n, b, t, d] - h[NL - 1, b, t, d])
// The caller only uses the softmax-projected combination w_n (d_n - sum_m w_m d_m), which is invariant under a common shift of
// the d_n: subtracting the LAST layer element-wise BEFORE the accumulation removes the large common part <g, h> that the
// projection would cancel afterwards (the differences between layers of a residual stream are one to two orders of magnitude
// smaller than the states themselves: summing first and subtracting later costs that many digits of the fp32 accumulators).
template <typename GT>
__global__ __launch_bounds__(256) void wsum_bwd_kernel(const uint16_t* __restrict__ h, const GT* __restrict__ g,
                                                       int NL, float* __restrict__ dw_partial, int B, int R, int D,
                                                       int row_off) {
    __shared__ float red[4][32];
    const int64_t chunks_per_row = D >> 3;
    const int64_t total = (int64_t)B * R * chunks_per_row;
    const int64_t plane = (int64_t)B * R * D;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.f;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = q / chunks_per_row;
        const int cc = (int)(q % chunks_per_row);
        const int t = (int)(row % R);
        if (t + row_off >= R) continue;
        f32x4 g0, g1;
        load_g8(g + (row + row_off) * D + cc * 8, g0, g1);
        const uint16_t* src = h + row * D + cc * 8;
        const uint4 r = *(const uint4*)(src + (int64_t)(NL - 1) * plane);
#pragma unroll
        for (int n = 0; n < 31; ++n) {
            if (n < NL - 1) {
                const uint4 u = *(const uint4*)(src + n * plane);
                acc[n] += g0[0] * (bflo(u.x) - bflo(r.x)) + g0[1] * (bfhi(u.x) - bfhi(r.x)) + g0[2] * (bflo(u.y) - bflo(r.y)) +
                          g0[3] * (bfhi(u.y) - bfhi(r.y)) + g1[0] * (bflo(u.z) - bflo(r.z)) + g1[1] * (bfhi(u.z) - bfhi(r.z)) +
                          g1[2] * (bflo(u.w) - bflo(r.w)) + g1[3] * (bfhi(u.w) - bfhi(r.w));
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        const float s = wave_sum(acc[n]);
        if (lane == 0) red[wave][n] = s;
    }
    __syncthreads();
    if (threadIdx.x < NL)
        dw_partial[(int64_t)blockIdx.x * NL + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------- weighted sum over RAW hidden states
// Round 3 (LayerNorm folded into the encoder GEMMs, csrc/gemm256_bf16.hip "LN"): layers n >= first_lazy of h hold the RAW rows in
// front of the layer's final LayerNorm, with their row statistics in stats[n][row][8][2] (ns valid strips of (sum, sum of
// squares)); the hidden state the reference sums is LN(raw) = (raw - mean) rstd gamma_n + beta_n, evaluated here in fp32.
struct LazyLn {
    const float* stats;      // [NL][B*R][8][2]
    const float* gamma;      // [NL][D]
    const float* beta;       // [NL][D]
    int first_lazy, ns;
    float eps;
};

__device__ __forceinline__ void lazy_row_stats(const LazyLn& z, int n, int64_t rows_total, int64_t row, int D, float& mean, float& rstd) {
    const float* sp = z.stats + ((int64_t)n * rows_total + row) * 16;
    float s1 = 0.f, s2 = 0.f;
    for (int st = 0; st < z.ns; ++st) {
        const f32x2 t = *(const f32x2*)(sp + 2 * st);
        s1 += t.x;
        s2 += t.y;
    }
    const float inv = 1.f / (float)D;
    mean = s1 * inv;
    rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + z.eps);
}

__device__ __forceinline__ void lazy_load8(const LazyLn& z, const uint16_t* __restrict__ src, int n, int64_t rows_total, int64_t row, int D,
                                           int cc, float (&x)[8]) {
    const uint4 u = *(const uint4*)src;
    x[0] = bflo(u.x); x[1] = bfhi(u.x); x[2] = bflo(u.y); x[3] = bfhi(u.y);
    x[4] = bflo(u.z); x[5] = bfhi(u.z); x[6] = bflo(u.w); x[7] = bfhi(u.w);
    if (n >= z.first_lazy) {
        float mean, rstd;
        lazy_row_stats(z, n, rows_total, row, D, mean, rstd);
        const float* gp = z.gamma + (int64_t)n * D + cc * 8;
        const float* bp = z.beta + (int64_t)n * D + cc * 8;
        const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4), b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[e] = fmaf((x[e] - mean) * rstd, g0[e], b0[e]);
            x[4 + e] = fmaf((x[4 + e] - mean) * rstd, g1[e], b1[e]);
        }
    }
}

__global__ __launch_bounds__(256) void wsum_lazy_fwd_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w, int NL,
                                                            uint16_t* __restrict__ out, int B, int R, int D, int row_off, LazyLn z) {
    const int64_t chunks_per_row = D >> 3;
    const int64_t rows_total = (int64_t)B * R;
    const int64_t total = rows_total * chunks_per_row;
    const int64_t plane = rows_total * D;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = q / chunks_per_row;
        const int cc = (int)(q % chunks_per_row);
        const int t = (int)(row % R);
        if (t + row_off >= R) continue;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint16_t* src = h + row * D + cc * 8;
        for (int n = 0; n < NL; ++n) {
            float x[8];
            lazy_load8(z, src + n * plane, n, rows_total, row, D, cc, x);
            const float wn = w[n];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(wn, x[e], acc[e]);
        }
        uint4 o;
        o.x = pack2bf(acc[0], acc[1]); o.y = pack2bf(acc[2], acc[3]);
        o.z = pack2bf(acc[4], acc[5]); o.w = pack2bf(acc[6], acc[7]);
        *(uint4*)(out + (row + row_off) * D + cc * 8) = o;
    }
}

// as wsum_bwd_kernel (differences against the last layer, see there), hidden states evaluated lazily
__global__ __launch_bounds__(256) void wsum_lazy_bwd_kernel(const uint16_t* __restrict__ h, const float* __restrict__ g, int NL,
                                                            float* __restrict__ dw_partial, int B, int R, int D, int row_off, LazyLn z) {
    __shared__ float red[4][32];
    const int64_t chunks_per_row = D >> 3;
    const int64_t rows_total = (int64_t)B * R;
    const int64_t total = rows_total * chunks_per_row;
    const int64_t plane = rows_total * D;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.f;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = q / chunks_per_row;
        const int cc = (int)(q % chunks_per_row);
        const int t = (int)(row % R);
        if (t + row_off >= R) continue;
        f32x4 g0, g1;
        load_g8(g + (row + row_off) * D + cc * 8, g0, g1);
        const uint16_t* src = h + row * D + cc * 8;
        float r[8];
        lazy_load8(z, src + (int64_t)(NL - 1) * plane, NL - 1, rows_total, row, D, cc, r);
#pragma unroll
        for (int n = 0; n < 31; ++n) {
            if (n < NL - 1) {
                float x[8];
                lazy_load8(z, src + n * plane, n, rows_total, row, D, cc, x);
                acc[n] += g0[0] * (x[0] - r[0]) + g0[1] * (x[1] - r[1]) + g0[2] * (x[2] - r[2]) + g0[3] * (x[3] - r[3]) +
                          g1[0] * (x[4] - r[4]) + g1[1] * (x[5] - r[5]) + g1[2] * (x[6] - r[6]) + g1[3] * (x[7] - r[7]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        const float sm = wave_sum(acc[n]);
        if (lane == 0) red[wave][n] = sm;
    }
    __syncthreads();
    if (threadIdx.x < NL)
        dw_partial[(int64_t)blockIdx.x * NL + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------- weighted sum, normalised
// normalize_features = True (HuBERT-large recipes): out = sum_n w[n] * LayerNorm_noaffine(h[n, row, :]).
// One wave per row (D <= 1024 -> <= 16 elements per lane held in registers), wavefront reductions per layer.
template <int NE>   // 8-element chunks per lane
__device__ __forceinline__ void load_row_norm(const uint16_t* __restrict__ src, int lane, int nchunks, int D, float eps,
                                              float (&xh)[NE][8]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
            const uint4 u = *(const uint4*)(src + ch * 8);
            xh[i][0] = bflo(u.x); xh[i][1] = bfhi(u.x); xh[i][2] = bflo(u.y); xh[i][3] = bfhi(u.y);
            xh[i][4] = bflo(u.z); xh[i][5] = bfhi(u.z); xh[i][6] = bflo(u.w); xh[i][7] = bfhi(u.w);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += xh[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xh[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[i][j] -= mean;
                sq += xh[i][j] * xh[i][j];
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) xh[i][j] *= rstd;
}

template <int NE>
__global__ __launch_bounds__(256) void wsum_norm_fwd_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w,
                                                            int NL, uint16_t* __restrict__ out, int B, int R, int D,
                                                            int row_off) {
    const int lane = threadIdx.x & 63;
    const int nchunks = D >> 3;
    const int64_t rows = (int64_t)B * R, plane = rows * D;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        if ((int)(row % R) + row_off >= R) continue;
        float acc[NE][8];
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        for (int n = 0; n < NL; ++n) {
            float xh[NE][8];
            load_row_norm<NE>(h + n * plane + row * D, lane, nchunks, D, 1e-5f, xh);
            const float wn = w[n];
#pragma unroll
            for (int i = 0; i < NE; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] += wn * xh[i][j];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunks) {
                uint4 o;
                o.x = pack2bf(acc[i][0], acc[i][1]); o.y = pack2bf(acc[i][2], acc[i][3]);
                o.z = pack2bf(acc[i][4], acc[i][5]); o.w = pack2bf(acc[i][6], acc[i][7]);
                *(uint4*)(out + (row + row_off) * D + ch * 8) = o;
            }
        }
    }
}

template <int NE, typename GT>
__global__ __launch_bounds__(256) void wsum_norm_bwd_kernel(const uint16_t* __restrict__ h, const GT* __restrict__ g,
                                                            int NL, float* __restrict__ dw_partial, int B, int R, int D,
                                                            int row_off) {
    __shared__ float red[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = D >> 3;
    const int64_t rows = (int64_t)B * R, plane = rows * D;
    float accn[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) accn[n] = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        if ((int)(row % R) + row_off >= R) continue;
        float gv[NE][8];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunks) {
                f32x4 g0, g1;
                load_g8(g + (row + row_off) * D + ch * 8, g0, g1);
                gv[i][0] = g0[0]; gv[i][1] = g0[1]; gv[i][2] = g0[2]; gv[i][3] = g0[3];
                gv[i][4] = g1[0]; gv[i][5] = g1[1]; gv[i][6] = g1[2]; gv[i][7] = g1[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[i][j] = 0.f;
            }
        }
        float xr[NE][8];                 // reference layer (the last one): see wsum_bwd_kernel
        load_row_norm<NE>(h + (int64_t)(NL - 1) * plane + row * D, lane, nchunks, D, 1e-5f, xr);
#pragma unroll
        for (int n = 0; n < 31; ++n) {
            if (n < NL - 1) {
                float xh[NE][8];
                load_row_norm<NE>(h + n * plane + row * D, lane, nchunks, D, 1e-5f, xh);
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < NE; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += gv[i][j] * (xh[i][j] - xr[i][j]);
                accn[n] += d;        // per-lane partial; reduced over the wave once at the end
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        const float s = wave_sum(accn[n]);
        if (lane == 0) red[wave][n] = s;
    }
    __syncthreads();
    if (threadIdx.x < NL)
        dw_partial[(int64_t)blockIdx.x * NL + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------- pos_conv prep
// xz[b*R + t, :] = t < valid[b] ? x : 0 ;  xg[g][b][halo + t][0:Dg] = same, group-major with zero halos.
__global__ __launch_bounds__(256) void posconv_prep_kernel(const uint16_t* __restrict__ x,
                                                           const int32_t* __restrict__ valid_len,
                                                           uint16_t* __restrict__ xz, uint16_t* __restrict__ xg,
                                                           int B, int R, int D, int G, int halo) {
    const int Dg = D / G;
    const int cpg = Dg >> 3;               // 16-B chunks per group slice (Dg % 8 == 0)
    const int64_t chunks_per_row = D >> 3;
    const int64_t total = (int64_t)B * R * chunks_per_row;
    const int Rp = R + 2 * halo;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = q / chunks_per_row;
        const int cc = (int)(q % chunks_per_row);
        const int b = (int)(row / R), t = (int)(row % R);
        uint4 u = make_uint4(0, 0, 0, 0);
        if (t < valid_len[b]) u = *(const uint4*)(x + row * D + cc * 8);
        *(uint4*)(xz + row * D + cc * 8) = u;
        const int g = cc / cpg, cg = cc % cpg;
        *(uint4*)(xg + (((int64_t)g * B + b) * Rp + halo + t) * Dg + cg * 8) = u;
    }
}


// ---------------------------------------------------------------------------------------- ragged rows (sc_segments, round 4)
// Weighted sum from the segment layout (utterance b's frames at rows row0[b] + t of every h[n]) into a UNIFORM [B, Rout, D] buffer:
// out[b, s] = sum_n w[n] h[n, row0[b] + s - row_off] for 0 <= s - row_off < pitch_b, zero elsewhere (every row of out is written).
__global__ __launch_bounds__(256) void wsum_fwd_seg_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w, int NL,
                                                           uint16_t* __restrict__ out, const int32_t* __restrict__ row0, int B, int Rout,
                                                           int D, int row_off, int64_t plane) {
    const int64_t chunks_per_row = D >> 3;
    const int64_t total = (int64_t)B * Rout * chunks_per_row;
    float wl[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) wl[n] = n < NL ? w[n] : 0.f;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t orow = q / chunks_per_row;
        const int cc = (int)(q % chunks_per_row);
        const int b = (int)(orow / Rout), t = (int)(orow % Rout) - row_off;
        const int r0 = row0[b], pitch = row0[b + 1] - r0;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t >= 0 && t < pitch) {
            const uint16_t* src = h + (int64_t)(r0 + t) * D + cc * 8;
#pragma unroll 4
            for (int n = 0; n < NL; ++n) {
                const uint4 u = *(const uint4*)(src + n * plane);
                const float wn = wl[n];
                acc[0] += wn * bflo(u.x); acc[1] += wn * bfhi(u.x); acc[2] += wn * bflo(u.y); acc[3] += wn * bfhi(u.y);
                acc[4] += wn * bflo(u.z); acc[5] += wn * bfhi(u.z); acc[6] += wn * bflo(u.w); acc[7] += wn * bfhi(u.w);
            }
        }
        uint4 o;
        o.x = pack2bf(acc[0], acc[1]); o.y = pack2bf(acc[2], acc[3]);
        o.z = pack2bf(acc[4], acc[5]); o.w = pack2bf(acc[6], acc[7]);
        *(uint4*)(out + orow * D + cc * 8) = o;
    }
}

// The same sum for a layer count known at compile time (13 / 25: base / large), one 16-byte chunk per thread (round 5): grid (chunks of an
// utterance's Rout rows / 256, B) - no 64-bit divisions, no grid-stride loop -, the NLT weights in registers under constant indices (the
// generic kernel above indexes its weight array with a loop variable: v_movrel sequences), all NLT loads of a chunk in flight before
// the first multiply-add.  Same additions in the same order: same bits.
template <int NLT>
__global__ __launch_bounds__(256) void wsum_fwd_seg_fixed_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w,
                                                                 uint16_t* __restrict__ out, const int32_t* __restrict__ row0, int Rout,
                                                                 int D, int row_off, int64_t plane) {
    const int cpr = D >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Rout * cpr) return;
    const int b = blockIdx.y;
    const int s = idx / cpr, cc = idx - s * cpr, t = s - row_off;
    const int r0 = row0[b], pitch = row0[b + 1] - r0;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t >= 0 && t < pitch) {
        const uint16_t* src = h + (int64_t)(r0 + t) * D + cc * 8;
        uint4 u[NLT];
#pragma unroll
        for (int n = 0; n < NLT; ++n) u[n] = *(const uint4*)(src + n * plane);
#pragma unroll
        for (int n = 0; n < NLT; ++n) {
            const float wn = w[n];
            acc[0] += wn * bflo(u[n].x); acc[1] += wn * bfhi(u[n].x); acc[2] += wn * bflo(u[n].y); acc[3] += wn * bfhi(u[n].y);
            acc[4] += wn * bflo(u[n].z); acc[5] += wn * bfhi(u[n].z); acc[6] += wn * bflo(u[n].w); acc[7] += wn * bfhi(u[n].w);
        }
    }
    uint4 o;
    o.x = pack2bf(acc[0], acc[1]); o.y = pack2bf(acc[2], acc[3]);
    o.z = pack2bf(acc[4], acc[5]); o.w = pack2bf(acc[6], acc[7]);
    *(uint4*)(out + ((int64_t)b * Rout + s) * D + cc * 8) = o;
}

// as wsum_bwd_kernel, g in the uniform [B, Rout, D] layout, h in the segment layout
template <typename GT>
__global__ __launch_bounds__(256) void wsum_bwd_seg_kernel(const uint16_t* __restrict__ h, const GT* __restrict__ g, int NL,
                                                           float* __restrict__ dw_partial, const int32_t* __restrict__ row0, int B, int Rout,
                                                           int D, int row_off, int64_t plane) {
    __shared__ float red[4][32];
    const int64_t chunks_per_row = D >> 3;
    const int64_t total = (int64_t)B * Rout * chunks_per_row;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.f;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t orow = q / chunks_per_row;
        const int cc = (int)(q % chunks_per_row);
        const int b = (int)(orow / Rout), t = (int)(orow % Rout) - row_off;
        const int r0 = row0[b], pitch = row0[b + 1] - r0;
        if (t < 0 || t >= pitch) continue;
        f32x4 g0, g1;
        load_g8(g + orow * D + cc * 8, g0, g1);
        const uint16_t* src = h + (int64_t)(r0 + t) * D + cc * 8;
        const uint4 r = *(const uint4*)(src + (int64_t)(NL - 1) * plane);
#pragma unroll
        for (int n = 0; n < 31; ++n) {
            if (n < NL - 1) {
                const uint4 u = *(const uint4*)(src + n * plane);
                acc[n] += g0[0] * (bflo(u.x) - bflo(r.x)) + g0[1] * (bfhi(u.x) - bfhi(r.x)) + g0[2] * (bflo(u.y) - bflo(r.y)) +
                          g0[3] * (bfhi(u.y) - bfhi(r.y)) + g1[0] * (bflo(u.z) - bflo(r.z)) + g1[1] * (bfhi(u.z) - bfhi(r.z)) +
                          g1[2] * (bflo(u.w) - bflo(r.w)) + g1[3] * (bfhi(u.w) - bfhi(r.w));
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        const float s = wave_sum(acc[n]);
        if (lane == 0) red[wave][n] = s;
    }
    __syncthreads();
    if (threadIdx.x < NL)
        dw_partial[(int64_t)blockIdx.x * NL + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// wsum_bwd_seg_kernel for a layer count known at compile time (round 5): the same elements per thread in the same order (grid-stride over the
// 16-byte chunks of the uniform gradient), but 32-bit index arithmetic (the generic kernel spends two 64-bit divisions per chunk), every load of a
// chunk in flight before the first use, and NLT - 1 wavefront reductions instead of 32.
template <int NLT, typename GT>
__global__ __launch_bounds__(256) void wsum_bwd_seg_fixed_kernel(const uint16_t* __restrict__ h, const GT* __restrict__ g,
                                                                 float* __restrict__ dw_partial, const int32_t* __restrict__ row0, int B, int Rout,
                                                                 int D, int row_off, int64_t plane) {
    __shared__ float red[4][32];
    const uint32_t cpr = (uint32_t)D >> 3, total = (uint32_t)B * (uint32_t)Rout * cpr, step = gridDim.x * 256u;
    float acc[NLT - 1];
#pragma unroll
    for (int n = 0; n < NLT - 1; ++n) acc[n] = 0.f;
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < total; q += step) {
        const uint32_t orow = q / cpr, cc = q - orow * cpr;
        const uint32_t b = orow / (uint32_t)Rout;
        const int t = (int)(orow - b * (uint32_t)Rout) - row_off;
        const int r0 = row0[b], pitch = row0[b + 1] - r0;
        if (t < 0 || t >= pitch) continue;
        f32x4 g0, g1;
        load_g8(g + (int64_t)orow * D + cc * 8, g0, g1);
        const uint16_t* src = h + (int64_t)(r0 + t) * D + cc * 8;
        const uint4 r = *(const uint4*)(src + (int64_t)(NLT - 1) * plane);
        uint4 u[NLT - 1];
#pragma unroll
        for (int n = 0; n < NLT - 1; ++n) u[n] = *(const uint4*)(src + n * plane);
#pragma unroll
        for (int n = 0; n < NLT - 1; ++n)
            acc[n] += g0[0] * (bflo(u[n].x) - bflo(r.x)) + g0[1] * (bfhi(u[n].x) - bfhi(r.x)) + g0[2] * (bflo(u[n].y) - bflo(r.y)) +
                      g0[3] * (bfhi(u[n].y) - bfhi(r.y)) + g1[0] * (bflo(u[n].z) - bflo(r.z)) + g1[1] * (bfhi(u[n].z) - bfhi(r.z)) +
                      g1[2] * (bflo(u[n].w) - bflo(r.w)) + g1[3] * (bfhi(u[n].w) - bfhi(r.w));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < NLT - 1; ++n) {
        const float sn = wave_sum(acc[n]);
        if (lane == 0) red[wave][n] = sn;
    }
    if (lane == 0) red[wave][NLT - 1] = 0.f;          // the last layer's entry: zero (everything is relative to it)
    __syncthreads();
    if (threadIdx.x < NLT)
        dw_partial[(int64_t)blockIdx.x * NLT + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <int NE>
__global__ __launch_bounds__(256) void wsum_norm_fwd_seg_kernel(const uint16_t* __restrict__ h, const float* __restrict__ w, int NL,
                                                                uint16_t* __restrict__ out, const int32_t* __restrict__ row0, int B, int Rout,
                                                                int D, int row_off, int64_t plane) {
    const int lane = threadIdx.x & 63;
    const int nchunks = D >> 3;
    const int64_t orows = (int64_t)B * Rout;
    for (int64_t orow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); orow < orows; orow += (int64_t)gridDim.x * 4) {
        const int b = (int)(orow / Rout), t = (int)(orow % Rout) - row_off;
        const int r0 = row0[b], pitch = row0[b + 1] - r0;
        float acc[NE][8];
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        if (t >= 0 && t < pitch) {
            for (int n = 0; n < NL; ++n) {
                float xh[NE][8];
                load_row_norm<NE>(h + n * plane + (int64_t)(r0 + t) * D, lane, nchunks, D, 1e-5f, xh);
                const float wn = w[n];
#pragma unroll
                for (int i = 0; i < NE; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] += wn * xh[i][j];
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunks) {
                uint4 o;
                o.x = pack2bf(acc[i][0], acc[i][1]); o.y = pack2bf(acc[i][2], acc[i][3]);
                o.z = pack2bf(acc[i][4], acc[i][5]); o.w = pack2bf(acc[i][6], acc[i][7]);
                *(uint4*)(out + orow * D + ch * 8) = o;
            }
        }
    }
}

template <int NE, typename GT>
__global__ __launch_bounds__(256) void wsum_norm_bwd_seg_kernel(const uint16_t* __restrict__ h, const GT* __restrict__ g, int NL,
                                                                float* __restrict__ dw_partial, const int32_t* __restrict__ row0, int B,
                                                                int Rout, int D, int row_off, int64_t plane) {
    __shared__ float red[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = D >> 3;
    const int64_t orows = (int64_t)B * Rout;
    float accn[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) accn[n] = 0.f;
    for (int64_t orow = (int64_t)blockIdx.x * 4 + wave; orow < orows; orow += (int64_t)gridDim.x * 4) {
        const int b = (int)(orow / Rout), t = (int)(orow % Rout) - row_off;
        const int r0 = row0[b], pitch = row0[b + 1] - r0;
        if (t < 0 || t >= pitch) continue;
        const int64_t row = r0 + t;
        float gv[NE][8];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunks) {
                f32x4 g0, g1;
                load_g8(g + orow * D + ch * 8, g0, g1);
                gv[i][0] = g0[0]; gv[i][1] = g0[1]; gv[i][2] = g0[2]; gv[i][3] = g0[3];
                gv[i][4] = g1[0]; gv[i][5] = g1[1]; gv[i][6] = g1[2]; gv[i][7] = g1[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[i][j] = 0.f;
            }
        }
        float xr[NE][8];
        load_row_norm<NE>(h + (int64_t)(NL - 1) * plane + row * D, lane, nchunks, D, 1e-5f, xr);
#pragma unroll
        for (int n = 0; n < 31; ++n) {
            if (n < NL - 1) {
                float xh[NE][8];
                load_row_norm<NE>(h + n * plane + row * D, lane, nchunks, D, 1e-5f, xh);
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < NE; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += gv[i][j] * (xh[i][j] - xr[i][j]);
                accn[n] += d;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        const float s = wave_sum(accn[n]);
        if (lane == 0) red[wave][n] = s;
    }
    __syncthreads();
    if (threadIdx.x < NL)
        dw_partial[(int64_t)blockIdx.x * NL + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// pos_conv input on ragged rows: xz[row] = masked x[row]; slab xg[g][row0[b] + 2 halo b + halo + t] = the same.  The slab layout moves
// with the batch's lengths, so the halo rows are re-zeroed here: the rows t < halo of an utterance also clear its leading halo, the
// last `halo` rows its trailing one (strided by the pitch when an utterance is shorter than a halo).
__global__ __launch_bounds__(256) void posconv_prep_seg_kernel(const uint16_t* __restrict__ x, const int32_t* __restrict__ valid_len,
                                                               uint16_t* __restrict__ xz, uint16_t* __restrict__ xg,
                                                               const int32_t* __restrict__ chunk, int rows, int B, int D, int G, int halo) {
    const int Dg = D / G;
    const int cpg = Dg >> 3;
    const int64_t chunks_per_row = D >> 3;
    const int64_t total = (int64_t)rows * chunks_per_row;
    const int64_t slab_rows = (int64_t)rows + 2 * halo * B;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(q / chunks_per_row);
        const int cc = (int)(q % chunks_per_row);
        const int4 sg = *(const int4*)(chunk + 4 * (row >> 3));       // (first row, pitch, utterance, -)
        const int t = row - sg.x, pitch = sg.y, b = sg.z;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (t < valid_len[b]) u = *(const uint4*)(x + (int64_t)row * D + cc * 8);
        *(uint4*)(xz + (int64_t)row * D + cc * 8) = u;
        const int g = cc / cpg, cg = cc % cpg;
        uint16_t* slab = xg + ((int64_t)g * slab_rows + sg.x + 2 * halo * b) * Dg + cg * 8;     // leading halo row 0 of (g, b)
        *(uint4*)(slab + (int64_t)(halo + t) * Dg) = u;
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int v = t; v < halo; v += pitch) {
            *(uint4*)(slab + (int64_t)v * Dg) = z;
            *(uint4*)(slab + (int64_t)(halo + pitch + v) * Dg) = z;
        }
    }
}

}  // namespace

extern "C" int sc_layernorm_bf16(const sc_bf16* x, int64_t ldx, const float* gamma, const float* beta, sc_bf16* y,
                                 int64_t ldy, int64_t rows, int32_t D, float eps, int32_t act, void* stream) {
    SC_CHECK(x && gamma && beta && y, "sc_layernorm_bf16: null pointer");
    SC_CHECK(D > 0 && D % 4 == 0 && D <= 1024, "sc_layernorm_bf16: D=%d must be a multiple of 4, <= 1024", D);
    SC_CHECK(ldx % 4 == 0 && ldy % 4 == 0 && rows > 0, "sc_layernorm_bf16: bad ld/rows");
    SC_CHECK(((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                 ((uintptr_t)beta % 16) == 0, "sc_layernorm_bf16: alignment");
    hipStream_t s = (hipStream_t)stream;
    if (D % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0) {
        // 16-byte path: half a wave per row, 8 rows per workgroup
        dim3 grid8((unsigned)((rows + 7) / 8));
        const int nch8 = (D / 8 + 31) / 32;
        if (nch8 <= 2) hipLaunchKernelGGL(layernorm16_kernel<2>, grid8, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
        else if (nch8 == 3) hipLaunchKernelGGL(layernorm16_kernel<3>, grid8, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
        else hipLaunchKernelGGL(layernorm16_kernel<4>, grid8, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
        SC_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)((rows + 3) / 4));
    const int nch = (D / 4 + 63) / 64;
    if (nch <= 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
    else if (nch == 3) hipLaunchKernelGGL(layernorm_kernel<3>, grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
    else hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, D, eps, act);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wsum_fwd(const sc_bf16* h, const float* w, int32_t NL, sc_bf16* out, int32_t B, int32_t R, int32_t D,
                           int32_t row_off, int32_t normalize, void* stream) {
    SC_CHECK(h && w && out, "sc_wsum_fwd: null pointer");
    SC_CHECK(NL >= 1 && NL <= 32 && D % 8 == 0 && row_off >= 0 && row_off < R, "sc_wsum_fwd: bad NL/D/row_off");
    if (normalize) {
        SC_CHECK(D <= 1024, "sc_wsum_fwd: normalised variant needs D <= 1024 (got %d)", D);
        const int64_t rows = (int64_t)B * R;
        const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
        if (D <= 512) hipLaunchKernelGGL(wsum_norm_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w, NL, out, B, R, D, row_off);
        else hipLaunchKernelGGL(wsum_norm_fwd_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w, NL, out, B, R, D, row_off);
        SC_LAUNCH_CHECK();
        return 0;
    }
    const int64_t total = (int64_t)B * R * (D / 8);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wsum_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w, NL, out, B, R, D, row_off);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wsum_bwd(const sc_bf16* h, const void* gv, int32_t NL, float* dw_partial, int32_t nblk, int32_t B,
                           int32_t R, int32_t D, int32_t row_off, int32_t flags, void* stream) {
    SC_CHECK(h && gv && dw_partial, "sc_wsum_bwd: null pointer");
    SC_CHECK(NL >= 1 && NL <= 32 && D % 8 == 0 && nblk >= 1 && row_off >= 0 && row_off < R, "sc_wsum_bwd: bad args");
    SC_CHECK(((uintptr_t)gv % 16) == 0, "sc_wsum_bwd: g must be 16-byte aligned");
    const bool normalize = flags & 1, g16 = flags & 2;         // bit 1: g is bf16
    const float* g = (const float*)gv;
    const uint16_t* gh = (const uint16_t*)gv;
    hipStream_t s = (hipStream_t)stream;
    if (normalize) {
        SC_CHECK(D <= 1024, "sc_wsum_bwd: normalised variant needs D <= 1024 (got %d)", D);
        if (D <= 512) {
            if (g16) hipLaunchKernelGGL((wsum_norm_bwd_kernel<1, uint16_t>), dim3(nblk), dim3(256), 0, s, h, gh, NL, dw_partial, B, R, D, row_off);
            else hipLaunchKernelGGL((wsum_norm_bwd_kernel<1, float>), dim3(nblk), dim3(256), 0, s, h, g, NL, dw_partial, B, R, D, row_off);
        } else {
            if (g16) hipLaunchKernelGGL((wsum_norm_bwd_kernel<2, uint16_t>), dim3(nblk), dim3(256), 0, s, h, gh, NL, dw_partial, B, R, D, row_off);
            else hipLaunchKernelGGL((wsum_norm_bwd_kernel<2, float>), dim3(nblk), dim3(256), 0, s, h, g, NL, dw_partial, B, R, D, row_off);
        }
        SC_LAUNCH_CHECK();
        return 0;
    }
    if (g16) hipLaunchKernelGGL(wsum_bwd_kernel<uint16_t>, dim3(nblk), dim3(256), 0, s, h, gh, NL, dw_partial, B, R, D, row_off);
    else hipLaunchKernelGGL(wsum_bwd_kernel<float>, dim3(nblk), dim3(256), 0, s, h, g, NL, dw_partial, B, R, D, row_off);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wsum_lazy_fwd(const sc_bf16* h, const float* w, int32_t NL, sc_bf16* out, int32_t B, int32_t R, int32_t D,
                                int32_t row_off, const float* stats, const float* gamma, const float* beta, int32_t first_lazy,
                                int32_t ns, float eps, void* stream) {
    SC_CHECK(h && w && out && stats && gamma && beta, "sc_wsum_lazy_fwd: null pointer");
    SC_CHECK(NL >= 1 && NL <= 32 && D % 8 == 0 && row_off >= 0 && row_off < R && ns >= 1 && ns <= 8 && first_lazy >= 0 && eps > 0.f,
             "sc_wsum_lazy_fwd: bad NL / D / row_off / strips");
    const LazyLn z{stats, gamma, beta, first_lazy, ns, eps};
    const int64_t total = (int64_t)B * R * (D / 8);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wsum_lazy_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w, NL, out, B, R, D, row_off, z);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wsum_lazy_bwd(const sc_bf16* h, const float* g, int32_t NL, float* dw_partial, int32_t nblk, int32_t B, int32_t R,
                                int32_t D, int32_t row_off, const float* stats, const float* gamma, const float* beta,
                                int32_t first_lazy, int32_t ns, float eps, void* stream) {
    SC_CHECK(h && g && dw_partial && stats && gamma && beta, "sc_wsum_lazy_bwd: null pointer");
    SC_CHECK(NL >= 1 && NL <= 32 && D % 8 == 0 && nblk >= 1 && row_off >= 0 && row_off < R && ns >= 1 && ns <= 8 && eps > 0.f,
             "sc_wsum_lazy_bwd: bad args");
    const LazyLn z{stats, gamma, beta, first_lazy, ns, eps};
    hipLaunchKernelGGL(wsum_lazy_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, h, g, NL, dw_partial, B, R, D, row_off, z);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_posconv_prep(const sc_bf16* x, const int32_t* valid_len, sc_bf16* xz, sc_bf16* xg, int32_t B,
                               int32_t R, int32_t D, int32_t G, int32_t halo, void* stream) {
    SC_CHECK(x && valid_len && xz && xg, "sc_posconv_prep: null pointer");
    SC_CHECK(G > 0 && D % G == 0 && (D / G) % 8 == 0, "sc_posconv_prep: D/G=%d must be a multiple of 8", D / G);
    const int64_t total = (int64_t)B * R * (D / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(posconv_prep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, valid_len, xz, xg, B, R, D, G, halo);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wsum_fwd_seg(const sc_bf16* h, const float* w, int32_t NL, sc_bf16* out, const sc_segments* seg, int32_t Rout, int32_t D,
                               int32_t row_off, int32_t normalize, void* stream) {
    SC_CHECK(h && w && out && seg && seg->row0, "sc_wsum_fwd_seg: null pointer");
    SC_CHECK(NL >= 1 && NL <= 32 && D % 8 == 0 && row_off >= 0 && Rout > row_off && seg->B > 0 && seg->rows > 0, "sc_wsum_fwd_seg: bad NL/D/row_off");
    const int B = seg->B;
    const int64_t plane = (int64_t)seg->rows * D;
    hipStream_t s = (hipStream_t)stream;
    if (normalize) {
        SC_CHECK(D <= 1024, "sc_wsum_fwd_seg: normalised variant needs D <= 1024 (got %d)", D);
        const int64_t rows = (int64_t)B * Rout;
        const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
        if (D <= 512) hipLaunchKernelGGL(wsum_norm_fwd_seg_kernel<1>, dim3(grid), dim3(256), 0, s, h, w, NL, out, seg->row0, B, Rout, D, row_off, plane);
        else hipLaunchKernelGGL(wsum_norm_fwd_seg_kernel<2>, dim3(grid), dim3(256), 0, s, h, w, NL, out, seg->row0, B, Rout, D, row_off, plane);
        SC_LAUNCH_CHECK();
        return 0;
    }
    const int64_t total = (int64_t)B * Rout * (D / 8);
    if ((NL == 13 || NL == 25) && (int64_t)Rout * (D / 8) < (1ll << 30) && B <= 65535 && !sc_option(5)) {      // option 5: A/B switch (tools/)
        const dim3 g2((unsigned)(((int64_t)Rout * (D / 8) + 255) / 256), (unsigned)B);
        if (NL == 13) hipLaunchKernelGGL(wsum_fwd_seg_fixed_kernel<13>, g2, dim3(256), 0, s, h, w, out, seg->row0, Rout, D, row_off, plane);
        else hipLaunchKernelGGL(wsum_fwd_seg_fixed_kernel<25>, g2, dim3(256), 0, s, h, w, out, seg->row0, Rout, D, row_off, plane);
        SC_LAUNCH_CHECK();
        return 0;
    }
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wsum_fwd_seg_kernel, dim3(grid), dim3(256), 0, s, h, w, NL, out, seg->row0, B, Rout, D, row_off, plane);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wsum_bwd_seg(const sc_bf16* h, const void* gv, int32_t NL, float* dw_partial, int32_t nblk, const sc_segments* seg,
                               int32_t Rout, int32_t D, int32_t row_off, int32_t flags, void* stream) {
    SC_CHECK(h && gv && dw_partial && seg && seg->row0, "sc_wsum_bwd_seg: null pointer");
    SC_CHECK(((uintptr_t)gv % 16) == 0, "sc_wsum_bwd_seg: g must be 16-byte aligned");
    const bool normalize = flags & 1, g16 = flags & 2;         // bit 1: g is bf16
    const float* g = (const float*)gv;
    const uint16_t* gh = (const uint16_t*)gv;
    SC_CHECK(NL >= 1 && NL <= 32 && D % 8 == 0 && nblk >= 1 && row_off >= 0 && Rout > row_off && seg->B > 0 && seg->rows > 0, "sc_wsum_bwd_seg: bad args");
    const int B = seg->B;
    const int64_t plane = (int64_t)seg->rows * D;
    hipStream_t s = (hipStream_t)stream;
    if (normalize) {
        SC_CHECK(D <= 1024, "sc_wsum_bwd_seg: normalised variant needs D <= 1024 (got %d)", D);
        if (D <= 512) {
            if (g16) hipLaunchKernelGGL((wsum_norm_bwd_seg_kernel<1, uint16_t>), dim3(nblk), dim3(256), 0, s, h, gh, NL, dw_partial, seg->row0, B, Rout, D, row_off, plane);
            else hipLaunchKernelGGL((wsum_norm_bwd_seg_kernel<1, float>), dim3(nblk), dim3(256), 0, s, h, g, NL, dw_partial, seg->row0, B, Rout, D, row_off, plane);
        } else {
            if (g16) hipLaunchKernelGGL((wsum_norm_bwd_seg_kernel<2, uint16_t>), dim3(nblk), dim3(256), 0, s, h, gh, NL, dw_partial, seg->row0, B, Rout, D, row_off, plane);
            else hipLaunchKernelGGL((wsum_norm_bwd_seg_kernel<2, float>), dim3(nblk), dim3(256), 0, s, h, g, NL, dw_partial, seg->row0, B, Rout, D, row_off, plane);
        }
        SC_LAUNCH_CHECK();
        return 0;
    }
    if ((NL == 13 || NL == 25) && (int64_t)B * Rout * (D / 8) < (1ll << 31) && !sc_option(5)) {      // option 5: A/B switch (tools/)
        if (NL == 13) {
            if (g16) hipLaunchKernelGGL((wsum_bwd_seg_fixed_kernel<13, uint16_t>), dim3(nblk), dim3(256), 0, s, h, gh, dw_partial, seg->row0, B, Rout, D, row_off, plane);
            else hipLaunchKernelGGL((wsum_bwd_seg_fixed_kernel<13, float>), dim3(nblk), dim3(256), 0, s, h, g, dw_partial, seg->row0, B, Rout, D, row_off, plane);
        } else {
            if (g16) hipLaunchKernelGGL((wsum_bwd_seg_fixed_kernel<25, uint16_t>), dim3(nblk), dim3(256), 0, s, h, gh, dw_partial, seg->row0, B, Rout, D, row_off, plane);
            else hipLaunchKernelGGL((wsum_bwd_seg_fixed_kernel<25, float>), dim3(nblk), dim3(256), 0, s, h, g, dw_partial, seg->row0, B, Rout, D, row_off, plane);
        }
        SC_LAUNCH_CHECK();
        return 0;
    }
    if (g16) hipLaunchKernelGGL(wsum_bwd_seg_kernel<uint16_t>, dim3(nblk), dim3(256), 0, s, h, gh, NL, dw_partial, seg->row0, B, Rout, D, row_off, plane);
    else hipLaunchKernelGGL(wsum_bwd_seg_kernel<float>, dim3(nblk), dim3(256), 0, s, h, g, NL, dw_partial, seg->row0, B, Rout, D, row_off, plane);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_posconv_prep_seg(const sc_bf16* x, const int32_t* valid_len, sc_bf16* xz, sc_bf16* xg, const sc_segments* seg, int32_t D,
                                   int32_t G, int32_t halo, void* stream) {
    SC_CHECK(x && valid_len && xz && xg && seg && seg->chunk, "sc_posconv_prep_seg: null pointer");
    SC_CHECK(G > 0 && D % G == 0 && (D / G) % 8 == 0 && seg->rows > 0 && seg->rows % SC_SEG_ROWS == 0 && seg->B > 0 && halo > 0,
             "sc_posconv_prep_seg: D/G=%d must be a multiple of 8, rows a multiple of 8", D / G);
    SC_CHECK(((uintptr_t)seg->chunk % 16) == 0, "sc_posconv_prep_seg: chunk table must be 16-byte aligned");
    const int64_t total = (int64_t)seg->rows * (D / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(posconv_prep_seg_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, valid_len, xz, xg, seg->chunk, seg->rows, seg->B, D, G, halo);
    SC_LAUNCH_CHECK();
    return 0;
}
