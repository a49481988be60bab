// Round 3: the one-row-per-utterance tail of the parallel head (B x D matrices, B = per-GPU batch <= a few hundred rows) in ~40
// launches per train step instead of ~125.  fp32 throughout, on the master weights.
//
//   sc_rt_gemm   skinny fp32 GEMM on the matrix pipe (v_mfma_f32_32x32x2_f32, exact fp32): 64 x 64 output tiles, 4 waves of 32 x 32,
//                K-steps of 16 through LDS.  Few rows mean few tiles, so the contraction is SPLIT over blockIdx.z and every slice
//                writes its own partial tile [S][M][N]; no reduce launch follows - the CONSUMER adds the slices in order when it
//                loads them (this kernel's A prologue, or a row kernel below), together with the bias the producer could not add.
//                Un-split products finish in the epilogue instead: alpha, bias, erf-GELU (pre-activation kept for the backward),
//                hash dropout, accumulate (beta).  Weight-gradient form (A and B contraction-major over the batch rows): the bias
//                gradient (column sums of A) is a by-product of the tiles with blockIdx.x == 0.
//   sc_rt_ln_fwd / _bwd   row LayerNorm over slices: z = (sum_s y_s + bias) * dropout + residual -> LN [-> second LN] ; backward with the
//                residual gradient added and the parameter gradients accumulated by the column blocks of the same launch.
//   sc_rt_l2norm_fwd / _bwd   unit rows (the two embedding matrices in front of the loss) and its backward.
//
// Mirrors nn.TransformerEncoderLayer (post-LN, GELU) + final LayerNorm + Linear as instantiated by
// avssl/module/kw_modules/TransformerModels.py:48-97 and consumed at avssl/model/kw_branches.py:266-280; the L2 normalisation is
// avssl/model/kwClip.py:857,913-915.
#include <algorithm>

#include "sc_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int RT = 64, RTK = 32, RLD = RT + 4;

struct RtGemm {
    const float* A; int64_t lda, a_slice, a_z; int a_kmajor, a_ns;
    const float* a_bias;            // [K] added to the summed A slices (row-major A only)
    const float* a_rowscale;        // [M][G]: a_bias[k] is multiplied by a_rowscale[row][k / a_group] (NULL: 1)
    int a_group, a_nscale;
    const float* B; int64_t ldb, b_z; int b_kmajor;
    float* C; int64_t ldc, c_slice, c_z;
    float* U;                       // pre-activation copy (act = 1), same layout as C, or NULL
    int M, N, K, S, Kc;
    float alpha, beta;
    const float* bias; int64_t bias_z;
    int act;
    float drop_scale; uint32_t drop_thr, drop_seed;
    float* gb; int64_t gb_z;        // bias gradient (+= column sums of A over the contraction), a_kmajor form, S == 1
};

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// element (row, col) kept?  Same bits as sc_dropout_mult_f32 / sc_keep8 on the row-major index row * N + col.
__device__ __forceinline__ bool rt_keep(uint32_t idx, uint32_t seed, uint32_t thr) {
    return (sc_keep8(idx & ~7u, seed, thr) >> (idx & 7u)) & 1u;
}

__global__ __launch_bounds__(256) void rt_gemm_kernel(const RtGemm p) {
    __shared__ __attribute__((aligned(16))) float As[2][RTK][RLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][RTK][RLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const int z = blockIdx.z / p.S, ks = blockIdx.z - z * p.S;
    const int m0 = blockIdx.y * RT, n0 = blockIdx.x * RT;
    const float* A = p.A + z * p.a_z;
    const float* Bm = p.B + z * p.b_z;
    const int k_begin = ks * p.Kc, k_end = min(p.K, k_begin + p.Kc);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float gsum = 0.f;               // bias-gradient partial (thread = column of A), first N-tile only

    // One RTK x 64 operand slab per K-step, NV = RTK / 16 pieces of 4 consecutive elements per thread along the operand's unit-stride
    // dimension.  Interior, 16-byte aligned tiles take ONE 16-byte load per piece; edges fall back to guarded scalar loads.
    constexpr int NV = RTK / 16;
    const bool a_vec = (p.lda % 4 == 0) && (p.a_slice % 4 == 0) && (((uintptr_t)A) % 16 == 0) && m0 + RT <= p.M &&
                       (p.a_kmajor || !p.a_bias);
    const bool b_vec = (p.ldb % 4 == 0) && (((uintptr_t)Bm) % 16 == 0) && n0 + RT <= p.N;
    auto load_a = [&](int k0, f32x4 (&v)[NV]) {
        const bool full = k0 + RTK <= k_end;
        if (p.a_kmajor) {           // A[k][m]: 4 consecutive m of one k
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int kk = (tid >> 4) + 16 * i, mm = (tid & 15) * 4, k = k0 + kk;
                if (a_vec && full) {
                    v[i] = *(const f32x4*)(A + (int64_t)k * p.lda + m0 + mm);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = m0 + mm + e;
                        v[i][e] = (k < k_end && m < p.M) ? A[(int64_t)k * p.lda + m] : 0.f;
                    }
                }
            }
        } else {                    // A[m][k] (+ slices, bias): 4 consecutive k of one row
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int mm = tid >> 2, kk = (tid & 3) * 4 + 16 * i, m = m0 + mm;
                if (a_vec && full) {
                    f32x4 t = *(const f32x4*)(A + (int64_t)m * p.lda + k0 + kk);
                    for (int sl = 1; sl < p.a_ns; ++sl) t += *(const f32x4*)(A + sl * p.a_slice + (int64_t)m * p.lda + k0 + kk);
                    v[i] = t;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = k0 + kk + e;
                        float sm = 0.f;
                        if (m < p.M && k < k_end) {
                            for (int sl = 0; sl < p.a_ns; ++sl) sm += A[sl * p.a_slice + (int64_t)m * p.lda + k];
                            if (p.a_bias) sm += p.a_bias[k] * (p.a_rowscale ? p.a_rowscale[(int64_t)m * p.a_nscale + k / p.a_group] : 1.f);
                        }
                        v[i][e] = sm;
                    }
                }
            }
        }
    };
    auto stage_a = [&](const f32x4 (&v)[NV], float (*dst)[RLD]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (p.a_kmajor) {
                *(f32x4*)&dst[(tid >> 4) + 16 * i][(tid & 15) * 4] = v[i];
            } else {
                const int mm = tid >> 2, kk = (tid & 3) * 4 + 16 * i;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[kk + e][mm] = v[i][e];
            }
        }
    };
    auto load_b = [&](int k0, f32x4 (&v)[NV]) {
        const bool full = k0 + RTK <= k_end;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (p.b_kmajor) {       // B[k][n]
                const int kk = (tid >> 4) + 16 * i, nn = (tid & 15) * 4, k = k0 + kk;
                if (b_vec && full) {
                    v[i] = *(const f32x4*)(Bm + (int64_t)k * p.ldb + n0 + nn);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = n0 + nn + e;
                        v[i][e] = (k < k_end && n < p.N) ? Bm[(int64_t)k * p.ldb + n] : 0.f;
                    }
                }
            } else {                // B[n][k]
                const int nn = tid >> 2, kk = (tid & 3) * 4 + 16 * i, n = n0 + nn;
                if (b_vec && full) {
                    v[i] = *(const f32x4*)(Bm + (int64_t)n * p.ldb + k0 + kk);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = k0 + kk + e;
                        v[i][e] = (n < p.N && k < k_end) ? Bm[(int64_t)n * p.ldb + k] : 0.f;
                    }
                }
            }
        }
    };
    auto stage_b = [&](const f32x4 (&v)[NV], float (*dst)[RLD]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (p.b_kmajor) {
                *(f32x4*)&dst[(tid >> 4) + 16 * i][(tid & 15) * 4] = v[i];
            } else {
                const int nn = tid >> 2, kk = (tid & 3) * 4 + 16 * i;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[kk + e][nn] = v[i][e];
            }
        }
    };

    f32x4 ra[NV], rb[NV];
    load_a(k_begin, ra);
    load_b(k_begin, rb);
    stage_a(ra, As[0]);
    stage_b(rb, Bs[0]);
    __syncthreads();
    const int nk = (k_end - k_begin + RTK - 1) / RTK;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nk;
        if (more) {
            load_a(k_begin + (t + 1) * RTK, ra);
            load_b(k_begin + (t + 1) * RTK, rb);
        }
#pragma unroll
        for (int kk = 0; kk < RTK; kk += 2) {
            const int kr = kk + half;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[cur][kr][wm * 32 + l31], Bs[cur][kr][wn * 32 + l31], acc, 0, 0, 0);
        }
        if (p.gb && blockIdx.x == 0 && tid < RT) {
#pragma unroll
            for (int kk = 0; kk < RTK; ++kk) gsum += As[cur][kk][tid];
        }
        if (more) {
            stage_a(ra, As[cur ^ 1]);
            stage_b(rb, Bs[cur ^ 1]);
        }
        __syncthreads();
    }
    // accumulator element r of lane l: row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    const int col = n0 + wn * 32 + l31;
    const int rbase = m0 + wm * 32 + 4 * half;
    if (col < p.N) {
        if (p.S > 1) {
            float* C = p.C + z * p.c_z + (int64_t)ks * p.c_slice;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < p.M) C[(int64_t)row * p.ldc + col] = acc[r];
            }
        } else {
            float* C = p.C + z * p.c_z;
            const float bv = p.bias ? p.bias[z * p.bias_z + col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                float v = p.alpha * acc[r] + bv;
                if (p.act) {
                    if (p.U) p.U[z * p.c_z + (int64_t)row * p.ldc + col] = v;
                    v = gelu_exact(v);
                }
                if (p.drop_thr) v = rt_keep((uint32_t)row * (uint32_t)p.N + (uint32_t)col, p.drop_seed, p.drop_thr) ? v * p.drop_scale : 0.f;
                if (p.beta != 0.f) v += p.beta * C[(int64_t)row * p.ldc + col];
                C[(int64_t)row * p.ldc + col] = v;
            }
        }
    }
    if (p.gb && blockIdx.x == 0 && tid < RT && m0 + tid < p.M) p.gb[z * p.gb_z + m0 + tid] += gsum;
}

// ------------------------------------------------------------------------------------------------ row LayerNorm over slices
struct RtLn {
    const float* y; int64_t y_slice; int ns;      // [ns][rows][D] partial sums of the producer (added in order)
    const float* bias;                            // [D] or NULL
    float drop_scale; uint32_t drop_thr, drop_seed;   // multiplier on (sum + bias), hash index row * D + j
    const float* res; int64_t res_stride;         // residual rows (stride 0: one broadcast row) or NULL
    const float *g1, *b1; float eps1;
    float *out1, *xhat1, *rstd1;
    const float *g2, *b2; float eps2;             // optional second LayerNorm on out1 (g2 == NULL: none)
    float *out2, *xhat2, *rstd2;
    int rows, D;
};

// sum over the 256 threads of a block (4 waves), result on every thread
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                               // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ONE WORKGROUP PER ROW (B rows only: a wave per row would leave the chip to 16 waves and serialise D x slices loads per lane)
__global__ __launch_bounds__(256) void rt_ln_fwd_kernel(const RtLn p) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = tid + i * 256;
        float t = 0.f;
        if (j < p.D) {
            for (int sl = 0; sl < p.ns; ++sl) t += p.y[sl * p.y_slice + (int64_t)row * p.D + j];
            if (p.bias) t += p.bias[j];
            if (p.drop_thr) t = rt_keep((uint32_t)row * (uint32_t)p.D + (uint32_t)j, p.drop_seed, p.drop_thr) ? t * p.drop_scale : 0.f;
            if (p.res) t += p.res[row * p.res_stride + j];
        }
        v[i] = t;
        s += t;
    }
    auto norm = [&](const float* g, const float* b, float eps, float* out, float* xhat, float* rstd) {
        const float mean = block_sum256(s, red) / (float)p.D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * 256;
            const float d = j < p.D ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float rs = rsqrtf(block_sum256(q, red) / (float)p.D + eps);
        if (tid == 0) rstd[row] = rs;
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * 256;
            if (j < p.D) {
                const float h = (v[i] - mean) * rs;
                xhat[(int64_t)row * p.D + j] = h;
                v[i] = h * g[j] + b[j];
                out[(int64_t)row * p.D + j] = v[i];
                s += v[i];
            }
        }
    };
    norm(p.g1, p.b1, p.eps1, p.out1, p.xhat1, p.rstd1);
    if (p.g2) norm(p.g2, p.b2, p.eps2, p.out2, p.xhat2, p.rstd2);
}

struct RtLnBwd {
    const float* dy; int64_t dy_slice; int ns;    // [ns][rows][D] partial sums of the incoming gradient (added in order)
    const float* add;                             // [rows][D] added to the summed gradient (a second path into the same LN output) or NULL
    const float *xhat, *gamma, *rstd;
    float* dx;                                    // [rows][D]
    float drop_scale; uint32_t drop_thr, drop_seed;   // the forward's multiplier on the LN INPUT's producer term: dx_masked (see below)
    float* dx_masked;                             // optional second output: dx * mask (gradient of the producer term in front of the dropout)
    float *dgamma, *dbeta;                        // += over rows
    int rows, D, row_blocks;
};

// blocks [0, rows): one workgroup per row (the summed gradient stays in registers: slices are read once);
// blocks after: 32 columns x 8 row groups per workgroup, rows r = g, g + 8, ... in order, the eight group sums added in order
__global__ __launch_bounds__(256) void rt_ln_bwd_kernel(const RtLnBwd p) {
    __shared__ float red[4];
    __shared__ float part[2][8][32];
    if ((int)blockIdx.x < p.row_blocks) {
        const int row = blockIdx.x, tid = threadIdx.x;
        const float* hr = p.xhat + (int64_t)row * p.D;
        float gv[4], hv[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * 256;
            float g = 0.f, h = 0.f;
            if (j < p.D) {
                float d = 0.f;
                for (int sl = 0; sl < p.ns; ++sl) d += p.dy[sl * p.dy_slice + (int64_t)row * p.D + j];
                if (p.add) d += p.add[(int64_t)row * p.D + j];
                g = d * p.gamma[j];
                h = hr[j];
            }
            gv[i] = g;
            hv[i] = h;
            s1 += g;
            s2 += g * h;
        }
        s1 = block_sum256(s1, red) / (float)p.D;
        s2 = block_sum256(s2, red) / (float)p.D;
        const float rs = p.rstd[row];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * 256;
            if (j >= p.D) continue;
            const float dxv = rs * (gv[i] - s1 - hv[i] * s2);
            p.dx[(int64_t)row * p.D + j] = dxv;
            if (p.dx_masked)
                p.dx_masked[(int64_t)row * p.D + j] =
                    (!p.drop_thr || rt_keep((uint32_t)row * (uint32_t)p.D + (uint32_t)j, p.drop_seed, p.drop_thr)) ? dxv * p.drop_scale : 0.f;
        }
    } else {
        const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
        const int j = (blockIdx.x - p.row_blocks) * 32 + c;
        float sg = 0.f, sb = 0.f;
        if (j < p.D)
            for (int r = g; r < p.rows; r += 8) {
                float d = 0.f;
                for (int sl = 0; sl < p.ns; ++sl) d += p.dy[sl * p.dy_slice + (int64_t)r * p.D + j];
                if (p.add) d += p.add[(int64_t)r * p.D + j];
                sg += d * p.xhat[(int64_t)r * p.D + j];
                sb += d;
            }
        part[0][g][c] = sg;
        part[1][g][c] = sb;
        __syncthreads();
        if (g == 0 && j < p.D) {
            float tg = 0.f, tb = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                tg += part[0][k][c];
                tb += part[1][k][c];
            }
            p.dgamma[j] += tg;
            p.dbeta[j] += tb;
        }
    }
}

// ------------------------------------------------------------------------------------------------ elementwise over slices
// mode 0: out = sum_s y_s + bias[j] * (rowscale ? rowscale[row][j / group] : 1)
// mode 1: u = sum_s y_s + bias[j] ; f = gelu_erf(u) * dropout          (u and f kept for the backward)
// mode 2: du = (sum_s y_s) * dropout * gelu_erf'(u)
__global__ __launch_bounds__(256) void rt_elem_kernel(const float* __restrict__ y, int64_t y_slice, int ns, const float* __restrict__ bias,
                                                      const float* __restrict__ rowscale, int group, int nscale, float* __restrict__ out,
                                                      float* __restrict__ u, int rows, int D, int mode, float drop_scale, uint32_t drop_thr,
                                                      uint32_t drop_seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * D) return;
    const int row = (int)(i / D), j = (int)(i - (int64_t)row * D);
    float t = 0.f;
    for (int sl = 0; sl < ns; ++sl) t += y[sl * y_slice + i];
    const float mask = (!drop_thr || rt_keep((uint32_t)i, drop_seed, drop_thr)) ? drop_scale : 0.f;
    if (mode == 0) {
        if (bias) t += bias[j] * (rowscale ? rowscale[(int64_t)row * nscale + j / group] : 1.f);
        out[i] = t;
    } else if (mode == 1) {
        if (bias) t += bias[j];
        u[i] = t;
        out[i] = gelu_exact(t) * mask;
    } else {
        const float x = u[i];
        const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
        out[i] = t * mask * (cdf + x * pdf);
    }
}

// ------------------------------------------------------------------------------------------------ unit rows
// e = x / |x| with x = sum_s y_s + bias  (x itself is kept: encode_speech returns the un-normalised embedding)
__global__ __launch_bounds__(256) void rt_l2norm_fwd_kernel(const float* __restrict__ y, int64_t y_slice, int ns, const float* __restrict__ bias,
                                                            float* __restrict__ x, float* __restrict__ e, float* __restrict__ rnorm, int rows,
                                                            int D) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float v[4];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = tid + i * 256;
        float t = 0.f;
        if (j < D) {
            for (int sl = 0; sl < ns; ++sl) t += y[sl * y_slice + (int64_t)row * D + j];
            if (bias) t += bias[j];
            if (x) x[(int64_t)row * D + j] = t;
        }
        v[i] = t;
        q += t * t;
    }
    const float rn = 1.f / sqrtf(block_sum256(q, red));
    if (tid == 0) rnorm[row] = rn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = tid + i * 256;
        if (j < D) e[(int64_t)row * D + j] = v[i] * rn;
    }
}
// dx = (g - e <g, e>) / |x|
__global__ __launch_bounds__(256) void rt_l2norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ e, const float* __restrict__ rnorm,
                                                            float* __restrict__ dx, int rows, int D) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float d = 0.f;
    for (int j = tid; j < D; j += 256) d += g[(int64_t)row * D + j] * e[(int64_t)row * D + j];
    d = block_sum256(d, red);
    const float rn = rnorm[row];
    for (int j = tid; j < D; j += 256) dx[(int64_t)row * D + j] = (g[(int64_t)row * D + j] - e[(int64_t)row * D + j] * d) * rn;
}

// blocks [0, B): cbias[b, h] (wave h of 4, heads h, h + 4, ...) ; blocks after: gbv over 256 columns each
__global__ __launch_bounds__(256) void rt_value_bias_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ bv,
                                                                const float* __restrict__ psum, float* __restrict__ cbias,
                                                                float* __restrict__ gbv, int B, int D, int H) {
    const int dh = D / H;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int h = wave; h < H; h += 4) {
            float t = 0.f;
            for (int j = lane; j < dh; j += 64) t += dctx[(int64_t)b * D + h * dh + j] * bv[h * dh + j];
            t = wave_sum(t);
            if (lane == 0) cbias[(int64_t)b * H + h] = t;
        }
    } else {
        const int j = (blockIdx.x - B) * 256 + threadIdx.x;
        if (j >= D) return;
        const int h = j / dh;
        float t = 0.f;
        for (int b = 0; b < B; ++b) t += dctx[(int64_t)b * D + j] * psum[(int64_t)b * H + h];
        gbv[j] += t;
    }
}

__global__ __launch_bounds__(256) void rt_softmax_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int NL, const float* __restrict__ w,
                                                                   float* __restrict__ out) {
    // thread t adds the partials of blocks t, t + 256, ... for one layer at a time; the 256 sums of a layer are added by wave
    // reductions in a fixed order; then the softmax backward on NL values
    __shared__ float red[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int n = 0; n < NL; ++n) {
        float d = 0.f;
        for (int b = tid; b < nblk; b += 256) d += part[(int64_t)b * NL + n];
        d = wave_sum(d);
        if (lane == 0) red[wave][n] = d;
    }
    __syncthreads();
    if (tid < 64) {
        const float d = tid < NL ? (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]) : 0.f;
        const float wn = tid < NL ? w[tid] : 0.f;
        const float dot = wave_sum(wn * d);
        if (tid < NL) out[tid] = wn * (d - dot);
    }
}

}  // namespace

extern "C" int sc_rt_gemm(const sc_rt_gemm_args* a, void* stream) {
    SC_CHECK(a && a->A && a->B && a->C, "sc_rt_gemm: null pointer");
    SC_CHECK(a->M > 0 && a->N > 0 && a->K > 0 && a->nbatch > 0 && a->S >= 1 && a->a_ns >= 1, "sc_rt_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    SC_CHECK(a->S == 1 || (!a->bias && !a->act && a->beta == 0.f && a->alpha == 1.f && !a->gb && a->drop_p == 0.f),
             "sc_rt_gemm: a split product (S = %d) writes raw partial slices: no epilogue", a->S);
    SC_CHECK(!a->a_kmajor || (a->a_ns == 1 && !a->a_bias), "sc_rt_gemm: slices / bias on a row-major A only");
    SC_CHECK(!a->gb || a->a_kmajor, "sc_rt_gemm: the bias gradient is a by-product of the weight-gradient form (A contraction-major)");
    SC_CHECK(a->drop_p >= 0.f && a->drop_p < 1.f && (a->drop_p == 0.f || (int64_t)a->M * a->N < ((int64_t)1 << 32)), "sc_rt_gemm: drop_p");
    RtGemm p;
    p.A = a->A; p.lda = a->lda; p.a_slice = a->a_slice; p.a_z = a->a_z; p.a_kmajor = a->a_kmajor; p.a_ns = a->a_ns;
    p.a_bias = a->a_bias; p.a_rowscale = a->a_rowscale; p.a_group = a->a_group > 0 ? a->a_group : 1; p.a_nscale = a->a_nscale;
    p.B = a->B; p.ldb = a->ldb; p.b_z = a->b_z; p.b_kmajor = a->b_kmajor;
    p.C = a->C; p.ldc = a->ldc; p.c_slice = a->c_slice; p.c_z = a->c_z; p.U = a->U;
    p.M = a->M; p.N = a->N; p.K = a->K; p.S = a->S;
    p.Kc = ((a->K + a->S - 1) / a->S + RTK - 1) / RTK * RTK;
    SC_CHECK((int64_t)p.Kc * (a->S - 1) < a->K, "sc_rt_gemm: S = %d leaves an empty slice for K = %d", a->S, a->K);
    p.alpha = a->alpha; p.beta = a->beta; p.bias = a->bias; p.bias_z = a->bias_z; p.act = a->act;
    p.drop_thr = a->drop_p > 0.f ? (uint32_t)(a->drop_p * 65536.f + 0.5f) : 0u;
    p.drop_scale = a->drop_p > 0.f ? 1.f / (1.f - a->drop_p) : 1.f;
    p.drop_seed = a->drop_seed;
    p.gb = a->gb; p.gb_z = a->gb_z;
    dim3 grid((a->N + RT - 1) / RT, (a->M + RT - 1) / RT, a->nbatch * a->S);
    hipLaunchKernelGGL(rt_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t sc_rt_gemm_slices(int32_t M, int32_t N, int32_t K, int32_t nbatch) {
    // enough workgroups to cover the chip, K slices of at least 64
    const int tiles = ((N + RT - 1) / RT) * ((M + RT - 1) / RT) * std::max(nbatch, 1);
    int S = std::min((sc_num_cus() + tiles - 1) / tiles, std::max(K / 128, 1));
    S = std::max(1, std::min(S, 8));          // every slice is re-read by the consumer: a few fat ones
    int Kc = ((K + S - 1) / S + RTK - 1) / RTK * RTK;
    return (K + Kc - 1) / Kc;
}

extern "C" int sc_rt_ln_fwd(const sc_rt_ln_args* a, void* stream) {
    SC_CHECK(a && a->y && a->g1 && a->b1 && a->out1 && a->xhat1 && a->rstd1, "sc_rt_ln_fwd: null pointer");
    SC_CHECK(a->rows > 0 && a->D > 0 && a->D <= 1024 && a->ns >= 1, "sc_rt_ln_fwd: rows=%d D=%d (<= 1024)", a->rows, a->D);
    SC_CHECK(!a->g2 || (a->b2 && a->out2 && a->xhat2 && a->rstd2), "sc_rt_ln_fwd: second LayerNorm needs b2, out2, xhat2, rstd2");
    RtLn p;
    p.y = a->y; p.y_slice = a->y_slice; p.ns = a->ns; p.bias = a->bias;
    p.drop_thr = a->drop_p > 0.f ? (uint32_t)(a->drop_p * 65536.f + 0.5f) : 0u;
    p.drop_scale = a->drop_p > 0.f ? 1.f / (1.f - a->drop_p) : 1.f;
    p.drop_seed = a->drop_seed;
    p.res = a->res; p.res_stride = a->res_stride;
    p.g1 = a->g1; p.b1 = a->b1; p.eps1 = a->eps1; p.out1 = a->out1; p.xhat1 = a->xhat1; p.rstd1 = a->rstd1;
    p.g2 = a->g2; p.b2 = a->b2; p.eps2 = a->eps2; p.out2 = a->out2; p.xhat2 = a->xhat2; p.rstd2 = a->rstd2;
    p.rows = a->rows; p.D = a->D;
    hipLaunchKernelGGL(rt_ln_fwd_kernel, dim3(a->rows), dim3(256), 0, (hipStream_t)stream, p);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rt_ln_bwd(const sc_rt_ln_bwd_args* a, void* stream) {
    SC_CHECK(a && a->dy && a->xhat && a->gamma && a->rstd && a->dx && a->dgamma && a->dbeta, "sc_rt_ln_bwd: null pointer");
    SC_CHECK(a->rows > 0 && a->D > 0 && a->D <= 1024 && a->ns >= 1, "sc_rt_ln_bwd: rows=%d D=%d (<= 1024)", a->rows, a->D);
    RtLnBwd p;
    p.dy = a->dy; p.dy_slice = a->dy_slice; p.ns = a->ns; p.add = a->add; p.xhat = a->xhat; p.gamma = a->gamma; p.rstd = a->rstd;
    p.dx = a->dx; p.dx_masked = a->dx_masked;
    p.drop_thr = a->drop_p > 0.f ? (uint32_t)(a->drop_p * 65536.f + 0.5f) : 0u;
    p.drop_scale = a->drop_p > 0.f ? 1.f / (1.f - a->drop_p) : 1.f;
    p.drop_seed = a->drop_seed;
    p.dgamma = a->dgamma; p.dbeta = a->dbeta;
    p.rows = a->rows; p.D = a->D; p.row_blocks = a->rows;
    hipLaunchKernelGGL(rt_ln_bwd_kernel, dim3(p.row_blocks + (a->D + 31) / 32), dim3(256), 0, (hipStream_t)stream, p);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rt_l2norm_fwd(const float* y, int64_t y_slice, int32_t ns, const float* bias, float* x, float* e, float* rnorm, int32_t rows,
                                int32_t D, void* stream) {
    SC_CHECK(y && e && rnorm && rows > 0 && D > 0 && ns >= 1, "sc_rt_l2norm_fwd: bad args");
    SC_CHECK(D <= 1024, "sc_rt_l2norm_fwd: D=%d (<= 1024)", D);
    hipLaunchKernelGGL(rt_l2norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, y, y_slice, ns, bias, x, e, rnorm, rows, D);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rt_l2norm_bwd(const float* g, const float* e, const float* rnorm, float* dx, int32_t rows, int32_t D, void* stream) {
    SC_CHECK(g && e && rnorm && dx && rows > 0 && D > 0, "sc_rt_l2norm_bwd: bad args");
    hipLaunchKernelGGL(rt_l2norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g, e, rnorm, dx, rows, D);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rt_elem(const float* y, int64_t y_slice, int32_t ns, const float* bias, const float* rowscale, int32_t group, int32_t nscale,
                          float* out, float* u, int32_t rows, int32_t D, int32_t mode, float drop_p, uint32_t drop_seed, void* stream) {
    SC_CHECK(y && out && rows > 0 && D > 0 && ns >= 1 && mode >= 0 && mode <= 2 && (mode == 0 || u), "sc_rt_elem: bad args");
    SC_CHECK(drop_p >= 0.f && drop_p < 1.f && (int64_t)rows * D < ((int64_t)1 << 32), "sc_rt_elem: drop_p / size");
    const int64_t n = (int64_t)rows * D;
    hipLaunchKernelGGL(rt_elem_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, y_slice, ns, bias, rowscale,
                       group > 0 ? group : 1, nscale, out, u, rows, D, mode, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f,
                       drop_p > 0.f ? (uint32_t)(drop_p * 65536.f + 0.5f) : 0u, drop_seed);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rt_value_bias_bwd(const float* dctx, const float* bv, const float* psum, float* cbias, float* gbv, int32_t B, int32_t D,
                                    int32_t H, void* stream) {
    SC_CHECK(dctx && bv && psum && cbias && gbv && B > 0 && D > 0 && H > 0 && D % H == 0, "sc_rt_value_bias_bwd: bad args");
    hipLaunchKernelGGL(rt_value_bias_bwd_kernel, dim3(B + (D + 255) / 256), dim3(256), 0, (hipStream_t)stream, dctx, bv, psum, cbias, gbv, B, D, H);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rt_softmax_bwd_reduce(const float* part, int32_t nblk, int32_t NL, const float* w_soft, float* out, void* stream) {
    SC_CHECK(part && w_soft && out && nblk > 0 && NL > 0 && NL <= 64, "sc_rt_softmax_bwd_reduce: NL=%d (<= 64)", NL);
    hipLaunchKernelGGL(rt_softmax_bwd_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nblk, NL, w_soft, out);
    SC_LAUNCH_CHECK();
    return 0;
}
