// Keyword -> sub-word vector quantiser of the cascaded+/hybrid+ tails (scope rows a11 / f3):
//   cosine of every keyword against every (reduced-vocabulary) CLIP token embedding        avssl/model/kw_branches.py:158-179
//   special tokens masked, argmax / softmax(x / temp) straight-through, code statistics      my_vector_quantizer.py:64-165
//   keywords = subword_prob @ token_embedding and its backward                               kw_branches.py:181-197
//
// The cosine scores decide a DISCRETE token (argmax over 8112 / 19787 scores): they are computed in exact fp32 on the matrix
// pipe (v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulate, 64 FLOP/clk/SIMD), both operands K-major so that every global
// load and every LDS fragment read is unit-stride:
//   sc_vq_prep_f32        kw [Nk, Et] -> kwn_T [Et][Nkp] = kw / max(|kw|, eps) transposed, rnorm [Nk]
//   sc_sgemm_mfma_f32     C [M, N] = A . B^T (+ bias), each operand row-major or K-major   (128 x 128 tiles, 4 waves of 64 x 64,
//                         K-tile 16, register-staged double buffer)
//   sc_vq_rowstats        per row: mask columns -> -inf (written back, as the reference's in-place `x[:, i] += -inf`), argmax,
//                         LSE(x / temp), LSE(x), entropy of softmax(x)
//   sc_vq_colprob         partial column sums of softmax(x) (prob_perplexity)      grid = column blocks x row chunks, fixed order
//   sc_vq_perplexity      code_perplexity (histogram of the argmax indices) + prob_perplexity: count, per-workgroup entropy sums, final
//   sc_vq_gather_f32      out[n] = table[idx[n]]    (value of hard @ token_embedding)
//   sc_vq_onehot_f32      dense subword_prob for the module-level API (hard one-hot; the straight-through term is value-neutral)
//   sc_vq_soft_bwd        dx = softmax(x / temp) * (t - <softmax, t>) / temp   (t = d subword_prob), bf16 or fp32 out
//   sc_vq_norm_bwd_f32    gradient through x / max(|x|, eps)
// Everything is deterministic: fixed-order reductions, integer atomics only.
#include "sc_common.h"

namespace {

// ------------------------------------------------------------------------------------------ prep
// grid (row blocks of 64, column blocks of 64): every workgroup recomputes the norms of its 64 rows (four threads per row, independent
// 16-byte loads; the rows come out of the L2) and normalises + transposes ONE 64 x 64 tile.  (The first version walked all column
// tiles of a row block in one workgroup, 16 rows per wave one after the other: 26 workgroups, 67 - 102 us for 3 - 5 MB.)
__global__ __launch_bounds__(256) void vq_prep_kernel(const float* __restrict__ kw, int64_t ldk, int Nk, int Et, float eps,
                                                      float* __restrict__ kwn_T, int64_t ldt, float* __restrict__ rnorm) {
    __shared__ float rn[64];
    __shared__ float tile[64][65];
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    {
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3, n = n0 + r;
        float ss = 0.f;
        if (n < Nk) {
            const float* row = kw + (int64_t)n * ldk;
            if ((Et & 3) == 0 && (ldk & 3) == 0 && (((uintptr_t)kw) & 15) == 0) {
                for (int k = q * 4; k < Et; k += 16) {
                    const f32x4 v = *(const f32x4*)(row + k);
                    ss = fmaf(v[0], v[0], ss); ss = fmaf(v[1], v[1], ss); ss = fmaf(v[2], v[2], ss); ss = fmaf(v[3], v[3], ss);
                }
            } else {
                for (int k = q; k < Et; k += 4) ss = fmaf(row[k], row[k], ss);
            }
        }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        if (q == 0) {
            const float r_ = (n < Nk) ? 1.f / fmaxf(sqrtf(ss), eps) : 0.f;
            rn[r] = r_;
            if (n < Nk && blockIdx.y == 0) rnorm[n] = r_;
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int r = idx >> 6, c = idx & 63;
        const int n = n0 + r, k = k0 + c;
        tile[r][c] = (n < Nk && k < Et) ? kw[(int64_t)n * ldk + k] * rn[r] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int c = idx >> 6, r = idx & 63;
        const int k = k0 + c;
        if (k < Et) kwn_T[(int64_t)k * ldt + n0 + r] = tile[r][c];
    }
}

// ------------------------------------------------------------------------------------------ three-way bf16 split (round 6)
// x (times an optional per-row scale) = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 significant bits in
// three bf16 values.  A product of two such numbers to fp32 accuracy is the six bf16 x bf16 products (1,1) (1,2) (2,1) (1,3) (3,1) (2,2) -
// each EXACT in fp32 (8 x 8 significant bits) - summed in fp32; the dropped terms are below 2^-24 of the product.  Laid out as six
// K-blocks, side 0 = [x1 | x1 | x2 | x1 | x3 | x2], side 1 = [x1 | x2 | x1 | x3 | x1 | x2], the whole sum is ONE bf16 GEMM with K = 6 Ep
// on the bf16 matrix pipe (2.5 PF) instead of an fp32 GEMM on the fp32 pipe (157 TF): the cosine scores of the keyword quantiser.
// out [Rp][6 Ep] bf16; rows >= R and columns >= E of every block are written as zeros.  One thread = 4 consecutive columns of a row.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ row_scale, int R, int E,
                                                     uint16_t* __restrict__ out, int Rp, int Ep, int side) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int q4 = Ep >> 2;
    if (idx >= (int64_t)Rp * q4) return;
    const int r = (int)(idx / q4), c = (int)(idx % q4) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R) {
        const float s = row_scale ? row_scale[r] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < E) v[j] = x[(int64_t)r * ldx + c + j] * s;
    }
    uint2 part[3];
    float res[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const uint32_t lo = pack2bf(res[0], res[1]), hi = pack2bf(res[2], res[3]);
        part[t] = make_uint2(lo, hi);
        res[0] -= bflo(lo); res[1] -= bfhi(lo); res[2] -= bflo(hi); res[3] -= bfhi(hi);      // exact: the difference of a value and its rounding
    }
    uint16_t* o = out + (int64_t)r * 6 * Ep + c;
    const int pat0[6] = {0, 0, 1, 0, 2, 1}, pat1[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
    for (int b = 0; b < 6; ++b) *(uint2*)(o + (int64_t)b * Ep) = part[side ? pat1[b] : pat0[b]];
}

// ------------------------------------------------------------------------------------------ fp32 MFMA GEMM
// C[m, n] = sum_k A(m, k) B(n, k) (+ bias[n]).  Each operand is either ROW-major ([rows][K], the nn.Linear layout) or K-major ([K][rows]);
// any M, N, K (16-byte loads when base and leading dimension are 16-byte multiples, element loads otherwise).
// LDS image [k][LDT floats], LDT = 132: a fragment read is lane -> column (32 consecutive floats per half wave, conflict free);
// row-major tiles are transposed on the way in (4 scalar LDS stores per 16-byte global load, 2-way conflicts = free on b32 stores).
constexpr int LDT = 132;

template <bool KMAJOR>
__device__ __forceinline__ void sg_load(float4 (&r)[2], const float* __restrict__ P, int64_t ld, int rows, int K, int r0, int k0,
                                        int tid, bool vec) {
    if constexpr (KMAJOR) {          // P[k][row]: thread -> (k = tid / 32 (+8), 4 consecutive rows)
        const int kr = tid >> 5, c = (tid & 31) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* p = P + (int64_t)(k0 + kr + 8 * h) * ld + r0 + c;
            if (k0 + kr + 8 * h >= K) r[h] = float4{0.f, 0.f, 0.f, 0.f};
            else if (vec && r0 + c + 3 < rows) r[h] = *(const float4*)p;
            else {
                r[h].x = r0 + c < rows ? p[0] : 0.f;
                r[h].y = r0 + c + 1 < rows ? p[1] : 0.f;
                r[h].z = r0 + c + 2 < rows ? p[2] : 0.f;
                r[h].w = r0 + c + 3 < rows ? p[3] : 0.f;
            }
        }
    } else {                         // P[row][k]: thread -> (row = tid / 4 (+64), 4 consecutive k)
        const int rr = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = r0 + rr + 64 * h;
            const float* p = P + (int64_t)row * ld + k0 + kq;
            if (row >= rows || k0 + kq >= K) r[h] = float4{0.f, 0.f, 0.f, 0.f};
            else if (vec && k0 + kq + 3 < K) r[h] = *(const float4*)p;
            else {
                r[h].x = p[0];
                r[h].y = k0 + kq + 1 < K ? p[1] : 0.f;
                r[h].z = k0 + kq + 2 < K ? p[2] : 0.f;
                r[h].w = k0 + kq + 3 < K ? p[3] : 0.f;
            }
        }
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void sg_stage(const float4 (&r)[2], float (*S)[LDT], int tid) {
    if constexpr (KMAJOR) {
        const int kr = tid >> 5, c = (tid & 31) * 4;
        *(float4*)&S[kr][c] = r[0];
        *(float4*)&S[kr + 8][c] = r[1];
    } else {
        const int rr = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            S[kq + 0][rr + 64 * h] = r[h].x;
            S[kq + 1][rr + 64 * h] = r[h].y;
            S[kq + 2][rr + 64 * h] = r[h].z;
            S[kq + 3][rr + 64 * h] = r[h].w;
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void sgemm_mfma_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                           int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                           const float* __restrict__ bias, int vec_a, int vec_b, int Kc, int64_t c_slice) {
    // blockIdx.z = slice of the contraction (k in [z Kc, min(K, (z + 1) Kc)), Kc a multiple of 16); with more than one slice the
    // raw partial tile goes to C + z c_slice (no bias) and sgemm_reduce_kernel adds the slices in order
    __shared__ float As[2][16][LDT];
    __shared__ float Bs[2][16][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;
    float4 ra[2], rb[2];
    const int kb = blockIdx.z * Kc;
    K = min(K, kb + Kc);
    C += blockIdx.z * c_slice;
    sg_load<A_KMAJOR>(ra, A, lda, M, K, m0, kb, tid, vec_a);
    sg_load<B_KMAJOR>(rb, Bm, ldb, N, K, n0, kb, tid, vec_b);
    sg_stage<A_KMAJOR>(ra, As[0], tid);
    sg_stage<B_KMAJOR>(rb, Bs[0], tid);
    __syncthreads();
    const int nk = (K - kb + 15) >> 4;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nk;
        if (more) {
            sg_load<A_KMAJOR>(ra, A, lda, M, K, m0, kb + (t + 1) * 16, tid, vec_a);
            sg_load<B_KMAJOR>(rb, Bm, ldb, N, K, n0, kb + (t + 1) * 16, tid, vec_b);
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const int kr = kk + half;
            const float a0 = As[cur][kr][wm * 64 + l31], a1 = As[cur][kr][wm * 64 + 32 + l31];
            const float b0 = Bs[cur][kr][wn * 64 + l31], b1 = Bs[cur][kr][wn * 64 + 32 + l31];
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
        if (more) {
            sg_stage<A_KMAJOR>(ra, As[cur ^ 1], tid);
            sg_stage<B_KMAJOR>(rb, Bs[cur ^ 1], tid);
        }
        __syncthreads();
    }
    // accumulator element r of lane l: row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    auto store = [&](const f32x16& a, int mi, int nj) {
        const int col = n0 + wn * 64 + nj * 32 + l31;
        if (col >= N) return;
        const float bv = bias ? bias[col] : 0.f;
        const int rbase = m0 + wm * 64 + mi * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (row < M) C[(int64_t)row * ldc + col] = a[r] + bv;
        }
    };
    store(acc00, 0, 0);
    store(acc01, 0, 1);
    store(acc10, 1, 0);
    store(acc11, 1, 1);
}

// C[m, n] = part[0][m, n] + part[1][m, n] + ... + part[S - 1][m, n] (+ bias[n]): the slices of a split product, added in slice order
__global__ __launch_bounds__(256) void sgemm_reduce_kernel(const float* __restrict__ part, int64_t slice, int S, int M, int N,
                                                           const float* __restrict__ bias, float* __restrict__ C, int64_t ldc) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)M * N) return;
    const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
    float a = part[e];
    for (int s2 = 1; s2 < S; ++s2) a += part[s2 * slice + e];
    C[(int64_t)m * ldc + n] = a + (bias ? bias[n] : 0.f);
}

// ------------------------------------------------------------------------------------------ row statistics
struct MaskCols { int c[4]; };

__device__ __forceinline__ bool is_masked(int v, const MaskCols& m) { return v == m.c[0] || v == m.c[1] || v == m.c[2] || v == m.c[3]; }

template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {      // red: NW floats of LDS; all threads get the total
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(256) void vq_rowstats_kernel(float* __restrict__ x, int64_t ldx, int V, float inv_temp, MaskCols mask,
                                                          int64_t* __restrict__ idx, float* __restrict__ lse_t,
                                                          float* __restrict__ lse_1, float* __restrict__ ent) {
    __shared__ float red[4];
    __shared__ float redm[4];
    __shared__ int redi[4];
    float* row = x + (int64_t)blockIdx.x * ldx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 1: write the mask, max + first arg max
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        float xv = row[v];
        if (is_masked(v, mask)) {
            xv = -INFINITY;
            row[v] = xv;
        }
        if (xv > m || (xv == m && v < mi)) { m = xv; mi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o);
        const int oi = __shfl_xor(mi, o);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) { redm[wave] = m; redi[wave] = mi; }
    __syncthreads();
    m = redm[0];
    mi = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (redm[w] > m || (redm[w] == m && redi[w] < mi)) { m = redm[w]; mi = redi[w]; }
    // pass 2: sums of exp((x - m) / temp) and exp(x - m)
    float st = 0.f, s1 = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float d = row[v] - m;             // -inf for masked columns: exp -> 0
        st += __expf(d * inv_temp);
        s1 += __expf(d);
    }
    st = block_sum<4>(st, red);
    s1 = block_sum<4>(s1, red);
    const float l1 = m + __logf(s1);
    // pass 3: entropy exactly as the reference writes it: - sum p log(p + 1e-9), p = softmax(x)
    float e = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float p = __expf(row[v] - l1);
        e -= p * __logf(p + 1e-9f);
    }
    e = block_sum<4>(e, red);
    if (threadIdx.x == 0) {
        idx[blockIdx.x] = mi;
        lse_t[blockIdx.x] = m * inv_temp + __logf(st);
        lse_1[blockIdx.x] = l1;
        ent[blockIdx.x] = e;
    }
}

// partial[chunk][v] = sum over the chunk's rows of exp(x[n, v] - lse_1[n]); the blocks of chunk 0 also clear the histogram word of
// their column (the counting launch comes next)
__global__ __launch_bounds__(256) void vq_colprob_kernel(const float* __restrict__ x, int64_t ldx, int Nk, int V,
                                                         const float* __restrict__ lse_1, int rows_per_chunk,
                                                         float* __restrict__ partial, int* __restrict__ hist) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * rows_per_chunk;
    const int n1 = min(Nk, n0 + rows_per_chunk);
    if (v >= V) return;
    if (blockIdx.y == 0) hist[v] = 0;
    float s = 0.f;
    for (int n = n0; n < n1; ++n) s += __expf(x[(int64_t)n * ldx + v] - lse_1[n]);
    partial[(int64_t)blockIdx.y * V + v] = s;
}

__global__ __launch_bounds__(256) void vq_hist_kernel(const int64_t* __restrict__ idx, int Nk, int* __restrict__ hist) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < Nk) atomicAdd(&hist[(int)idx[n]], 1);               // integer atomics: order-independent
}

// per workgroup w: ent[2 w] = sum over its columns of hp log(hp + 1e-7), ent[2 w + 1] = sum of ap log(ap + 1e-7),
// hp = histogram(idx) / Nk, ap = sum_chunks partial / Nk   (one workgroup walked all V columns x nchunk partials: 40 - 93 us)
__global__ __launch_bounds__(1024) void vq_entropy_kernel(int Nk, int V, const float* __restrict__ partial, int nchunk,
                                                          const int* __restrict__ hist, float* __restrict__ ent) {
    __shared__ float red[16];
    const float inv = 1.f / (float)Nk;
    float sc = 0.f, sp = 0.f;
    for (int v = blockIdx.x * 1024 + threadIdx.x; v < V; v += gridDim.x * 1024) {
        const float hp = (float)hist[v] * inv;
        sc += hp * __logf(hp + 1e-7f);
        float a = 0.f;
        for (int c = 0; c < nchunk; ++c) a += partial[(int64_t)c * V + v];
        a *= inv;
        sp += a * __logf(a + 1e-7f);
    }
    sc = block_sum<16>(sc, red);
    sp = block_sum<16>(sp, red);
    if (threadIdx.x == 0) {
        ent[2 * blockIdx.x] = sc;
        ent[2 * blockIdx.x + 1] = sp;
    }
}

// out[0] = code_perplexity = exp(-sum hp log(hp + 1e-7)), out[1] = prob_perplexity = exp(-sum ap log(ap + 1e-7)): the workgroups'
// sums added in order
__global__ void vq_perplexity_kernel(const float* __restrict__ ent, int nw, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float sc = 0.f, sp = 0.f;
        for (int w = 0; w < nw; ++w) { sc += ent[2 * w]; sp += ent[2 * w + 1]; }
        out[0] = __expf(-sc);
        out[1] = __expf(-sp);
    }
}

__global__ __launch_bounds__(256) void vq_gather_kernel(const float* __restrict__ table, int64_t ldt, const int64_t* __restrict__ idx,
                                                        float* __restrict__ out, int64_t ldo, int Nk, int Et) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= Nk) return;
    const float* src = table + idx[n] * ldt;
    float* dst = out + (int64_t)n * ldo;
    for (int k = (threadIdx.x & 63) * 4; k < Et; k += 256) *(float4*)(dst + k) = *(const float4*)(src + k);
}

__global__ __launch_bounds__(256) void vq_onehot_kernel(const int64_t* __restrict__ idx, float* __restrict__ out, int64_t ldo, int V) {
    float* row = out + (int64_t)blockIdx.x * ldo;
    const int k = (int)idx[blockIdx.x];
    for (int v = threadIdx.x; v < V; v += 256) row[v] = (v == k) ? 1.f : 0.f;
}

// dx[v] = softmax(x / temp)[v] * (t[v] - sum_u softmax[u] t[u]) / temp ; columns V .. ldd_pad - 1 are zero-filled
template <typename OutT>
__global__ __launch_bounds__(256) void vq_soft_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ lse_t,
                                                          const float* __restrict__ t, int64_t ldt, int V, int Vpad, float inv_temp,
                                                          OutT* __restrict__ dx, int64_t ldd) {
    __shared__ float red[4];
    const float* xr = x + (int64_t)blockIdx.x * ldx;
    const float* tr = t + (int64_t)blockIdx.x * ldt;
    OutT* dr = dx + (int64_t)blockIdx.x * ldd;
    const float l = lse_t[blockIdx.x];
    float m = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float s = __expf(fmaf(xr[v], inv_temp, -l));
        if (s > 0.f) m = fmaf(s, tr[v], m);             // masked columns: s = 0 (and t may be anything)
    }
    m = block_sum<4>(m, red);
    for (int v = threadIdx.x; v < Vpad; v += 256) {
        float g = 0.f;
        if (v < V) {
            const float s = __expf(fmaf(xr[v], inv_temp, -l));
            if (s > 0.f) g = s * (tr[v] - m) * inv_temp;
        }
        if constexpr (sizeof(OutT) == 2) dr[v] = f2bf(g);
        else dr[v] = g;
    }
}

// y = x * r, r = 1 / max(|x|, eps) (saved):  dx = r (dy - y <y, dy>)   [|x| >= eps; below eps the norm is the constant eps: dx = dy / eps]
__global__ __launch_bounds__(256) void vq_norm_bwd_kernel(const float* __restrict__ kw, int64_t ldk, const float* __restrict__ rnorm,
                                                          const float* __restrict__ dy, int64_t ldy, float eps,
                                                          float* __restrict__ dx, int64_t ldd, int Nk, int Et) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= Nk) return;
    const int lane = threadIdx.x & 63;
    const float r = rnorm[n];
    const float* xr = kw + (int64_t)n * ldk;
    const float* gr = dy + (int64_t)n * ldy;
    float dot = 0.f;
    for (int k = lane; k < Et; k += 64) dot = fmaf(xr[k] * r, gr[k], dot);
    dot = wave_sum(dot);
    const bool clamped = r >= 1.f / eps;
    for (int k = lane; k < Et; k += 64) dx[(int64_t)n * ldd + k] = clamped ? gr[k] * r : r * (gr[k] - xr[k] * r * dot);
}

// ------------------------------------------------------------------------------------------ keyword BatchNorm (kw_bn.py:167-228)
// nn.BatchNorm1d over the keyword positions: x [N, E] fp32 (N = batch x keyword slots, padded slots included, as the reference's
// permute(0, 2, 1) view feeds them), statistics per channel.  One workgroup per BN_CB = 8 channels, 1024 threads = 128 row lanes x 8
// columns: 64 - 96 workgroups and a 13-row serial walk per pass at N = 1600 (32 channels x 32 row lanes: 16 - 24 workgroups, 50 rows,
// 30 - 39 us; 8 row lanes: 102 us); two-pass mean / variance, then the normalisation, all out of the L2.  The row lanes are added in
// fixed order.
constexpr int BN_CB = 8, BN_RL = 1024 / BN_CB;                                    // channels per workgroup, row lanes
__device__ __forceinline__ float col_reduce8(float v, float (*red)[BN_CB + 1]) {  // sum over the row lanes of a column
    const int c = threadIdx.x % BN_CB, r = threadIdx.x / BN_CB;
    __syncthreads();
    red[r][c] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BN_RL; ++i) s += red[i][c];
    return s;
}

__global__ __launch_bounds__(1024) void bn_fwd_kernel(const float* __restrict__ x, int64_t ldx, int N, int E, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ run_mean,
                                                     float* __restrict__ run_var, int training, float momentum, float eps,
                                                     float* __restrict__ y, int64_t ldy, float* __restrict__ save_mean,
                                                     float* __restrict__ save_rstd) {
    __shared__ float red[BN_RL][BN_CB + 1];
    const int c = threadIdx.x % BN_CB, r = threadIdx.x / BN_CB;
    const int e = blockIdx.x * BN_CB + c;
    const bool ok = e < E;
    float mean, rstd;
    if (training) {
        float s = 0.f;
        for (int n = r; n < N; n += BN_RL) s += ok ? x[(int64_t)n * ldx + e] : 0.f;
        mean = col_reduce8(s, red) / (float)N;
        float q = 0.f;
        for (int n = r; n < N; n += BN_RL) {
            const float d = ok ? x[(int64_t)n * ldx + e] - mean : 0.f;
            q = fmaf(d, d, q);
        }
        const float var = col_reduce8(q, red) / (float)N;
        rstd = rsqrtf(var + eps);
        if (ok && r == 0) {
            save_mean[e] = mean;
            save_rstd[e] = rstd;
            run_mean[e] = (1.f - momentum) * run_mean[e] + momentum * mean;
            run_var[e] = (1.f - momentum) * run_var[e] + momentum * var * ((float)N / (float)max(N - 1, 1));
        }
    } else {
        mean = ok ? run_mean[e] : 0.f;
        rstd = ok ? rsqrtf(run_var[e] + eps) : 0.f;
    }
    if (!ok) return;
    const float g = gamma[e] * rstd, b = beta[e] - mean * gamma[e] * rstd;
    for (int n = r; n < N; n += BN_RL) y[(int64_t)n * ldy + e] = fmaf(x[(int64_t)n * ldx + e], g, b);
}

// dx = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)) ; dgamma = sum dy xhat ; dbeta = sum dy       (training statistics)
__global__ __launch_bounds__(1024) void bn_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t ldg,
                                                     int N, int E, const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                     const float* __restrict__ save_rstd, float* __restrict__ dx, int64_t ldd,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[BN_RL][BN_CB + 1];
    const int c = threadIdx.x % BN_CB, r = threadIdx.x / BN_CB;
    const int e = blockIdx.x * BN_CB + c;
    const bool ok = e < E;
    const float mean = ok ? save_mean[e] : 0.f, rstd = ok ? save_rstd[e] : 0.f;
    float sb = 0.f, sg = 0.f;
    for (int n = r; n < N; n += BN_RL) {
        const float g = ok ? dy[(int64_t)n * ldg + e] : 0.f;
        const float xh = ok ? (x[(int64_t)n * ldx + e] - mean) * rstd : 0.f;
        sb += g;
        sg = fmaf(g, xh, sg);
    }
    sb = col_reduce8(sb, red);
    sg = col_reduce8(sg, red);
    if (!ok) return;
    if (r == 0) {
        dgamma[e] = sg;
        dbeta[e] = sb;
    }
    const float k = gamma[e] * rstd, mb = sb / (float)N, mg = sg / (float)N;
    for (int n = r; n < N; n += BN_RL) {
        const float xh = (x[(int64_t)n * ldx + e] - mean) * rstd;
        dx[(int64_t)n * ldd + e] = k * (dy[(int64_t)n * ldg + e] - mb - xh * mg);
    }
}

}  // namespace

extern "C" int sc_vq_prep_f32(const float* kw, int64_t ldk, int32_t Nk, int32_t Et, float eps, float* kwn_T, int64_t ldt, float* rnorm,
                              void* stream) {
    SC_CHECK(kw && kwn_T && rnorm, "sc_vq_prep_f32: null pointer");
    SC_CHECK(Nk > 0 && Et > 0 && ldt % 64 == 0 && ldt >= Nk, "sc_vq_prep_f32: ldt must be a multiple of 64 >= Nk");
    hipLaunchKernelGGL(vq_prep_kernel, dim3((unsigned)(ldt / 64), (unsigned)((Et + 63) / 64)), dim3(256), 0, (hipStream_t)stream, kw, ldk, Nk, Et, eps, kwn_T, ldt, rnorm);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_split3_bf16(const float* x, int64_t ldx, const float* row_scale, int32_t R, int32_t E, sc_bf16* out, int32_t Rp, int32_t Ep,
                              int32_t side, void* stream) {
    SC_CHECK(x && out, "sc_split3_bf16: null pointer");
    SC_CHECK(R > 0 && E > 0 && Rp >= R && Ep >= E && Ep % 4 == 0 && (side == 0 || side == 1) && ((uintptr_t)out % 8) == 0,
             "sc_split3_bf16: R=%d E=%d Rp=%d Ep=%d (Ep %% 4 == 0, out 8-byte aligned) side=%d", R, E, Rp, Ep, side);
    const int64_t n = (int64_t)Rp * (Ep / 4);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, row_scale, R, E, out, Rp, Ep, side);
    SC_LAUNCH_CHECK();
    return 0;
}

// slices the split entry point wants for this shape (1 = the product already fills the chip): a few workgroups walking a long K
// one 16-wide tile at a time are latency-bound (64 rows x 768 columns x K = 1024: 6 workgroups, 100 us)
extern "C" int32_t sc_sgemm_mfma_slices(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    const int64_t wgs = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    const int cus = sc_num_cus();
    if (wgs > cus || K < 256) return 1;
    int64_t S = 2 * cus / wgs;                       // two workgroups per CU are resident (launch bounds)
    if (S > K / 64) S = K / 64;
    if (S > 16) S = 16;
    return S < 1 ? 1 : (int32_t)S;
}

static int sgemm_launch(const float* A, int64_t lda, int32_t a_kmajor, const float* Bm, int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc,
                        int32_t M, int32_t N, int32_t K, const float* bias, float* partials, int32_t S, void* stream) {
    SC_CHECK(A && Bm && C, "sc_sgemm_mfma_f32: null pointer");
    SC_CHECK(M > 0 && N > 0 && K > 0, "sc_sgemm_mfma_f32: M=%d N=%d K=%d", M, N, K);
    SC_CHECK(ldc >= N, "sc_sgemm_mfma_f32: ldc");
    SC_CHECK(S >= 1 && S <= 64 && (S == 1 || partials), "sc_sgemm_mfma_f32: S=%d slices need a [S, M, N] workspace", S);
    // 16-byte loads when the base and the leading dimension allow it, element loads otherwise (tiny / ragged operands)
    const int vec_a = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0), vec_b = (ldb % 4 == 0) && (((uintptr_t)Bm & 15) == 0);
    int Kc = K;
    if (S > 1) {
        Kc = ((K + S - 1) / S + 15) / 16 * 16;
        S = (K + Kc - 1) / Kc;                           // no empty slice
    }
    const dim3 grid((N + 127) / 128, (M + 127) / 128, S);
    float* out = S > 1 ? partials : C;
    const int64_t ldo = S > 1 ? N : ldc, slice = S > 1 ? (int64_t)M * N : 0;
    const float* kb = S > 1 ? nullptr : bias;
#define SG_LAUNCH(AK, BK) hipLaunchKernelGGL((sgemm_mfma_kernel<AK, BK>), grid, dim3(256), 0, (hipStream_t)stream, A, lda, Bm, ldb, out, ldo, M, N, K, kb, vec_a, vec_b, Kc, slice)
    if (a_kmajor && b_kmajor) SG_LAUNCH(true, true);
    else if (a_kmajor) SG_LAUNCH(true, false);
    else if (b_kmajor) SG_LAUNCH(false, true);
    else SG_LAUNCH(false, false);
#undef SG_LAUNCH
    SC_LAUNCH_CHECK();
    if (S > 1) {
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(sgemm_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, slice, S, M, N, bias, C, ldc);
        SC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sc_sgemm_mfma_f32(const float* A, int64_t lda, int32_t a_kmajor, const float* Bm, int64_t ldb, int32_t b_kmajor,
                                 float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, void* stream) {
    return sgemm_launch(A, lda, a_kmajor, Bm, ldb, b_kmajor, C, ldc, M, N, K, bias, nullptr, 1, stream);
}

extern "C" int sc_sgemm_mfma_f32_split(const float* A, int64_t lda, int32_t a_kmajor, const float* Bm, int64_t ldb, int32_t b_kmajor,
                                       float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, float* partials,
                                       int32_t S, void* stream) {
    return sgemm_launch(A, lda, a_kmajor, Bm, ldb, b_kmajor, C, ldc, M, N, K, bias, partials, S, stream);
}

extern "C" int sc_vq_rowstats(float* x, int64_t ldx, int32_t Nk, int32_t V, float temp, const int32_t* mask_cols_host, int32_t n_mask,
                              int64_t* idx, float* lse_t, float* lse_1, float* ent, void* stream) {
    SC_CHECK(x && idx && lse_t && lse_1 && ent, "sc_vq_rowstats: null pointer");
    SC_CHECK(Nk > 0 && V > 0 && temp > 0.f && n_mask >= 0 && n_mask <= 4 && ldx >= V, "sc_vq_rowstats: bad arguments");
    MaskCols m;
    for (int i = 0; i < 4; ++i) m.c[i] = (i < n_mask) ? mask_cols_host[i] : -1;
    hipLaunchKernelGGL(vq_rowstats_kernel, dim3(Nk), dim3(256), 0, (hipStream_t)stream, x, ldx, V, 1.f / temp, m, idx, lse_t, lse_1, ent);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_vq_perplexity(const float* x, int64_t ldx, int32_t Nk, int32_t V, const int64_t* idx, const float* lse_1,
                                float* partial, int32_t nchunk, int32_t* hist, float* out2, void* stream) {
    SC_CHECK(x && idx && lse_1 && partial && hist && out2, "sc_vq_perplexity: null pointer");
    SC_CHECK(Nk > 0 && V > 0 && nchunk > 0, "sc_vq_perplexity: bad arguments");
    const int rpc = (Nk + nchunk - 1) / nchunk;
    hipStream_t s = (hipStream_t)stream;
    // hist [V + 64] int32: the 64 words behind the histogram hold the entropy partials of <= 32 workgroups (as floats)
    float* ent = (float*)(hist + V);
    const int nw = (V + 1023) / 1024 < 32 ? (V + 1023) / 1024 : 32;
    hipLaunchKernelGGL(vq_colprob_kernel, dim3((V + 255) / 256, nchunk), dim3(256), 0, s, x, ldx, Nk, V, lse_1, rpc, partial, hist);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(vq_hist_kernel, dim3((Nk + 255) / 256), dim3(256), 0, s, idx, Nk, hist);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(vq_entropy_kernel, dim3(nw), dim3(1024), 0, s, Nk, V, partial, nchunk, (const int*)hist, ent);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(vq_perplexity_kernel, dim3(1), dim3(64), 0, s, (const float*)ent, nw, out2);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_vq_gather_f32(const float* table, int64_t ldt, const int64_t* idx, float* out, int64_t ldo, int32_t Nk, int32_t Et,
                                void* stream) {
    SC_CHECK(table && idx && out, "sc_vq_gather_f32: null pointer");
    SC_CHECK(Nk > 0 && Et > 0 && Et % 4 == 0 && ldt % 4 == 0 && ldo % 4 == 0, "sc_vq_gather_f32: Et, ldt, ldo %% 4");
    hipLaunchKernelGGL(vq_gather_kernel, dim3((Nk + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ldt, idx, out, ldo, Nk, Et);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_vq_onehot_f32(const int64_t* idx, float* out, int64_t ldo, int32_t Nk, int32_t V, void* stream) {
    SC_CHECK(idx && out && Nk > 0 && V > 0 && ldo >= V, "sc_vq_onehot_f32: bad arguments");
    hipLaunchKernelGGL(vq_onehot_kernel, dim3(Nk), dim3(256), 0, (hipStream_t)stream, idx, out, ldo, V);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_vq_soft_bwd(const float* x, int64_t ldx, const float* lse_t, const float* t, int64_t ldt, int32_t Nk, int32_t V,
                              int32_t Vpad, float temp, void* dx, int64_t ldd, int32_t out_bf16, void* stream) {
    SC_CHECK(x && lse_t && t && dx, "sc_vq_soft_bwd: null pointer");
    SC_CHECK(Nk > 0 && V > 0 && Vpad >= V && ldd >= Vpad && temp > 0.f, "sc_vq_soft_bwd: bad arguments");
    if (out_bf16)
        hipLaunchKernelGGL(vq_soft_bwd_kernel<uint16_t>, dim3(Nk), dim3(256), 0, (hipStream_t)stream, x, ldx, lse_t, t, ldt, V, Vpad,
                           1.f / temp, (uint16_t*)dx, ldd);
    else
        hipLaunchKernelGGL(vq_soft_bwd_kernel<float>, dim3(Nk), dim3(256), 0, (hipStream_t)stream, x, ldx, lse_t, t, ldt, V, Vpad,
                           1.f / temp, (float*)dx, ldd);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_vq_norm_bwd_f32(const float* kw, int64_t ldk, const float* rnorm, const float* dy, int64_t ldy, float eps, float* dx,
                                  int64_t ldd, int32_t Nk, int32_t Et, void* stream) {
    SC_CHECK(kw && rnorm && dy && dx && Nk > 0 && Et > 0, "sc_vq_norm_bwd_f32: bad arguments");
    hipLaunchKernelGGL(vq_norm_bwd_kernel, dim3((Nk + 3) / 4), dim3(256), 0, (hipStream_t)stream, kw, ldk, rnorm, dy, ldy, eps, dx, ldd, Nk, Et);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_bn_rows_fwd(const float* x, int64_t ldx, int32_t N, int32_t E, const float* gamma, const float* beta, float* run_mean,
                              float* run_var, int32_t training, float momentum, float eps, float* y, int64_t ldy, float* save_mean,
                              float* save_rstd, void* stream) {
    SC_CHECK(x && gamma && beta && run_mean && run_var && y, "sc_bn_rows_fwd: null pointer");
    SC_CHECK(N > 0 && E > 0 && (!training || (save_mean && save_rstd)), "sc_bn_rows_fwd: bad arguments");
    hipLaunchKernelGGL(bn_fwd_kernel, dim3((E + BN_CB - 1) / BN_CB), dim3(BN_CB * BN_RL), 0, (hipStream_t)stream, x, ldx, N, E, gamma, beta, run_mean, run_var,
                       training, momentum, eps, y, ldy, save_mean, save_rstd);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_bn_rows_bwd(const float* x, int64_t ldx, const float* dy, int64_t ldg, int32_t N, int32_t E, const float* gamma,
                              const float* save_mean, const float* save_rstd, float* dx, int64_t ldd, float* dgamma, float* dbeta,
                              void* stream) {
    SC_CHECK(x && dy && gamma && save_mean && save_rstd && dx && dgamma && dbeta, "sc_bn_rows_bwd: null pointer");
    SC_CHECK(N > 0 && E > 0, "sc_bn_rows_bwd: bad arguments");
    hipLaunchKernelGGL(bn_bwd_kernel, dim3((E + BN_CB - 1) / BN_CB), dim3(BN_CB * BN_RL), 0, (hipStream_t)stream, x, ldx, dy, ldg, N, E, gamma, save_mean,
                       save_rstd, dx, ldd, dgamma, dbeta);
    SC_LAUNCH_CHECK();
    return 0;
}
