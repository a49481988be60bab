// fp32 kernels of the contrastive loss and the optimiser step.
//   sc_sgemm_f32     generic strided fp32 GEMM (LDS-tiled 64x64x16, 4x4 register blocking)
//   sc_infonce_fwd   logits (exact-fp32 MFMA) + masked row / column log-sum-exp + loss scalar in ONE launch
//   sc_infonce_grad  dL/dlogits (x inv_temp) + the d inv_temp terms
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
// Masked contrastive loss (avssl/module/losses.py:185-245) with every option of the reference: margin on the positives, decoupled
// form (dcl: the positive is left out of its own denominator), one-sided (a2b / b2a) or symmetric.
//   logits[i,j] = inv_temp <A_i, B_j> - margin [i == j] ;   neg[i,j] = (ids ? ids[i] != ids[j] : i != j) || (!dcl && i == j)
//   loss = 1/(Bg n_dir) sum_i [ a2b (-l_ii + log sum_j neg_ij e^{l_ij}) + b2a (-l_ii + log sum_j neg_ji e^{l_ji}) ]
// FORWARD = ONE kernel.  Each workgroup owns a 64 x 64 tile of the logits: A and B rows staged through LDS (K-tiles of 16,
// transposed on the way in), the product on the matrix pipe in exact fp32 (v_mfma_f32_32x32x2_f32, one 32 x 32 block per wave),
// the scaled tile written out (the backward reads it) and reduced in LDS to per-tile row / column (max, sum exp) partials.  The
// workgroup that takes the last ticket merges the partials of all tiles into the row / column log-sum-exps and the scalar loss
// (fixed order: deterministic).  Cross-workgroup visibility: stores -> agent-scope release fence (+ explicit vmcnt drain) ->
// barrier -> ticket (returning atomic) ; last arriver: agent-scope acquire fence -> barrier -> plain loads.
// inv_temp comes from DEVICE memory (a trainable temperature never forces a host read).
constexpr int LT = 68;       // LDS row pitch of the [k][64 rows] operand images (2-way conflicts on the b32 stores only: free)
constexpr int KTL = 64;      // K-tile of the logits product

__global__ __launch_bounds__(256) void infonce_fwd_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int Bg, int E,
                                                          const int64_t* __restrict__ ids, const float* __restrict__ inv_temp_p,
                                                          float margin, int dcl, int a2b, int b2a, float* __restrict__ logits,
                                                          float* __restrict__ part, float* __restrict__ diag,
                                                          float* __restrict__ lse_row, float* __restrict__ lse_col,
                                                          float* __restrict__ loss, unsigned int* __restrict__ ticket) {
    __shared__ float As[KTL][LT];
    __shared__ float Bs[KTL][LT];
    __shared__ float Ls[64][65];
    __shared__ int last_flag;
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const int nt = gridDim.x;
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int i0 = ti * 64, j0 = tj * 64;
    const float it = inv_temp_p[0];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // K-tiles of 64: a workgroup's K loop is a chain of dependent global-load round trips (one tile = the whole problem at
    // Bg = 64), so each trip carries 4 x 16 bytes per thread and operand, and the next tile's loads fly under this tile's MFMAs
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    float4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + c * 16 + kq;
            ra[c] = rb[c] = float4{0.f, 0.f, 0.f, 0.f};
            if (k < E) {                           // E % 4 == 0 (checked on the host)
                if (i0 + lr < Bg) ra[c] = *(const float4*)(A + (int64_t)(i0 + lr) * E + k);
                if (j0 + lr < Bg) rb[c] = *(const float4*)(Bm + (int64_t)(j0 + lr) * E + k);
            }
        }
    };
    load_tile(0);
    for (int k0 = 0; k0 < E; k0 += KTL) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int kk = c * 16 + kq;
            As[kk + 0][lr] = ra[c].x; As[kk + 1][lr] = ra[c].y; As[kk + 2][lr] = ra[c].z; As[kk + 3][lr] = ra[c].w;
            Bs[kk + 0][lr] = rb[c].x; Bs[kk + 1][lr] = rb[c].y; Bs[kk + 2][lr] = rb[c].z; Bs[kk + 3][lr] = rb[c].w;
        }
        __syncthreads();
        if (k0 + KTL < E) load_tile(k0 + KTL);
#pragma unroll
        for (int kk = 0; kk < KTL; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + half][wm * 32 + l31], Bs[kk + half][wn * 32 + l31], acc, 0, 0, 0);
    }
    // scaled tile -> LDS (+ global)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, cc = wn * 32 + l31;
        const int gi = i0 + rr, gj = j0 + cc;
        const float l = acc[r] * it - ((gi == gj) ? margin : 0.f);
        Ls[rr][cc] = l;
        if (gi < Bg && gj < Bg) logits[(int64_t)gi * Bg + gj] = l;
    }
    __syncthreads();
    // per-tile partials: threads 0..63 one row each, 64..127 one column each
    float* prm = part;                                   // [nt][Bg] row max   (indexed by column tile)
    float* prs = part + (int64_t)nt * Bg;                // row sum exp
    float* pcm = part + (int64_t)2 * nt * Bg;            // [nt][Bg] col max   (indexed by row tile)
    float* pcs = part + (int64_t)3 * nt * Bg;
    // the tile's 64 row ids and 64 column ids once into LDS (the mask test below used to re-load them from global memory inside both
    // inner loops: 256 dependent loads per thread on a kernel that is ONE workgroup at the per-GPU batch of the recipes)
    __shared__ int64_t idl[2][64];
    if (tid < 128) {
        const int g = (tid < 64 ? i0 : j0) + (tid & 63);
        idl[tid >> 6][tid & 63] = g < Bg ? (ids ? ids[g] : (int64_t)g) : (int64_t)-1 - g;
    }
    __syncthreads();
    if (tid < 128) {
        const bool is_row = tid < 64;
        const int o = tid & 63;
        const int gfix = (is_row ? i0 : j0) + o;
        if (gfix < Bg) {
            const int64_t idf = idl[is_row ? 0 : 1][o];
            const int64_t* idv = idl[is_row ? 1 : 0];
            float m = -INFINITY;
            for (int q = 0; q < 64; ++q) {
                const int gvar = (is_row ? j0 : i0) + q;
                if (gvar >= Bg) break;
                const bool neg = idv[q] != idf || (!dcl && gvar == gfix);
                if (neg) m = fmaxf(m, is_row ? Ls[o][q] : Ls[q][o]);
            }
            float sum = 0.f;
            for (int q = 0; q < 64; ++q) {
                const int gvar = (is_row ? j0 : i0) + q;
                if (gvar >= Bg) break;
                const bool neg = idv[q] != idf || (!dcl && gvar == gfix);
                if (neg) sum += __expf((is_row ? Ls[o][q] : Ls[q][o]) - m);
            }
            if (is_row) { prm[(int64_t)tj * Bg + gfix] = m; prs[(int64_t)tj * Bg + gfix] = sum; }
            else { pcm[(int64_t)ti * Bg + gfix] = m; pcs[(int64_t)ti * Bg + gfix] = sum; }
            if (is_row && ti == tj) diag[gfix] = Ls[o][o];
        }
    }
    // ---- ticket: the last workgroup to arrive finalises.  A single-tile launch (Bg <= 64, the per-GPU batch of the recipes) is
    // its own last arriver: no fence - an agent-scope release on this part writes the XCD's L2 back, ~30 us after a GEMM
    if (gridDim.x * gridDim.y > 1) {
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned total = gridDim.x * gridDim.y;
            const unsigned t = atomicAdd(ticket, 1u);
            last_flag = (t == total - 1);
            if (t == total - 1) *ticket = 0;            // ready for the next launch on this stream
        }
        __syncthreads();
        if (!last_flag) return;
        __threadfence();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float term = 0.f;
    for (int i = tid; i < Bg; i += 256) {
        float mr = -INFINITY, mc = -INFINITY;
        for (int t = 0; t < nt; ++t) {
            mr = fmaxf(mr, prm[(int64_t)t * Bg + i]);
            mc = fmaxf(mc, pcm[(int64_t)t * Bg + i]);
        }
        float sr = 0.f, sc = 0.f;
        for (int t = 0; t < nt; ++t) {
            const float pm = prm[(int64_t)t * Bg + i], qm = pcm[(int64_t)t * Bg + i];
            if (pm > -INFINITY) sr += prs[(int64_t)t * Bg + i] * __expf(pm - mr);
            if (qm > -INFINITY) sc += pcs[(int64_t)t * Bg + i] * __expf(qm - mc);
        }
        const float lrow = mr + __logf(sr), lcol = mc + __logf(sc);     // no negatives at all: log 0 = -inf, as the reference
        lse_row[i] = lrow;
        lse_col[i] = lcol;
        const float d = diag[i];
        if (a2b) term += lrow - d;
        if (b2a) term += lcol - d;
    }
    term = wave_sum(term);
    if (lane == 0) red[wave] = term;
    __syncthreads();
    if (tid == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / ((float)Bg * (float)((a2b && b2a) ? 2 : 1));
}

// G[i,j] = inv_temp gscale / (Bg n_dir) (neg_ij (a2b e^{l - lse_row_i} + b2a e^{l - lse_col_j}) - (a2b + b2a) [i == j])
//   (already multiplied by inv_temp: dA = G . B, dB = G^T . A need no further scaling);
// dlogit_dot[i] = sum_j G[i,j] / inv_temp * <A_i, B_j> ... expressed through the stored logits: (l_ij + margin [i == j]) / inv_temp
//   -> d loss / d inv_temp = sum_i dlogit_dot[i]
__global__ __launch_bounds__(256) void infonce_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ ids,
                                                           const float* __restrict__ lse_row, const float* __restrict__ lse_col,
                                                           int Bg, const float* __restrict__ gscale,
                                                           const float* __restrict__ inv_temp_p, float margin, int dcl, int a2b,
                                                           int b2a, float* __restrict__ G, float* __restrict__ dlogit_dot) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    const int64_t idi = ids ? ids[i] : (int64_t)i;
    const float lr = lse_row[i];
    const float it = inv_temp_p[0];
    const float gs = gscale[0] / ((float)Bg * (float)((a2b && b2a) ? 2 : 1));
    const float npos = (float)((a2b ? 1 : 0) + (b2a ? 1 : 0));
    float dot = 0.f;
    for (int j = threadIdx.x; j < Bg; j += 256) {
        const float l = logits[(int64_t)i * Bg + j];
        const bool neg = (ids ? ids[j] != idi : j != i) || (!dcl && j == i);
        float g = 0.f;
        if (neg) g = (a2b ? __expf(l - lr) : 0.f) + (b2a ? __expf(l - lse_col[j]) : 0.f);
        if (j == i) g -= npos;
        g *= gs;
        G[(int64_t)i * Bg + j] = g * it;
        dot += g * (l + ((j == i) ? margin : 0.f));
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) dlogit_dot[i] = ((red[0] + red[1]) + (red[2] + red[3])) / it;
}

// ------------------------------------------------------------------------------------------ optimiser
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    if (((uintptr_t)x % 16) == 0) {          // 16-byte loads over the aligned body, scalar tail
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const f32x4 v = *(const f32x4*)(x + i * 4);
            s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
    }
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
        // every workgroup re-derives the global norm from the partials: lanes of the first wave take them in a fixed interleave and
        // add in a fixed order (deterministic), instead of every THREAD walking all of them
        __shared__ float clip_s;
        if (threadIdx.x < 64) {
            float tot = 0.f;
            for (int i = threadIdx.x; i < nblk; i += 64) tot += gn_partial[i];
            tot = wave_sum(tot);
            if (threadIdx.x == 0) clip_s = fminf(1.f, max_norm / (sqrtf(tot) + 1e-6f));      // torch.nn.utils.clip_grad_norm_
        }
        __syncthreads();
        clip = clip_s;
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

extern "C" int64_t sc_infonce_workspace_floats(int32_t Bg) {
    const int64_t nt = (Bg + 63) / 64;
    return 4 * nt * Bg + Bg + 4;        // per-tile partials, diagonal, ticket word (+ padding)
}

extern "C" int sc_infonce_fwd(const float* A, const float* Bm, int32_t Bg, int32_t E, const int64_t* ids, const float* inv_temp,
                              float margin, int32_t dcl, int32_t a2b, int32_t b2a, float* logits, float* lse_row, float* lse_col,
                              float* loss, float* workspace, void* stream) {
    SC_CHECK(A && Bm && inv_temp && logits && lse_row && lse_col && loss && workspace, "sc_infonce_fwd: null pointer");
    SC_CHECK(Bg > 0 && E > 0 && E % 4 == 0 && (a2b || b2a), "sc_infonce_fwd: Bg=%d E=%d (E %% 4), a2b | b2a", Bg, E);
    SC_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)Bm & 15) == 0, "sc_infonce_fwd: A / B must be 16-byte aligned");
    const int nt = (Bg + 63) / 64;
    float* part = workspace;
    float* diag = workspace + (int64_t)4 * nt * Bg;
    unsigned int* ticket = (unsigned int*)(diag + Bg);      // zero before the first launch; every launch leaves it zero
    hipLaunchKernelGGL(infonce_fwd_kernel, dim3(nt, nt), dim3(256), 0, (hipStream_t)stream, A, Bm, Bg, E, ids, inv_temp, margin, dcl,
                       a2b, b2a, logits, part, diag, lse_row, lse_col, loss, ticket);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_infonce_grad(const float* logits, const int64_t* ids, const float* lse_row, const float* lse_col, int32_t Bg,
                               const float* gscale, const float* inv_temp, float margin, int32_t dcl, int32_t a2b, int32_t b2a,
                               float* G, float* dlogit_dot, void* stream) {
    SC_CHECK(logits && lse_row && lse_col && gscale && inv_temp && G && dlogit_dot, "sc_infonce_grad: null pointer");
    SC_CHECK(Bg > 0 && (a2b || b2a), "sc_infonce_grad: bad arguments");
    hipLaunchKernelGGL(infonce_grad_kernel, dim3(Bg), dim3(256), 0, (hipStream_t)stream, logits, ids, lse_row, lse_col, Bg, gscale,
                       inv_temp, margin, dcl, a2b, b2a, G, dlogit_dot);
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
