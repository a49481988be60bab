// CLS attention pooling of the parallel branch without materialising K / V.
//
// The branch consumes only the CLS row of its transformer layer, so per utterance b and head h
//   scores[s] = (Wk_h^T q_h) . X[b,s]   (+ a constant that cancels in the softmax)
//   ctx_h     = Wv_h (sum_s p[s] X[b,s]) + bv_h
// i.e. the two (B*S x D x D) projections collapse into two HBM-bound sweeps over X (bf16 [B, R, D]):
// sc_cls_scores (X . vec^T, H <= 16 vectors) and sc_cls_pool_fwd (masked softmax + p-weighted row sum).
// The backward uses the same two sweeps: dp = X . dm^T, then dX / da in sc_cls_pool_bwd.
#include "sc_common.h"

namespace {

constexpr int MAXH = 16;

// sum NH per-lane values over the 64 lanes with NH - 1 + 6 - log2(NH) shuffles instead of 6 NH: each halving round trades half of the
// values with the partner lane; afterwards lane L holds the total of value index bits(L) (bit 5 = most significant index bit)
template <int NH>
__device__ __forceinline__ float wave_sum_multi(float (&v)[NH], int lane, int& which) {
    int n = NH, off = 32;
    which = 0;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
        if (n > 1) {
            const bool hi = (lane & off) != 0;
            const int half = n >> 1;
#pragma unroll
            for (int i = 0; i < NH / 2; ++i) {
                if (i < half) {
                    const float keep = hi ? v[i + half] : v[i];
                    const float send = hi ? v[i] : v[i + half];
                    v[i] = keep + __shfl_xor(send, off);
                }
            }
            which = which * 2 + (hi ? 1 : 0);
            n = half;
        } else {
            v[0] += __shfl_xor(v[0], off);
        }
        off >>= 1;
    }
    return v[0];
}

// scores[b, h, s] = vec[b?, h, :] . X[b, s, :]      one wave per row, lanes across D.  The NH head vectors are row-invariant:
// each lane keeps its slices of them in registers (NH x NCH float4) instead of re-reading NH x 16 B per 8 B of X, and the
// wave's rows of a block are loaded before any of them is reduced (more loads in flight).
template <int NH, int NCH>     // NCH: 4-element chunks per lane (D <= 256 NCH)
__global__ __launch_bounds__(256) void cls_scores_kernel(const uint16_t* __restrict__ X,
                                                         const float* __restrict__ vec, int64_t vec_bstride,
                                                         float* __restrict__ scores, int R, int D, int Htot, int h0) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = D >> 2;              // 4-element chunks
    const float* vb = vec + (int64_t)b * vec_bstride + (int64_t)h0 * D;      // heads h0 .. h0 + NH - 1 of Htot
    // the four waves need the same slices: one cooperative copy into LDS (NH D floats), then registers from there
    __shared__ __attribute__((aligned(16))) float vs[NH * 256 * NCH];
    for (int i = threadIdx.x; i < NH * nchunks; i += 256) *(f32x4*)(vs + i * 4) = *(const f32x4*)(vb + (int64_t)i * 4);
    __syncthreads();
    f32x4 vr[NH][NCH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + 64 * c;
            vr[h][c] = ch < nchunks ? *(const f32x4*)(vs + (h * nchunks + ch) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    constexpr int ROWS = 64, RPW = ROWS / 4;          // rows per block / per wave
    const int s0 = blockIdx.x * ROWS + wave * RPW;
#pragma unroll
    for (int g = 0; g < RPW; g += 4) {
        uint2 u[4][NCH];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = min(s0 + g + r, R - 1);
            const uint16_t* xr = X + ((int64_t)b * R + s) * D;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = lane + 64 * c;
                u[r][c] = ch < nchunks ? *(const uint2*)(xr + ch * 4) : uint2{0u, 0u};
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = s0 + g + r;
            float acc[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) acc[h] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float x0 = bflo(u[r][c].x), x1 = bfhi(u[r][c].x), x2 = bflo(u[r][c].y), x3 = bfhi(u[r][c].y);
#pragma unroll
                for (int h = 0; h < NH; ++h) acc[h] += x0 * vr[h][c][0] + x1 * vr[h][c][1] + x2 * vr[h][c][2] + x3 * vr[h][c][3];
            }
            int h;
            const float t = wave_sum_multi<NH>(acc, lane, h);          // lane holds head h's total once its low 6 - log2(NH) bits are 0
            constexpr int LOWMASK = 63 >> (NH == 1 ? 0 : NH == 2 ? 1 : NH == 4 ? 2 : 3);
            if ((lane & LOWMASK) == 0 && s < R) scores[((int64_t)b * Htot + h0 + h) * R + s] = t;
        }
    }
}

// grid (D/64, B): masked softmax over s < len (redundant per block, tiny) then
// m[b,h,d] = sum_s p[h,s] X[b,s,d] for this block's 64 columns.
template <int NH>     // NH = H (heads), one of 1, 2, 4, 8, 16
__global__ __launch_bounds__(256) void cls_pool_fwd_kernel(const uint16_t* __restrict__ X,
                                                           const float* __restrict__ scores,
                                                           const int32_t* __restrict__ len, float* __restrict__ p,
                                                           float* __restrict__ m, int R, int D, int H,
                                                           const float* __restrict__ mult, float* __restrict__ psum) {
    extern __shared__ float sm[];            // p[H][R] then red[4][H][64]
    float* ps = sm;
    float* red = sm + H * R;
    const int b = blockIdx.y, d0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = max(1, min(len[b], R));
    for (int h = wave; h < H; h += 4) {
        const float* sr = scores + ((int64_t)b * H + h) * R;
        float mx = -INFINITY;
        for (int s = lane; s < n; s += 64) mx = fmaxf(mx, sr[s]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int s = lane; s < n; s += 64) {
            const float e = __expf(sr[s] - mx);
            ps[h * R + s] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float kept = 0.f;
        for (int s = lane; s < R; s += 64) {
            const float v = s < n ? ps[h * R + s] * inv : 0.f;
            // attention-weight dropout (train mode): the pooling uses p * mult (mult = 0 or 1 / (1 - p_drop)), p itself is kept
            const float w = mult ? v * mult[((int64_t)b * H + h) * R + s] : v;
            ps[h * R + s] = w;
            kept += w;
            if (blockIdx.x == 0) p[((int64_t)b * H + h) * R + s] = v;
        }
        if (psum) {                          // sum_s p mult: the weight of the value bias (dropped weights no longer sum to 1)
            kept = wave_sum(kept);
            if (blockIdx.x == 0 && lane == 0) psum[(int64_t)b * H + h] = kept;
        }
    }
    __syncthreads();
    // pooling: a lane owns 8 columns (one 16-byte load per row) of every 8th row of its wave: 32 rows of 128 B per sweep of the
    // workgroup instead of 4 rows of 2-byte elements
    float acc[NH][8];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
    const int cchunk = lane & 7, rsub = lane >> 3;
    const uint16_t* xc = X + (int64_t)b * R * D + d0 + cchunk * 8;
    for (int s = wave * 8 + rsub; s < n; s += 32) {
        const uint4 u = *(const uint4*)(xc + (int64_t)s * D);
        const float x[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float w = ps[h * R + s];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[h][e] = fmaf(w, x[e], acc[h][e]);
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[h][e];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (rsub == 0) red[(wave * H + h) * 64 + cchunk * 8 + e] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < H * 64; i += 256) {
        const int h = i >> 6, d = i & 63;
        const float v = (red[(0 * H + h) * 64 + d] + red[(1 * H + h) * 64 + d]) +
                        (red[(2 * H + h) * 64 + d] + red[(3 * H + h) * 64 + d]);
        m[((int64_t)b * H + h) * D + d0 + d] = v;
    }
}

// grid (D/64, B).  ds[h,s] = p (dp - sum_s p dp);  dX[b,s,d] = sum_h p dm[b,h,d] + ds a[h,d];
// da_partial[b,h,d] = sum_s ds[h,s] X[b,s,d]
template <int NH>     // NH = H
__global__ __launch_bounds__(256) void cls_pool_bwd_kernel(const uint16_t* __restrict__ X,
                                                           const float* __restrict__ p,
                                                           const float* __restrict__ dp,
                                                           const float* __restrict__ dm,
                                                           const float* __restrict__ a,
                                                           const int32_t* __restrict__ len, float* __restrict__ dX,
                                                           float* __restrict__ da_partial, int R, int D, int H,
                                                           const float* __restrict__ mult, const float* __restrict__ cbias) {
    extern __shared__ float sm[];            // ps[H][R] (p * mult: the weights the pooling used), dss[H][R], red[4][H][64]
    float* ps = sm;
    float* dss = sm + H * R;
    float* red = sm + 2 * H * R;
    const int b = blockIdx.y, d0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = max(1, min(len[b], R));
    for (int h = wave; h < H; h += 4) {
        const float* pr = p + ((int64_t)b * H + h) * R;
        const float* dpr = dp + ((int64_t)b * H + h) * R;
        const float* mr = mult ? mult + ((int64_t)b * H + h) * R : nullptr;
        // cbias given: dp is the RAW score gradient X . dm; the gradient of the attention weight in front of the dropout is
        // (dp + cbias[b,h]) * mult (cbias = the value-bias path through sum_s p mult).  Otherwise dp arrives already in that form.
        const float cb = cbias ? cbias[(int64_t)b * H + h] : 0.f;
        auto dpe = [&](int s) { return cbias ? (dpr[s] + cb) * (mr ? mr[s] : 1.f) : dpr[s]; };
        float dot = 0.f;
        for (int s = lane; s < n; s += 64) dot += pr[s] * dpe(s);
        dot = wave_sum(dot);
        for (int s = lane; s < R; s += 64) {
            const float pv = s < n ? pr[s] : 0.f;
            ps[h * R + s] = mr ? pv * mr[s] : pv;
            dss[h * R + s] = s < n ? pv * (dpe(s) - dot) : 0.f;
        }
    }
    __syncthreads();
    // a lane owns 4 columns (one 8-byte load, one 16-byte store per row) of every 4th row of its wave: 16 rows per sweep
    const int cchunk = lane & 15, rsub = lane >> 4, c0 = d0 + cchunk * 4;
    f32x4 dmv[NH], av[NH], acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        dmv[h] = *(const f32x4*)(dm + ((int64_t)b * H + h) * D + c0);
        av[h] = *(const f32x4*)(a + (int64_t)h * D + c0);
        acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const uint16_t* xc = X + (int64_t)b * R * D + c0;
    float* gx = dX + (int64_t)b * R * D + c0;
    for (int s = wave * 4 + rsub; s < R; s += 16) {
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (s < n) {
            const uint2 u = *(const uint2*)(xc + (int64_t)s * D);
            const f32x4 x = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y)};
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float dsv = dss[h * R + s], pw = ps[h * R + s];
                g += pw * dmv[h] + dsv * av[h];
                acc[h] += dsv * x;
            }
        }
        *(f32x4*)(gx + (int64_t)s * D) = g;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[h][e];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (rsub == 0) red[(wave * H + h) * 64 + cchunk * 4 + e] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < H * 64; i += 256) {
        const int h = i >> 6, d = i & 63;
        const float v = (red[(0 * H + h) * 64 + d] + red[(1 * H + h) * 64 + d]) +
                        (red[(2 * H + h) * 64 + d] + red[(3 * H + h) * 64 + d]);
        da_partial[((int64_t)b * H + h) * D + d0 + d] = v;
    }
}

}  // namespace

extern "C" int sc_cls_scores(const sc_bf16* X, const float* vec, int64_t vec_bstride, float* scores, int32_t B,
                             int32_t R, int32_t D, int32_t H, void* stream) {
    SC_CHECK(X && vec && scores, "sc_cls_scores: null pointer");
    SC_CHECK(D % 4 == 0 && ((uintptr_t)X % 8) == 0 && ((uintptr_t)vec % 16) == 0 && vec_bstride % 4 == 0,
             "sc_cls_scores: alignment");
    SC_CHECK(D <= 1024, "sc_cls_scores: D=%d must be <= 1024", D);
    dim3 grid((R + 63) / 64, B);
    hipStream_t s = (hipStream_t)stream;
    const int nch = (D / 4 + 63) / 64;
#define SC_CS(NH, NC) hipLaunchKernelGGL((cls_scores_kernel<NH, NC>), grid, dim3(256), 0, s, X, vec, vec_bstride, scores, R, D, H, h0)
#define SC_CS_H(NH)                                                        \
    do {                                                                   \
        if (nch <= 1) SC_CS(NH, 1); else if (nch == 2) SC_CS(NH, 2); else if (nch == 3) SC_CS(NH, 3); else SC_CS(NH, 4); \
    } while (0)
    int h0 = 0;
    switch (H) {
        case 1: SC_CS_H(1); break;
        case 2: SC_CS_H(2); break;
        case 4: SC_CS_H(4); break;
        case 8: SC_CS_H(8); break;
        case 16:                                     // 16 heads x 4 chunks do not fit the register file: two passes of 8
            SC_CS_H(8);
            h0 = 8;
            SC_CS_H(8);
            break;
        default: sc_set_error("sc_cls_scores: H=%d not in {1,2,4,8,16}", H); return -1;
    }
#undef SC_CS_H
#undef SC_CS
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cls_pool_fwd(const sc_bf16* X, const float* scores, const int32_t* len, float* p, float* m,
                               int32_t B, int32_t R, int32_t D, int32_t H, const float* mult, float* psum, void* stream) {
    SC_CHECK(X && scores && len && p && m, "sc_cls_pool_fwd: null pointer");
    SC_CHECK(D % 64 == 0 && H >= 1 && H <= MAXH && ((uintptr_t)X % 16) == 0, "sc_cls_pool_fwd: D %% 64, H <= 16, 16-byte aligned X required (D=%d H=%d)", D, H);
    const size_t lds = (size_t)(H * R + 4 * H * 64) * sizeof(float);
    SC_CHECK(lds <= 64 * 1024, "sc_cls_pool_fwd: H*R too large for LDS");
    #define SC_CPF(NH) hipLaunchKernelGGL(cls_pool_fwd_kernel<NH>, dim3(D / 64, B), dim3(256), lds, (hipStream_t)stream, X, scores, len, p, m, R, D, H, mult, psum)
    switch (H) {
        case 1: SC_CPF(1); break;
        case 2: SC_CPF(2); break;
        case 4: SC_CPF(4); break;
        case 8: SC_CPF(8); break;
        case 16: SC_CPF(16); break;
        default: sc_set_error("sc_cls_pool_fwd: H=%d not in {1,2,4,8,16}", H); return -1;
    }
#undef SC_CPF
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cls_pool_bwd(const sc_bf16* X, const float* p, const float* dp, const float* dm, const float* a,
                               const int32_t* len, float* dX, float* da_partial, int32_t B, int32_t R, int32_t D,
                               int32_t H, const float* mult, const float* cbias, void* stream) {
    SC_CHECK(X && p && dp && dm && a && len && dX && da_partial, "sc_cls_pool_bwd: null pointer");
    SC_CHECK(D % 64 == 0 && H >= 1 && H <= MAXH, "sc_cls_pool_bwd: D %% 64, H <= 16 required");
    SC_CHECK(((uintptr_t)X % 8) == 0 && ((uintptr_t)dX % 16) == 0 && ((uintptr_t)dm % 16) == 0 && ((uintptr_t)a % 16) == 0, "sc_cls_pool_bwd: alignment");
    const size_t lds = (size_t)(2 * H * R + 4 * H * 64) * sizeof(float);
    SC_CHECK(lds <= 64 * 1024, "sc_cls_pool_bwd: H*R too large for LDS");
    #define SC_CPB(NH) hipLaunchKernelGGL(cls_pool_bwd_kernel<NH>, dim3(D / 64, B), dim3(256), lds, (hipStream_t)stream, X, p, dp, dm, a, len, dX, da_partial, R, D, H, mult, cbias)
    switch (H) {
        case 1: SC_CPB(1); break;
        case 2: SC_CPB(2); break;
        case 4: SC_CPB(4); break;
        case 8: SC_CPB(8); break;
        case 16: SC_CPB(16); break;
        default: sc_set_error("sc_cls_pool_bwd: H=%d not in {1,2,4,8,16}", H); return -1;
    }
#undef SC_CPB
    SC_LAUNCH_CHECK();
    return 0;
}
