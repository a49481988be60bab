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

// scores[b, h, s] = vec[b?, h, :] . X[b, s, :]      one wave per row, lanes across D
template <int NH>
__global__ __launch_bounds__(256) void cls_scores_kernel(const uint16_t* __restrict__ X,
                                                         const float* __restrict__ vec, int64_t vec_bstride,
                                                         float* __restrict__ scores, int R, int D) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = D >> 2;              // 4-element chunks
    const float* vb = vec + (int64_t)b * vec_bstride;
    const int rows_per_block = 16;
    const int s_begin = blockIdx.x * rows_per_block;
    for (int s = s_begin + wave; s < min(R, s_begin + rows_per_block); s += 4) {
        const uint16_t* xr = X + ((int64_t)b * R + s) * D;
        float acc[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[h] = 0.f;
        for (int ch = lane; ch < nchunks; ch += 64) {
            const uint2 u = *(const uint2*)(xr + ch * 4);
            const float x0 = bflo(u.x), x1 = bfhi(u.x), x2 = bflo(u.y), x3 = bfhi(u.y);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const f32x4 v = *(const f32x4*)(vb + (int64_t)h * D + ch * 4);
                acc[h] += x0 * v[0] + x1 * v[1] + x2 * v[2] + x3 * v[3];
            }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float t = wave_sum(acc[h]);
            if (lane == 0) scores[((int64_t)b * NH + h) * R + s] = t;
        }
    }
}

// grid (D/64, B): masked softmax over s < len (redundant per block, tiny) then
// m[b,h,d] = sum_s p[h,s] X[b,s,d] for this block's 64 columns.
__global__ __launch_bounds__(256) void cls_pool_fwd_kernel(const uint16_t* __restrict__ X,
                                                           const float* __restrict__ scores,
                                                           const int32_t* __restrict__ len, float* __restrict__ p,
                                                           float* __restrict__ m, int R, int D, int H,
                                                           const float* __restrict__ mult) {
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
        for (int s = lane; s < R; s += 64) {
            const float v = s < n ? ps[h * R + s] * inv : 0.f;
            // attention-weight dropout (train mode): the pooling uses p * mult (mult = 0 or 1 / (1 - p_drop)), p itself is kept
            ps[h * R + s] = mult ? v * mult[((int64_t)b * H + h) * R + s] : v;
            if (blockIdx.x == 0) p[((int64_t)b * H + h) * R + s] = v;
        }
    }
    __syncthreads();
    float acc[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) acc[h] = 0.f;
    const uint16_t* xc = X + (int64_t)b * R * D + d0 + lane;
    for (int s = wave; s < n; s += 4) {
        const float x = bf2f(xc[(int64_t)s * D]);
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < H) acc[h] += ps[h * R + s] * x;
    }
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
        if (h < H) red[(wave * H + h) * 64 + lane] = acc[h];
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
__global__ __launch_bounds__(256) void cls_pool_bwd_kernel(const uint16_t* __restrict__ X,
                                                           const float* __restrict__ p,
                                                           const float* __restrict__ dp,
                                                           const float* __restrict__ dm,
                                                           const float* __restrict__ a,
                                                           const int32_t* __restrict__ len, float* __restrict__ dX,
                                                           float* __restrict__ da_partial, int R, int D, int H,
                                                           const float* __restrict__ mult) {
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
        float dot = 0.f;
        for (int s = lane; s < n; s += 64) dot += pr[s] * dpr[s];
        dot = wave_sum(dot);
        for (int s = lane; s < R; s += 64) {
            const float pv = s < n ? pr[s] : 0.f;
            ps[h * R + s] = mult ? pv * mult[((int64_t)b * H + h) * R + s] : pv;
            dss[h * R + s] = s < n ? pv * (dpr[s] - dot) : 0.f;     // dp arrives already multiplied by mult
        }
    }
    __syncthreads();
    float dmv[MAXH], av[MAXH], acc[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) {
        dmv[h] = h < H ? dm[((int64_t)b * H + h) * D + d0 + lane] : 0.f;
        av[h] = h < H ? a[(int64_t)h * D + d0 + lane] : 0.f;
        acc[h] = 0.f;
    }
    const uint16_t* xc = X + (int64_t)b * R * D + d0 + lane;
    float* gx = dX + (int64_t)b * R * D + d0 + lane;
    for (int s = wave; s < R; s += 4) {
        float g = 0.f;
        if (s < n) {
            const float x = bf2f(xc[(int64_t)s * D]);
#pragma unroll
            for (int h = 0; h < MAXH; ++h)
                if (h < H) {
                    const float dsv = dss[h * R + s];
                    g += ps[h * R + s] * dmv[h] + dsv * av[h];
                    acc[h] += dsv * x;
                }
        }
        gx[(int64_t)s * D] = g;
    }
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
        if (h < H) red[(wave * H + h) * 64 + lane] = acc[h];
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
    dim3 grid((R + 15) / 16, B);
    hipStream_t s = (hipStream_t)stream;
    switch (H) {
        case 1: hipLaunchKernelGGL(cls_scores_kernel<1>, grid, dim3(256), 0, s, X, vec, vec_bstride, scores, R, D); break;
        case 2: hipLaunchKernelGGL(cls_scores_kernel<2>, grid, dim3(256), 0, s, X, vec, vec_bstride, scores, R, D); break;
        case 4: hipLaunchKernelGGL(cls_scores_kernel<4>, grid, dim3(256), 0, s, X, vec, vec_bstride, scores, R, D); break;
        case 8: hipLaunchKernelGGL(cls_scores_kernel<8>, grid, dim3(256), 0, s, X, vec, vec_bstride, scores, R, D); break;
        case 16: hipLaunchKernelGGL(cls_scores_kernel<16>, grid, dim3(256), 0, s, X, vec, vec_bstride, scores, R, D); break;
        default: sc_set_error("sc_cls_scores: H=%d not in {1,2,4,8,16}", H); return -1;
    }
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cls_pool_fwd(const sc_bf16* X, const float* scores, const int32_t* len, float* p, float* m,
                               int32_t B, int32_t R, int32_t D, int32_t H, const float* mult, void* stream) {
    SC_CHECK(X && scores && len && p && m, "sc_cls_pool_fwd: null pointer");
    SC_CHECK(D % 64 == 0 && H >= 1 && H <= MAXH, "sc_cls_pool_fwd: D %% 64, H <= 16 required (D=%d H=%d)", D, H);
    const size_t lds = (size_t)(H * R + 4 * H * 64) * sizeof(float);
    SC_CHECK(lds <= 64 * 1024, "sc_cls_pool_fwd: H*R too large for LDS");
    hipLaunchKernelGGL(cls_pool_fwd_kernel, dim3(D / 64, B), dim3(256), lds, (hipStream_t)stream, X, scores, len, p, m, R, D, H, mult);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cls_pool_bwd(const sc_bf16* X, const float* p, const float* dp, const float* dm, const float* a,
                               const int32_t* len, float* dX, float* da_partial, int32_t B, int32_t R, int32_t D,
                               int32_t H, const float* mult, void* stream) {
    SC_CHECK(X && p && dp && dm && a && len && dX && da_partial, "sc_cls_pool_bwd: null pointer");
    SC_CHECK(D % 64 == 0 && H >= 1 && H <= MAXH, "sc_cls_pool_bwd: D %% 64, H <= 16 required");
    const size_t lds = (size_t)(2 * H * R + 4 * H * 64) * sizeof(float);
    SC_CHECK(lds <= 64 * 1024, "sc_cls_pool_bwd: H*R too large for LDS");
    hipLaunchKernelGGL(cls_pool_bwd_kernel, dim3(D / 64, B), dim3(256), lds, (hipStream_t)stream, X, p, dp, dm, a, len, dX, da_partial, R, D, H, mult);
    SC_LAUNCH_CHECK();
    return 0;
}
