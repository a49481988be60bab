// Continuous integrate-and-fire accumulation (avssl/module/cif.py:157-240, the scatter_add_ form of the reference's frame loop;
// cascaded+/hybrid+ branches), forward and backward.
//
// Per utterance, frame s with weight alpha_s and cumulative weight c_s = sum_{j<=s} alpha_j (given: torch.cumsum, so the slot
// boundaries are exactly the host framework's):
//     right_s = clip(floor(c_s / thr), 0, T) ;  left_s = right_{s-1} (0 for s = 0) ;  fire_s = right_s - left_s
//     rw_s = fire_s > 0 ? c_s - right_s thr : 0 ;  extra_s = max(fire_s - 1, 0) ;  lw_s = alpha_s - rw_s - extra_s thr
//     out[left_s] += lw_s x_s ;  out[left_s + j] += thr x_s  (j = 1 .. extra_s) ;  out[right_s] += rw_s x_s          (T + 1 slots)
// Slots are visited in non-decreasing order, so a thread that owns 4 channels walks the frames once with a running accumulator
// and writes every slot exactly once, in frame order: deterministic (the reference's scatter_add_ uses float atomics) and one
// pass over x.  Backward (indices are constants of the graph, as in the reference's no_grad block):
//     dx_s = lw_s g[left_s] + rw_s g[right_s] + thr sum_j g[left_s + j]
//     d alpha_s (direct) = x_s . g[left_s]            d c_s = fire_s > 0 ? x_s . (g[right_s] - g[left_s]) : 0
// (the cumsum's own backward turns d c into the remaining part of d alpha); the two dot products are reduced over the wave and
// written per channel block: pa / pb [nblk, B, S].
// HBM-bound: x once forward; x once + dx once backward.  One wave per (utterance, 256-channel block).
#include "sc_common.h"

namespace {

struct frame_meta {
    int left, right;
    float lw, rw;
};
constexpr int MAXS = 2048;      // frames per utterance held in LDS (32 KiB)

// every lane derives the metadata of frames lane, lane + 64, ... (each needs only c_s and c_{s-1}); the frame loop then reads
// one LDS word group per frame instead of two dependent global loads
__device__ __forceinline__ void stage_meta(frame_meta* sm, const float* __restrict__ alpha, const float* __restrict__ csum, int S,
                                           float thr, int T) {
    for (int s = threadIdx.x; s < S; s += 64) {
        const float c = csum[s];
        frame_meta m;
        m.right = min(max((int)floorf(c / thr), 0), T);
        m.left = s > 0 ? min(max((int)floorf(csum[s - 1] / thr), 0), T) : 0;
        const int fire = m.right - m.left;
        m.rw = fire > 0 ? c - (float)m.right * thr : 0.f;
        m.lw = alpha[s] - m.rw - (float)max(fire - 1, 0) * thr;
        sm[s] = m;
    }
    __syncthreads();
}

constexpr int UNROLL = 8;

// the backward's arithmetic spelled out (no compiler-chosen contraction): the sequential and the parallel kernels, fp32 and bf16 frames,
// must produce the same bits
__device__ __forceinline__ float dot4(const float4 v, const float4 g) {
    return fmaf(v.w, g.w, fmaf(v.z, g.z, fmaf(v.y, g.y, __fmul_rn(v.x, g.x))));
}
__device__ __forceinline__ float4 sub4(const float4 a, const float4 b) {
    return float4{__fsub_rn(a.x, b.x), __fsub_rn(a.y, b.y), __fsub_rn(a.z, b.z), __fsub_rn(a.w, b.w)};
}
__device__ __forceinline__ float4 scale4(const float s, const float4 g) {
    return float4{__fmul_rn(s, g.x), __fmul_rn(s, g.y), __fmul_rn(s, g.z), __fmul_rn(s, g.w)};
}
// d + rw gr + thr mid
__device__ __forceinline__ float4 fire4(const float4 d, const float rw, const float4 gr, const float thr, const float4 mid) {
    return float4{fmaf(rw, gr.x, fmaf(thr, mid.x, d.x)), fmaf(rw, gr.y, fmaf(thr, mid.y, d.y)), fmaf(rw, gr.z, fmaf(thr, mid.z, d.z)),
                  fmaf(rw, gr.w, fmaf(thr, mid.w, d.w))};
}


// the frames arrive as fp32 [B, S, C] or as the bf16 rows the attention block's LayerNorm wrote (round 4: row pitch of the block's
// buffer, no fp32 copy of the activations in between); the gradient leaves the same way
// A prefetched row stays in registers AS LOADED (float4 / the 8 raw bytes of 4 bf16) and is converted where it is consumed: converting
// at load time would make the wave wait for the load it has just issued (177 instead of 75 us for the forward of 64 x 499 x 1024).
template <typename XT> struct RawRow;
template <> struct RawRow<float> { typedef float4 type; };
template <> struct RawRow<uint16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 row_f4(const float4& r) { return r; }
__device__ __forceinline__ float4 row_f4(const uint2& u) { return float4{bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y)}; }
__device__ __forceinline__ void st4(float* p, const float4& v) { *(float4*)p = v; }
__device__ __forceinline__ void st4(uint16_t* p, const float4& v) { *(uint2*)p = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w)); }

template <typename XT>
__device__ __forceinline__ void load_batch(typename RawRow<XT>::type (&xv)[UNROLL], const XT* __restrict__ xb, int s0, int S, int C, bool active) {
    typedef typename RawRow<XT>::type R;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        R r;
        __builtin_memset(&r, 0, sizeof(R));
        if (active && s0 + u < S) r = *(const R*)(xb + (int64_t)(s0 + u) * C);
        xv[u] = r;
    }
}

// frames are consumed in batches of UNROLL rows; the next batch's loads are issued before the current one is processed
template <typename XT>
__global__ __launch_bounds__(64) void cif_fwd_kernel(const XT* __restrict__ x, int64_t xbs, const float* __restrict__ alpha,
                                                     const float* __restrict__ csum, float* __restrict__ out, int S, int C, int T,
                                                     float thr) {
    const int b = blockIdx.y, c0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const bool active = c0 < C;
    const XT* xb = x + (int64_t)b * xbs + (active ? c0 : 0);
    float* ob = out + (int64_t)b * (T + 1) * C + (active ? c0 : 0);
    __shared__ frame_meta sm[MAXS];
    typename RawRow<XT>::type xv[UNROLL], xn[UNROLL];
    load_batch(xv, xb, 0, S, C, active);
    stage_meta(sm, alpha + (int64_t)b * S, csum + (int64_t)b * S, S, thr, T);
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int s0 = 0; s0 < S; s0 += UNROLL) {
        load_batch(xn, xb, s0 + UNROLL, S, C, active);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (s0 + u < S) {
                const frame_meta m = sm[s0 + u];
                const int extra = m.right - m.left - 1;
                const float4 v = row_f4(xv[u]);
                acc.x = fmaf(m.lw, v.x, acc.x); acc.y = fmaf(m.lw, v.y, acc.y);
                acc.z = fmaf(m.lw, v.z, acc.z); acc.w = fmaf(m.lw, v.w, acc.w);
                if (m.right != m.left) {
                    if (active) {
                        *(float4*)(ob + (int64_t)m.left * C) = acc;
                        const float4 w = scale4(thr, v);
                        for (int j = 1; j <= extra; ++j) *(float4*)(ob + (int64_t)(m.left + j) * C) = w;
                    }
                    acc = scale4(m.rw, v);
                    cur = m.right;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) xv[u] = xn[u];
    }
    if (active) {
        *(float4*)(ob + (int64_t)cur * C) = acc;
        const float4 z = {0.f, 0.f, 0.f, 0.f};
        for (int t = cur + 1; t <= T; ++t) *(float4*)(ob + (int64_t)t * C) = z;
    }
}

// dx: frame s of utterance b at dx + b dxbs + s C; the zlo rows in front of an utterance's frames and the zhi rows behind them are
// zero-filled (the rows of the attention block's buffer that are not frames: CLS slot, padding)
template <typename XT, typename DT>
__global__ __launch_bounds__(64) void cif_bwd_kernel(const XT* __restrict__ x, int64_t xbs, const float* __restrict__ alpha,
                                                     const float* __restrict__ csum, const float* __restrict__ g,
                                                     DT* __restrict__ dx, int64_t dxbs, int zlo, int zhi, float* __restrict__ pa,
                                                     float* __restrict__ pb, int B, int S, int C, int T, float thr) {
    const int b = blockIdx.y, lane = threadIdx.x, c0 = (blockIdx.x * 64 + lane) * 4;
    const bool active = c0 < C;
    const XT* xb = x + (int64_t)b * xbs + (active ? c0 : 0);
    DT* dxb = dx + (int64_t)b * dxbs + (active ? c0 : 0);
    if (active) {
        for (int r = -zlo; r < 0; ++r) st4(dxb + (int64_t)r * C, float4{0.f, 0.f, 0.f, 0.f});
        for (int r = S; r < S + zhi; ++r) st4(dxb + (int64_t)r * C, float4{0.f, 0.f, 0.f, 0.f});
    }
    const float* gb = g + (int64_t)b * (T + 1) * C + (active ? c0 : 0);
    float* pab = pa + ((int64_t)blockIdx.x * B + b) * S;
    float* pbb = pb + ((int64_t)blockIdx.x * B + b) * S;
    const float4 zero = {0.f, 0.f, 0.f, 0.f};
    __shared__ frame_meta sm[MAXS];
    __shared__ float red[2 * UNROLL][65];        // per-lane partial dot products of a batch (row stride 65: conflict-free column sums)
    typename RawRow<XT>::type xv[UNROLL], xn[UNROLL];
    load_batch(xv, xb, 0, S, C, active);
    stage_meta(sm, alpha + (int64_t)b * S, csum + (int64_t)b * S, S, thr, T);
    float4 gl = active ? *(const float4*)gb : zero;                               // g[slot 0]
    float4 gn = active ? *(const float4*)(gb + (int64_t)min(1, T) * C) : zero;    // prefetched g[current slot + 1]
    for (int s0 = 0; s0 < S; s0 += UNROLL) {
        load_batch(xn, xb, s0 + UNROLL, S, C, active);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int s = s0 + u;
            float a = 0.f, bq = 0.f;
            if (s < S) {
                const frame_meta m = sm[s];
                const int extra = m.right - m.left - 1;
                const float4 v = row_f4(xv[u]);
                float4 d = scale4(m.lw, gl);
                a = dot4(v, gl);
                if (m.right != m.left) {
                    const float4 gr = extra == 0 ? gn : (active ? *(const float4*)(gb + (int64_t)m.right * C) : zero);
                    float4 mid = zero;
                    for (int j = 1; j <= extra; ++j) {
                        const float4 t = active ? *(const float4*)(gb + (int64_t)(m.left + j) * C) : zero;
                        mid.x += t.x; mid.y += t.y; mid.z += t.z; mid.w += t.w;
                    }
                    d = fire4(d, m.rw, gr, thr, mid);
                    bq = dot4(v, sub4(gr, gl));
                    gl = gr;
                    gn = active ? *(const float4*)(gb + (int64_t)min(m.right + 1, T) * C) : zero;
                }
                if (active) st4(dxb + (int64_t)s * C, d);
            }
            red[2 * u][lane] = a;
            red[2 * u + 1][lane] = bq;
        }
        __syncthreads();
        if (lane < 2 * UNROLL) {                 // lane k sums row k over the 64 lanes: fixed order, no shuffles in the frame loop
            float t = 0.f;
#pragma unroll 16
            for (int j = 0; j < 64; ++j) t += red[lane][j];
            const int s = s0 + (lane >> 1);
            if (s < S) ((lane & 1) ? pbb : pab)[s] = t;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) xv[u] = xn[u];
    }
}

// ------------------------------------------------------------------------------------------ the same, parallel over slots / frames
// The frame walk above is one dependent chain of S frames per wave (~150 ns a frame: 71 us forward, 128 us backward at S = 499 whatever
// the width).  But an output slot only sums a CONTIGUOUS run of frames - the frame that fired into it (weight rw) and the frames whose
// left slot it is (weights lw), in frame order - and the backward of a frame only reads the slots it touched.  One wave per (slot,
// 256-channel block, utterance) forward, one per (8 frames, 256-channel block, utterance) backward: the same operations on the same
// values in the same order (bitwise the sequential kernels' results), chains of ~S / T frames.
__device__ __forceinline__ int cif_right(float c, float thr, int T) { return min(max((int)floorf(c / thr), 0), T); }

// first s in [0, S) with right_s >= t, or S (right is non-decreasing): the 64 lanes test 64 candidates at a time
__device__ __forceinline__ int cif_first_right_ge(const float* __restrict__ cs, int S, float thr, int T, int t, int lane) {
    int lo = 0, hi = S;                                      // answer in [lo, hi]; hi < S means right_hi >= t is already known
    while (hi > lo) {
        const int step = (hi - lo + 63) >> 6;                // lane l tests the LAST index of its sub-range [lo + l step, lo + (l+1) step)
        const int idx = min(lo + (lane + 1) * step - 1, hi - 1);
        const unsigned long long m = __ballot(cif_right(cs[idx], thr, T) >= t);
        if (m == 0ull) return hi;                            // not even hi - 1: nothing in [lo, hi)
        const int l0 = __ffsll((long long)m) - 1;             // the first sub-range whose end satisfies it holds the answer
        hi = min(lo + (l0 + 1) * step - 1, hi - 1);
        lo = lo + l0 * step;
    }
    return lo;
}

template <typename XT>
__global__ __launch_bounds__(64) void cif_fwd_slot_kernel(const XT* __restrict__ x, int64_t xbs, const float* __restrict__ alpha,
                                                          const float* __restrict__ csum, float* __restrict__ out, int S, int C, int T,
                                                          float thr) {
    const int b = blockIdx.z, t = blockIdx.y, lane = threadIdx.x, c0 = (blockIdx.x * 64 + lane) * 4;
    const bool active = c0 < C;
    const XT* xb = x + (int64_t)b * xbs + (active ? c0 : 0);
    const float* cs = csum + (int64_t)b * S;
    const float* al = alpha + (int64_t)b * S;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    // f = first frame whose LEFT slot is >= t (left_s = right_{s-1}, left_0 = 0)
    int f = 0;
    bool live = true;                                        // does the sequential walk write this slot at all?
    if (t > 0) {
        const int a = cif_first_right_ge(cs, S, thr, T, t, lane);        // the frame that fires into / across slot t
        if (a >= S) {
            live = false;                                     // behind the last slot: zero
        } else {
            f = a + 1;
            const float ca = cs[a];
            const int ra = cif_right(ca, thr, T);
            const float4 v = active ? row_f4(*(const typename RawRow<XT>::type*)(xb + (int64_t)a * C)) : float4{0.f, 0.f, 0.f, 0.f};
            if (ra == t) {
                const float rw = ca - (float)ra * thr;        // (fire > 0 here: left_a < t = right_a)
                acc = scale4(rw, v);
            } else {                                          // left_a < t < right_a: one of the frame's extra slots
                if (active) *(float4*)(out + ((int64_t)b * (T + 1) + t) * C + c0) = scale4(thr, v);
                return;
            }
        }
    }
    if (live) {
        int left = f > 0 ? cif_right(cs[f - 1], thr, T) : 0;
        for (int s2 = f; s2 < S && left == t; ++s2) {
            const float c = cs[s2];
            const int right = cif_right(c, thr, T), fire = right - left;
            const float rw = fire > 0 ? c - (float)right * thr : 0.f;
            const float lw = al[s2] - rw - (float)max(fire - 1, 0) * thr;
            if (active) {
                const float4 v = row_f4(*(const typename RawRow<XT>::type*)(xb + (int64_t)s2 * C));
                acc.x = fmaf(lw, v.x, acc.x); acc.y = fmaf(lw, v.y, acc.y);
                acc.z = fmaf(lw, v.z, acc.z); acc.w = fmaf(lw, v.w, acc.w);
            }
            left = right;
        }
    }
    if (active) *(float4*)(out + ((int64_t)b * (T + 1) + t) * C + c0) = acc;
}

template <typename XT, typename DT>
__global__ __launch_bounds__(64) void cif_bwd_frame_kernel(const XT* __restrict__ x, int64_t xbs, const float* __restrict__ alpha,
                                                           const float* __restrict__ csum, const float* __restrict__ g,
                                                           DT* __restrict__ dx, int64_t dxbs, int zlo, int zhi, float* __restrict__ pa,
                                                           float* __restrict__ pb, int B, int S, int C, int T, float thr) {
    const int b = blockIdx.z, lane = threadIdx.x, c0 = (blockIdx.x * 64 + lane) * 4, s0 = blockIdx.y * UNROLL;
    const bool active = c0 < C;
    const XT* xb = x + (int64_t)b * xbs + (active ? c0 : 0);
    DT* dxb = dx + (int64_t)b * dxbs + (active ? c0 : 0);
    const float* gb = g + (int64_t)b * (T + 1) * C + (active ? c0 : 0);
    const float* cs = csum + (int64_t)b * S;
    const float* al = alpha + (int64_t)b * S;
    float* pab = pa + ((int64_t)blockIdx.x * B + b) * S;
    float* pbb = pb + ((int64_t)blockIdx.x * B + b) * S;
    const float4 zero = {0.f, 0.f, 0.f, 0.f};
    __shared__ float red[2 * UNROLL][65];
    if (active) {                                            // the rows of the buffer that are not frames: first / last block of an utterance
        if (blockIdx.y == 0)
            for (int r = -zlo; r < 0; ++r) st4(dxb + (int64_t)r * C, zero);
        if (s0 + UNROLL >= S)
            for (int r = S; r < S + zhi; ++r) st4(dxb + (int64_t)r * C, zero);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int s2 = s0 + u;
        float a = 0.f, bq = 0.f;
        if (s2 < S) {
            const float c = cs[s2];
            const int right = cif_right(c, thr, T), left = s2 > 0 ? cif_right(cs[s2 - 1], thr, T) : 0, fire = right - left;
            const float rw = fire > 0 ? c - (float)right * thr : 0.f;
            const float lw = al[s2] - rw - (float)max(fire - 1, 0) * thr;
            const int extra = fire - 1;
            const float4 v = active ? row_f4(*(const typename RawRow<XT>::type*)(xb + (int64_t)s2 * C)) : zero;
            const float4 gl = active ? *(const float4*)(gb + (int64_t)left * C) : zero;
            float4 d = scale4(lw, gl);
            a = dot4(v, gl);
            if (right != left) {
                const float4 gr = active ? *(const float4*)(gb + (int64_t)right * C) : zero;
                float4 mid = zero;
                for (int j = 1; j <= extra; ++j) {
                    const float4 tt = active ? *(const float4*)(gb + (int64_t)(left + j) * C) : zero;
                    mid.x += tt.x; mid.y += tt.y; mid.z += tt.z; mid.w += tt.w;
                }
                d = fire4(d, rw, gr, thr, mid);
                bq = dot4(v, sub4(gr, gl));
            }
            if (active) st4(dxb + (int64_t)s2 * C, d);
        }
        red[2 * u][lane] = a;
        red[2 * u + 1][lane] = bq;
    }
    __syncthreads();
    if (lane < 2 * UNROLL) {                     // lane k sums row k over the 64 lanes: fixed order
        float tsum = 0.f;
#pragma unroll 16
        for (int j = 0; j < 64; ++j) tsum += red[lane][j];
        const int s2 = s0 + (lane >> 1);
        if (s2 < S) ((lane & 1) ? pbb : pab)[s2] = tsum;
    }
}

// ------------------------------------------------------------------------------------------ bookkeeping (cif.py:106-175)
// One workgroup per utterance, no host round trip:  a = clip(alpha_raw, 0, 1) with padded frames zeroed ; quantity = sum a ;
// (train, apply_scaling) a *= (thr * target + eps) / quantity ; csum = inclusive scan(a) ; feat_len = clip(floor(sum a / thr), 1, 75) ;
// fired marks.  Sums and the scan run in fp64 and are rounded once: the scaled weights sum to target + 1e-5 by construction, and an
// fp32 summation error of that size would turn floor() into target - 1.
constexpr int PREP_PER = MAXS / 256;

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void cif_prepare_kernel(const float* __restrict__ a_raw, int64_t lda, const uint8_t* __restrict__ pad,
                                                          int64_t ldp, const int64_t* __restrict__ target, int S, float thr, float eps,
                                                          int scale, int max_feat, int T, float* __restrict__ a_clip,
                                                          float* __restrict__ alpha, float* __restrict__ csum, float* __restrict__ quantity,
                                                          float* __restrict__ ratio_out, int64_t* __restrict__ feat_len,
                                                          uint8_t* __restrict__ fired, int* __restrict__ flags) {
    __shared__ double red[4];
    __shared__ double tot[256];
    const int b = blockIdx.x, s0 = threadIdx.x * PREP_PER;
    float a[PREP_PER];
    double loc = 0.0;
#pragma unroll
    for (int i = 0; i < PREP_PER; ++i) {
        const int s = s0 + i;
        float v = 0.f;
        if (s < S && !pad[(int64_t)b * ldp + s]) v = fminf(fmaxf(a_raw[(int64_t)b * lda + s], 0.f), 1.f);
        a[i] = v;
        loc += (double)v;
        if (s < S) a_clip[(int64_t)b * S + s] = v;
    }
    const float q = (float)block_sum_d(loc, red);
    float ratio = 1.f;
    // an utterance whose clipped weights are all zero (q = 0) cannot be rescaled: its weights stay 0 and it yields feat_len 1
    // (the reference divides by zero here; its per-call assert only asks for ONE positive utterance) - no inf / NaN downstream
    if (scale) ratio = q > 0.f ? (thr * (float)target[b] + eps) / q : 0.f;
    loc = 0.0;
#pragma unroll
    for (int i = 0; i < PREP_PER; ++i) {
        a[i] *= ratio;
        loc += (double)a[i];
    }
    tot[threadIdx.x] = loc;
    const double total = block_sum_d(loc, red);            // (its barriers also publish tot[])
    double pre = 0.0;
    for (int j = 0; j < (int)threadIdx.x; ++j) pre += tot[j];
    float prev = (float)pre;                                // csum of the frame before this thread's first one
#pragma unroll
    for (int i = 0; i < PREP_PER; ++i) {
        const int s = s0 + i;
        pre += (double)a[i];
        const float c = (float)pre;
        if (s < S) {
            alpha[(int64_t)b * S + s] = a[i];
            csum[(int64_t)b * S + s] = c;
            const int right = min(max((int)floorf(c / thr), 0), T);
            const int left = s > 0 ? min(max((int)floorf(prev / thr), 0), T) : 0;
            fired[(int64_t)b * S + s] = right > left;
        }
        prev = c;
    }
    if (threadIdx.x == 0) {
        int64_t fl = (int64_t)floorf((float)total / thr);
        const int64_t cap = max_feat < T ? max_feat : T;    // T = slots the host allocated (sized from the targets when it knows them)
        fl = fl < 1 ? 1 : (fl > cap ? cap : fl);
        feat_len[b] = fl;
        quantity[b] = q;
        ratio_out[b] = ratio;
        if (q > 0.f) {
            atomicAdd(&flags[0], 1);                        // cumulative count of utterances with a positive weight sum
            atomicAdd(&flags[5], 1);                        // ... of THIS call
        } else if (scale) {
            atomicAdd(&flags[6], 1);                        // an utterance that could not be rescaled (ratio forced to 0, see above)
        }
        if (scale) {
            int64_t t = target[b];
            t = t < 1 ? 1 : (t > max_feat ? max_feat : t);
            if (t != fl) atomicAdd(&flags[1], 1);           // the host sized the output from target_len
        }
        // the reference asserts (alpha_sum > 0).any() on EVERY call (avssl/module/cif.py:121): the last utterance of the call to arrive
        // (ticket in flags[4]) looks at the call's own count and records a call without any positive utterance in flags[3]; both
        // scratch words are left at 0 for the next call (calls on one stream are ordered)
        __threadfence();
        if (atomicAdd(&flags[4], 1) + 1 == (int)gridDim.x) {
            __threadfence();
            const int pos = atomicExch(&flags[5], 0);
            atomicExch(&flags[4], 0);
            if (pos == 0) atomicAdd(&flags[3], 1);
        }
    }
}

// d a_raw from the fire kernel's per-channel-block partials (pa: direct d alpha, pb: d csum) and d quantity:
//   g'_i = sum_blk pa_i + sum_{j >= i} sum_blk pb_j ;  scaled: d a_k = r g'_k - r <g', a> / Q ; + d quantity ; padded frames 0
__global__ __launch_bounds__(256) void cif_prepare_bwd_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int nblk, int B,
                                                              int S, const float* __restrict__ a_clip, const uint8_t* __restrict__ pad,
                                                              int64_t ldp, const float* __restrict__ ratio, const float* __restrict__ quantity,
                                                              const float* __restrict__ gq, int scale, float* __restrict__ da) {
    __shared__ double red[4];
    __shared__ double tot[256];
    const int b = blockIdx.x, s0 = threadIdx.x * PREP_PER;
    float ga[PREP_PER], gc[PREP_PER];
    double loc = 0.0;
#pragma unroll
    for (int i = 0; i < PREP_PER; ++i) {
        const int s = s0 + i;
        float x = 0.f, y = 0.f;
        if (s < S)
            for (int k = 0; k < nblk; ++k) {
                x += pa[((int64_t)k * B + b) * S + s];
                y += pb[((int64_t)k * B + b) * S + s];
            }
        ga[i] = x;
        gc[i] = y;
        loc += (double)y;
    }
    tot[threadIdx.x] = loc;
    block_sum_d(loc, red);
    double suf = 0.0;                                       // sum of d csum over the frames AFTER this thread's chunk
    for (int j = threadIdx.x + 1; j < 256; ++j) suf += tot[j];
    double dot = 0.0;
#pragma unroll
    for (int i = PREP_PER - 1; i >= 0; --i) {
        suf += (double)gc[i];
        ga[i] += (float)suf;
        const int s = s0 + i;
        if (s < S) dot += (double)ga[i] * (double)a_clip[(int64_t)b * S + s];
    }
    const float r = ratio[b];
    const float corr = scale ? (float)block_sum_d(dot, red) * r / quantity[b] : 0.f;
    const float g_q = gq ? gq[b] : 0.f;
#pragma unroll
    for (int i = 0; i < PREP_PER; ++i) {
        const int s = s0 + i;
        if (s < S) da[(int64_t)b * S + s] = pad[(int64_t)b * ldp + s] ? 0.f : (r * ga[i] - corr + g_q);
    }
}

// inference-time tail handling (cif.py:244-297): the weight left in slot feat_len fires an extra keyword when it reaches the tail
// threshold (its row is rescaled to a full threshold); rows >= the final length are zeroed.
__global__ __launch_bounds__(256) void cif_tail_kernel(const float* __restrict__ alpha, const float* __restrict__ csum, int S, int C, int T,
                                                       float thr, float tail_thr, int max_feat, int64_t* __restrict__ feat_len,
                                                       float* __restrict__ out, float* __restrict__ factor, uint8_t* __restrict__ extend) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    const float* al = alpha + (int64_t)b * S;
    const float* cs = csum + (int64_t)b * S;
    const int fl = (int)feat_len[b];
    double w = 0.0;
    for (int s = threadIdx.x; s < S; s += 256) {
        const float c = cs[s];
        const int right = min(max((int)floorf(c / thr), 0), T);
        const int left = s > 0 ? min(max((int)floorf(cs[s - 1] / thr), 0), T) : 0;
        const int fire = right - left;
        const float rw = fire > 0 ? c - (float)right * thr : 0.f;
        const float lw = al[s] - rw - (float)max(fire - 1, 0) * thr;
        if (right == fl) w += (double)rw;
        if (left == fl) w += (double)lw;
    }
    const float tw = (float)block_sum_d(w, red);
    const bool ext = tw >= tail_thr;
    const float f = ext ? thr / tw : 1.f;
    const int fl_new = min(fl + (ext ? 1 : 0), max_feat);
    float* ob = out + (int64_t)b * (T + 1) * C;
    if (ext && fl <= T)
        for (int c = threadIdx.x; c < C; c += 256) ob[(int64_t)fl * C + c] *= f;
    __syncthreads();
    for (int64_t i = (int64_t)fl_new * C + threadIdx.x; i < (int64_t)(T + 1) * C; i += 256) ob[i] = 0.f;
    if (threadIdx.x == 0) {
        feat_len[b] = fl_new;
        factor[b] = f;
        extend[b] = ext;
    }
}

// ------------------------------------------------------------------------------------------ weight head (cif.py:106-129)
// The reference's weight generator ends  Conv1d -> Dropout(0.5) -> ReLU -> Dropout(0.5) -> Linear(C, 1) -> Sigmoid.  Everything behind
// the conv GEMM is one row kernel here (was: two dropout launches, a ReLU, a vendor GEMV, a sigmoid and casts, forward and backward):
//   alpha[row] = sigmoid(b + sum_c w[c] m2(row, c) relu(m1(row, c) y[row, c]))
// m1 / m2 = hash dropout multipliers (0 or 1 / (1 - p); keep bits = sc_keep8 on row * C + c, the library's stateless masks), p = 0 in eval.
struct CifHead {
    const float* y; int64_t ldy; const float* w; const float* bias; int rows, C;
    float sc1, sc2; uint32_t thr1, thr2, seed1, seed2;
};
__device__ __forceinline__ f32x4 cif_head_mult(uint32_t idx, uint32_t seed, uint32_t thr, float sc) {      // idx % 4 == 0
    if (!thr) return f32x4{1.f, 1.f, 1.f, 1.f};
    const uint32_t k = sc_keep8(idx & ~7u, seed, thr) >> (idx & 4u);
    return f32x4{(k & 1u) ? sc : 0.f, (k & 2u) ? sc : 0.f, (k & 4u) ? sc : 0.f, (k & 8u) ? sc : 0.f};
}

__global__ __launch_bounds__(256) void cif_head_fwd_kernel(const CifHead p, float* __restrict__ alpha) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.rows) return;
    const float* yr = p.y + (int64_t)row * p.ldy;
    float s = 0.f;
    for (int c = lane * 4; c < p.C; c += 256) {
        const f32x4 yv = *(const f32x4*)(yr + c), wv = *(const f32x4*)(p.w + c);
        const uint32_t idx = (uint32_t)row * (uint32_t)p.C + (uint32_t)c;
        const f32x4 m1 = cif_head_mult(idx, p.seed1, p.thr1, p.sc1), m2 = cif_head_mult(idx, p.seed2, p.thr2, p.sc2);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += wv[e] * m2[e] * fmaxf(m1[e] * yv[e], 0.f);
    }
    s = wave_sum(s);
    if (lane == 0) alpha[row] = 1.f / (1.f + __expf(-(s + p.bias[0])));
}

// g = dalpha a (1 - a) ; dy[row, c] = g w[c] m2 m1 [y > 0] ; dw_partial[blk][c] = sum over the block's rows of g m2 relu(m1 y) ;
// db_partial[blk] = sum g     (rows blk * 4 + wave, + 4 gridDim.x, ... ; partials reduced in order by sc_colsum_f32)
template <typename DT>
__global__ __launch_bounds__(256) void cif_head_bwd_kernel(const CifHead p, const float* __restrict__ alpha, const float* __restrict__ dalpha,
                                                           DT* __restrict__ dy, int64_t lddy, float* __restrict__ dw_partial,
                                                           float* __restrict__ db_partial) {
    __shared__ float red[4][1024];
    __shared__ float redb[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 dw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < p.rows; row += 4 * gridDim.x) {
        const float a = alpha[row];
        const float g = dalpha[row] * a * (1.f - a);
        db += g;
        const float* yr = p.y + (int64_t)row * p.ldy;
        DT* dr = dy + (int64_t)row * lddy;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane * 4 + i * 256;
            if (c >= p.C) break;
            const f32x4 yv = *(const f32x4*)(yr + c), wv = *(const f32x4*)(p.w + c);
            const uint32_t idx = (uint32_t)row * (uint32_t)p.C + (uint32_t)c;
            const f32x4 m1 = cif_head_mult(idx, p.seed1, p.thr1, p.sc1), m2 = cif_head_mult(idx, p.seed2, p.thr2, p.sc2);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = fmaxf(m1[e] * yv[e], 0.f);
                dw[i][e] += g * m2[e] * h;
                o[e] = h > 0.f ? g * wv[e] * m2[e] * m1[e] : 0.f;
            }
            st4(dr + c, float4{o[0], o[1], o[2], o[3]});
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)&red[wave][lane * 4 + i * 256] = dw[i];
    if (lane == 0) redb[wave] = db;              // every lane of the wave carries the same per-row g: no wave reduction
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256)
        dw_partial[(int64_t)blockIdx.x * p.C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (threadIdx.x == 0) db_partial[blockIdx.x] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
}

}  // namespace

extern "C" int sc_cif_fwd_rows(const void* x, int32_t x_bf16, int64_t xbs, const float* alpha, const float* csum, float* out, int32_t B,
                               int32_t S, int32_t C, int32_t T, float thr, void* stream) {
    SC_CHECK(x && alpha && csum && out, "sc_cif_fwd: null pointer");
    SC_CHECK(B > 0 && S > 0 && S <= MAXS && T >= 0 && C > 0 && C % 4 == 0 && thr > 0.f && xbs >= (int64_t)S * C && xbs % 4 == 0,
             "sc_cif_fwd: B=%d S=%d (<= 2048) C=%d (C %% 4) T=%d thr=%f", B, S, C, T, (double)thr);
    SC_CHECK(((uintptr_t)x % (x_bf16 ? 8 : 16)) == 0 && ((uintptr_t)out % 16) == 0, "sc_cif_fwd: alignment");
    if (!sc_option(6)) {            // one wave per output slot (sc_set_option(6, 1): the sequential frame walk, for A/B)
        const dim3 grid((C + 255) / 256, T + 1, B);
        if (x_bf16) hipLaunchKernelGGL(cif_fwd_slot_kernel<uint16_t>, grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)x, xbs, alpha, csum, out, S, C, T, thr);
        else hipLaunchKernelGGL(cif_fwd_slot_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, (const float*)x, xbs, alpha, csum, out, S, C, T, thr);
        SC_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((C + 255) / 256, B);
    if (x_bf16) hipLaunchKernelGGL(cif_fwd_kernel<uint16_t>, grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)x, xbs, alpha, csum, out, S, C, T, thr);
    else hipLaunchKernelGGL(cif_fwd_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, (const float*)x, xbs, alpha, csum, out, S, C, T, thr);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cif_fwd(const float* x, const float* alpha, const float* csum, float* out, int32_t B, int32_t S, int32_t C, int32_t T,
                          float thr, void* stream) {
    return sc_cif_fwd_rows(x, 0, (int64_t)S * C, alpha, csum, out, B, S, C, T, thr, stream);
}

extern "C" int sc_cif_bwd_rows(const void* x, int32_t x_bf16, int64_t xbs, const float* alpha, const float* csum, const float* g, void* dx,
                               int32_t dx_bf16, int64_t dxbs, int32_t zlo, int32_t zhi, float* pa, float* pb, int32_t B, int32_t S, int32_t C,
                               int32_t T, float thr, void* stream) {
    SC_CHECK(x && alpha && csum && g && dx && pa && pb, "sc_cif_bwd: null pointer");
    SC_CHECK(B > 0 && S > 0 && S <= MAXS && T >= 0 && C > 0 && C % 4 == 0 && thr > 0.f && xbs >= (int64_t)S * C && xbs % 4 == 0 && zlo >= 0 &&
                 zhi >= 0 && dxbs >= (int64_t)(S + zhi) * C && dxbs % 4 == 0,
             "sc_cif_bwd: B=%d S=%d (<= 2048) C=%d (C %% 4) T=%d thr=%f", B, S, C, T, (double)thr);
    SC_CHECK(((uintptr_t)x % (x_bf16 ? 8 : 16)) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)dx % (dx_bf16 ? 8 : 16)) == 0, "sc_cif_bwd: alignment");
    SC_CHECK((x_bf16 != 0) == (dx_bf16 != 0), "sc_cif_bwd: the gradient leaves in the dtype the frames came in");
    if (!sc_option(6)) {            // one wave per 8 frames
        const dim3 gridf((C + 255) / 256, (S + UNROLL - 1) / UNROLL, B);
        if (x_bf16)
            hipLaunchKernelGGL((cif_bwd_frame_kernel<uint16_t, uint16_t>), gridf, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)x, xbs, alpha,
                               csum, g, (uint16_t*)dx, dxbs, zlo, zhi, pa, pb, B, S, C, T, thr);
        else
            hipLaunchKernelGGL((cif_bwd_frame_kernel<float, float>), gridf, dim3(64), 0, (hipStream_t)stream, (const float*)x, xbs, alpha, csum, g,
                               (float*)dx, dxbs, zlo, zhi, pa, pb, B, S, C, T, thr);
        SC_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((C + 255) / 256, B);
    if (x_bf16)
        hipLaunchKernelGGL((cif_bwd_kernel<uint16_t, uint16_t>), grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)x, xbs, alpha, csum, g,
                           (uint16_t*)dx, dxbs, zlo, zhi, pa, pb, B, S, C, T, thr);
    else
        hipLaunchKernelGGL((cif_bwd_kernel<float, float>), grid, dim3(64), 0, (hipStream_t)stream, (const float*)x, xbs, alpha, csum, g, (float*)dx,
                           dxbs, zlo, zhi, pa, pb, B, S, C, T, thr);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cif_bwd(const float* x, const float* alpha, const float* csum, const float* g, float* dx, float* pa, float* pb,
                          int32_t B, int32_t S, int32_t C, int32_t T, float thr, void* stream) {
    return sc_cif_bwd_rows(x, 0, (int64_t)S * C, alpha, csum, g, dx, 0, (int64_t)S * C, 0, 0, pa, pb, B, S, C, T, thr, stream);
}

extern "C" int sc_cif_prepare(const float* alpha_raw, int64_t lda, const uint8_t* pad, int64_t ldp, const int64_t* target,
                              int32_t apply_scaling, int32_t B, int32_t S, float thr, float eps, int32_t max_feat, int32_t T, float* a_clip, float* alpha, float* csum,
                              float* quantity, float* ratio, int64_t* feat_len, uint8_t* fired, int32_t* flags, void* stream) {
    SC_CHECK(alpha_raw && pad && a_clip && alpha && csum && quantity && ratio && feat_len && fired && flags, "sc_cif_prepare: null pointer");
    SC_CHECK(B > 0 && S > 0 && S <= MAXS && thr > 0.f && T >= 0 && max_feat >= 1, "sc_cif_prepare: B=%d S=%d (<= 2048) T=%d", B, S, T);
    hipLaunchKernelGGL(cif_prepare_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, alpha_raw, lda, pad, ldp, target, S, thr, eps,
                       (target && apply_scaling) ? 1 : 0, max_feat, T, a_clip, alpha, csum, quantity, ratio, feat_len, fired, flags);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cif_prepare_bwd(const float* pa, const float* pb, int32_t nblk, int32_t B, int32_t S, const float* a_clip,
                                  const uint8_t* pad, int64_t ldp, const float* ratio, const float* quantity, const float* gq,
                                  int32_t scaled, float* da, void* stream) {
    SC_CHECK(pa && pb && a_clip && pad && ratio && quantity && da, "sc_cif_prepare_bwd: null pointer");
    SC_CHECK(B > 0 && S > 0 && S <= MAXS && nblk > 0, "sc_cif_prepare_bwd: B=%d S=%d nblk=%d", B, S, nblk);
    hipLaunchKernelGGL(cif_prepare_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pa, pb, nblk, B, S, a_clip, pad, ldp, ratio,
                       quantity, gq, scaled, da);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cif_tail(const float* alpha, const float* csum, int32_t B, int32_t S, int32_t C, int32_t T, float thr, float tail_thr,
                           int32_t max_feat, int64_t* feat_len, float* out, float* factor, uint8_t* extend, void* stream) {
    SC_CHECK(alpha && csum && feat_len && out && factor && extend, "sc_cif_tail: null pointer");
    SC_CHECK(B > 0 && S > 0 && C > 0 && T >= 0 && thr > 0.f, "sc_cif_tail: bad arguments");
    hipLaunchKernelGGL(cif_tail_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, alpha, csum, S, C, T, thr, tail_thr, max_feat, feat_len,
                       out, factor, extend);
    SC_LAUNCH_CHECK();
    return 0;
}

static CifHead cif_head_args(const float* y, int64_t ldy, const float* w, const float* bias, int32_t rows, int32_t C, float p1, uint32_t seed1,
                             float p2, uint32_t seed2) {
    CifHead a;
    a.y = y; a.ldy = ldy; a.w = w; a.bias = bias; a.rows = rows; a.C = C;
    a.thr1 = p1 > 0.f ? (uint32_t)(p1 * 65536.f + 0.5f) : 0u; a.sc1 = p1 > 0.f ? 1.f / (1.f - p1) : 1.f; a.seed1 = seed1;
    a.thr2 = p2 > 0.f ? (uint32_t)(p2 * 65536.f + 0.5f) : 0u; a.sc2 = p2 > 0.f ? 1.f / (1.f - p2) : 1.f; a.seed2 = seed2;
    return a;
}

extern "C" int sc_cif_head_fwd(const float* y, int64_t ldy, const float* w, const float* bias, float* alpha, int32_t rows, int32_t C, float p1,
                               uint32_t seed1, float p2, uint32_t seed2, void* stream) {
    SC_CHECK(y && w && bias && alpha && rows > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldy % 4 == 0, "sc_cif_head_fwd: C=%d (%% 4, <= 1024)", C);
    SC_CHECK(((uintptr_t)y % 16) == 0 && ((uintptr_t)w % 16) == 0 && (int64_t)rows * C < ((int64_t)1 << 32) && p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f,
             "sc_cif_head_fwd: alignment / size / p");
    const CifHead a = cif_head_args(y, ldy, w, bias, rows, C, p1, seed1, p2, seed2);
    hipLaunchKernelGGL(cif_head_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, alpha);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cif_head_bwd_rows(const float* y, int64_t ldy, const float* w, const float* alpha, const float* dalpha, void* dy,
                                    int32_t dy_bf16, int64_t lddy, float* dw_partial, float* db_partial, int32_t nblk, int32_t rows, int32_t C,
                                    float p1, uint32_t seed1, float p2, uint32_t seed2, void* stream) {
    SC_CHECK(y && w && alpha && dalpha && dy && dw_partial && db_partial && nblk > 0 && rows > 0 && C > 0 && C % 4 == 0 && C <= 1024,
             "sc_cif_head_bwd: bad args (C=%d)", C);
    SC_CHECK(ldy % 4 == 0 && lddy % 4 == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)dy % (dy_bf16 ? 8 : 16)) == 0 && ((uintptr_t)w % 16) == 0,
             "sc_cif_head_bwd: alignment");
    const CifHead a = cif_head_args(y, ldy, w, nullptr, rows, C, p1, seed1, p2, seed2);
    if (dy_bf16)
        hipLaunchKernelGGL(cif_head_bwd_kernel<uint16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, alpha, dalpha, (uint16_t*)dy, lddy, dw_partial, db_partial);
    else
        hipLaunchKernelGGL(cif_head_bwd_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, alpha, dalpha, (float*)dy, lddy, dw_partial, db_partial);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_cif_head_bwd(const float* y, int64_t ldy, const float* w, const float* alpha, const float* dalpha, float* dy, int64_t lddy,
                               float* dw_partial, float* db_partial, int32_t nblk, int32_t rows, int32_t C, float p1, uint32_t seed1, float p2,
                               uint32_t seed2, void* stream) {
    return sc_cif_head_bwd_rows(y, ldy, w, alpha, dalpha, dy, 0, lddy, dw_partial, db_partial, nblk, rows, C, p1, seed1, p2, seed2, stream);
}

// rows of a [lead + B P + trail, D] bf16 row buffer that are not frames <- 0: the `lead` rows in front, per utterance the rows
// [0, head) and [stop, P), the `trail` rows behind (the zero padding a k-tap conv GEMM reads in place between utterances)
__global__ __launch_bounds__(256) void rows_zero_pad_kernel(uint16_t* __restrict__ buf, int lead, int B, int P, int head, int stop, int trail,
                                                            int D) {
    const int per = head + (P - stop), nz = lead + B * per + trail;
    const int cpr = D >> 3;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < (int64_t)nz * cpr; q += (int64_t)gridDim.x * 256) {
        const int z = (int)(q / cpr), c = (int)(q - (int64_t)z * cpr);
        int row;
        if (z < lead) row = z;
        else if (z < lead + B * per) {
            const int b = (z - lead) / per, r = (z - lead) - b * per;
            row = lead + b * P + (r < head ? r : stop + (r - head));
        } else row = lead + B * P + (z - lead - B * per);
        *(uint4*)(buf + (int64_t)row * D + c * 8) = make_uint4(0, 0, 0, 0);
    }
}

extern "C" int sc_rows_zero_pad_bf16(uint16_t* buf, int32_t lead, int32_t B, int32_t P, int32_t head, int32_t stop, int32_t trail, int32_t D,
                                     void* stream) {
    SC_CHECK(buf && lead >= 0 && B >= 0 && trail >= 0 && D > 0 && D % 8 == 0 && (B == 0 || (P > 0 && head >= 0 && head <= stop && stop <= P)),
             "sc_rows_zero_pad_bf16: bad arguments (B=%d P=%d head=%d stop=%d D=%d)", B, P, head, stop, D);
    SC_CHECK(((uintptr_t)buf % 16) == 0, "sc_rows_zero_pad_bf16: alignment");
    const int64_t nz = (int64_t)lead + (int64_t)B * (head + (P - stop)) + trail;
    if (nz == 0) return 0;
    const int64_t chunks = nz * (D >> 3);
    hipLaunchKernelGGL(rows_zero_pad_kernel, dim3((unsigned)((chunks + 255) / 256 < 2048 ? (chunks + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, buf, lead, B, P, head,
                       stop, trail, D);
    SC_LAUNCH_CHECK();
    return 0;
}
