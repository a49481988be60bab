// Waveform front end: padding / optional utterance layer-norm, conv layer 0 (C_in = 1, k = 10, s = 5) with
// GroupNorm(512 groups) + GELU fused, channels-last bf16 output.
//
// GroupNorm with one channel per group needs mean/var over time of every conv-0 output channel.  Because
// conv 0 is linear in a 10-sample window, those statistics follow exactly from the 10x10 Gram matrix of the
// strided waveform:   sum_t y_c[t]   = w_c . s,        s[j]    = sum_t x[5t + j]
//                     sum_t y_c[t]^2 = w_c^T G w_c,    G[j,j'] = sum_t x[5t + j] x[5t + j']
// accumulated in fp64 (deterministic two-stage reduction).  The 2.1 GB (B=64, 10 s) conv-0 activation is
// therefore written exactly once, already normalised and activated; the waveform is read from L2.
#include "sc_common.h"

namespace {

// one block per utterance: optional layer_norm over the utterance's own samples, zero padding to ldw_out
// (normalise: 1024 threads, four independent fp64 accumulator pairs per thread - the statistics pass is a latency-bound
// serial loop otherwise: 328 us at 10 s / 256 threads)
// row0 != NULL (ragged rows, sc_segments): utterance b's region is [spr * row0[b], spr * row0[b + 1]) of ONE flat buffer
__global__ __launch_bounds__(1024) void wav_prep_kernel(const float* __restrict__ wav, int64_t ldw_in,
                                                        const int64_t* __restrict__ wav_len, float* __restrict__ out,
                                                        int64_t ldw_out, int L, int normalize, const int32_t* __restrict__ row0, int spr,
                                                        const int64_t* __restrict__ wav_off) {
    __shared__ double red[2][16];
    const int b = blockIdx.x;
    int len = (int)wav_len[b];
    len = max(0, min(len, L));
    // wav_off != NULL (the in-forward training crop, speech_encoder_plus.py:548-552): utterance b is wav[b, off_b : off_b + len_b]
    const float* x = wav + (int64_t)b * ldw_in + (wav_off ? wav_off[b] : (int64_t)0);
    float* o = out + (int64_t)b * ldw_out;
    if (row0) {
        const int r0 = row0[b];
        o = out + (int64_t)r0 * spr;
        ldw_out = (int64_t)(row0[b + 1] - r0) * spr;
    }
    float mean = 0.f, rstd = 1.f;
    if (normalize) {
        double a[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
        const int nt = blockDim.x;
        int i = threadIdx.x;
        for (; i + 3 * nt < len; i += 4 * nt) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double v = x[i + u * nt];
                a[u] += v;
                a2[u] += v * v;
            }
        }
        for (; i < len; i += nt) {
            const double v = x[i];
            a[0] += v;
            a2[0] += v * v;
        }
        double s = wave_sum_d((a[0] + a[1]) + (a[2] + a[3]));
        double s2 = wave_sum_d((a2[0] + a2[1]) + (a2[2] + a2[3]));
        const int nw = blockDim.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = s;
            red[1][threadIdx.x >> 6] = s2;
        }
        __syncthreads();
        double S = 0.0, S2 = 0.0;
        for (int w = 0; w < nw; ++w) {
            S += red[0][w];
            S2 += red[1][w];
        }
        const double mu = S / (double)max(len, 1);
        const double var = fmax(S2 / (double)max(len, 1) - mu * mu, 0.0);
        mean = (float)mu;
        rstd = (float)(1.0 / sqrt(var + 1e-5));
    }
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < ldw_out; i += (int64_t)gridDim.y * blockDim.x) {
        float v = 0.f;
        if (i < len) v = normalize ? (x[i] - mean) * rstd : x[i];
        o[i] = v;
    }
}

// partial[(b*nchunk + chunk)*66 + e]: e < 55 Gram (j <= j'), 55..64 sums
// wav_len != NULL: the caller's un-prepared batch - samples at and past wav_len[b] read as 0 (windows that lie wholly behind the
// utterance add nothing and are skipped); the sums are those of the zero-padded waveform, term for term
// wav_off != NULL: utterance b starts at sample wav_off[b] of its row (the in-forward crop)
__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ wav, int64_t ldw, int T0,
                                                          int nchunk, double* __restrict__ partial, const int64_t* __restrict__ wav_len,
                                                          const int64_t* __restrict__ wav_off) {
    __shared__ double red[4][SC_CONV0_NSTAT];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (T0 + nchunk - 1) / nchunk;
    const int len = wav_len ? (int)wav_len[b] : 0x7fffffff;
    const int t_begin = chunk * per, t_end = min(min(T0, t_begin + per), wav_len ? (len + 4) / 5 : 0x7fffffff);
    const float* x = wav + (int64_t)b * ldw + (wav_off ? wav_off[b] : (int64_t)0);
    if (t_begin >= t_end) {                      // a chunk wholly behind the utterance (ragged batches): its sums are zero
        if (threadIdx.x < 65) partial[((int64_t)b * nchunk + chunk) * SC_CONV0_NSTAT + threadIdx.x] = 0.0;
        return;
    }
    double acc[65];
#pragma unroll
    for (int e = 0; e < 65; ++e) acc[e] = 0.0;
    for (int t = t_begin + threadIdx.x; t < t_end; t += blockDim.x) {
        float v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = (5 * t + j < len) ? x[5 * t + j] : 0.f;
        int e = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int k = j; k < 10; ++k) acc[e++] += (double)v[j] * (double)v[k];
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[55 + j] += (double)v[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 65; ++e) {
        const double s = wave_sum_d(acc[e]);
        if (lane == 0) red[wave][e] = s;
    }
    __syncthreads();
    if (threadIdx.x < 65)
        partial[((int64_t)b * nchunk + chunk) * SC_CONV0_NSTAT + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one block per utterance: reduce partials in fixed order, then per-channel scale / shift
__global__ __launch_bounds__(256) void conv0_finalize_kernel(const double* __restrict__ partial, int nchunk,
                                                             const float* __restrict__ w0,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int C, int T0, float eps,
                                                             float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double st[SC_CONV0_NSTAT];
    const int b = blockIdx.x;
    if (threadIdx.x < 65) {
        double s = 0.0;
        for (int c = 0; c < nchunk; ++c) s += partial[((int64_t)b * nchunk + c) * SC_CONV0_NSTAT + threadIdx.x];
        st[threadIdx.x] = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double w[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) w[j] = (double)w0[c * 10 + j];
        double s1 = 0.0, s2 = 0.0;
        int e = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            s1 += w[j] * st[55 + j];
#pragma unroll
            for (int k = j; k < 10; ++k) {
                const double term = w[j] * w[k] * st[e++];
                s2 += (k == j) ? term : 2.0 * term;
            }
        }
        const double mu = s1 / (double)T0;
        const double var = fmax(s2 / (double)T0 - mu * mu, 0.0);
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double g = (double)gamma[c];
        scale[(int64_t)b * C + c] = (float)(g * rstd);
        shift[(int64_t)b * C + c] = (float)((double)beta[c] - mu * g * rstd);
    }
}

// out[b*R0 + t, c] = gelu(conv0(x)[t, c] * scale[b, c] + shift[b, c]); one wave = one output row per step
// (64 lanes x 8 channels = 512 channels = 1 KiB contiguous store).
template <bool F32OUT>      // F32OUT: the fp32 debug mode of the encoder (speechclip_plus_amd/debug_fp32.py) - same arithmetic, unrounded stores
__global__ __launch_bounds__(256) void conv0_gn_gelu_kernel(const float* __restrict__ wav, int64_t ldw,
                                                            const float* __restrict__ w0,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            void* __restrict__ out_, int R0, int C,
                                                            int rows_per_block, const int32_t* __restrict__ row0, int spr) {
    uint16_t* out = (uint16_t*)out_;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    // wave id made provably uniform: the 10-sample window x[5t .. 5t+9] is then fetched with scalar loads (SMEM,
    // scalar cache) instead of 10 broadcast vector loads per output row
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cgroups = C / 512;   // C is a multiple of 512 (checked on the host)
    const float* x = wav + (int64_t)b * ldw;
    int64_t orow = (int64_t)b * R0;                 // first output row of the utterance
    if (row0) {                                     // ragged rows: flat waveform, utterance at sample spr * row0[b], spr / 5 rows per row
        const int r0 = row0[b];
        x = wav + (int64_t)r0 * spr;
        orow = (int64_t)r0 * (spr / 5);
        R0 = (row0[b + 1] - r0) * (spr / 5);
    }
    const int t_begin = blockIdx.x * rows_per_block;
    if (t_begin >= R0) return;
    const int t_end = min(R0, t_begin + rows_per_block);
    for (int cg = 0; cg < cgroups; ++cg) {
        const int c0 = cg * 512 + lane * 8;
        // channel pairs in packed fp32 (v_pk_fma_f32): 4 pairs x 10 taps per output row
        f32x2 w[4][10], sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 10; ++j) w[i][j] = f32x2{w0[(c0 + 2 * i) * 10 + j], w0[(c0 + 2 * i + 1) * 10 + j]};
            sc[i] = *(const f32x2*)(scale + (int64_t)b * C + c0 + 2 * i);
            sh[i] = *(const f32x2*)(shift + (int64_t)b * C + c0 + 2 * i);
        }
#ifndef SC_CONV0_ROWS
#define SC_CONV0_ROWS 4
#endif
        // SC_CONV0_ROWS adjacent rows per wave iteration (round 5: neighbours share 5 of their 10 samples - 25 scalar loads instead of 40 for
        // four rows, ONE wait for them per iteration, independent chains for the scheduler: 780 -> 694 us at B = 64 x 10 s, 6 / 8 rows are
        // slower again); per element the same operations in the same order as the one-row form: same bits
        constexpr int NR = SC_CONV0_ROWS, NV = 5 * NR + 5;
        auto fetch = [&](int t, float (&v)[NV]) {
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = x[5 * t + j];   // wave-uniform address: scalar loads
#pragma unroll
            for (int j = 10; j < NV; ++j) v[j] = (t + (j - 5) / 5 < t_end) ? x[5 * t + j] : 0.f;      // (nothing behind the last row is read)
        };
        // (fetching the NEXT iteration's samples under this one's arithmetic: 804 us - not kept)
        for (int t = t_begin + NR * wave; t < t_end; t += 4 * NR) {
            float v[NV];
            fetch(t, v);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r > 0 && t + r >= t_end) break;                 // wave-uniform
                float o[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x2 a = f32x2{0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 10; ++j) a = __builtin_elementwise_fma(w[i][j], f32x2{v[5 * r + j], v[5 * r + j]}, a);
                    const f32x2 u_ = __builtin_elementwise_fma(a, sc[i], sh[i]);
                    const f32x2 g = F32OUT ? gelu_erf2(u_) : gelu_bf2(u_);      // (fp32 debug mode: the 3e-7 form)
                    o[2 * i] = g.x;
                    o[2 * i + 1] = g.y;
                }
                if constexpr (F32OUT) {
                    float* of = (float*)out_ + (orow + t + r) * C + c0;
                    *(f32x4*)of = f32x4{o[0], o[1], o[2], o[3]};
                    *(f32x4*)(of + 4) = f32x4{o[4], o[5], o[6], o[7]};
                } else {
                    uint4 u;
                    u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]);
                    u.z = pack2bf(o[4], o[5]); u.w = pack2bf(o[6], o[7]);
                    *(uint4*)(out + (orow + t + r) * C + c0) = u;
                }
            }
        }
    }
}

// "layer_norm" extractor mode (HuBERT-large): out[b*R0 + t, :] = gelu(LayerNorm_c(conv0(x)[t, :] + bias)), C = 512:
// one wave per output row, 8 channels per lane, wavefront reductions for the channel statistics.
template <bool F32OUT>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* __restrict__ wav, int64_t ldw,
                                                            const float* __restrict__ w0,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            void* __restrict__ out_, int R0, int rows_per_block,
                                                            const int32_t* __restrict__ row0, int spr) {
    constexpr int C = 512;
    uint16_t* out = (uint16_t*)out_;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* x = wav + (int64_t)b * ldw;
    int64_t orow = (int64_t)b * R0;
    if (row0) {                                     // ragged rows: see conv0_gn_gelu_kernel
        const int r0 = row0[b];
        x = wav + (int64_t)r0 * spr;
        orow = (int64_t)r0 * (spr / 5);
        R0 = (row0[b + 1] - r0) * (spr / 5);
    }
    const int t_begin = blockIdx.x * rows_per_block;
    if (t_begin >= R0) return;
    const int t_end = min(R0, t_begin + rows_per_block);
    const int c0 = lane * 8;
    // Round 5: channel PAIRS in packed fp32 for the taps (v_pk_fma_f32, as conv0_gn_gelu_kernel) and SC_CONV0LN_ROWS adjacent output rows per wave
    // iteration (neighbours share 5 of their 10 samples; their statistics reductions interleave) - per element the same
    // operations in the same order as the one-row scalar form, so the same bits.
    f32x2 w[4][10], bs[4];
    float gm[8], bt[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) w[i][j] = f32x2{w0[(c0 + 2 * i) * 10 + j], w0[(c0 + 2 * i + 1) * 10 + j]};
        bs[i] = bias ? f32x2{bias[c0 + 2 * i], bias[c0 + 2 * i + 1]} : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        gm[i] = gamma[c0 + i];
        bt[i] = beta[c0 + i];
    }
    auto finish = [&](const f32x2 (&a2)[4], float mean, float rstd, int t) {
        const float a[8] = {a2[0].x, a2[0].y, a2[1].x, a2[1].y, a2[2].x, a2[2].y, a2[3].x, a2[3].y};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f32x2 u_ = f32x2{(a[i] - mean) * rstd * gm[i] + bt[i], (a[i + 1] - mean) * rstd * gm[i + 1] + bt[i + 1]};
            const f32x2 g = F32OUT ? gelu_erf2(u_) : gelu_bf2(u_);
            o[i] = g.x;
            o[i + 1] = g.y;
        }
        if constexpr (F32OUT) {
            float* of = (float*)out_ + (orow + t) * C + c0;
            *(f32x4*)of = f32x4{o[0], o[1], o[2], o[3]};
            *(f32x4*)(of + 4) = f32x4{o[4], o[5], o[6], o[7]};
        } else {
            uint4 u;
            u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]);
            u.z = pack2bf(o[4], o[5]); u.w = pack2bf(o[6], o[7]);
            *(uint4*)(out + (orow + t) * C + c0) = u;
        }
    };
#ifndef SC_CONV0LN_ROWS
#define SC_CONV0LN_ROWS 2
#endif
    constexpr int NR = SC_CONV0LN_ROWS, NV = 5 * NR + 5;
    for (int t = t_begin + NR * wave; t < t_end; t += 4 * NR) {
        float v[NV];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = x[5 * t + j];
#pragma unroll
        for (int j = 10; j < NV; ++j) v[j] = (t + (j - 5) / 5 < t_end) ? x[5 * t + j] : 0.f;     // (nothing behind the last row is read)
        f32x2 a[NR][4];
        float sm[NR], sq[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) sm[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                f32x2 acc = bs[i];
#pragma unroll
                for (int j = 0; j < 10; ++j) acc = __builtin_elementwise_fma(w[i][j], f32x2{v[5 * r + j], v[5 * r + j]}, acc);
                a[r][i] = acc;
                sm[r] += acc.x;
                sm[r] += acc.y;
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) sm[r] = wave_sum(sm[r]) * (1.0f / C);          // mean
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            sq[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sq[r] += (a[r][i].x - sm[r]) * (a[r][i].x - sm[r]);
                sq[r] += (a[r][i].y - sm[r]) * (a[r][i].y - sm[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) sq[r] = rsqrtf(wave_sum(sq[r]) * (1.0f / C) + eps);   // rstd
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r == 0 || t + r < t_end) finish(a[r], sm[r], sq[r], t + r);
    }
}


// Round 6: the layer_norm mode WITHOUT cross-lane reductions.  conv0_ln_gelu_kernel above spends more than half of its 1.43 ms (B = 64 x
// 10 s; the GroupNorm-mode kernel does the same taps + GELU in 0.61 ms) in two wavefront reductions per row.  But a row's channel
// statistics are closed forms of its 10-sample window x:  y_c = w_c . x + b_c  =>
//     mean_c y = wbar . x + bbar,        var_c y = x^T G x + 2 h . x + s
// with wbar / bbar the channel means of the taps / bias, G = cov_c(w_c) (10 x 10), h = cov_c(w_c, b_c), s = var_c(b_c) - every term
// CENTRED, so the quadratic form has no cancellation beyond its own PSD structure; it is evaluated in fp64 (65 FMAs per row).  A wave
// owns 32 rows of its workgroup's 128: lane i computes the statistics of the wave's i-th row up front (one strided 10-sample gather per
// lane), and the main loop - the GroupNorm kernel's: four rows per iteration, packed taps, scalar sample loads - fetches (mean, rstd)
// of the current row with v_readlane.  The 77 raw channel sums behind wbar .. s come from a 77-workgroup pre-kernel.
// Same arithmetic per element as the two-pass kernel, (a - mean) rstd gamma + beta; mean and rstd differ from its fp32 two-pass values in
// the last bits (they are now the fp64 closed forms rounded once).
constexpr int C0LN_RAW = 10 + 1 + 55 + 10 + 1;      // sum w_i | sum b | sum w_i w_j (i <= j) | sum w_i b | sum b^2

__global__ __launch_bounds__(256) void conv0_ln_consts_kernel(const float* __restrict__ w0, const float* __restrict__ bias, double* __restrict__ raw) {
    constexpr int C = 512;
    __shared__ double red[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    int i = -1, j = -1, kind;            // kind 0: w_i, 1: b, 2: w_i w_j, 3: w_i b, 4: b b
    if (k < 10) { kind = 0; i = k; }
    else if (k == 10) kind = 1;
    else if (k < 66) {
        kind = 2;
        int e = k - 11;
        for (i = 0; e >= 10 - i; ++i) e -= 10 - i;
        j = i + e;
    } else if (k < 76) { kind = 3; i = k - 66; }
    else kind = 4;
    double acc = 0.0;
    for (int c = tid; c < C; c += 256) {
        const double b = bias ? (double)bias[c] : 0.0;
        const double wi = i >= 0 ? (double)w0[c * 10 + i] : 0.0, wj = j >= 0 ? (double)w0[c * 10 + j] : 0.0;
        acc += kind == 0 ? wi : kind == 1 ? b : kind == 2 ? wi * wj : kind == 3 ? wi * b : b * b;
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) raw[k] = red[0];
}

__global__ __launch_bounds__(256) void conv0_ln_gelu_stats_kernel(const float* __restrict__ wav, int64_t ldw, const float* __restrict__ w0,
                                                                  const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, const double* __restrict__ raw,
                                                                  uint16_t* __restrict__ out, int R0, const int32_t* __restrict__ row0, int spr) {
    constexpr int C = 512, NR = 4, RPB = 128;
    __shared__ double cst[C0LN_RAW];                 // wbar[10] | bbar | G[55] (i <= j, off-diagonal entries doubled) | 2 h[10] | s
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* x = wav + (int64_t)b * ldw;
    int64_t orow = (int64_t)b * R0;
    if (row0) {                                     // ragged rows: see conv0_gn_gelu_kernel
        const int r0 = row0[b];
        x = wav + (int64_t)r0 * spr;
        orow = (int64_t)r0 * (spr / 5);
        R0 = (row0[b + 1] - r0) * (spr / 5);
    }
    const int t_begin = blockIdx.x * RPB;
    if (t_begin >= R0) return;                      // uniform for the workgroup, before the barrier
    const int t_end = min(R0, t_begin + RPB);
    if (threadIdx.x < C0LN_RAW) {
        const int k = threadIdx.x;
        const double inv = 1.0 / C, bbar = raw[10] * inv;
        double v;
        if (k < 10) v = raw[k] * inv;
        else if (k == 10) v = bbar;
        else if (k < 66) {
            int e = k - 11, i = 0;
            for (; e >= 10 - i; ++i) e -= 10 - i;
            const int j = i + e;
            v = raw[k] * inv - (raw[i] * inv) * (raw[j] * inv);
            if (i != j) v *= 2.0;
        } else if (k < 76) v = 2.0 * (raw[k] * inv - (raw[k - 66] * inv) * bbar);
        else v = raw[76] * inv - bbar * bbar;
        cst[k] = v;
    }
    __syncthreads();
    // ---- statistics of the wave's rows: lane i < 32 <-> row t_begin + NR wave + 4 NR (i / NR) + i % NR
    float mu_l = 0.f, rs_l = 0.f;
    {
        const int i = lane & 31, t = t_begin + NR * wave + 4 * NR * (i / NR) + (i % NR);
        if (t < t_end) {
            double xv[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) xv[j] = (double)x[5 * t + j];
            double mu = cst[10], var = cst[76];
            int e = 11;
#pragma unroll
            for (int p = 0; p < 10; ++p) {
                mu = fma(cst[p], xv[p], mu);
                var = fma(cst[66 + p], xv[p], var);
                double row = 0.0;
#pragma unroll
                for (int q = p; q < 10; ++q) row = fma(cst[e++], xv[q], row);
                var = fma(row, xv[p], var);
            }
            mu_l = (float)mu;
            rs_l = (float)(1.0 / sqrt(fmax(var, 0.0) + (double)eps));
        }
    }
    const int c0 = lane * 8;
    f32x2 w[4][10], bs[4], gm[4], bt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) w[i][j] = f32x2{w0[(c0 + 2 * i) * 10 + j], w0[(c0 + 2 * i + 1) * 10 + j]};
        bs[i] = bias ? f32x2{bias[c0 + 2 * i], bias[c0 + 2 * i + 1]} : f32x2{0.f, 0.f};
        gm[i] = f32x2{gamma[c0 + 2 * i], gamma[c0 + 2 * i + 1]};
        bt[i] = f32x2{beta[c0 + 2 * i], beta[c0 + 2 * i + 1]};
    }
    constexpr int NV = 5 * NR + 5;
    int it = 0;
    for (int t = t_begin + NR * wave; t < t_end; t += 4 * NR, ++it) {
        float v[NV];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = x[5 * t + j];   // wave-uniform address: scalar loads
#pragma unroll
        for (int j = 10; j < NV; ++j) v[j] = (t + (j - 5) / 5 < t_end) ? x[5 * t + j] : 0.f;      // (nothing behind the last row is read)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r > 0 && t + r >= t_end) break;                 // wave-uniform
            const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mu_l), it * NR + r));
            const float rs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs_l), it * NR + r));
            float o[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 a = bs[i];
#pragma unroll
                for (int j = 0; j < 10; ++j) a = __builtin_elementwise_fma(w[i][j], f32x2{v[5 * r + j], v[5 * r + j]}, a);
                // the two-pass kernel's association, (a - mean) rstd gamma + beta: the deviation first, so a row whose mean is large
                // against its spread loses nothing to cancellation (regrouped as a (rstd gamma) + (beta - mean rstd gamma) the paired
                // GELU test of the large recipe read 1.9e-2 instead of 1.1e-2 on the weighted-sum logit gradient)
                const f32x2 dv = (a - f32x2{mu, mu}) * rs;
                const f32x2 g = gelu_bf2(__builtin_elementwise_fma(dv, gm[i], bt[i]));
                o[2 * i] = g.x;
                o[2 * i + 1] = g.y;
            }
            uint4 u;
            u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]);
            u.z = pack2bf(o[4], o[5]); u.w = pack2bf(o[6], o[7]);
            *(uint4*)(out + (orow + t + r) * C + c0) = u;
        }
    }
}

// ---------------------------------------------------------------------------------------- conv layer 0: backward (GroupNorm form)
// Fully trainable encoder (avssl/module/speech_encoder_plus.py:556-562), "default" extractor: y = gelu(n), n = gamma (u - mu) / sigma + beta,
// u[b, t, c] = sum_j W[c][j] x[b, 5t + j], statistics over t < T0 per (b, c).  The input is the waveform: no input gradient, only
// dW [C, 10], dgamma, dbeta.  With dn = dy gelu'(n) everything follows from 12 time sums per (b, c) -
//   A1 = sum_t dn,  A2 = sum_t dn n,  V_j = sum_t dn x[5t + j]  (j = 0..9)
// - and the forward's 10 x 10 Gram matrix G and sums S of the strided waveform (sc_conv0_stats):
//   Q = sum_t dn uhat = (A2 - beta A1) / gamma ;  dgamma = sum_b Q ;  dbeta = sum_b A1 ;
//   dW[c][j] = sum_b (gamma / sigma) [V_j - (A1 / T0) S_j - (Q / T0) (sum_i W[c][i] G_ij - mu S_j) / sigma]
// One pass over dy (u is recomputed from the waveform, 10 FMAs per element), then a per-utterance finalisation in fp64.
// gelu_grad_as: sc_common.h
constexpr int C0_NS = 12;
// grid (nblk, B), 4 waves per block; wave chunk wc = blockIdx.x * 4 + wave owns rows [wc * rpw, min(T0, (wc + 1) * rpw)); a lane owns
// 8 channels (C = 512).  partial[((b * nwc + wc) * 512 + c) * 12 + e]
__global__ __launch_bounds__(256) void conv0_gn_bwd_kernel(const float* __restrict__ wav, int64_t ldw, const float* __restrict__ w0,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const uint16_t* __restrict__ dy, int T0, int R0, int rpw, int nwc,
                                                           float* __restrict__ partial) {
    constexpr int C = 512;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = blockIdx.x * 4 + wave;
    const float* x = wav + (int64_t)b * ldw;
    const int c0 = lane * 8;
    // channel PAIRS in packed fp32 (v_pk_fma_f32), as the forward kernel: 4 pairs x (10 taps + 12 running sums) per row
    f32x2 w[4][10], sc[4], sh[4], acc[4][C0_NS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) w[i][j] = f32x2{w0[(c0 + 2 * i) * 10 + j], w0[(c0 + 2 * i + 1) * 10 + j]};
        sc[i] = *(const f32x2*)(scale + (int64_t)b * C + c0 + 2 * i);
        sh[i] = *(const f32x2*)(shift + (int64_t)b * C + c0 + 2 * i);
#pragma unroll
        for (int e = 0; e < C0_NS; ++e) acc[i][e] = f32x2{0.f, 0.f};
    }
    const int t_end = min(T0, (wc + 1) * rpw);
    for (int t = wc * rpw; t < t_end; ++t) {
        float v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = x[5 * t + j];              // wave-uniform address: scalar loads
        const uint4 d = *(const uint4*)(dy + ((int64_t)b * R0 + t) * C + c0);
        const f32x2 g[4] = {f32x2{bflo(d.x), bfhi(d.x)}, f32x2{bflo(d.y), bfhi(d.y)}, f32x2{bflo(d.z), bfhi(d.z)}, f32x2{bflo(d.w), bfhi(d.w)}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 u = f32x2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 10; ++j) u = __builtin_elementwise_fma(w[i][j], f32x2{v[j], v[j]}, u);
            const f32x2 n = __builtin_elementwise_fma(u, sc[i], sh[i]);
            const f32x2 dn = g[i] * gelu_grad_as2(n);
            acc[i][0] += dn;
            acc[i][1] = __builtin_elementwise_fma(dn, n, acc[i][1]);
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[i][2 + j] = __builtin_elementwise_fma(dn, f32x2{v[j], v[j]}, acc[i][2 + j]);
        }
    }
    if (wc < nwc) {
        float* pp = partial + (((int64_t)b * nwc + wc) * C + c0) * C0_NS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < C0_NS; e += 4) {
                *(f32x4*)(pp + (2 * i) * C0_NS + e) = f32x4{acc[i][e].x, acc[i][e + 1].x, acc[i][e + 2].x, acc[i][e + 3].x};
                *(f32x4*)(pp + (2 * i + 1) * C0_NS + e) = f32x4{acc[i][e].y, acc[i][e + 1].y, acc[i][e + 2].y, acc[i][e + 3].y};
            }
    }
}

// ---------------------------------------------------------------------------------------- conv layer 0: backward (LayerNorm form)
// "layer_norm" extractor (HuBERT-large): y = gelu(n), n = gamma xhat + beta, xhat = (u - mean_c u) rstd, u[t, c] = b[c] + sum_j W[c][j] x[5t + j]
// with the statistics over the 512 channels of a row.  Per row: dn = dy gelu'(n); dgamma += dn xhat; dbeta += dn;
//   g = dn gamma;  du = rstd (g - mean_c g - xhat mean_c(g xhat));  db += du;  dW[c][j] += du[c] x[5t + j].
// Same walk as conv0_gn_bwd_kernel (wave chunk of rows, 8 channels per lane, u recomputed from the waveform), 13 sums per channel:
// partial[((b * nwc + wc) * 512 + c) * 16 + e], e = 0..9 dW, 10 db, 11 dgamma, 12 dbeta (13..15 unused).
constexpr int C0L_NS = 16;
__global__ __launch_bounds__(256) void conv0_ln_bwd_kernel(const float* __restrict__ wav, int64_t ldw, const float* __restrict__ w0,
                                                           const float* __restrict__ bias, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, const uint16_t* __restrict__ dy,
                                                           int T0, int R0, int rpw, int nwc, float* __restrict__ partial) {
    constexpr int C = 512;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = blockIdx.x * 4 + wave;
    const float* x = wav + (int64_t)b * ldw;
    const int c0 = lane * 8;
    float w[8][10], bs[8], gm[8], bt[8], acc[8][13];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) w[i][j] = w0[(c0 + i) * 10 + j];
        bs[i] = bias ? bias[c0 + i] : 0.f;
        gm[i] = gamma[c0 + i];
        bt[i] = beta[c0 + i];
#pragma unroll
        for (int e = 0; e < 13; ++e) acc[i][e] = 0.f;
    }
    const int t_end = min(T0, (wc + 1) * rpw);
    for (int t = wc * rpw; t < t_end; ++t) {
        float v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = x[5 * t + j];              // wave-uniform address: scalar loads
        const uint4 d = *(const uint4*)(dy + ((int64_t)b * R0 + t) * C + c0);
        const float gy[8] = {bflo(d.x), bfhi(d.x), bflo(d.y), bfhi(d.y), bflo(d.z), bfhi(d.z), bflo(d.w), bfhi(d.w)};
        float a[8], s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float u = bs[i];
#pragma unroll
            for (int j = 0; j < 10; ++j) u = fmaf(w[i][j], v[j], u);
            a[i] = u;
            s += u;
        }
        const float mean = wave_sum(s) * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sq += (a[i] - mean) * (a[i] - mean);
        const float rstd = rsqrtf(wave_sum(sq) * (1.0f / C) + eps);
        float g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (a[i] - mean) * rstd;
            const float dn = gy[i] * gelu_grad_as(fmaf(xh, gm[i], bt[i]));
            acc[i][11] = fmaf(dn, xh, acc[i][11]);
            acc[i][12] += dn;
            a[i] = xh;
            g[i] = dn * gm[i];
            s1 += g[i];
            s2 = fmaf(g[i], xh, s2);
        }
        const float m1 = wave_sum(s1) * (1.0f / C), m2 = wave_sum(s2) * (1.0f / C);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float du = rstd * (g[i] - m1 - a[i] * m2);
            acc[i][10] += du;
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[i][j] = fmaf(du, v[j], acc[i][j]);
        }
    }
    if (wc < nwc) {
        float* pp = partial + (((int64_t)b * nwc + wc) * C + c0) * C0L_NS;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *(f32x4*)(pp + i * C0L_NS) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
            *(f32x4*)(pp + i * C0L_NS + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
            *(f32x4*)(pp + i * C0L_NS + 8) = f32x4{acc[i][8], acc[i][9], acc[i][10], acc[i][11]};
            *(f32x4*)(pp + i * C0L_NS + 12) = f32x4{acc[i][12], 0.f, 0.f, 0.f};
        }
    }
}

// one block per utterance, thread = channel (two rounds for 512): wave chunks added in order, then the per-(b, c) algebra in fp64;
// contrib[b][c][12] = (dW[c][0..9], dgamma, dbeta) of utterance b (summed over b by sc_colsum_f32)
__global__ __launch_bounds__(256) void conv0_gn_bwd_finalize_kernel(const float* __restrict__ partial, int nwc, const double* __restrict__ stats,
                                                                    int nchunk, const float* __restrict__ w0, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int T0, float eps,
                                                                    float* __restrict__ contrib) {
    constexpr int C = 512;
    __shared__ double st[SC_CONV0_NSTAT];
    const int b = blockIdx.x;
    if (threadIdx.x < 65) {
        double sm = 0.0;
        for (int c = 0; c < nchunk; ++c) sm += stats[((int64_t)b * nchunk + c) * SC_CONV0_NSTAT + threadIdx.x];
        st[threadIdx.x] = sm;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double a[C0_NS];
#pragma unroll
        for (int e = 0; e < C0_NS; ++e) a[e] = 0.0;
        for (int k = 0; k < nwc; ++k) {
            const float* pp = partial + (((int64_t)b * nwc + k) * C + c) * C0_NS;
#pragma unroll
            for (int e = 0; e < C0_NS; ++e) a[e] += (double)pp[e];
        }
        double w[10], WG[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) w[j] = (double)w0[c * 10 + j];
        // G is stored as its upper triangle (j <= k): e = index of (min, max)
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int lo = i < j ? i : j, hi = i < j ? j : i;
                const int e = lo * 10 - lo * (lo - 1) / 2 + (hi - lo);
                t += w[i] * st[e];
            }
            WG[j] = t;                                                   // sum_i W[c][i] G_ij
            s1 += w[j] * st[55 + j];
            s2 += w[j] * t;
        }
        const double T = (double)T0;
        const double mu = s1 / T;
        const double var = fmax(s2 / T - mu * mu, 0.0);
        const double rs = 1.0 / sqrt(var + (double)eps);
        const double g = (double)gamma[c], be = (double)beta[c];
        const double A1 = a[0], Q = (a[1] - be * A1) / g;
        float* out = contrib + ((int64_t)b * C + c) * C0_NS;
#pragma unroll
        for (int j = 0; j < 10; ++j)
            out[j] = (float)(g * rs * (a[2 + j] - (A1 / T) * st[55 + j] - (Q / T) * rs * (WG[j] - mu * st[55 + j])));
        out[10] = (float)Q;
        out[11] = (float)A1;
    }
}

}  // namespace

extern "C" int sc_wav_prep(const float* wav, int64_t ldw_in, const int64_t* wav_len, float* out, int64_t ldw_out,
                           int32_t B, int32_t L, int32_t normalize, void* stream) {
    SC_CHECK(wav && wav_len && out, "sc_wav_prep: null pointer");
    SC_CHECK(B > 0 && L > 0 && ldw_out >= L && ldw_in >= L, "sc_wav_prep: bad sizes");
    // without normalisation the kernel is a pure copy: spread each utterance over 32 blocks
    hipLaunchKernelGGL(wav_prep_kernel, dim3(B, normalize ? 4 : 32), dim3(normalize ? 1024 : 256), 0, (hipStream_t)stream, wav, ldw_in, wav_len, out, ldw_out, L, normalize, (const int32_t*)nullptr, 0, (const int64_t*)nullptr);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wav_prep_seg_crop(const float* wav, int64_t ldw_in, const int64_t* wav_len, const int64_t* wav_off, float* out,
                                    const sc_segments* seg, int32_t samples_per_row, int32_t L, int32_t normalize, void* stream) {
    SC_CHECK(wav && wav_len && out && seg && seg->row0, "sc_wav_prep_seg: null pointer");
    SC_CHECK(seg->B > 0 && L > 0 && ldw_in >= L && samples_per_row > 0 && samples_per_row % 5 == 0, "sc_wav_prep_seg: bad sizes");
    hipLaunchKernelGGL(wav_prep_kernel, dim3(seg->B, normalize ? 4 : 32), dim3(normalize ? 1024 : 256), 0, (hipStream_t)stream, wav, ldw_in, wav_len,
                       out, (int64_t)0, L, normalize, seg->row0, samples_per_row, wav_off);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_wav_prep_seg(const float* wav, int64_t ldw_in, const int64_t* wav_len, float* out, const sc_segments* seg,
                               int32_t samples_per_row, int32_t L, int32_t normalize, void* stream) {
    return sc_wav_prep_seg_crop(wav, ldw_in, wav_len, nullptr, out, seg, samples_per_row, L, normalize, stream);
}

extern "C" int sc_wav_prep_crop(const float* wav, int64_t ldw_in, const int64_t* wav_len, const int64_t* wav_off, float* out,
                                int64_t ldw_out, int32_t B, int32_t L, int32_t normalize, void* stream) {
    SC_CHECK(wav && wav_len && out, "sc_wav_prep_crop: null pointer");
    SC_CHECK(B > 0 && L > 0 && ldw_out >= L && ldw_in >= L, "sc_wav_prep_crop: bad sizes");
    hipLaunchKernelGGL(wav_prep_kernel, dim3(B, normalize ? 4 : 32), dim3(normalize ? 1024 : 256), 0, (hipStream_t)stream, wav, ldw_in, wav_len, out,
                       ldw_out, L, normalize, (const int32_t*)nullptr, 0, wav_off);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_stats(const float* wav, int64_t ldw, int32_t B, int32_t T0, int32_t nchunk, double* partial,
                              void* stream) {
    SC_CHECK(wav && partial, "sc_conv0_stats: null pointer");
    SC_CHECK(B > 0 && T0 > 0 && nchunk > 0 && ldw >= 5 * (int64_t)(T0 - 1) + 10, "sc_conv0_stats: bad sizes");
    hipLaunchKernelGGL(conv0_stats_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, wav, ldw, T0, nchunk, partial, (const int64_t*)nullptr, (const int64_t*)nullptr);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_stats_len_crop(const float* wav, int64_t ldw, const int64_t* wav_len, const int64_t* wav_off, int32_t B, int32_t T0,
                                       int32_t nchunk, double* partial, void* stream) {
    SC_CHECK(wav && wav_len && partial, "sc_conv0_stats_len: null pointer");
    SC_CHECK(B > 0 && T0 > 0 && nchunk > 0, "sc_conv0_stats_len: bad sizes");
    hipLaunchKernelGGL(conv0_stats_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, wav, ldw, T0, nchunk, partial, wav_len, wav_off);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_stats_len(const float* wav, int64_t ldw, const int64_t* wav_len, int32_t B, int32_t T0, int32_t nchunk,
                                  double* partial, void* stream) {
    return sc_conv0_stats_len_crop(wav, ldw, wav_len, nullptr, B, T0, nchunk, partial, stream);
}

extern "C" int sc_conv0_finalize(const double* partial, int32_t nchunk, const float* w0, const float* gamma,
                                 const float* beta, int32_t B, int32_t C, int32_t T0, float eps, float* scale,
                                 float* shift, void* stream) {
    SC_CHECK(partial && w0 && gamma && beta && scale && shift, "sc_conv0_finalize: null pointer");
    hipLaunchKernelGGL(conv0_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, partial, nchunk, w0, gamma, beta, C, T0, eps, scale, shift);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_gn_gelu(const float* wav, int64_t ldw, const float* w0, const float* scale, const float* shift,
                                sc_bf16* out, int32_t B, int32_t R0, int32_t C, void* stream) {
    SC_CHECK(wav && w0 && scale && shift && out, "sc_conv0_gn_gelu: null pointer");
    SC_CHECK(C % 512 == 0 && ldw >= 5 * (int64_t)(R0 - 1) + 10, "sc_conv0_gn_gelu: C=%d must be a multiple of 512; ldw too small", C);
    SC_CHECK(((uintptr_t)out % 16) == 0, "sc_conv0_gn_gelu: alignment");
    const int rows_per_block = 128;
    dim3 grid((R0 + rows_per_block - 1) / rows_per_block, B);
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, wav, ldw, w0, scale, shift, (void*)out, R0, C, rows_per_block,
                       (const int32_t*)nullptr, 0);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_gn_gelu_seg(const float* wav_flat, const sc_segments* seg, int32_t samples_per_row, const float* w0, const float* scale,
                                    const float* shift, sc_bf16* out, int32_t C, void* stream) {
    SC_CHECK(wav_flat && seg && seg->row0 && w0 && scale && shift && out, "sc_conv0_gn_gelu_seg: null pointer");
    SC_CHECK(C % 512 == 0 && samples_per_row > 0 && samples_per_row % 5 == 0 && seg->B > 0 && seg->max_pitch > 0,
             "sc_conv0_gn_gelu_seg: C=%d must be a multiple of 512, samples_per_row a multiple of 5", C);
    SC_CHECK(((uintptr_t)out % 16) == 0, "sc_conv0_gn_gelu_seg: alignment");
    const int rows_per_block = 128;
    const int R0max = seg->max_pitch * (samples_per_row / 5);
    dim3 grid((R0max + rows_per_block - 1) / rows_per_block, seg->B);
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, wav_flat, (int64_t)0, w0, scale, shift, (void*)out, 0, C,
                       rows_per_block, seg->row0, samples_per_row);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_gn_gelu_f32(const float* wav, int64_t ldw, const float* w0, const float* scale, const float* shift,
                                    float* out, int32_t B, int32_t R0, int32_t C, void* stream) {
    SC_CHECK(wav && w0 && scale && shift && out, "sc_conv0_gn_gelu_f32: null pointer");
    SC_CHECK(C % 512 == 0 && ldw >= 5 * (int64_t)(R0 - 1) + 10, "sc_conv0_gn_gelu_f32: C=%d must be a multiple of 512; ldw too small", C);
    SC_CHECK(((uintptr_t)out % 16) == 0, "sc_conv0_gn_gelu_f32: alignment");
    const int rows_per_block = 128;
    dim3 grid((R0 + rows_per_block - 1) / rows_per_block, B);
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, wav, ldw, w0, scale, shift, (void*)out, R0, C, rows_per_block,
                       (const int32_t*)nullptr, 0);
    SC_LAUNCH_CHECK();
    return 0;
}

// layer_norm-mode conv 0 through the closed-form row statistics (round 6): 77 raw channel sums into a stream-ordered scratch of 616
// bytes (hipMallocAsync / hipFreeAsync on the caller's stream: nothing outlives the call, nothing is shared between streams), then the
// main kernel.  sc_set_option(2, 1): the two-pass reduction kernel, for A/B (tools/bench_conv0ln.py).  Returns 1 (caller falls back to
// the two-pass kernel) when the runtime has no stream-ordered allocator.
static int conv0_ln_closed_form(const float* wav, int64_t ldw, const float* w0, const float* bias, const float* gamma, const float* beta, float eps,
                                sc_bf16* out, dim3 grid, int R0, const int32_t* row0, int spr, hipStream_t s) {
    double* raw = nullptr;
    if (hipMallocAsync((void**)&raw, C0LN_RAW * sizeof(double), s) != hipSuccess) {
        (void)hipGetLastError();                    // no stream-ordered allocator on this runtime / device: the two-pass kernel needs no scratch
        return 1;
    }
    hipLaunchKernelGGL(conv0_ln_consts_kernel, dim3(C0LN_RAW), dim3(256), 0, s, w0, bias, raw);
    hipLaunchKernelGGL(conv0_ln_gelu_stats_kernel, grid, dim3(256), 0, s, wav, ldw, w0, bias, gamma, beta, eps, (const double*)raw, (uint16_t*)out, R0,
                       row0, spr);
    const hipError_t le = hipGetLastError();
    const hipError_t fe = hipFreeAsync(raw, s);
    if (le != hipSuccess || fe != hipSuccess) {
        sc_set_error("sc_conv0_ln_gelu: launch / free failed: %s", hipGetErrorString(le != hipSuccess ? le : fe));
        return -2;
    }
    return 0;
}

extern "C" int sc_conv0_ln_gelu(const float* wav, int64_t ldw, const float* w0, const float* bias, const float* gamma,
                                const float* beta, float eps, sc_bf16* out, int32_t B, int32_t R0, int32_t C, void* stream) {
    SC_CHECK(wav && w0 && gamma && beta && out, "sc_conv0_ln_gelu: null pointer");
    SC_CHECK(C == 512 && ldw >= 5 * (int64_t)(R0 - 1) + 10, "sc_conv0_ln_gelu: C must be 512 (got %d); ldw too small", C);
    SC_CHECK(((uintptr_t)out % 16) == 0, "sc_conv0_ln_gelu: alignment");
    const int rows_per_block = 128;
    dim3 grid((R0 + rows_per_block - 1) / rows_per_block, B);
    if (!sc_option(2)) {
        const int rc = conv0_ln_closed_form(wav, ldw, w0, bias, gamma, beta, eps, out, grid, R0, nullptr, 0, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }
    hipLaunchKernelGGL(conv0_ln_gelu_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, wav, ldw, w0, bias, gamma, beta, eps, (void*)out, R0, rows_per_block,
                       (const int32_t*)nullptr, 0);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_ln_gelu_seg(const float* wav_flat, const sc_segments* seg, int32_t samples_per_row, const float* w0, const float* bias,
                                    const float* gamma, const float* beta, float eps, sc_bf16* out, int32_t C, void* stream) {
    SC_CHECK(wav_flat && seg && seg->row0 && w0 && gamma && beta && out, "sc_conv0_ln_gelu_seg: null pointer");
    SC_CHECK(C == 512 && samples_per_row > 0 && samples_per_row % 5 == 0 && seg->B > 0 && seg->max_pitch > 0,
             "sc_conv0_ln_gelu_seg: C must be 512 (got %d), samples_per_row a multiple of 5", C);
    SC_CHECK(((uintptr_t)out % 16) == 0, "sc_conv0_ln_gelu_seg: alignment");
    const int rows_per_block = 128;
    const int R0max = seg->max_pitch * (samples_per_row / 5);
    dim3 grid((R0max + rows_per_block - 1) / rows_per_block, seg->B);
    if (!sc_option(2)) {
        const int rc = conv0_ln_closed_form(wav_flat, 0, w0, bias, gamma, beta, eps, out, grid, 0, seg->row0, samples_per_row, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }
    hipLaunchKernelGGL(conv0_ln_gelu_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, wav_flat, (int64_t)0, w0, bias, gamma, beta, eps, (void*)out, 0,
                       rows_per_block, seg->row0, samples_per_row);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_ln_gelu_f32(const float* wav, int64_t ldw, const float* w0, const float* bias, const float* gamma,
                                    const float* beta, float eps, float* out, int32_t B, int32_t R0, int32_t C, void* stream) {
    SC_CHECK(wav && w0 && gamma && beta && out, "sc_conv0_ln_gelu_f32: null pointer");
    SC_CHECK(C == 512 && ldw >= 5 * (int64_t)(R0 - 1) + 10, "sc_conv0_ln_gelu_f32: C must be 512 (got %d); ldw too small", C);
    SC_CHECK(((uintptr_t)out % 16) == 0, "sc_conv0_ln_gelu_f32: alignment");
    const int rows_per_block = 128;
    dim3 grid((R0 + rows_per_block - 1) / rows_per_block, B);
    hipLaunchKernelGGL(conv0_ln_gelu_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, wav, ldw, w0, bias, gamma, beta, eps, (void*)out, R0, rows_per_block,
                       (const int32_t*)nullptr, 0);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_gn_bwd(const float* wav, int64_t ldw, const float* w0, const float* scale, const float* shift, const sc_bf16* dy,
                               const double* stats, int32_t nchunk, const float* gamma, const float* beta, int32_t B, int32_t T0, int32_t R0,
                               int32_t C, float eps, float* partial, int32_t nwc, float* contrib, void* stream) {
    SC_CHECK(wav && w0 && scale && shift && dy && stats && gamma && beta && partial && contrib, "sc_conv0_gn_bwd: null pointer");
    SC_CHECK(C == 512 && B > 0 && T0 > 0 && T0 <= R0 && nwc > 0 && nwc % 4 == 0 && nchunk > 0 && ldw >= 5 * (int64_t)(T0 - 1) + 10,
             "sc_conv0_gn_bwd: C must be 512 (got %d), nwc a multiple of 4", C);
    SC_CHECK(((uintptr_t)dy % 16) == 0 && ((uintptr_t)partial % 16) == 0, "sc_conv0_gn_bwd: alignment");
    const int rpw = (T0 + nwc - 1) / nwc;
    hipLaunchKernelGGL(conv0_gn_bwd_kernel, dim3(nwc / 4, B), dim3(256), 0, (hipStream_t)stream, wav, ldw, w0, scale, shift, dy, T0, R0, rpw, nwc,
                       partial);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv0_gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, partial, nwc, stats, nchunk, w0, gamma, beta, T0,
                       eps, contrib);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_conv0_ln_bwd(const float* wav, int64_t ldw, const float* w0, const float* bias, const float* gamma, const float* beta,
                               float eps, const sc_bf16* dy, int32_t B, int32_t T0, int32_t R0, int32_t C, float* partial, int32_t nwc,
                               void* stream) {
    SC_CHECK(wav && w0 && gamma && beta && dy && partial, "sc_conv0_ln_bwd: null pointer");
    SC_CHECK(C == 512 && B > 0 && T0 > 0 && T0 <= R0 && nwc > 0 && nwc % 4 == 0 && ldw >= 5 * (int64_t)(T0 - 1) + 10,
             "sc_conv0_ln_bwd: C must be 512 (got %d), nwc a multiple of 4", C);
    SC_CHECK(((uintptr_t)dy % 16) == 0 && ((uintptr_t)partial % 16) == 0, "sc_conv0_ln_bwd: alignment");
    const int rpw = (T0 + nwc - 1) / nwc;
    hipLaunchKernelGGL(conv0_ln_bwd_kernel, dim3(nwc / 4, B), dim3(256), 0, (hipStream_t)stream, wav, ldw, w0, bias, gamma, beta, eps,
                       (const uint16_t*)dy, T0, R0, rpw, nwc, partial);
    SC_LAUNCH_CHECK();
    return 0;
}
