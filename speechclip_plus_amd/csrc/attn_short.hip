// Causal self-attention of SHORT sequences: the frozen CLIP text tower on the keyword prompts (clip_text_hip.py;
// avssl/module/clip_official.py:222-279 -> openai/CLIP Transformer, nn.MultiheadAttention with head_dim 64).  encode_keywords hands
// the tower the prompt prefix only (2 + keywords <= 32 tokens), so a sequence is ONE 32-row segment and a (sequence, head) problem is
// 32 x 32 scores on 64 channels: 8 matrix instructions.  The 128-row flash kernels spend a launch each on V^T, the forward, the
// backward's row sums, dQ and dK/dV and mask 7/8 of every score tile; here one wave owns a (sequence, head) from its global loads to
// its stores - forward 1 launch (no V^T, no LSE), backward 1 launch that recomputes the probabilities (no saved output, no delta pass).
//
// Layout (v_mfma_f32_32x32x16_bf16; lane l: r = l & 31, h = l >> 5; accumulator register g <-> row (g & 3) + 8 (g >> 2) + 4 h, column r):
//   X  = S^T = K Q^T   [key][query]   A = K rows, B = Q rows: both 16-byte global loads, the same fragments give
//   X' = S   = Q K^T   [query][key]   with the operands swapped.  A softmax over keys is a reduction over X's registers (+ one lane
//   swap) per query lane.  Products that sum over an accumulator's ROW index take it as the B operand straight from the registers
//   (registers 8s..8s+7 = k-step s, k order permuted: element j of half h is row 16 s + 8 (j >> 2) + 4 h + (j & 3)); the A operand of
//   those products is a TRANSPOSED tile (V^T, K^T, Q^T, dO^T), gathered in the same k order from the wave's row-major LDS copy:
//     forward    O^T  [d][query] = V^T  P^T          (X  as B)        -> 8-byte stores of 4 channels per (query, register group)
//     backward   dQ^T [d][query] = K^T  dS^T         (X  as B)
//                dV^T [d][key]   = dO^T P            (X' as B)
//                dK^T [d][key]   = Q^T  dS           (X' as B)
//   dP^T = V dO^T and dP = dO V^T come from the row fragments like the scores; delta = sum_k P dP is a register sum in the X layout
//   and reaches the X' layout (queries on registers) through 32 floats of LDS, like the row maximum and 1 / sum.
// Rows behind a prompt (scratch rows of the segment) are ordinary causal queries / keys: finite values, and zero gradient as long as
// the caller's d out is zero there (clip_text_hip.tower_backward).
#include "sc_common.h"

namespace {

constexpr int SEG = 32, DH = 64, PITCH = 72;      // LDS row pitch in elements: 144 bytes (16-byte aligned rows)

__device__ __forceinline__ int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }
__device__ __forceinline__ __bf16 to_bf(float f) { return __builtin_bit_cast(__bf16, f2bf(f)); }

// the A operand of a product that sums over an accumulator's row index: column (32 nb + r) of the wave's row-major tile, rows in the
// accumulator's k order
__device__ __forceinline__ bf16x8 gather_col(const uint16_t (*tile)[PITCH], int nb, int s, int r, int h) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = __builtin_bit_cast(__bf16, tile[16 * s + 8 * (j >> 2) + 4 * h + (j & 3)][32 * nb + r]);
    return a;
}

__device__ __forceinline__ void load_rows(bf16x8 f[4], const uint16_t* base, int64_t ld, int r, int h) {
#pragma unroll
    for (int t = 0; t < 4; ++t) f[t] = *(const bf16x8*)(base + (int64_t)r * ld + 16 * t + 8 * h);
}
__device__ __forceinline__ void stage_rows(uint16_t (*tile)[PITCH], const bf16x8 f[4], int r, int h) {
#pragma unroll
    for (int t = 0; t < 4; ++t) *(bf16x8*)&tile[r][16 * t + 8 * h] = f[t];
}
__device__ __forceinline__ f32x16 mma4(const bf16x8 a[4], const bf16x8 b[4]) {
    f32x16 c;
#pragma unroll
    for (int g = 0; g < 16; ++g) c[g] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[t], c, 0, 0, 0);
    return c;
}
// y^T [d][column on the lane] -> y[column][hd 64 + d]: register group gq holds 4 consecutive channels
__device__ __forceinline__ void store_T(uint16_t* dst, int64_t ld, int r, int h, int nb, const f32x16& y, float mul) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        uint2 o;
        o.x = pack2bf(y[4 * gq] * mul, y[4 * gq + 1] * mul);
        o.y = pack2bf(y[4 * gq + 2] * mul, y[4 * gq + 3] * mul);
        *(uint2*)(dst + (int64_t)r * ld + 32 * nb + 8 * gq + 4 * h) = o;
    }
}

__global__ __launch_bounds__(256) void attn32_fwd_kernel(const uint16_t* __restrict__ qkv, int64_t ld, uint16_t* __restrict__ out,
                                                         int64_t ldo, int nprob, int heads, int W, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) uint16_t vt[4][SEG][PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int prob = min(blockIdx.x * 4 + wave, nprob - 1);          // (a surplus wave repeats the last problem: same values, same address)
    const int seq = prob / heads, hd = prob - seq * heads;
    const uint16_t* base = qkv + (int64_t)seq * SEG * ld + hd * DH;
    bf16x8 qf[4], kf[4], vf[4];
    load_rows(qf, base, ld, r, h);
    load_rows(kf, base + W, ld, r, h);
    load_rows(vf, base + 2 * W, ld, r, h);
    stage_rows(vt[wave], vf, r, h);
    f32x16 x = mma4(kf, qf);                                          // [key][query r]
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        x[g] = acc_row(g, h) <= r ? x[g] * scale_log2e : -INFINITY;
        mx = fmaxf(mx, x[g]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));                               // key 0 is visible to every query: finite
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        x[g] = exp2f(x[g] - mx);
        sum += x[g];
    }
    sum += __shfl_xor(sum, 32);
    bf16x8 pb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[s][j] = to_bf(x[8 * s + j]);
    __syncthreads();
    const float inv = 1.f / sum;
    uint16_t* dst = out + (int64_t)seq * SEG * ldo + hd * DH;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        f32x16 o;
#pragma unroll
        for (int g = 0; g < 16; ++g) o[g] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_col(vt[wave], nb, s, r, h), pb[s], o, 0, 0, 0);
        store_T(dst, ldo, r, h, nb, o, inv);
    }
}

__global__ __launch_bounds__(256) void attn32_bwd_kernel(const uint16_t* __restrict__ qkv, int64_t ld, const uint16_t* __restrict__ dout,
                                                         int64_t ldd, uint16_t* __restrict__ dqkv, int64_t ldg, int nprob, int heads,
                                                         int W, float scale) {
    __shared__ __attribute__((aligned(16))) uint16_t tq[4][SEG][PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t tk[4][SEG][PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t tdo[4][SEG][PITCH];
    __shared__ float st[4][3][SEG];                                    // per query: row maximum, 1 / sum, delta
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int prob = min(blockIdx.x * 4 + wave, nprob - 1);
    const int seq = prob / heads, hd = prob - seq * heads;
    const uint16_t* base = qkv + (int64_t)seq * SEG * ld + hd * DH;
    const float scale_log2e = scale * 1.4426950408889634f;
    bf16x8 qf[4], kf[4], vf[4], df[4];
    load_rows(qf, base, ld, r, h);
    load_rows(kf, base + W, ld, r, h);
    load_rows(vf, base + 2 * W, ld, r, h);
    load_rows(df, dout + (int64_t)seq * SEG * ldd + hd * DH, ldd, r, h);
    stage_rows(tq[wave], qf, r, h);
    stage_rows(tk[wave], kf, r, h);
    stage_rows(tdo[wave], df, r, h);
    uint16_t* gbase = dqkv + (int64_t)seq * SEG * ldg + hd * DH;
    // ---- keys on the registers, query r on the lane: statistics, delta, dQ
    {
        f32x16 x = mma4(kf, qf);
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            x[g] = acc_row(g, h) <= r ? x[g] * scale_log2e : -INFINITY;
            mx = fmaxf(mx, x[g]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            x[g] = exp2f(x[g] - mx);
            sum += x[g];
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
        const f32x16 dp = mma4(vf, df);                               // dP^T [key][query r]
        float delta = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            x[g] *= inv;                                              // P^T (0 above the diagonal)
            delta = fmaf(x[g], dp[g], delta);
        }
        delta += __shfl_xor(delta, 32);
        if (h == 0) {
            st[wave][0][r] = mx;
            st[wave][1][r] = inv;
            st[wave][2][r] = delta;
        }
        bf16x8 sb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) sb[s][j] = to_bf(scale * x[8 * s + j] * (dp[8 * s + j] - delta));
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x16 y;
#pragma unroll
            for (int g = 0; g < 16; ++g) y[g] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_col(tk[wave], nb, s, r, h), sb[s], y, 0, 0, 0);
            store_T(gbase, ldg, r, h, nb, y, 1.f);                    // dQ
        }
    }
    // ---- queries on the registers, key r on the lane: dV, dK
    {
        f32x16 x = mma4(qf, kf);                                      // [query][key r]
        const f32x16 dp = mma4(df, vf);                               // dP [query][key r]
        bf16x8 pb[2], sb[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int q = acc_row(g, h);
            const float p = r <= q ? exp2f(x[g] * scale_log2e - st[wave][0][q]) * st[wave][1][q] : 0.f;
            pb[g >> 3][g & 7] = to_bf(p);
            sb[g >> 3][g & 7] = to_bf(scale * p * (dp[g] - st[wave][2][q]));
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x16 yv, yk;
#pragma unroll
            for (int g = 0; g < 16; ++g) yv[g] = yk[g] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                yv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_col(tdo[wave], nb, s, r, h), pb[s], yv, 0, 0, 0);
                yk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_col(tq[wave], nb, s, r, h), sb[s], yk, 0, 0, 0);
            }
            store_T(gbase + 2 * W, ldg, r, h, nb, yv, 1.f);           // dV
            store_T(gbase + W, ldg, r, h, nb, yk, 1.f);               // dK
        }
    }
}

}  // namespace

extern "C" int sc_attn32_fwd_bf16(const sc_bf16* qkv, int64_t ld, sc_bf16* out, int64_t ldo, int32_t nseq, int32_t heads, float scale,
                                  void* stream) {
    SC_CHECK(qkv && out && nseq > 0 && heads > 0, "sc_attn32_fwd_bf16: bad arguments");
    SC_CHECK(ld % 8 == 0 && ldo % 4 == 0 && ld >= 3 * heads * DH && ldo >= heads * DH && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0,
             "sc_attn32_fwd_bf16: qkv rows must be 16-byte aligned [.., 3 heads 64], out rows 8-byte aligned (ld=%lld ldo=%lld)", (long long)ld,
             (long long)ldo);
    const int nprob = nseq * heads;
    hipLaunchKernelGGL(attn32_fwd_kernel, dim3((nprob + 3) / 4), dim3(256), 0, (hipStream_t)stream, qkv, ld, out, ldo, nprob, heads,
                       heads * DH, scale * 1.4426950408889634f);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_attn32_bwd_bf16(const sc_bf16* qkv, int64_t ld, const sc_bf16* dout, int64_t ldd, sc_bf16* dqkv, int64_t ldg, int32_t nseq,
                                  int32_t heads, float scale, void* stream) {
    SC_CHECK(qkv && dout && dqkv && nseq > 0 && heads > 0, "sc_attn32_bwd_bf16: bad arguments");
    SC_CHECK(ld % 8 == 0 && ldd % 8 == 0 && ldg % 4 == 0 && ld >= 3 * heads * DH && ldd >= heads * DH && ldg >= 3 * heads * DH &&
                 ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)dqkv % 8) == 0,
             "sc_attn32_bwd_bf16: qkv / dout rows must be 16-byte aligned, dqkv rows 8-byte aligned");
    const int nprob = nseq * heads;
    hipLaunchKernelGGL(attn32_bwd_kernel, dim3((nprob + 3) / 4), dim3(256), 0, (hipStream_t)stream, qkv, ld, dout, ldd, dqkv, ldg, nprob,
                       heads, heads * DH, scale);
    SC_LAUNCH_CHECK();
    return 0;
}
