// Keyword prompt of the cascaded branches (avssl/module/clip_official.py:222-279) in the row layout of the text tower.
//
// The reference builds, per sample, [SOT, kw_1 .. kw_n, EOT, token 0 ...] + positional_embedding with ~20 element-wise torch ops
// (zeros / scatter / embedding / where / cat / add) and gathers the end-of-text row of the transformer output with an advanced
// index; here one launch writes the tower's packed bf16 rows [Bp * SEG, W] directly (rows behind the prefix and pad samples zero),
// one launch reads the B end-of-text rows back as fp32, and the two backward launches are their exact adjoints.  HBM-bound on
// a few MB: latency-sized kernels whose point is the ~35 launches they replace.
#include "sc_common.h"

namespace {

// X[b SEG + t] = bf16(e(b, t) + pos[t]) for t < n_pos, b < B; 0 elsewhere.  e: t = 0 SOT; 1 <= t < index = count_b + 1: keyword t - 1
// (zero past the keyword tensor); t = index: EOT; behind it token 0 - exactly what clip_official.py:233-262 assembles.
__global__ __launch_bounds__(128) void prompt_assemble_kernel(const float* __restrict__ kw, int64_t ldb, const int64_t* __restrict__ count,
                                                              const float* __restrict__ tok, const float* __restrict__ pos,
                                                              uint16_t* __restrict__ X, int32_t* __restrict__ eot_row,
                                                              unsigned long long* __restrict__ clamped, int B, int N, int W, int SEG,
                                                              int n_pos) {
    const int r = blockIdx.x, b = r / SEG, t = r - b * SEG;
    uint16_t* x = X + (int64_t)r * W;
    if (b >= B || t >= n_pos) {
        for (int c = threadIdx.x * 8; c < W; c += 128 * 8) *(uint4*)(x + c) = make_uint4(0, 0, 0, 0);
        return;
    }
    const int64_t index = count[b] + 1;
    if (t == 0 && threadIdx.x == 0) {
        // the row the head reads: the end-of-text position, clamped into the prefix the tower sees (a count beyond the keyword
        // tensor is a caller error the reference reports as a shape mismatch; here it is clamped and counted)
        const bool over = index > n_pos - 1;
        eot_row[b] = b * SEG + (int)(over ? n_pos - 1 : (index < 0 ? 0 : index));
        if (over && clamped) atomicAdd(clamped, 1ull);
    }
    const float* e;
    bool zero = false;
    if (t == 0) e = tok;
    else if (t < index) { e = kw + (int64_t)b * ldb + (int64_t)(t - 1) * W; zero = (t - 1 >= N); }
    else if (t == index) e = tok + W;
    else e = tok + 2 * W;
    const float* pp = pos + (int64_t)t * W;
    for (int c = threadIdx.x * 4; c < W; c += 128 * 4) {
        const f32x4 pv = *(const f32x4*)(pp + c);
        f32x4 ev = {0.f, 0.f, 0.f, 0.f};
        if (!zero) ev = *(const f32x4*)(e + c);
        uint2 o;
        o.x = pack2bf(ev[0] + pv[0], ev[1] + pv[1]);
        o.y = pack2bf(ev[2] + pv[2], ev[3] + pv[3]);
        *(uint2*)(x + c) = o;
    }
}

// dkw[b, j] = float(dX[b SEG + j + 1]) for j + 1 < min(index_b, n_pos), else 0        (every element of dkw is written)
__global__ __launch_bounds__(128) void prompt_assemble_bwd_kernel(const uint16_t* __restrict__ dX, const int64_t* __restrict__ count,
                                                                  float* __restrict__ dkw, int64_t ldb, int N, int W, int SEG, int n_pos) {
    const int b = blockIdx.x / N, j = blockIdx.x - b * N;
    const int64_t index = count[b] + 1;
    const bool live = (j + 1 < index) && (j + 1 < n_pos);
    float* d = dkw + (int64_t)b * ldb + (int64_t)j * W;
    const uint16_t* g = dX + ((int64_t)b * SEG + j + 1) * W;
    for (int c = threadIdx.x * 4; c < W; c += 128 * 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const uint2 u = *(const uint2*)(g + c);
            v = f32x4{bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y)};
        }
        *(f32x4*)(d + c) = v;
    }
}

__global__ __launch_bounds__(128) void rows_gather_kernel(const uint16_t* __restrict__ X, const int32_t* __restrict__ row, float* __restrict__ out, int W) {
    const int b = blockIdx.x;
    const uint16_t* x = X + (int64_t)row[b] * W;
    for (int c = threadIdx.x * 4; c < W; c += 128 * 4) {
        const uint2 u = *(const uint2*)(x + c);
        *(f32x4*)(out + (int64_t)b * W + c) = f32x4{bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y)};
    }
}

// dX[r] = bf16(d[b]) if r is sample b's selected row (b = r / SEG < B), else 0            (every row of dX is written)
__global__ __launch_bounds__(128) void rows_scatter_kernel(const float* __restrict__ d, const int32_t* __restrict__ row, uint16_t* __restrict__ dX,
                                                           int B, int W, int SEG) {
    const int r = blockIdx.x, b = r / SEG;
    const bool hit = b < B && row[b] == r;
    uint16_t* x = dX + (int64_t)r * W;
    for (int c = threadIdx.x * 4; c < W; c += 128 * 4) {
        uint2 o = make_uint2(0, 0);
        if (hit) {
            const f32x4 v = *(const f32x4*)(d + (int64_t)b * W + c);
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
        }
        *(uint2*)(x + c) = o;
    }
}

// mask[b, k] = k >= lens[b] + add (1 = padding), k < n: the key-padding masks of the branch heads (avssl/util/data_utils.py:6-22) from the
// lengths in one launch (was arange + compare, and ones + copy for the attention block's padded pitch)
__global__ __launch_bounds__(256) void len_mask_kernel(const int64_t* __restrict__ lens, int add, uint8_t* __restrict__ mask, int B, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * n) return;
    const int b = (int)(i / n), k = (int)(i - (int64_t)b * n);
    mask[i] = (int64_t)k >= lens[b] + add ? 1 : 0;
}

}  // namespace

extern "C" int sc_prompt_assemble(const float* keywords, int64_t ldb, const int64_t* count, const float* tok, const float* pos, uint16_t* X,
                                  int32_t* eot_row, int64_t* clamped, int32_t B, int32_t Bp, int32_t N, int32_t W, int32_t SEG, int32_t n_pos,
                                  void* stream) {
    SC_CHECK(keywords && count && tok && pos && X && eot_row, "sc_prompt_assemble: null pointer");
    SC_CHECK(B > 0 && Bp >= B && N >= 0 && SEG > 0 && n_pos >= 2 && n_pos <= SEG, "sc_prompt_assemble: B=%d Bp=%d N=%d SEG=%d n_pos=%d", B, Bp, N, SEG, n_pos);
    SC_CHECK(W > 0 && W % 8 == 0 && ldb % 4 == 0, "sc_prompt_assemble: W=%d must be a multiple of 8 (ldb of 4)", W);
    SC_CHECK(((uintptr_t)keywords % 16) == 0 && ((uintptr_t)tok % 16) == 0 && ((uintptr_t)pos % 16) == 0 && ((uintptr_t)X % 16) == 0,
             "sc_prompt_assemble: operands must be 16-byte aligned");
    hipLaunchKernelGGL(prompt_assemble_kernel, dim3(Bp * SEG), dim3(128), 0, (hipStream_t)stream, keywords, ldb, count, tok, pos, X, eot_row,
                       (unsigned long long*)clamped, B, N, W, SEG, n_pos);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_prompt_assemble_bwd(const uint16_t* dX, const int64_t* count, float* dkeywords, int64_t ldb, int32_t B, int32_t N, int32_t W,
                                      int32_t SEG, int32_t n_pos, void* stream) {
    SC_CHECK(dX && count && dkeywords, "sc_prompt_assemble_bwd: null pointer");
    SC_CHECK(B > 0 && N > 0 && SEG > 0 && n_pos >= 2 && n_pos <= SEG && W > 0 && W % 8 == 0 && ldb % 4 == 0,
             "sc_prompt_assemble_bwd: B=%d N=%d W=%d SEG=%d n_pos=%d", B, N, W, SEG, n_pos);
    SC_CHECK(((uintptr_t)dX % 8) == 0 && ((uintptr_t)dkeywords % 16) == 0, "sc_prompt_assemble_bwd: alignment");
    hipLaunchKernelGGL(prompt_assemble_bwd_kernel, dim3(B * N), dim3(128), 0, (hipStream_t)stream, dX, count, dkeywords, ldb, N, W, SEG, n_pos);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rows_gather_bf16(const uint16_t* X, const int32_t* row, float* out, int32_t B, int32_t W, void* stream) {
    SC_CHECK(X && row && out && B > 0 && W > 0 && W % 8 == 0, "sc_rows_gather_bf16: bad arguments (B=%d W=%d)", B, W);
    SC_CHECK(((uintptr_t)X % 8) == 0 && ((uintptr_t)out % 16) == 0, "sc_rows_gather_bf16: alignment");
    hipLaunchKernelGGL(rows_gather_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, X, row, out, W);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_rows_scatter_bf16(const float* d, const int32_t* row, uint16_t* dX, int32_t M, int32_t B, int32_t W, int32_t SEG, void* stream) {
    SC_CHECK(d && row && dX && M > 0 && B > 0 && SEG > 0 && W > 0 && W % 8 == 0, "sc_rows_scatter_bf16: bad arguments (M=%d B=%d W=%d SEG=%d)", M, B, W, SEG);
    SC_CHECK(((uintptr_t)d % 16) == 0 && ((uintptr_t)dX % 8) == 0, "sc_rows_scatter_bf16: alignment");
    hipLaunchKernelGGL(rows_scatter_kernel, dim3(M), dim3(128), 0, (hipStream_t)stream, d, row, dX, B, W, SEG);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_len_mask_u8(const int64_t* lens, int32_t add, uint8_t* mask, int32_t B, int32_t n, void* stream) {
    SC_CHECK(lens && mask && B > 0 && n > 0, "sc_len_mask_u8: bad arguments");
    hipLaunchKernelGGL(len_mask_kernel, dim3((unsigned)(((int64_t)B * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lens, add, mask, B, n);
    SC_LAUNCH_CHECK();
    return 0;
}
