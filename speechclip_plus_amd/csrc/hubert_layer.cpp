// sc_hubert_layer_fwd: one C-ABI call that enqueues a whole frozen HuBERT encoder layer on the caller's stream
// (fairseq TransformerSentenceEncoderLayer as invoked at avssl/module/speech_encoder_plus.py:49-53), post-LN (base) or pre-LN
// (large) order, eval or train-mode dropout:
//     QKV GEMM (V stored transposed per head) -> flash attention -> out_proj GEMM (+bias, dropout, +residual) -> LayerNorm
//     -> FC1 GEMM (+bias, erf-GELU) -> FC2 GEMM (+bias, dropout, +residual) -> LayerNorm
// It only sequences the library's own entry points (no new arithmetic), so a binding does 1 FFI call per layer instead of 7.
// sc_workspace_bytes reports the scratch a caller has to provide.
#include <hip/hip_runtime.h>
#include <string.h>

#include "sc_common.h"

static int gemm(const sc_bf16* A, int64_t lda, const sc_bf16* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K, const float* bias,
                const sc_bf16* residual, int64_t ldr, int act, float drop_p, uint32_t drop_seed, sc_bf16* Ct, int n_split, int R, int dh,
                void* stream, const int32_t* seg_chunk = nullptr) {
    sc_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    a.bias = bias; a.residual = residual; a.ldr = ldr; a.act = act;
    a.Ct = Ct; a.n_split = n_split; a.R = R; a.dh = dh;
    a.nb1 = a.nb2 = 1;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    a.seg_chunk = seg_chunk;
    return sc_gemm_bf16(&a, stream);
}

extern "C" int64_t sc_workspace_bytes(int32_t what, int64_t a, int64_t b, int64_t c) {
    switch (what) {
        case SC_WS_INFONCE:          // a = Bg
            return 4 * sc_infonce_workspace_floats((int32_t)a);
        case SC_WS_HUBERT_LAYER: {   // a = B * R rows, b = D, c = F: qk [M, 2D] + vt [M, D] + ctx [M, D] + pre [M, D] + x1 [M, D] + ffn [M, F], bf16
            return 2 * a * (6 * b + c);
        }
        default:
            sc_set_error("sc_workspace_bytes: unknown kind %d", what);
            return -1;
    }
}

extern "C" int sc_hubert_layer_fwd(const sc_hubert_layer_args* p, void* stream) {
    SC_CHECK(p && p->x && p->out && p->valid_len && p->qk && p->vt && p->ctx && p->pre && (p->x1 || p->fused_ln) && p->ffn,
             "sc_hubert_layer_fwd: null pointer");
    const sc_segments* seg = p->seg;
    if (seg) {
        SC_CHECK(seg->row0 && seg->chunk && seg->B > 0 && seg->rows > 0 && seg->rows % SC_SEG_ROWS == 0 && p->H > 0 && p->D == p->H * 64 && p->F > 0,
                 "sc_hubert_layer_fwd: segment table B=%d rows=%d D=%d (= 64 H) F=%d", seg->B, seg->rows, p->D, p->F);
    } else {
        SC_CHECK(p->B > 0 && p->R > 0 && p->R % 8 == 0 && p->H > 0 && p->D == p->H * 64 && p->F > 0 && p->T > 0 && p->T <= p->R,
                 "sc_hubert_layer_fwd: B=%d R=%d (%% 8) T=%d D=%d (= 64 H) F=%d", p->B, p->R, p->T, p->D, p->F);
    }
    const int M = seg ? seg->rows : p->B * p->R, D = p->D, F = p->F, H = p->H;
    const float scale = 0.125f;       // head_dim 64
    const int32_t* chunk = seg ? seg->chunk : nullptr;
    auto attention = [&]() -> int {
        if (seg)
            return sc_attn_fwd_seg_bf16(p->qk, 2 * D, p->vt, p->valid_len, p->ctx, D, seg, p->attn_work, p->n_attn_work, H, D, scale, nullptr, 0,
                                        p->p_attn, p->seed_attn, stream);
        return sc_attn_fwd_bf16(p->qk, 2 * D, p->vt, p->valid_len, p->ctx, D, p->B, p->R, H, D, scale, nullptr, 0, p->p_attn, p->seed_attn, stream);
    };
    int rc;
    if (p->fused_ln) {
        // LayerNorm-free form: QKV (A raw or materialised) -> attention -> out_proj (+ lazily normalised residual, statistics of
        // `pre`) -> FC1 on the raw `pre` with the LN1-folded weights -> FC2 (+ LN1(pre) as residual, statistics of `out`)
        SC_CHECK(!p->pre_ln, "sc_hubert_layer_fwd: fused_ln is built for the post-LN order");
        SC_CHECK(p->fc1_colsum && p->stats1 && p->out_stats && (!p->x_stats || (p->x_ln_g && p->x_ln_b && p->qkv_colsum && p->x_ns > 0)),
                 "sc_hubert_layer_fwd: fused_ln needs fc1_colsum, stats1, out_stats (and x_ln_g / x_ln_b / qkv_colsum with x_stats)");
        sc_gemm_args a;
        auto fill = [&](const sc_bf16* A, int64_t lda, const sc_bf16* W, int64_t ldw, void* C, int64_t ldc, int N, int K, const float* bias) {
            memset(&a, 0, sizeof(a));
            a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc;
            a.M = M; a.N = N; a.K = K; a.bias = bias;
            a.n_split = -1; a.nb1 = a.nb2 = 1; a.ln_eps = p->eps;
        };
        fill(p->x, D, p->qkv_w, D, p->qk, 2 * D, 3 * D, D, p->qkv_b);
        a.Ct = p->vt; a.n_split = 2 * D; a.R = p->R; a.dh = 64; a.seg_chunk = chunk;
        if (p->x_stats) { a.ln_stats = p->x_stats; a.ln_ns = p->x_ns; a.ln_colsum = p->qkv_colsum; }
        if ((rc = sc_gemm_bf16(&a, stream))) return rc;
        if ((rc = attention())) return rc;
        fill(p->ctx, D, p->o_w, D, p->pre, D, D, D, p->o_b);
        a.residual = p->x; a.ldr = D; a.drop_p = p->p_res; a.drop_seed = p->seed_o; a.stats_out = p->stats1;
        if (p->x_stats) { a.res_stats = p->x_stats; a.res_ns = p->x_ns; a.res_gamma = p->x_ln_g; a.res_beta = p->x_ln_b; }
        const int ns1 = sc_gemm_stats_strips(&a);
        if ((rc = sc_gemm_bf16(&a, stream))) return rc;
        fill(p->pre, D, p->fc1_w, D, p->ffn, F, F, D, p->fc1_b);
        a.act = 1; a.ln_stats = p->stats1; a.ln_ns = ns1; a.ln_colsum = p->fc1_colsum;
        if ((rc = sc_gemm_bf16(&a, stream))) return rc;
        fill(p->ffn, F, p->fc2_w, F, p->out, D, D, F, p->fc2_b);
        a.residual = p->pre; a.ldr = D; a.drop_p = p->p_res; a.drop_seed = p->seed_fc2; a.stats_out = p->out_stats;
        a.res_stats = p->stats1; a.res_ns = ns1; a.res_gamma = p->ln1_g; a.res_beta = p->ln1_b;
        return sc_gemm_bf16(&a, stream);
    }
    const sc_bf16* attn_in = p->x;
    if (p->pre_ln) {                  // x1 = LN1(x)
        if ((rc = sc_layernorm_bf16(p->x, D, p->ln1_g, p->ln1_b, p->x1, D, M, D, p->eps, 0, stream))) return rc;
        attn_in = p->x1;
    }
    if ((rc = gemm(attn_in, D, p->qkv_w, D, p->qk, 2 * D, M, 3 * D, D, p->qkv_b, nullptr, 0, 0, 0.f, 0, p->vt, 2 * D, p->R, 64, stream, chunk))) return rc;
    if ((rc = attention())) return rc;
    if ((rc = gemm(p->ctx, D, p->o_w, D, p->pre, D, M, D, D, p->o_b, p->x, D, 0, p->p_res, p->seed_o, nullptr, -1, 0, 0, stream))) return rc;
    if (p->pre_ln) {                  // pre = x + attn ; x1 = LN2(pre) ; out = pre + ffn(x1)
        if ((rc = sc_layernorm_bf16(p->pre, D, p->ln2_g, p->ln2_b, p->x1, D, M, D, p->eps, 0, stream))) return rc;
        if ((rc = gemm(p->x1, D, p->fc1_w, D, p->ffn, F, M, F, D, p->fc1_b, nullptr, 0, 1, 0.f, 0, nullptr, -1, 0, 0, stream))) return rc;
        return gemm(p->ffn, F, p->fc2_w, F, p->out, D, M, D, F, p->fc2_b, p->pre, D, 0, p->p_res, p->seed_fc2, nullptr, -1, 0, 0, stream);
    }
    // post-LN: x1 = LN1(x + attn) ; out = LN2(x1 + ffn(x1))
    if ((rc = sc_layernorm_bf16(p->pre, D, p->ln1_g, p->ln1_b, p->x1, D, M, D, p->eps, 0, stream))) return rc;
    if ((rc = gemm(p->x1, D, p->fc1_w, D, p->ffn, F, M, F, D, p->fc1_b, nullptr, 0, 1, 0.f, 0, nullptr, -1, 0, 0, stream))) return rc;
    if ((rc = gemm(p->ffn, F, p->fc2_w, F, p->pre, D, M, D, F, p->fc2_b, p->x1, D, 0, p->p_res, p->seed_fc2, nullptr, -1, 0, 0, stream))) return rc;
    return sc_layernorm_bf16(p->pre, D, p->ln2_g, p->ln2_b, p->out, D, M, D, p->eps, 0, stream);
}
