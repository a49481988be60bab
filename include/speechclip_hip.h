/*
 * speechclip_hip.h - C ABI of libspeechclip_hip.so (gfx950 / MI355X).
 *
 * The reference (ShampooWang/SpeechCLIP_plus) is 100 % Python and has no FFI / operator layer of its own
 * (SURVEY.md F1); its seams are nn.Module.forward signatures.  This header is therefore the boundary a
 * maintainer would bind *under* those modules (ctypes stub: INTEGRATION.md).  Every entry point names the
 * reference computation it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all buffers
 *     (outputs and workspaces are allocated by the caller and passed in).
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, never synchronise, keep no
 *     global state.  Re-entrant per stream.
 *   - return 0 on success, <0 on error; sc_last_error() gives a thread-local message.
 *   - bf16 tensors are raw uint16 storage (`sc_bf16`); "rows" of activations live in the padded layout
 *     described in DESIGN.md: utterance b, frame t  ->  row  b*R + t (uniform pitch R), or - round 4, the `_seg` entry points -
 *     row  row0[b] + t  with a pitch per utterance (`sc_segments`): work follows the real lengths of a ragged batch.
 */
#ifndef SPEECHCLIP_HIP_H
#define SPEECHCLIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t sc_bf16;

const char* sc_last_error(void);
int sc_abi_version(void);     /* 4 since round 4 (sc_segments; sc_gemm_args / sc_hubert_layer_args grew the segment fields) */
int sc_is_diag_build(void);   /* 1: libspeechclip_hip_diag.so - the same sources built with SC_DIAG_BUILD: also holds the diagnostic kernels
                                 (sc_gemm_args.tile 32 = timing only, RESULTS WRONG; 34 = stamped) and the LayerNorm-folded GEMMs; the
                                 product library refuses both */
int64_t sc_sizeof(int32_t what);   /* sizeof of 0 sc_gemm_args, 1 sc_hubert_layer_args, 2 sc_rt_gemm_args, 3 sc_rt_ln_args, 4 sc_rt_ln_bwd_args, 5 sc_segments */
/* tuning switches for same-process A/B measurements (tools/); results never depend on them.  key 1: the 256-row GEMM uses
 * plain instead of non-temporal stores on tiles with a residual.  key 6: sc_cif_fwd* / sc_cif_bwd* walk the frames in one wave per
 * (utterance, 256 channels) instead of one wave per output slot / per 8 frames (the same bits, 3.7x / 4.5x the time at 64 x 499). */
int sc_set_option(int32_t key, int32_t value);

/* ------------------------------------------------------------------------------------------------
 * Ragged row layout (round 4).  The reference pads every batch to its longest utterance and computes all of it
 * (avssl/module/speech_encoder_plus.py:506-518, 548-552); here utterance b owns rows [row0[b], row0[b + 1]) of every
 * [rows, C] activation buffer - its own pitch, a multiple of SC_SEG_ROWS = 8 rows, sized by its own number of frames - so GEMM rows,
 * attention blocks and row kernels follow the real lengths.  Down the conv stack layer l has the same table scaled by
 * 2^(6-l) (row0 * 64 at conv layer 0), the waveform by `samples_per_row` (320): a strided Conv1d stays ONE flat GEMM.
 *   row0   device [B + 1] int32, multiples of 8, row0[0] = 0, row0[B] = rows
 *   chunk  device [rows / 8][4] int32: for every 8-row chunk (first row of its utterance, pitch of its utterance, utterance, 0)
 *          - the reverse lookup of the flat kernels (one 16-byte load)
 *   max_pitch = max_b (row0[b + 1] - row0[b])   (grid sizing on the host)
 * The struct itself lives in HOST memory (passed by pointer), its two tables in device memory.
 * ---------------------------------------------------------------------------------------------- */
#define SC_SEG_ROWS 8
typedef struct {
    const int32_t* row0;
    const int32_t* chunk;
    int32_t B, rows, max_pitch, reserved;
} sc_segments;

/* ------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogue:  C = epi(A . W^T)      (every Linear / Conv1d on the path)
 *   replaces: fairseq Linear/Conv1d calls reached from avssl/module/speech_encoder_plus.py:75-105
 *             (conv feature extractor layers 1-6 as strided-row GEMMs, post_extract_proj, q/k/v/out_proj,
 *             fc1, fc2, pos_conv per group).
 *   A   [M, K] bf16, row stride lda (elements; rows may overlap - this is how a channels-last strided
 *       Conv1d becomes a GEMM: lda = stride*C_in, K = k*C_in)
 *   W   [N, K] bf16, row stride ldw   (torch Linear layout [out, in])
 *   C   [M, N] bf16 (or fp32 if out_f32), row stride ldc
 *   epilogue order:  acc -> +bias[n] -> act (0 none, 1 GELU-erf) -> dropout (train mode) -> +residual[m, n] -> store
 *   dropout (fairseq dropout_input / the layers' residual dropouts, F.dropout semantics: keep with probability 1 - p, scale the
 *       kept values by 1 / (1 - p)): stateless counter-based mask - element (m, n) is kept iff the 16-bit lane of
 *       sc_hash32((m*N + n) / 2 ^ drop_seed) selected by (m*N + n) & 1 is >= round(p * 65536); not applied to Ct columns
 *   columns n >= n_split (if n_split >= 0) are stored TRANSPOSED per head into Ct:
 *       Ct[(((m / R) * H + (n-n_split)/dh) * dh + (n-n_split)%dh) * R + m % R]     (V^T for attention)
 *   batch: grid.z = nb1*nb2 ; operand pointers advance by  (z / nb2) * s?1 + (z % nb2) * s?2  elements.
 *   Requirements: K % 64 == 0, lda/ldw % 8 == 0, ldc % 8 == 0; A, W, C, residual AND bias 16-byte aligned (the epilogue reads the
 *   bias in groups of four floats), sBias1 / sBias2 multiples of 4.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const sc_bf16* A; int64_t lda;
    const sc_bf16* W; int64_t ldw;
    void* C; int64_t ldc;
    int32_t M, N, K;
    const float* bias;            /* [N] fp32 or NULL */
    const sc_bf16* residual; int64_t ldr;   /* [M, N] bf16 or NULL */
    int32_t act;                  /* 0 none, 1 gelu(erf) (fused epilogue of every tile family); 2 QuickGELU only with aux_mode */
    int32_t out_f32;              /* 0: C is bf16, 1: C is fp32 */
    sc_bf16* Ct; int32_t n_split; int32_t R; int32_t dh;   /* transposed-store region; n_split < 0 disables */
    int32_t nb1, nb2;             /* batch = nb1*nb2 (>= 1 each) */
    int64_t sA1, sA2, sW1, sW2, sC1, sC2, sBias1, sBias2, sR1, sR2;
    int32_t tile;                 /* 0 auto | 1: 128x128 (4 waves) | 2: 256-row tile, 8 waves, width 256 or 192 by wave quantisation
                                     (7 / 8 force 192 / 256) | 3: 128x64 (13: four LDS stages) | 14 / 15: 64x64 with two / four stages
                                     (few rows, narrow output, long K).  Every tile family accumulates k in the same order and
                                     shares the epilogue arithmetic: results do not depend on the choice.  (32 / 34: diagnostic
                                     kernels of libspeechclip_hip_diag.so only; the product library returns an error) */
    int32_t reserved;             /* store policy of the bf16 C tile (256-row kernels): 0 auto = non-temporal stores when a residual
                                     is given (the output is the next residual stream, not re-read by this kernel; measured
                                     -3.5 % GEMM time per step), 1 always non-temporal, 2 never */
    float drop_p;                 /* 0 = no dropout */
    uint32_t drop_seed;
    int32_t tap_c;                /* K-tile visiting-order hint for conv-shaped A (rows overlap: lda = 2*tap_c, K = 3*tap_c, i.e. a
                                     channels-last Conv1d with k = 3, stride 2): 0 = ascending k; tap_c = C_in visits, per 64-channel
                                     block, tap 0, tap 2, tap 1 - tap 2 of output row r is tap 0 of row r + 1, so its tile is
                                     re-read while still in the L2 instead of 16 K-tiles later from the fabric. Only the
                                     fp32 summation order over k changes; every tile family uses the same order, so results do not depend on
                                     the dispatcher's choice. */
    /* ---- LayerNorm folded into its neighbour GEMMs (256-row tile family only; see csrc/gemm256_bf16.hip, "LN").  A row-statistics
     * buffer is [M][8][2] fp32: per row 8 strip slots of (sum, sum of squares) - one strip per N-tile of the GEMM that wrote it, at most 4 used (N <= 1024); the
     * strips a producer does not write must be ZERO (allocate the buffer zeroed): consumers add the first four slots of a line.
     *   consumer (no residual, no dropout): A holds RAW (pre-LayerNorm) rows, W = W0 diag(gamma) folded by the caller,
     *       ln_stats / ln_ns = the statistics of A's rows and the number of valid strips, ln_colsum[n] = sum_k W[n,k] (of the bf16
     *       values), bias[n] = sum_k beta[k] W0[n,k] + bias0[n];   C = rstd_m (A W^T - mean_m ln_colsum) + bias  -> act -> store
     *   producer (residual given, act = 0): stats_out receives the statistics of the bf16 rows of C (strip = N-tile index;
     *       sc_gemm_stats_strips() tells how many the dispatcher's tile width gives); res_stats (+ res_ns, res_gamma, res_beta):
     *       the residual operand holds RAW rows too and enters as LayerNorm(residual row) = (r - mean) rstd gamma[n] + beta[n].
     *   ln_eps: the LayerNorm epsilon of both uses. */
    int32_t ln_ns;
    const float* ln_stats;
    const float* ln_colsum;
    const float* res_stats;
    const float* res_gamma;
    const float* res_beta;
    float* stats_out;
    int32_t res_ns;
    float ln_eps;
    /* ---- TN form (256 x 256 tiles only): tn = 1 reads BOTH operands with the reduction index as the row index,
     *       C[m, n] = sum_r A[r, m] W[r, n],   A [K, M] row stride lda, W [K, N] row stride ldw  (K = number of rows r),
     * i.e. a weight gradient dW = dY^T X straight from the row-major dY [rows, M] and X [rows, N] (rows may overlap: the im2col
     * view of a strided conv input), no transposed copies.  Requirements: M % 256 == 0, N % 256 == 0, K % 64 == 0, no bias /
     * activation / residual / dropout / transposed store / LayerNorm folding; split along r through the batch strides
     * (sA1 = rows_per_slice * lda, sW1 = rows_per_slice * ldw) into fp32 partials (out_f32 = 1) or a single bf16 / fp32 result.
     * k_total > 0: the slices are ragged - slice z1 covers rows [z1 K, min((z1 + 1) K, k_total)) (K, k_total multiples of 64). */
    int32_t tn;
    int32_t k_total;
    /* ---- activation fused with a second operand (Ct doubles as the aux pointer [M, N] bf16, row stride ldc; act = 1 erf-GELU - every
     * tile family, the 256-row one without a residual (round 4) - or 2 QuickGELU - 128-row tiles; no transposed store, bf16 output):
     *   aux_mode 1  dual store:  Ct <- u = bf16(acc + bias) (the pre-activation the backward needs),  C <- act(u)   (FFN fc1 of a layer
     *               that is differentiated: one launch instead of GEMM + sc_act_bf16)
     *   aux_mode 2  C <- bf16(acc) * act'(Ct)   (the input-gradient GEMM of fc2 followed by the activation's backward: Ct = the saved u)
     * Both reproduce the two-launch sequence bit for bit (the activation reads the ROUNDED values the first kernel would have stored). */
    int32_t aux_mode;
    int32_t reserved3;
    /* ---- ragged rows (round 4): seg_chunk != NULL replaces the uniform `R` of the transposed store - row m belongs to the
     * utterance of chunk m / 8 = (first row r0, pitch Rb, ...) (sc_segments.chunk) and goes to
     *       Ct[(N - n_split) * r0 + (n - n_split) * Rb + (m - r0)]          (V^T [H, dh, Rb] per utterance, utterances back to back) */
    const int32_t* seg_chunk;
} sc_gemm_args;
int32_t sc_gemm_stats_strips(const sc_gemm_args* args);   /* strips a producer launch with these args writes per row (0: not on the 256-row family) */
int sc_gemm_bf16(const sc_gemm_args* args, void* stream);
uint32_t sc_hash32(uint32_t x);   /* host twin of the kernels' dropout hash (lowbias32): reconstructs a mask exactly */

/* ------------------------------------------------------------------------------------------------
 * Self-attention forward (flash style, key-padding by length), head_dim 64.
 *   replaces: fairseq MultiheadAttention inside TransformerSentenceEncoderLayer, invoked at
 *             avssl/module/speech_encoder_plus.py:52  (softmax((q*dh^-.5) k^T + kpm(-inf)) v)
 *   qk   [B*R, ldqk] bf16 : columns [0, D) = q (unscaled), [D, 2D) = k ; head h owns 64 columns
 *   vt   [B, H, 64, R] bf16 : v transposed per head (written by sc_gemm_bf16's Ct path)
 *   valid_len [B] int32 : keys t >= valid_len[b] are masked (-inf)
 *   out  [B*R, ldo] bf16, columns h*64..h*64+63
 *   causal (CLIP text tower): 1 = key t visible to query q iff t <= q ; 32 / 64 = causal INSIDE aligned segments of that many
 *       rows (short sequences packed back to back into one 128-row block attend to their own segment only); same for the backward
 * ---------------------------------------------------------------------------------------------- */
int sc_attn_fwd_bf16(const sc_bf16* qk, int64_t ldqk, const sc_bf16* vt, const int32_t* valid_len,
                     sc_bf16* out, int64_t ldo, int32_t B, int32_t R, int32_t H, int32_t D, float scale,
                     float* lse2 /* [B,H,R] log2-domain log-sum-exp for the backward, or NULL */,
                     int32_t causal /* 1: key t' > query t masked too (CLIP text tower) */,
                     float drop_p, uint32_t drop_seed /* attention-probability dropout (train mode): P' = mask . P / (1 - p), element
                     ((b*H + h)*R + q)*R + k hashed as in sc_gemm_args; 0 = off */, void* stream);

/* The same over ragged rows (sc_segments): q / k rows of utterance b at row0[b] + t, vt = per utterance [H, 64, Rb] at element offset
 * D * row0[b] (what sc_gemm_bf16 writes with seg_chunk), out rows likewise; lse2 [H][rows].  One workgroup = 128 queries of one
 * (utterance, head); in a last, shorter block the waves (32 queries each) past the pitch idle and the rows past it are not stored.  work (optional, device
 * [nwork][4] int32, 16-byte aligned): the 128-query blocks to run, in launch order, one item = (utterance | q-block << 16, row0 of the
 * utterance, its pitch, its key count or -1 = read valid_len) - the host sorts them longest first (an utterance's cost grows with
 * its key count); NULL = every (b, q-block < max_pitch / 128), blocks past an utterance's pitch exit.  Dropout element index:
 * ((h * rows + row0[b] + q) * max_pitch + k).  Bit-identical to the uniform call on the rows they share. */
int sc_attn_fwd_seg_bf16(const sc_bf16* qk, int64_t ldqk, const sc_bf16* vt, const int32_t* valid_len, sc_bf16* out, int64_t ldo,
                         const sc_segments* seg, const int32_t* work, int32_t nwork, int32_t H, int32_t D, float scale, float* lse2,
                         int32_t causal, float drop_p, uint32_t drop_seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Self-attention backward, head_dim 64 (fairseq MultiheadAttention / nn.MultiheadAttention / CLIP ResidualAttentionBlock
 * with dh = 64): dq, dk, dv from q, k, v, out, dout and the forward's lse2.  Two kernels (dq ; dk + dv), no atomics.
 *   q, k, v, out, dout : row-major [B*R, ld] bf16, head h at columns h*64..h*64+63 of the given base pointers
 *   qT, kT, doT        : per-head transposed copies [B, H, 64, R] (sc_head_transpose_bf16)
 *   delta              : [B, H, R] fp32 workspace (rowsum(dout . out), written here)
 *   query rows t >= q_rows must carry dout = 0; keys t >= valid_len[b] (and t' > t when causal) get zero gradient
 * sc_head_transpose_bf16: xT[b, h, d, t] = x[b*R + t, h*64 + d]
 * ---------------------------------------------------------------------------------------------- */
int sc_attn_bwd_bf16(const sc_bf16* q, int64_t ldq, const sc_bf16* k, int64_t ldk, const sc_bf16* v, int64_t ldv,
                     const sc_bf16* out, int64_t ldo, const sc_bf16* dout, int64_t lddo, const sc_bf16* qT, const sc_bf16* kT,
                     const sc_bf16* doT, const float* lse2, float* delta, const int32_t* valid_len, sc_bf16* dq, int64_t lddq,
                     sc_bf16* dk, int64_t lddk, sc_bf16* dv, int64_t lddv, int32_t B, int32_t R, int32_t H,
                     int32_t q_rows /* query rows t >= q_rows carry dout = 0 (layout padding) */, float scale, int32_t causal,
                     float drop_p, uint32_t drop_seed /* the forward's probability dropout: same mask, regenerated */,
                     void* stream);
/* the same with the operand preparation inside: qT / kT / doT are scratch [B, H, 64, R] the call fills in ONE launch (three head
 * transposes + delta), followed by the dq and dk/dv kernels - 3 launches instead of 6 */
int sc_attn_bwd_fused_bf16(const sc_bf16* q, int64_t ldq, const sc_bf16* k, int64_t ldk, const sc_bf16* v, int64_t ldv,
                           const sc_bf16* out, int64_t ldo, const sc_bf16* dout, int64_t lddo, sc_bf16* qT, sc_bf16* kT,
                           sc_bf16* doT, const float* lse2, float* delta, const int32_t* valid_len, sc_bf16* dq,
                           int64_t lddq, sc_bf16* dk, int64_t lddk, sc_bf16* dv, int64_t lddv, int32_t B, int32_t R,
                           int32_t H, int32_t q_rows, float scale, int32_t causal, float drop_p, uint32_t drop_seed, void* stream);
int sc_head_transpose_bf16(const sc_bf16* x, int64_t ldx, sc_bf16* xT, int32_t B, int32_t R, int32_t H, void* stream);
/* Causal self-attention of 32-row sequences (the frozen CLIP text tower on keyword prompts of <= 32 tokens: clip_official.py:222-279 ->
 * CLIP Transformer / nn.MultiheadAttention, head_dim 64), one wave per (sequence, head), no workspaces:
 *   qkv  [nseq * 32, ld >= 3 heads 64] bf16 rows, Q | K | V side by side, head h at columns h*64.. of each third
 *   forward   out[nseq * 32, heads 64] = softmax(scale Q K^T | key <= query) V
 *   backward  dqkv (laid out like qkv) from qkv and dout alone: the probabilities are recomputed, delta = sum_k P dP in fp32
 * Rows behind a prompt are ordinary causal rows (finite values; zero gradient iff their dout rows are zero). */
int sc_attn32_fwd_bf16(const sc_bf16* qkv, int64_t ld, sc_bf16* out, int64_t ldo, int32_t nseq, int32_t heads, float scale, void* stream);
int sc_attn32_bwd_bf16(const sc_bf16* qkv, int64_t ld, const sc_bf16* dout, int64_t ldd, sc_bf16* dqkv, int64_t ldg, int32_t nseq,
                       int32_t heads, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row LayerNorm, bf16 in/out, fp32 statistics:  y = (x - mean) * rstd * gamma + beta  [-> GELU]
 *   replaces: fairseq LayerNorm calls (speech_encoder_plus.py:40,78 and inside each encoder layer),
 *             conv-extractor LayerNorm in "layer_norm" mode.
 *   D % 4 == 0, D <= 1024.  act: 0 none, 1 GELU(erf).
 * ---------------------------------------------------------------------------------------------- */
int sc_layernorm_bf16(const sc_bf16* x, int64_t ldx, const float* gamma, const float* beta, sc_bf16* y,
                      int64_t ldy, int64_t rows, int32_t D, float eps, int32_t act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Waveform front end.
 *   sc_wav_prep: speech_encoder_plus.py:506-518 (optional per-utterance layer_norm over the whole
 *     waveform, zero padding) -> padded fp32 [B, ldw_out], samples >= wav_len[b] are 0.
 *   sc_conv0_stats + sc_conv0_finalize: GroupNorm(512,512) statistics of conv layer 0 over the padded
 *     time axis T0 (fairseq Fp32GroupNorm, "default" extractor mode) computed analytically from the
 *     10x10 Gram matrix of the strided waveform (fp64 accumulation) -> scale/shift [B, C] fp32.
 *   sc_conv0_gn_gelu: conv0 (C_in=1, k=10, s=5, no bias) + GroupNorm affine + GELU, channels-last bf16
 *     out[b*R0 + t, c].
 * ---------------------------------------------------------------------------------------------- */
int sc_wav_prep(const float* wav, int64_t ldw_in, const int64_t* wav_len, float* out, int64_t ldw_out,
                int32_t B, int32_t L, int32_t normalize, void* stream);
#define SC_CONV0_NSTAT 66   /* 55 Gram entries + 10 sums + pad */
int sc_conv0_stats(const float* wav, int64_t ldw, int32_t B, int32_t T0, int32_t nchunk, double* partial,
                   void* stream);
int sc_conv0_finalize(const double* partial, int32_t nchunk, const float* w0 /*[C,10]*/, const float* gamma,
                      const float* beta, int32_t B, int32_t C, int32_t T0, float eps, float* scale,
                      float* shift, void* stream);
int sc_conv0_gn_gelu(const float* wav, int64_t ldw, const float* w0, const float* scale, const float* shift,
                     sc_bf16* out, int32_t B, int32_t R0, int32_t C, void* stream);
int sc_conv0_gn_gelu_f32(const float* wav, int64_t ldw, const float* w0, const float* scale, const float* shift,
                         float* out, int32_t B, int32_t R0, int32_t C, void* stream);      /* fp32 debug mode: unrounded stores */
/* backward of conv layer 0 + GroupNorm + GELU for the fully trainable encoder (speech_encoder_plus.py:556-562; the input is the
 * waveform: parameter gradients only).  dy [B*R0, 512] bf16 = gradient of the layer's output, scale / shift / stats = the forward's
 * (sc_conv0_finalize, sc_conv0_stats with nchunk chunks).  partial: scratch [B, nwc, 512, 12] fp32 (nwc % 4 == 0 wave chunks);
 * contrib [B, 512, 12] fp32 = per utterance (dW[c][0..9], dgamma[c], dbeta[c]) - sum over B with sc_colsum_f32. */
int sc_conv0_gn_bwd(const float* wav, int64_t ldw, const float* w0, const float* scale, const float* shift, const sc_bf16* dy,
                    const double* stats, int32_t nchunk, const float* gamma, const float* beta, int32_t B, int32_t T0, int32_t R0,
                    int32_t C, float eps, float* partial, int32_t nwc, float* contrib, void* stream);
/* the same for the "layer_norm" extractor (HuBERT-large: conv 0 (+bias) -> LayerNorm over the 512 channels -> GELU): per wave chunk of
 * rows 13 sums per channel - partial [B, nwc, 512, 16] fp32 with (dW[c][0..9], dbias[c], dgamma[c], dbeta[c], 3 unused); sum over
 * B * nwc with sc_colsum_f32 (nwc % 4 == 0). */
int sc_conv0_ln_bwd(const float* wav, int64_t ldw, const float* w0, const float* bias, const float* gamma, const float* beta, float eps,
                    const sc_bf16* dy, int32_t B, int32_t T0, int32_t R0, int32_t C, float* partial, int32_t nwc, void* stream);
/* Ragged rows (sc_segments; round 4).  The prepared waveform is ONE flat buffer, utterance b at sample samples_per_row * row0[b]
 * (320 = 5 * 64: conv layer 0 then reads x[5 r + j] for its flat output row r = 64 * row0[b] + t, like every later conv layer reads
 * rows 2 r ..), zero from wav_len[b] to the end of its region; out needs samples_per_row * rows + 16 floats.
 *   sc_wav_prep_seg    : as sc_wav_prep into that layout
 *   sc_conv0_stats_len : GroupNorm statistics straight from the caller's [B, ldw] batch, samples >= wav_len[b] read as 0 (the
 *                        statistics run over the PADDED batch length T0 - fairseq semantics - whatever the row layout)
 *   sc_conv0_gn_gelu_seg / sc_conv0_ln_gelu_seg : conv layer 0 on the flat waveform, out row 64 * row0[b] + t for t < 64 * pitch_b */
int sc_wav_prep_seg(const float* wav, int64_t ldw_in, const int64_t* wav_len, float* out, const sc_segments* seg,
                    int32_t samples_per_row, int32_t L, int32_t normalize, void* stream);
int sc_conv0_stats_len(const float* wav, int64_t ldw, const int64_t* wav_len, int32_t B, int32_t T0, int32_t nchunk, double* partial,
                       void* stream);
int sc_conv0_gn_gelu_seg(const float* wav_flat, const sc_segments* seg, int32_t samples_per_row, const float* w0, const float* scale,
                         const float* shift, sc_bf16* out, int32_t C, void* stream);
/* The in-forward training crop (round 5): avssl/module/speech_encoder_plus.py:548-552 calls random_crop_max_length
 * (avssl/data/audio_transforms.py:5-23) per utterance on the un-padded list and re-pads (:506-518).  Here the caller's [B, ldw] batch
 * stays where it is: utterance b is wav[b, wav_off[b] : wav_off[b] + wav_len[b]] (wav_len = the CROPPED lengths, L = their maximum,
 * wav_off[b] + wav_len[b] <= ldw is the caller's contract), read in place by the two kernels that touch the caller's batch.
 * wav_off == NULL: the un-cropped entry points above. */
int sc_wav_prep_crop(const float* wav, int64_t ldw_in, const int64_t* wav_len, const int64_t* wav_off, float* out, int64_t ldw_out,
                     int32_t B, int32_t L, int32_t normalize, void* stream);
int sc_wav_prep_seg_crop(const float* wav, int64_t ldw_in, const int64_t* wav_len, const int64_t* wav_off, float* out,
                         const sc_segments* seg, int32_t samples_per_row, int32_t L, int32_t normalize, void* stream);
int sc_conv0_stats_len_crop(const float* wav, int64_t ldw, const int64_t* wav_len, const int64_t* wav_off, int32_t B, int32_t T0,
                            int32_t nchunk, double* partial, void* stream);
int sc_conv0_ln_gelu_seg(const float* wav_flat, const sc_segments* seg, int32_t samples_per_row, const float* w0, const float* bias,
                         const float* gamma, const float* beta, float eps, sc_bf16* out, int32_t C, void* stream);
/* "layer_norm" extractor mode (HuBERT-large): conv0 (+bias) -> LayerNorm over the 512 channels -> GELU */
int sc_conv0_ln_gelu(const float* wav, int64_t ldw, const float* w0, const float* bias, const float* gamma,
                     const float* beta, float eps, sc_bf16* out, int32_t B, int32_t R0, int32_t C, void* stream);
int sc_conv0_ln_gelu_f32(const float* wav, int64_t ldw, const float* w0, const float* bias, const float* gamma,
                         const float* beta, float eps, float* out, int32_t B, int32_t R0, int32_t C, void* stream);   /* fp32 debug mode */

/* ------------------------------------------------------------------------------------------------
 * pos_conv input: zero padded frames (speech_encoder_plus.py:32-33 index_put(x, padding_mask, 0)) and
 * regroup channels-last [B*R, D] into the group-major, halo-padded slab layout the grouped conv GEMM reads:
 *   xz  [B*R, D]              : masked copy (residual operand of the pos_conv epilogue)
 *   xg  [G, B, R + 2*halo, D/G] : rows [halo, halo+R) hold the masked frames, everything else 0
 * ---------------------------------------------------------------------------------------------- */
int sc_posconv_prep(const sc_bf16* x, const int32_t* valid_len, sc_bf16* xz, sc_bf16* xg, int32_t B, int32_t R,
                    int32_t D, int32_t G, int32_t halo, void* stream);

/* ragged rows: xg [G][rows + 2 * halo * B][D / G], utterance b's slab at row  row0[b] + 2 * halo * b  (its frames at + halo) */
int sc_posconv_prep_seg(const sc_bf16* x, const int32_t* valid_len, sc_bf16* xz, sc_bf16* xg, const sc_segments* seg, int32_t D,
                        int32_t G, int32_t halo, void* stream);

/* HuBERT positional convolution on the slab layout above + bias + GELU + residual (speech_encoder_plus.py:32-37: grouped Conv1d,
 * kernel Kp = 128, padding 64, SamePad drops the last frame):
 *   out[b*R + t, g*Dg + n] = gelu(bias[g*Dg + n] + sum_j sum_ci w[g][n][j*Dg + ci] * xg[g][b][t + j][ci]) + residual[b*R + t, g*Dg + n]
 * w [G, Dg, Kp*Dg] tap-major per group, Dg = D / G in {48, 64}, Rp = rows of an xg slab (>= R + Kp - 1).  The input slab of a
 * (group, utterance, 512-frame block) stays in LDS for all taps; only the weights stream.  Same arithmetic as the sc_gemm_bf16
 * formulation (lda = Dg, K = Kp*Dg, act = 1, residual): bit-identical results. */
int sc_posconv_bf16(const sc_bf16* xg, const sc_bf16* w, const float* bias, const sc_bf16* residual, sc_bf16* out, int32_t B,
                    int32_t R, int32_t D, int32_t G, int32_t Kp, int32_t Rp, void* stream);

/* ragged rows: the slab layout of sc_posconv_prep_seg (halo = Kp / 2), out / residual rows row0[b] + t */
int sc_posconv_seg_bf16(const sc_bf16* xg, const sc_bf16* w, const float* bias, const sc_bf16* residual, sc_bf16* out,
                        const sc_segments* seg, int32_t D, int32_t G, int32_t Kp, void* stream);

/* Weight gradient of that convolution (fully trainable HuBERT; speech_encoder_plus.py:29-40 under trainable: true):
 *   part[z][g][co][tap*Dg + ci] = sum over the z-th slice of slab rows m of  du[g][m][co] * xg[g][m + tap][ci]
 * du, xg [G, rows, Dg] bf16 (rows = B * Rp, the slab row pitch; du must be ZERO on rows whose window straddles two utterances, i.e.
 * laid out like the slab with the frames at row offset 0: pass the du slab advanced by `halo` rows), Dg in {48, 64}, Kp % 16 == 0,
 * rows % (128 * Z) == 0; xg rows past the end of the buffer are never dereferenced (clamped; they meet du = 0).
 * part [Z, G, Dg, Kp*Dg] fp32: reduce over z with sc_colsum_f32. */
int sc_posconv_wgrad_bf16(const sc_bf16* du, const sc_bf16* xg, float* part, int32_t G, int64_t rows, int32_t Dg, int32_t Kp,
                          int32_t Z, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weighted sum over hidden states (avssl/module/weighted_sum.py:26-45).
 *   h    [NL, B*R, D] bf16 ; w [NL] fp32 = softmax(weights) (host computes the 13-element softmax)
 *   out  [B, R, D] bf16, written at row offset `row_off` inside each utterance (row_off = 1 leaves row 0
 *        for the CLS token of kw_branches.py:266-267), rows t in [0, R - row_off)
 *   bwd: dw[n] = sum_{b,t,d} g[b, t + row_off, d] * (h[n, b, t, d] - h[NL-1, b, t, d])   (g fp32 [B, R, D]).  The gradient of the
 *        13 (25) logits is the softmax projection w_n (dw_n - sum_m w_m dw_m), invariant under a common shift of dw: the last
 *        layer is subtracted element-wise BEFORE the accumulation, so the large common part <g, h> never enters the fp32 sums
 *        (layers of a residual stream differ by 1e-1 .. 1e-2 of their norm; summing first would cost those digits)
 *   normalize != 0: every h[n, row, :] passes through a non-affine LayerNorm(D, eps 1e-5) first
 *        (normalize_features=True, weighted_sum.py:41-42; used by the HuBERT-large recipes), D <= 1024
 * ---------------------------------------------------------------------------------------------- */
int sc_wsum_fwd(const sc_bf16* h, const float* w, int32_t NL, sc_bf16* out, int32_t B, int32_t R, int32_t D,
                int32_t row_off, int32_t normalize, void* stream);
/* bwd `flags`: bit 0 = normalize, bit 1 = g is bf16 [B, R, D] instead of fp32 (the attention block of the cascaded+/hybrid+ branches
 * returns its input gradient as the bf16 rows its GEMM wrote; g 16-byte aligned either way) */
int sc_wsum_bwd(const sc_bf16* h, const void* g, int32_t NL, float* dw_partial /*[nblk, NL]*/, int32_t nblk,
                int32_t B, int32_t R, int32_t D, int32_t row_off, int32_t flags, void* stream);
/* Ragged hidden states -> uniform-pitch output (round 4): h [NL, seg->rows, D] in the segment layout; out / g [B, Rout, D] with
 * utterance b's frame t at out[b, t + row_off] for t + row_off < min(pitch_b + row_off, Rout), every other row of out is ZEROED (the
 * consumers - the CLS pooling kernels, the cascaded+/hybrid+ branches - keep a uniform [B, Rout, D] view and mask by length). */
int sc_wsum_fwd_seg(const sc_bf16* h, const float* w, int32_t NL, sc_bf16* out, const sc_segments* seg, int32_t Rout, int32_t D,
                    int32_t row_off, int32_t normalize, void* stream);
int sc_wsum_bwd_seg(const sc_bf16* h, const void* g, int32_t NL, float* dw_partial /*[nblk, NL]*/, int32_t nblk,
                    const sc_segments* seg, int32_t Rout, int32_t D, int32_t row_off, int32_t flags, void* stream);
/* The same sums over RAW hidden states (LayerNorm folded into the encoder GEMMs, see sc_gemm_args): layers n >= first_lazy of h hold
 * the rows in FRONT of the layer's final LayerNorm; stats [NL][B*R][8][2] fp32 their row statistics (ns valid strips), gamma / beta
 * [NL][D] the LayerNorm affines (rows < first_lazy unused): the summed state is (raw - mean) rstd gamma_n + beta_n in fp32. */
int sc_wsum_lazy_fwd(const sc_bf16* h, const float* w, int32_t NL, sc_bf16* out, int32_t B, int32_t R, int32_t D, int32_t row_off,
                     const float* stats, const float* gamma, const float* beta, int32_t first_lazy, int32_t ns, float eps, void* stream);
int sc_wsum_lazy_bwd(const sc_bf16* h, const float* g, int32_t NL, float* dw_partial /*[nblk, NL]*/, int32_t nblk, int32_t B, int32_t R,
                     int32_t D, int32_t row_off, const float* stats, const float* gamma, const float* beta, int32_t first_lazy,
                     int32_t ns, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CLS attention pooling (the query row 0 of the parallel branch's TransformerEncoder layer;
 * avssl/model/kw_branches.py:266-280 -> nn.MultiheadAttention inside TransformerModels.py:48-97).
 * Only the CLS query is consumed by the branch, so K/V are never materialised:
 *   scores[b,h,s] = a[h] . X[b,s]            a[h] = Wk_h^T q_h * dh^-.5  (host side, tiny)
 *   p = softmax_s(scores | s < len[b]) ;  m[b,h,:] = sum_s p[b,h,s] X[b,s,:]
 *   X [B, R, D] bf16 ; a [H, D] fp32 ; len [B] int32 (valid keys incl. CLS) ; H <= 16
 * backward (dm [B,H,D] fp32 given):
 *   dp = dm . X ; ds = p (dp - sum p dp) ; dX = sum_h p dm + ds a ; da_partial[b,h,:] = sum_s ds X
 * train-mode dropout of the attention weights (nn.MultiheadAttention dropout=): mult [B,H,R] fp32 (0 or 1/(1-p_drop),
 *   NULL = none): m = sum_s p*mult X ; psum [B,H] (optional output) = sum_s p*mult, the weight of the value bias.  The backward
 *   takes dp already multiplied by mult, or - cbias [B,H] given - the RAW dp = dm . X and forms (dp + cbias[b,h]) * mult itself
 *   (cbias = the value-bias path, sc_rt_value_bias_bwd); it uses p*mult in the dX term.
 * ---------------------------------------------------------------------------------------------- */
int sc_cls_scores(const sc_bf16* X, const float* vec, int64_t vec_bstride, float* scores, int32_t B, int32_t R,
                  int32_t D, int32_t H, void* stream);
int sc_cls_pool_fwd(const sc_bf16* X, const float* scores, const int32_t* len, float* p, float* m, int32_t B,
                    int32_t R, int32_t D, int32_t H, const float* mult, float* psum, void* stream);
int sc_cls_pool_bwd(const sc_bf16* X, const float* p, const float* dp, const float* dm, const float* a,
                    const int32_t* len, float* dX, float* da_partial, int32_t B, int32_t R, int32_t D, int32_t H,
                    const float* mult, const float* cbias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row softmax of the GEMM-based attention core (attention block of the cascaded+/hybrid+ branches,
 * avssl/module/kw_modules/TransformerModels.py:101-126 -> nn.MultiheadAttention with head_dim 768 / 128):
 *   forward   P = softmax(scale * scores | key mask)   scores fp32 [rows, n] contiguous (a batched sc_gemm_bf16 output), P bf16;
 *             Pd = keep . P / (1 - p) (train mode, drop_p > 0: the hash mask of sc_gemm_args over element row * n + col)
 *             key_mask uint8 [rows / rows_per_batch, n], non-zero = padded key (probability exactly 0)
 *   backward  dS = scale * P (dP' - sum_k P dP'),  dP' = keep . dP / (1 - p)         dP fp32, dS bf16
 *   n % 4 == 0, n <= 1024.
 * ---------------------------------------------------------------------------------------------- */
int sc_softmax_fwd(const float* scores, const uint8_t* key_mask, sc_bf16* P, sc_bf16* Pd, int64_t rows, int32_t n,
                   int32_t rows_per_batch, float scale, float drop_p, uint32_t drop_seed, void* stream);
int sc_softmax_bwd(const float* dP, const sc_bf16* P, sc_bf16* dS, int64_t rows, int32_t n, float scale, float drop_p,
                   uint32_t drop_seed, void* stream);
/* fp32 DEBUG mode (SURVEY 8d "fp32 kernel mode (for debugging)"; host side speechclip_plus_amd/debug_fp32.py): the same kernels with
 * unrounded fp32 outputs - softmax probabilities here, the conv layer 0 activations below (sc_conv0_gn_gelu_f32, sc_conv0_ln_gelu_f32);
 * every product of that mode runs on sc_sgemm_mfma_f32 / sc_sgemm_f32_ex, the norms on sc_rowln_f32_fwd, GELU on sc_gelu_f32. */
int sc_softmax_fwd_f32(const float* scores, const uint8_t* key_mask, float* P, int64_t rows, int32_t n, int32_t rows_per_batch,
                       float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Continuous integrate-and-fire accumulation (avssl/module/cif.py:157-240; the downsampler of the cascaded+/hybrid+ branches).
 *   x [B,S,C] fp32, alpha [B,S] fp32 (weights, 0 on padded frames), csum = cumsum(alpha) [B,S] (computed by the caller so
 *   that the slot boundaries floor(csum / thr) are the host framework's), out [B,T+1,C] fp32 (slot T collects the tail;
 *   every slot is written).  Frame s adds lw_s x_s to slot left_s, thr x_s to the slots in between, rw_s x_s to slot right_s.
 *   backward: g [B,T+1,C] -> dx [B,S,C], pa / pb [ceil(C/256), B, S]: per-channel-block partials of
 *   d alpha_s (direct) = x_s . g[left_s]  and  d csum_s = fire_s ? x_s . (g[right_s] - g[left_s]) : 0.        C % 4 == 0.
 * ---------------------------------------------------------------------------------------------- */
int sc_cif_fwd(const float* x, const float* alpha, const float* csum, float* out, int32_t B, int32_t S, int32_t C, int32_t T,
               float thr, void* stream);
int sc_cif_bwd(const float* x, const float* alpha, const float* csum, const float* g, float* dx, float* pa, float* pb, int32_t B,
               int32_t S, int32_t C, int32_t T, float thr, void* stream);
/* The same kernels on the ROWS of the attention block in front of CIF (round 4): x = frame 0 of utterance 0, fp32 or bf16 (x_bf16), with
 * `xbs` elements between utterances (the block's row pitch x C) - no fp32 copy of the activations.  The gradient leaves in the dtype
 * the frames came in, frame s of utterance b at dx + b dxbs + s C; the zlo rows in front of an utterance's frames and the zhi rows
 * behind them are zero-filled (CLS slot / padding rows of the block's buffer). */
int sc_cif_fwd_rows(const void* x, int32_t x_bf16, int64_t xbs, const float* alpha, const float* csum, float* out, int32_t B, int32_t S,
                    int32_t C, int32_t T, float thr, void* stream);
int sc_cif_bwd_rows(const void* x, int32_t x_bf16, int64_t xbs, const float* alpha, const float* csum, const float* g, void* dx,
                    int32_t dx_bf16, int64_t dxbs, int32_t zlo, int32_t zhi, float* pa, float* pb, int32_t B, int32_t S, int32_t C, int32_t T,
                    float thr, void* stream);
/* buf [lead + B*P + trail, D] bf16: the rows that are not frames <- 0 (the `lead` rows in front, per utterance rows [0, head) and
 * [stop, P), the `trail` rows behind): the zero padding a k-tap conv GEMM reads in place between utterances.  B = 0: lead + trail only. */
int sc_rows_zero_pad_bf16(sc_bf16* buf, int32_t lead, int32_t B, int32_t P, int32_t head, int32_t stop, int32_t trail, int32_t D, void* stream);

/* CIF bookkeeping on the device (avssl/module/cif.py:106-175, 244-297), one workgroup per utterance, no host round trip.
 *   sc_cif_prepare: a = clip(alpha_raw, 0, 1), padded frames (pad != 0) zeroed -> a_clip [B,S] ("orig_alpha") ; quantity = sum a ;
 *     if target && apply_scaling: a *= (thr * target + eps) / quantity (ratio [B] kept for the backward) ; alpha [B,S] ;
 *     csum = inclusive scan [B,S] ; feat_len = clip(floor(sum alpha / thr), 1, max_feat) ; fired [B,S] = slot index advances at the
 *     frame (indices clipped at T) ; flags (8 int32, zeroed once by the caller): [0] += (quantity > 0) ; [1] += (feat_len !=
 *     clip(target, 1, max_feat)) when scaling (the caller sized the output from target) ; [3] += 1 when NO utterance of this call had
 *     a positive quantity (the reference asserts that on every call, cif.py:121) ; [6] += utterances whose zero quantity could not
 *     be rescaled ; [4], [5] scratch of the per-call check (left at 0) ; [2], [7] the caller's.  fp64 sums.
 *   sc_cif_prepare_bwd: pa / pb of sc_cif_bwd + d quantity (gq, may be NULL) -> d alpha_raw [B,S] (suffix sum of d csum, scaling).
 *   sc_cif_tail (inference): tail weight of slot feat_len >= tail_thr -> that row *= thr / weight, feat_len += 1 (clip max_feat),
 *     rows >= feat_len zeroed in out [B,T+1,C]; factor / extend [B] returned for the caller's backward / diagnostics.    S <= 2048. */
int sc_cif_prepare(const float* alpha_raw, int64_t lda, const uint8_t* pad, int64_t ldp, const int64_t* target, int32_t apply_scaling,
                   int32_t B, int32_t S, float thr, float eps, int32_t max_feat, int32_t T, float* a_clip, float* alpha, float* csum,
                   float* quantity, float* ratio, int64_t* feat_len, uint8_t* fired, int32_t* flags, void* stream);
int sc_cif_prepare_bwd(const float* pa, const float* pb, int32_t nblk, int32_t B, int32_t S, const float* a_clip, const uint8_t* pad,
                       int64_t ldp, const float* ratio, const float* quantity, const float* gq, int32_t scaled, float* da, void* stream);
int sc_cif_tail(const float* alpha, const float* csum, int32_t B, int32_t S, int32_t C, int32_t T, float thr, float tail_thr,
                int32_t max_feat, int64_t* feat_len, float* out, float* factor, uint8_t* extend, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Keyword -> CLIP sub-word vector quantiser (cascaded+/hybrid+ tails; csrc/vq.hip).
 *   replaces: GeneralBranch.get_keyword_cosine_score / vq_audio_features (avssl/model/kw_branches.py:158-197) and
 *             SimpleVectorQuantizer.forward (avssl/module/speechclip_c_modules/my_vector_quantizer.py:64-165).
 *   sc_vq_prep_f32     kw [Nk, Et] (row stride ldk) -> kwn_T [Et][ldt] = (kw / max(|kw|, eps))^T (columns >= Nk zero; ldt % 64 == 0),
 *                      rnorm [Nk] = 1 / max(|kw|, eps)                                       (F.normalize, kw_branches.py:170-173)
 *   sc_sgemm_mfma_f32  C [M, N] = A . B^T (+ bias[n]) in exact fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32): the cosine scores
 *                      decide an argmax over the vocabulary and the CIF weights a floor(), so no reduced-precision operand.
 *                      A: [M][K] row-major (a_kmajor 0) or [K][M] (a_kmajor 1); B: [N][K] (the nn.Linear layout) or [K][N].
 *                      any M, N, K (16-byte loads when base and leading dimension allow, element loads otherwise)
 *   sc_sgemm_mfma_f32_split  the same product with the contraction cut into S slices of whole 16-wide K-tiles (partials [S, M, N] fp32:
 *                      the caller's workspace), added in slice order with the bias by a second launch - for products with few output
 *                      tiles and a long K (the keyword projection's 64- and 1600-row products and their weight gradients), which the
 *                      one-slice kernel walks with a handful of workgroups.  sc_sgemm_mfma_slices(M, N, K) = the S to use (1: do not split)
 *   sc_vq_rowstats     x [Nk, ldx] fp32, V columns: columns listed in mask_cols_host (<= 4, host array: the special tokens 0, 2, 3) are
 *                      set to -inf IN PLACE (the reference's x[:, i] += -inf), idx = first argmax, lse_t = LSE(x / temp),
 *                      lse_1 = LSE(x), ent = - sum p log(p + 1e-9) with p = softmax(x)       (my_vector_quantizer.py:80-116)
 *   sc_vq_perplexity   out2[0] = code_perplexity (histogram of idx), out2[1] = prob_perplexity (column means of softmax(x));
 *                      workspaces: partial [nchunk, V] fp32, hist [V + 64] int32 (the last 64 words: per-workgroup entropy partials)
 *   sc_vq_gather_f32   out[n] = table[idx[n]]   (= hard one-hot @ token_embedding)
 *   sc_vq_onehot_f32   dense hard one-hot [Nk, ldo] (module-level subword_prob)
 *   sc_vq_soft_bwd     dx = softmax(x / temp) (t - <softmax, t>) / temp, t = d subword_prob [Nk, ldt]; out bf16 (feeds the bf16
 *                      GEMM dx . normalised table) or fp32; columns V .. Vpad - 1 zero-filled         (straight-through estimator)
 *   sc_vq_norm_bwd_f32 gradient through x / max(|x|, eps)
 * ---------------------------------------------------------------------------------------------- */
int sc_vq_prep_f32(const float* kw, int64_t ldk, int32_t Nk, int32_t Et, float eps, float* kwn_T, int64_t ldt, float* rnorm, void* stream);
/* round 6: x[r, :] * row_scale[r] (row_scale NULL: 1) as three bf16 addends x1 + x2 + x3 (24 significant bits), laid out as the six
 * K-blocks of a product to fp32 accuracy on the bf16 matrix pipe - side 0: [x1 | x1 | x2 | x1 | x3 | x2], side 1: [x1 | x2 | x1 | x3 | x1 | x2];
 * sc_gemm_bf16 (out_f32) over K = 6 Ep on a side-0 and a side-1 operand then sums the products (1,1) (1,2) (2,1) (1,3) (3,1) (2,2), each exact
 * in fp32.  out [Rp][6 Ep] bf16, zero outside [R, E].  Used for the cosine scores the keyword argmax is taken over (was sc_sgemm_mfma_f32). */
int sc_split3_bf16(const float* x, int64_t ldx, const float* row_scale, int32_t R, int32_t E, sc_bf16* out, int32_t Rp, int32_t Ep,
                   int32_t side, void* stream);
int sc_sgemm_mfma_f32(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc,
                      int32_t M, int32_t N, int32_t K, const float* bias, void* stream);
int sc_sgemm_mfma_f32_split(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, int32_t b_kmajor, float* C,
                            int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, float* partials, int32_t S, void* stream);
int32_t sc_sgemm_mfma_slices(int32_t M, int32_t N, int32_t K);
int sc_vq_rowstats(float* x, int64_t ldx, int32_t Nk, int32_t V, float temp, const int32_t* mask_cols_host, int32_t n_mask, int64_t* idx,
                   float* lse_t, float* lse_1, float* ent, void* stream);
int sc_vq_perplexity(const float* x, int64_t ldx, int32_t Nk, int32_t V, const int64_t* idx, const float* lse_1, float* partial,
                     int32_t nchunk, int32_t* hist, float* out2, void* stream);
int sc_vq_gather_f32(const float* table, int64_t ldt, const int64_t* idx, float* out, int64_t ldo, int32_t Nk, int32_t Et, void* stream);
int sc_vq_onehot_f32(const int64_t* idx, float* out, int64_t ldo, int32_t Nk, int32_t V, void* stream);
int sc_vq_soft_bwd(const float* x, int64_t ldx, const float* lse_t, const float* t, int64_t ldt, int32_t Nk, int32_t V, int32_t Vpad,
                   float temp, void* dx, int64_t ldd, int32_t out_bf16, void* stream);
int sc_vq_norm_bwd_f32(const float* kw, int64_t ldk, const float* rnorm, const float* dy, int64_t ldy, float eps, float* dx, int64_t ldd,
                       int32_t Nk, int32_t Et, void* stream);

/* Keyword BatchNorm: nn.BatchNorm1d over the keyword positions (avssl/module/speechclip_c_modules/kw_bn.py:167-228).
 *   x [N, E] fp32 (N = batch x keyword slots), per-channel statistics.  training: batch statistics (biased variance for the
 *   normalisation, unbiased for the running estimate; run_mean / run_var updated in place with `momentum`), save_mean / save_rstd
 *   kept for the backward; inference: the running estimates.  backward (training statistics): dx, dgamma [E], dbeta [E]. */
int sc_bn_rows_fwd(const float* x, int64_t ldx, int32_t N, int32_t E, const float* gamma, const float* beta, float* run_mean,
                   float* run_var, int32_t training, float momentum, float eps, float* y, int64_t ldy, float* save_mean,
                   float* save_rstd, void* stream);
int sc_bn_rows_bwd(const float* x, int64_t ldx, const float* dy, int64_t ldg, int32_t N, int32_t E, const float* gamma,
                   const float* save_mean, const float* save_rstd, float* dx, int64_t ldd, float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 strided GEMM  C[i,j] = alpha * sum_k A[i*sai + k*sak] * Bm[j*sbj + k*sbk]  (+ bias[j])
 *   small fp32 products of the loss and of the CLS-row tail (logits = A.B^T / tau, dA = G.B, ...)
 * ---------------------------------------------------------------------------------------------- */
int sc_sgemm_f32(const float* A, int64_t sai, int64_t sak, const float* Bm, int64_t sbj, int64_t sbk, float* C,
                 int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, const float* bias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row kernels of the backward pass over bf16 activations (CLIP text tower input gradient, HuBERT layer backward):
 *   sc_layernorm_bwd_bf16 : dx = LayerNorm'(x; gamma)(dy) (+ dres), statistics recomputed from the saved LN input x;
 *                           optional per-workgroup partial sums [n_partial, D] of dgamma = sum dy xhat and dbeta = sum dy
 *                           (n_partial = workgroups launched; reduce with sc_colsum_f32)
 *   sc_act_bf16           : df == NULL: out = act(u) ; else out = df * act'(u) ; act 1 = erf-GELU, 2 = QuickGELU
 * ---------------------------------------------------------------------------------------------- */
int sc_layernorm_bwd_bf16(const sc_bf16* x, int64_t ldx, const sc_bf16* dy, int64_t lddy, const float* gamma, const sc_bf16* dres,
                          int64_t lddres, sc_bf16* dx, int64_t lddx, int64_t rows, int32_t D, float eps, float* dgamma_partial,
                          float* dbeta_partial, int32_t n_partial, void* stream);
/* the same pass with what the NEXT products of a differentiated layer's backward read (round 4): dx_drop (may be NULL) = F.dropout of the
 * stored dx with the stateless mask of element row*D + col (drop_p = 0: a plain copy is not written - pass NULL), dsum_partial
 * [n_partial, D] (may be NULL; needs the parameter partial buffers) = partial column sums of dx_drop's values (of dx when drop_p = 0):
 * the bias gradient of the residual branch behind this LayerNorm - no dropout launch, no column-sum pass */
int sc_layernorm_bwd_drop_bf16(const sc_bf16* x, int64_t ldx, const sc_bf16* dy, int64_t lddy, const float* gamma, const sc_bf16* dres,
                               int64_t lddres, sc_bf16* dx, int64_t lddx, int64_t rows, int32_t D, float eps, float* dgamma_partial,
                               float* dbeta_partial, int32_t n_partial, sc_bf16* dx_drop, int64_t lddd, float drop_p, uint32_t drop_seed,
                               float* dsum_partial, void* stream);
/* out[b*R + t] = (prev ? prev[..] : 0) + bf16(w[0] * dX[b, t + row_off]) for t < T (rows t >= T: prev or 0): the gradient reaching hidden
 * state n of a differentiated encoder = what came down from layer n + 1 + its share of the weighted sum's gradient dX (fp32 [B, R, D]) */
int sc_wsum_share_bf16(const float* dX, const float* w, const sc_bf16* prev, sc_bf16* out, int32_t B, int32_t R, int32_t T, int32_t D,
                       int32_t row_off, void* stream);
int sc_act_bf16(const sc_bf16* u, const sc_bf16* df, sc_bf16* out, int64_t n, int32_t act, void* stream);
/*   sc_transpose_bf16 : y[c, r] = x[r, c]  - operands of the weight-gradient GEMMs (dW = dY^T X: the row index becomes the
 *                       contraction dimension of sc_gemm_bf16, split along K over the batch dimension, partials in fp32)
 *   sc_colsum_bf16    : partial[blk, c] = sum of the block's rows of x[:, c] in fp32 (bias gradients; reduce with sc_colsum_f32) */
/*   sc_transpose_batched_bf16 : nbatch equally shaped [rows, cols] matrices, matrix z starting sx / sy elements after matrix z - 1 */
int sc_transpose_batched_bf16(const sc_bf16* x, int64_t ldx, int64_t sx, sc_bf16* y, int64_t ldy, int64_t sy, int32_t rows, int32_t cols,
                              int32_t nbatch, void* stream);
/*   sc_cast_transpose_f32_bf16 : the bf16 working copies of an fp32 [rows, cols] weight in one pass - y = bf16(x) (same layout) and /
 *   or yT = bf16(x)^T [cols, rows]; either may be NULL.  rows, cols, leading dims multiples of 4. */
int sc_cast_transpose_f32_bf16(const float* x, int64_t ldx, sc_bf16* y, int64_t ldy, sc_bf16* yT, int64_t ldyT, int32_t rows, int32_t cols,
                               void* stream);
/*   sc_dropout_bf16   : out = dropout(x) (F.dropout semantics, the stateless mask of sc_gemm_args over element row*D + col):
 *                       fairseq's encoder dropout after pos_conv + LayerNorm (speech_encoder_plus.py:41), train mode only */
int sc_dropout_bf16(const sc_bf16* x, int64_t ldx, sc_bf16* out, int64_t ldo, int64_t rows, int32_t D, float p, uint32_t seed,
                    void* stream);
/* the multiplier F.dropout applies, as fp32: out[i] = keep(i, seed) ? 1 / (1 - p) : 0, keep = the stateless hash mask above
 * (element index i); n % 8 == 0.  The CLS head's four masks (nn.TransformerEncoderLayer dropout / dropout1 / dropout2 and the
 * attention-weight dropout, avssl/module/kw_modules/TransformerModels.py:61-81) are products with these. */
int sc_dropout_mult_f32(float* out, int64_t n, float p, uint32_t seed, void* stream);
int sc_transpose_bf16(const sc_bf16* x, int64_t ldx, sc_bf16* y, int64_t ldy, int32_t rows, int32_t cols,
                      float* colsum_partial /* NULL, or [ceil(rows / 64), cols] fp32: per-64-row-block column sums of x */, void* stream);
/* input gradient of a channels-last Conv1d(k = 3, stride 2) run as a strided-row GEMM (fully trainable HuBERT, conv layers 1-4):
 * dcols [M, 3C] = dy . W -> dx [2M, C]:  dx[2m] = dcols[m][0:C] + dcols[m-1][2C:3C],  dx[2m+1] = dcols[m][C:2C]   (one pass) */
int sc_conv_overlap_add_bf16(const sc_bf16* dcols, sc_bf16* dx, int64_t M, int32_t C, void* stream);
/* the same followed by the backward of the activation that produced the conv's input (u [2M, C] = its kept pre-activation):
 * dx <- bf16(dx) * act'(u), bit-identical to sc_conv_overlap_add_bf16 + sc_act_bf16(u, dx) in one pass */
int sc_conv_overlap_add_act_bf16(const sc_bf16* dcols, const sc_bf16* u, sc_bf16* dx, int64_t M, int32_t C, int32_t act, void* stream);
int sc_colsum_bf16(const sc_bf16* x, int64_t ldx, int64_t rows, int32_t cols, float* partial, int32_t nblk, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One-row-per-utterance tail of the parallel head in fp32 on the master weights:
 *   nn.TransformerEncoderLayer (post-LN, GELU) row 0 + final LayerNorm + Linear, as instantiated by
 *   avssl/module/kw_modules/TransformerModels.py:48-97 and consumed at avssl/model/kw_branches.py:266-280.
 *   sc_sgemm_f32_ex : C[z][i,j] = alpha sum_k A[z][i*sai + k*sak] B[z][j*sbj + k*sbk] (+ bias[z][j]) + beta C[z][i,j]
 *                     (beta = 1 accumulates weight gradients straight into the flat gradient buffer; with a workspace,
 *                     few-tile products are split along K and the slices reduced in order)
 *   sc_rowln_f32_fwd: y = LayerNorm(x + res) gamma + beta ; res_stride 0 broadcasts one residual row (the CLS token)
 *   sc_rowln_f32_bwd: dx ; dgamma += sum_rows dy xhat ; dbeta += sum_rows dy   (fixed row order, no atomics)
 *   sc_gelu_f32     : df == NULL: out = gelu(u) (erf form) ; else out = df * gelu'(u)
 *   sc_colsum_f32   : out[j] = beta out[j] + alpha sum_i x[i*ld + j]
 *   sc_headmask_f32 : dir 0: Qm[h,j] = q[j] if j / dh == h else 0 ; dir 1: q[j] = Qm[j / dh, j]
 * ---------------------------------------------------------------------------------------------- */
int sc_sgemm_f32_ex(const float* A, int64_t sai, int64_t sak, int64_t saz, const float* Bm, int64_t sbj, int64_t sbk,
                    int64_t sbz, float* C, int64_t ldc, int64_t scz, int32_t M, int32_t N, int32_t K, int32_t nbatch,
                    float alpha, float beta, const float* bias, int64_t sbiasz, float* workspace, int64_t workspace_floats,
                    void* stream);
int sc_rowln_f32_fwd(const float* x, const float* res, int64_t res_stride, const float* gamma, const float* beta, float* y,
                     float* xhat, float* rstd, int32_t rows, int32_t D, float eps, void* stream);
int sc_rowln_f32_bwd(const float* dy, const float* xhat, const float* gamma, const float* rstd, float* dx, float* dgamma_acc,
                     float* dbeta_acc, int32_t rows, int32_t D, void* stream);
int sc_gelu_f32(const float* u, const float* df, float* out, int64_t n, void* stream);
int sc_colsum_f32(const float* x, int64_t ld, int32_t rows, int32_t cols, float* out, float alpha, float beta, void* stream);
int sc_headmask_f32(float* q, float* Qm, int32_t H, int32_t D, int32_t dh, int32_t dir, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Masked contrastive loss: MaskedContrastiveLoss.forward, avssl/module/losses.py:185-245 (ctor options :130-168).
 *   logits[i,j] = inv_temp <A_i, B_j> - margin [i == j]        A, B [Bg, E] fp32 (unit-norm rows), inv_temp: DEVICE scalar
 *   neg[i,j] = (ids ? ids[i] != ids[j] : i != j) || (!dcl && i == j)           (no MAX_EYE = 256 cap, losses.py:126)
 *   loss = 1 / (Bg n_dir) sum_i [ a2b (-l_ii + log sum_j neg_ij e^{l_ij}) + b2a (-l_ii + log sum_j neg_ji e^{l_ji}) ]
 *   sc_infonce_fwd : ONE launch - 64 x 64 logit tiles on the matrix pipe in exact fp32, LDS-tiled operands, per-tile masked
 *                    (max, sum exp) partials, merged by the last-arriving workgroup (ticket) into lse_row / lse_col [Bg] and
 *                    loss[0]; the scaled logits [Bg, Bg] are kept for the backward.  workspace: sc_infonce_workspace_floats(Bg)
 *                    floats, ZEROED once by the caller before the first use (holds the ticket word; each launch leaves it 0),
 *                    one workspace per stream.  E % 4 == 0.
 *   sc_infonce_grad: G = inv_temp gscale dloss/dlogits [Bg, Bg] (dA = G . B, dB = G^T . A: sc_sgemm_mfma_f32), and
 *                    dlogit_dot[i] = sum_j dloss/dlogit_ij <A_i, B_j>  (d loss / d inv_temp = sum_i dlogit_dot[i])
 * ---------------------------------------------------------------------------------------------- */
int64_t sc_infonce_workspace_floats(int32_t Bg);
int sc_infonce_fwd(const float* A, const float* B, int32_t Bg, int32_t E, const int64_t* ids, const float* inv_temp, float margin,
                   int32_t dcl, int32_t a2b, int32_t b2a, float* logits, float* lse_row, float* lse_col, float* loss,
                   float* workspace, void* stream);
int sc_infonce_grad(const float* logits, const int64_t* ids, const float* lse_row, const float* lse_col, int32_t Bg,
                    const float* gscale /*device scalar*/, const float* inv_temp /*device scalar*/, float margin, int32_t dcl,
                    int32_t a2b, int32_t b2a, float* G, float* dlogit_dot /*[Bg]*/, void* stream);

/* CIF weight head (avssl/module/cif.py:106-129: ... Conv1d -> Dropout(0.5) -> ReLU -> Dropout(0.5) -> Linear(C, 1) -> Sigmoid): everything
 * behind the conv GEMM in one row kernel, forward and backward.  y [rows, C] fp32 = the conv output (bias included), w [C], bias [1]:
 *   alpha[row] = sigmoid(bias + sum_c w[c] m2 relu(m1 y[row, c]))     m1 / m2: dropout multipliers (p1 / p2, 0 in eval), keep bits =
 *   sc_dropout_mult_f32's for the seed on the index row * C + c.   backward: dy, and per-block partial sums dw_partial [nblk, C],
 *   db_partial [nblk] (reduce with sc_colsum_f32). */
int sc_cif_head_fwd(const float* y, int64_t ldy, const float* w, const float* bias, float* alpha, int32_t rows, int32_t C, float p1,
                    uint32_t seed1, float p2, uint32_t seed2, void* stream);
int sc_cif_head_bwd(const float* y, int64_t ldy, const float* w, const float* alpha, const float* dalpha, float* dy, int64_t lddy,
                    float* dw_partial, float* db_partial, int32_t nblk, int32_t rows, int32_t C, float p1, uint32_t seed1, float p2,
                    uint32_t seed2, void* stream);
/* dy as bf16 rows (dy_bf16 != 0: what the input-gradient GEMM of the weight conv reads) or fp32 */
int sc_cif_head_bwd_rows(const float* y, int64_t ldy, const float* w, const float* alpha, const float* dalpha, void* dy, int32_t dy_bf16,
                         int64_t lddy, float* dw_partial, float* db_partial, int32_t nblk, int32_t rows, int32_t C, float p1, uint32_t seed1,
                         float p2, uint32_t seed2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row tail of the parallel head, round 3 (csrc/rowtail.hip): the B-row products of nn.TransformerEncoderLayer + final LayerNorm +
 * Linear (avssl/module/kw_modules/TransformerModels.py:48-97, avssl/model/kw_branches.py:266-280) and the L2 normalisation of
 * avssl/model/kwClip.py:857,913-915, fp32 on the master weights.
 *   sc_rt_gemm: C[z] = epi(alpha A[z] . B[z]^T) on the matrix pipe in exact fp32, 64 x 64 tiles.  A [M, K] row-major (then it may be
 *     given as a_ns partial slices that are added on load, plus a_bias[k] * a_rowscale[row][k / a_group]) or contraction-major
 *     [K, M] (a_kmajor); B [N, K] or [K, N] (b_kmajor).  S > 1 splits the contraction: slice s writes its raw partial tile to
 *     C + s * c_slice and the CONSUMER adds the slices (no reduce launch, no epilogue).  S = 1: v = alpha acc + bias[n]; act = 1:
 *     U = v (if given), v = gelu_erf(v); dropout (keep bits = sc_dropout_mult_f32's on the index row * N + col); v += beta C; store.
 *     gb (a_kmajor only): gb[m] += sum_k A[k][m] - the bias gradient as a by-product of a weight-gradient product.
 *   sc_rt_gemm_slices: the S this library would pick for a few-row product (enough workgroups to cover the chip).
 *   sc_rt_ln_fwd: z = (sum_s y_s + bias) * dropout + residual ; out1 = LN1(z) [; out2 = LN2(out1)], xhat / rstd kept.
 *   sc_rt_ln_bwd: d = sum_s dy_s (+ add) ; dx = LayerNorm backward ; dx_masked = dx * dropout multiplier (optional) ; dgamma / dbeta +=.
 *   sc_rt_l2norm_fwd / _bwd: x = sum_s y_s + bias ; e = x / |x| ; backward dx = (g - e <g, e>) / |x|.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float* A; int64_t lda, a_slice, a_z; int32_t a_kmajor, a_ns;
    const float* a_bias; const float* a_rowscale; int32_t a_group, a_nscale;
    const float* B; int64_t ldb, b_z; int32_t b_kmajor, nbatch;
    float* C; int64_t ldc, c_slice, c_z;
    float* U;
    int32_t M, N, K, S;
    float alpha, beta;
    const float* bias; int64_t bias_z;
    int32_t act; float drop_p; uint32_t drop_seed; int32_t pad_;
    float* gb; int64_t gb_z;
} sc_rt_gemm_args;
int sc_rt_gemm(const sc_rt_gemm_args* args, void* stream);
int32_t sc_rt_gemm_slices(int32_t M, int32_t N, int32_t K, int32_t nbatch);
typedef struct {
    const float* y; int64_t y_slice; int32_t ns, rows;
    const float* bias;
    float drop_p; uint32_t drop_seed;
    const float* res; int64_t res_stride;
    const float *g1, *b1; float *out1, *xhat1, *rstd1;
    const float *g2, *b2; float *out2, *xhat2, *rstd2;
    float eps1, eps2; int32_t D, pad_;
} sc_rt_ln_args;
int sc_rt_ln_fwd(const sc_rt_ln_args* args, void* stream);
typedef struct {
    const float* dy; int64_t dy_slice; int32_t ns, rows;
    const float* add;
    const float *xhat, *gamma, *rstd;
    float *dx, *dx_masked;
    float drop_p; uint32_t drop_seed;
    float *dgamma, *dbeta;
    int32_t D, pad_;
} sc_rt_ln_bwd_args;
int sc_rt_ln_bwd(const sc_rt_ln_bwd_args* args, void* stream);
int sc_rt_l2norm_fwd(const float* y, int64_t y_slice, int32_t ns, const float* bias, float* x /*or NULL*/, float* e, float* rnorm,
                     int32_t rows, int32_t D, void* stream);
int sc_rt_l2norm_bwd(const float* g, const float* e, const float* rnorm, float* dx, int32_t rows, int32_t D, void* stream);
/* value-bias terms of the CLS head under attention-weight dropout (ctx_h = Wv_h m_h + bv_h * psum[b,h]):
 *   cbias[b,h] = sum_j dctx[b, h dh + j] bv[h dh + j]   (d psum)      gbv[h dh + j] += sum_b dctx[b, h dh + j] psum[b,h] */
int sc_rt_value_bias_bwd(const float* dctx, const float* bv, const float* psum, float* cbias, float* gbv, int32_t B, int32_t D, int32_t H,
                         void* stream);
/* end of the weighted-sum backward: d_soft[n] = sum_blk part[blk][n] ; out[n] = w[n] (d_soft[n] - sum_m w[m] d_soft[m])  (softmax backward) */
int sc_rt_softmax_bwd_reduce(const float* part, int32_t nblk, int32_t NL, const float* w_soft, float* out, void* stream);
/* elementwise over slices.  mode 0: out = sum_s y_s + bias[j] * rowscale[row][j / group] ; mode 1: u = sum_s y_s + bias, out = gelu_erf(u) *
 * dropout ; mode 2: out = (sum_s y_s) * dropout * gelu_erf'(u)   (dropout keep bits as sc_dropout_mult_f32 on the index row * D + j) */
int sc_rt_elem(const float* y, int64_t y_slice, int32_t ns, const float* bias, const float* rowscale, int32_t group, int32_t nscale, float* out,
               float* u, int32_t rows, int32_t D, int32_t mode, float drop_p, uint32_t drop_seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused driver: ONE call enqueues a whole frozen HuBERT encoder layer (fairseq TransformerSentenceEncoderLayer as invoked at
 * avssl/module/speech_encoder_plus.py:49-53) on the caller's stream - QKV GEMM (V^T epilogue) -> attention -> out_proj (+bias,
 * dropout, +residual) -> LayerNorm -> FC1 (+bias, GELU) -> FC2 (+bias, dropout, +residual) -> LayerNorm; pre_ln = 1: the
 * HuBERT-large order (LayerNorms in front of the two halves).  x / out: [B*R, D] bf16 in the padded row layout, head_dim 64.
 * Scratch (caller-owned, sc_workspace_bytes(SC_WS_HUBERT_LAYER, B*R, D, F) bytes in total): qk [B*R, 2D], vt [B, H, 64, R],
 * ctx / pre / x1 [B*R, D], ffn [B*R, F], all bf16.  p_attn / p_res = 0 in inference; seeds as in sc_gemm_args / sc_attn_fwd_bf16.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const sc_bf16* x; sc_bf16* out;
    const int32_t* valid_len;                 /* [B] keys per utterance */
    int32_t B, R, T, D, F, H, pre_ln, reserved;
    const sc_bf16 *qkv_w, *o_w, *fc1_w, *fc2_w;     /* [3D, D], [D, D], [F, D], [D, F] */
    const float *qkv_b, *o_b, *fc1_b, *fc2_b, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float eps, p_attn, p_res;
    uint32_t seed_attn, seed_o, seed_fc2;
    sc_bf16 *qk, *vt, *ctx, *pre, *x1, *ffn;        /* scratch */
    /* ---- fused_ln = 1 (post-LN order only): no LayerNorm launch - the residual stream stays RAW and the two LayerNorms are folded
     * into their neighbour GEMMs (sc_gemm_args, "LayerNorm folded").  `out` then receives the rows in FRONT of the layer's final
     * LayerNorm and out_stats their row statistics; x may itself be such a raw output (x_stats != NULL, x_ns strips, x_ln_g / x_ln_b =
     * the affine of the LayerNorm that applies to it; qkv_w / qkv_b / qkv_colsum are then the folded forms for THAT LayerNorm).
     * fc1_w / fc1_b / fc1_colsum are always the forms folded with ln1; ln1_g / ln1_b are still read (the residual of fc2).
     * stats1: scratch for the statistics of `pre`.  x1 is not used. */
    int32_t fused_ln, x_ns;
    const float* x_stats;
    const float *x_ln_g, *x_ln_b;
    const float *qkv_colsum, *fc1_colsum;
    float *stats1, *out_stats;
    /* ---- ragged rows (round 4): seg != NULL (host pointer, seg->row0 device) - x / out / scratch hold seg->rows rows in the
     * segment layout (R, T are then ignored; vt = per utterance [H, 64, pitch]); attn_work / n_attn_work as in sc_attn_fwd_seg_bf16 */
    const sc_segments* seg;
    const int32_t* attn_work;
    int32_t n_attn_work, reserved2;
} sc_hubert_layer_args;
int sc_hubert_layer_fwd(const sc_hubert_layer_args* args, void* stream);
#define SC_WS_INFONCE 0       /* a = Bg */
#define SC_WS_HUBERT_LAYER 1  /* a = B*R rows, b = D, c = F */
int64_t sc_workspace_bytes(int32_t what, int64_t a, int64_t b, int64_t c);

/* ------------------------------------------------------------------------------------------------
 * Optimiser step on a flat fp32 parameter buffer: torch.optim.Adam semantics (L2 weight decay added to
 * the gradient), gradient clipping by global norm folded in (avssl/model/kwClip.py:646-674 +
 * trainer.gradient_clip_val).  sumsq: partial sums of squares for the global norm.
 * ---------------------------------------------------------------------------------------------- */
int sc_sumsq_f32(const float* x, int64_t n, float* partial, int32_t nblk, void* stream);
int sc_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int32_t step, const float* gnorm_sq_partial, int32_t nblk,
                float max_norm, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Keyword prompt of the cascaded branches in the text tower's packed rows (round 4; replaces the ~35 element-wise torch launches of
 * ClipModel.encode_keywords, avssl/module/clip_official.py:222-279, and of the padding around the tower).
 *   sc_prompt_assemble: X [Bp*SEG, W] bf16 <- per sample b < B the prefix t < n_pos of
 *       [SOT, kw_1 .. kw_n, EOT, token 0 ...] + positional_embedding  (n = count[b]; keyword t-1 read from keywords[b*ldb + (t-1)*W],
 *       zero past the N keywords given), every other row (t >= n_pos, pad samples b >= B) zero.  tok [3, W] = the embeddings of SOT,
 *       EOT and token 0; pos [>= n_pos, W].  eot_row[b] <- b*SEG + min(count[b] + 1, n_pos - 1): the row the head reads; counts that
 *       pointed behind the prefix are clamped and added to *clamped (device counter, may be NULL).
 *   sc_prompt_assemble_bwd: dkeywords[b, j] <- float(dX[b*SEG + j + 1]) for j + 1 < min(count[b] + 1, n_pos), else 0 (every element
 *       of the [B, N, W] gradient is written).
 *   sc_rows_gather_bf16: out[b] <- float(X[row[b]]);   sc_rows_scatter_bf16: dX [M, W] <- 0 except dX[row[b]] = bf16(d[b]) (the row of
 *       sample b lies in its own segment b*SEG .. b*SEG + SEG - 1; every row of dX is written).
 * ---------------------------------------------------------------------------------------------- */
int sc_prompt_assemble(const float* keywords, int64_t ldb, const int64_t* count, const float* tok, const float* pos, sc_bf16* X,
                       int32_t* eot_row, int64_t* clamped, int32_t B, int32_t Bp, int32_t N, int32_t W, int32_t SEG, int32_t n_pos,
                       void* stream);
int sc_prompt_assemble_bwd(const sc_bf16* dX, const int64_t* count, float* dkeywords, int64_t ldb, int32_t B, int32_t N, int32_t W,
                           int32_t SEG, int32_t n_pos, void* stream);
int sc_rows_gather_bf16(const sc_bf16* X, const int32_t* row, float* out, int32_t B, int32_t W, void* stream);
int sc_rows_scatter_bf16(const float* d, const int32_t* row, sc_bf16* dX, int32_t M, int32_t B, int32_t W, int32_t SEG, void* stream);
/* mask[b, k] = (k >= lens[b] + add), k < n, as bytes (1 = padding): get_keypadding_mask (avssl/util/data_utils.py:6-22) and the attention
 * block's padded-pitch key mask from the lengths in one launch. */
int sc_len_mask_u8(const int64_t* lens, int32_t add, uint8_t* mask, int32_t B, int32_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
