// Flash-style self-attention forward for gfx950, head_dim 64, key padding by per-utterance length.
//
// One workgroup = 4 waves = 128 query rows of one (utterance, head); each wave owns 32 queries.
// Scores are computed TRANSPOSED, S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS,
// B = Q rows held in registers), so a lane holds one query column and its 16 (of 32) keys in registers:
// the row max / row sum are in-lane reductions plus one exchange with lane^32.  The S^T accumulator is then,
// converted to bf16, directly the B operand of  O^T += V^T . P^T  (no LDS round trip, no lane movement);
// V arrives already transposed per head (written by the QKV GEMM's epilogue), so the A operand V^T[d][keys]
// is two 8-byte LDS reads.  K and V^T tiles (64 keys) are staged through LDS with register prefetch
// (global loads for tile t+1 are issued before tile t's MFMAs, written to LDS after them).
// LDS images are XOR-swizzled so the b128 (K) and b64 (V^T) fragment reads are bank-conflict free.
#include "sc_common.h"

namespace {

constexpr int KT = 64;   // keys per LDS tile

// XOR swizzle of the V^T tile image ([64 d rows][64 keys] bf16, 128-byte rows, 8-byte units): unit u of row r lives at unit u ^ vsw(r).
//   (r >> 1) & 15 : the 32 rows a ds_read_b64 lane group touches cover all 64 banks once (rows 2j / 2j + 1 sit in the two 128-byte halves
//                   of the 256-byte bank row);
//   ^ (r & 1)     : the staging stores are ds_write_b64 - 16 contiguous lanes = rows r, r + 1, banks mod 32: without this bit both rows
//                   wrote the same banks (2-way);
//   ^ (r >> 5)    : round 4 - rows r and r + 32 used to sit exactly 4096 bytes apart, so the compiler fused the two fragment reads of a
//                   step into ds_read2st64_b64: 16-lane groups on 32 banks = half the rate of ds_read_b64 AND 2-way conflicts
//                   (rocprofv3: SQ_LDS_BANK_CONFLICT = 38 % of SQ_LDS_IDX_ACTIVE, profiles/r04_attn_pmc.json); with this bit the
//                   distance depends on the lane and the reads stay ds_read_b64.
// (sc_common.h: sc_tr_swizzle - shared with the backward kernels' transposed tiles)
#ifdef SC_AB_OLD_VSW
__device__ __forceinline__ int vsw(int row) { return (row >> 1) & 15; }
#else
__device__ __forceinline__ int vsw(int row) { return sc_tr_swizzle(row); }
#endif

// STAMP (diagnostics build only, sc_diag_attn_fwd_stamps): per wave, the shader-clock cycles spent between the landmarks of a key
// tile, summed over the tiles -> stamps[wg / 32][wave][8]: 0 staging writes + barrier, 1 next-tile loads + S^T MFMAs issued, 2 softmax
// of block 0 (waits for its S^T), 3 P.V of block 0 issued, 4 softmax of block 1, 5 P.V of block 1 issued, 6 closing barrier, 7 total.
// s_memtime needs an s_waitcnt lgkmcnt(0), i.e. the LDS reads in flight at a landmark are drained there: the stamped kernel is slower
// than the production one (tools/attn_stamps.py prints both) - the SPLIT between the sections is what it is for.
template <int DROP, int STAMP = 0>   // DROP: train-mode probability dropout as a compile-time variant: the eval kernel carries none of its registers
__global__ __launch_bounds__(256) void attn_fwd_kernel(const uint16_t* __restrict__ qk, int64_t ldqk,
                                                        const uint16_t* __restrict__ vt,
                                                        const int32_t* __restrict__ valid_len,
                                                        uint16_t* __restrict__ out, int64_t ldo, int R, int H, int D,
                                                        float c /* scale * log2(e) */, float* __restrict__ lse2, int causal,
                                                        float drop_p, uint32_t drop_seed, const int32_t* __restrict__ row0,
                                                        const int32_t* __restrict__ work, int npairs, int rows_total, int max_pitch,
                                                        long long* __restrict__ stamps = nullptr) {
    // Round 5: two K and two V^T tile buffers (32 KiB), filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write)
    // one tile ahead; ONE barrier per tile.  Both images are [64 rows][128 B] with the 16-byte chunk index XOR (row >> 1) & 7 - the
    // swizzle sits in the DMA's per-lane source address.
    __shared__ __attribute__((aligned(16))) char KVs[2][2][KT * 128];      // [buffer][K | V^T]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // 1-D grid, XCD-aware decode: blocks bid, bid + 8, ... share an XCD (and its L2); give each XCD a contiguous
    // range of (utterance, head, q-block) triples with the q-block fastest, so the R/128 workgroups that re-read the
    // same K / V^T of one (utterance, head) hit in that XCD's L2 instead of fetching it once per XCD.
    const int nqb = (R + 127) >> 7;
    int logical;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // uniform rows: utterance b at row b R.  Ragged rows (row0 != NULL, sc_segments): utterance b at row0[b] with its own pitch; the
    // (utterance, q-block) pairs come from the host's work list (longest first) or, without one, from the (B x max q-blocks) rectangle
    int qblk, h, b, r0, n_valid_item = -1;
    if (row0) {
        const int pair = logical % npairs;
        h = logical / npairs;
        if (work) {
            // one 16-byte item = (utterance | q-block << 16, first row, pitch, key count or -1): everything the workgroup needs to
            // form its addresses after ONE load (three dependent loads - item, row0[b], row0[b + 1] - cost 5 us per launch)
            const int4 it = *(const int4*)(work + 4 * pair);
            b = it.x & 0xffff;
            qblk = it.x >> 16;
            r0 = it.y;
            R = it.z;
            n_valid_item = it.w;
        } else {
            b = pair / nqb;
            qblk = pair % nqb;
            r0 = row0[b];
            R = row0[b + 1] - r0;                                    // this utterance's pitch (a multiple of 8)
        }
        if (qblk * 128 >= R) return;                                 // uniform for the workgroup, before any barrier
    } else {
        qblk = logical % nqb;
        const int bh = logical / nqb;
        h = bh % H;
        b = bh / H;
        r0 = b * R;
    }
    // a last, shorter block: the waves past the pitch still stage K / V^T and meet the barriers, but neither multiply nor store;
    // inside the wave that straddles the pitch the lanes past it compute on the next utterance's rows and skip their store
    const bool wave_on = qblk * 128 + wave * 32 < R;
    const int q0 = wave_on ? qblk * 128 + wave * 32 : 0;
    int n_valid = n_valid_item >= 0 ? n_valid_item : valid_len[b];
    n_valid = max(1, min(n_valid, R));

    // Q fragments (B operand): Q[q0 + l31][ks*16 + 8*half + j]
    bf16x8 qf[4];
    {
        const uint16_t* qp = qk + ((int64_t)r0 + q0 + l31) * ldqk + h * 64 + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // loader mapping: a tile = 8 pieces of 8 rows x 128 B (one LDS-DMA wave instruction each); wave w brings pieces w and w + 4
    // Round 6: a source address = wave-uniform 64-bit base (SGPRs: utterance, head, piece, tile) + ONE 32-bit per-lane byte offset for
    // the K pieces and one for the V^T pieces (row inside the piece and swizzled chunk: (row >> 1) & 7 does not depend on the piece):
    // global_load_lds_dwordx4 v, s[base] - 2 address VGPRs instead of four 64-bit pointers and no 64-bit vector adds per tile
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int srow = lane >> 3, sch = (lane & 7) ^ ((wave_u * 4 + (lane >> 4)) & 7);
    const uint32_t k_lane = (uint32_t)(srow * (int)ldqk + sch * 8) * 2u;            // bytes
    const uint32_t v_lane = (uint32_t)(srow * R + sch * 8) * 2u;
    const char* k_base = (const char*)(qk + ((int64_t)r0 + wave_u * 8) * ldqk + D + h * 64);
    const char* v_base = (const char*)(vt + (int64_t)D * r0 + (int64_t)(h * 64 + wave_u * 8) * R);   // V^T of the utterance: [H, 64, pitch] at D * r0
    auto stage = [&](int t, int buf) {
        uint32_t kl = k_lane, vl = v_lane;           // opaque: keeps base + lane offset from being re-associated into 64-bit vector pointers
        asm volatile("" : "+v"(kl), "+v"(vl));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* kb = k_base + ((int64_t)t * KT + 32 * i) * ldqk * 2;
            const char* vb = v_base + ((int64_t)32 * i * R + t * KT) * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + kl),
                                             (__attribute__((address_space(3))) void*)(&KVs[buf][0][(wave + 4 * i) * 1024]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + vl),
                                             (__attribute__((address_space(3))) void*)(&KVs[buf][1][(wave + 4 * i) * 1024]), 16, 0, 0);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;        // finite floor: a key block may be fully masked for a query (segment-causal packing)

    const int qrow = q0 + l31;                                       // this lane's query
    // attention-probability dropout (fairseq attention_dropout, train mode): P' = mask . P / (1 - p) with the row sum taken
    // over the un-masked P; element (b, h, q, k) -> byte of sc_hash32(quad) as in sc_common.h (sc_drop8_*)
    const uint32_t drop_thr = DROP ? sc_drop8_thr(drop_p) : 0u;        // 8-bit fields: one hash word per four probabilities (sc_common.h)
    const uint32_t drop_row = row0 ? (uint32_t)(h * rows_total + r0 + qrow) * (uint32_t)max_pitch
                                   : (uint32_t)(((b * H + h) * R + qrow)) * (uint32_t)R;
    if (causal) n_valid = min(n_valid, qblk * 128 + 128);           // keys beyond the block's last query are all masked
    const int ntiles = (n_valid + KT - 1) / KT;
    stage(0, 0);
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0, st_first = 0;
#define SC_ST(K)                                         \
    do {                                                 \
        if (STAMP) {                                     \
            const long long now_ = clock64();            \
            st_acc[K] += now_ - st_prev;                 \
            st_prev = now_;                              \
        }                                                \
    } while (0)
    if (STAMP) st_first = st_prev = clock64();
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        // tile t has been staged one iteration ago (or above); retire this wave's pieces, meet the others', start the next tile
        const char* Ks = KVs[t & 1][0];
        const char* Vs = KVs[t & 1][1];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // every wave's pieces of tile t are in; everyone has finished tile t - 1
        asm volatile("" ::: "memory");
        SC_ST(0);
        if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);               // into the buffer tile t - 1 was read from
        // Two 32-key blocks per tile, software-pipelined inside the wave: both S^T blocks are issued first, so the
        // matrix pipe works on block 1 while the VALU runs block 0's softmax, and on P.V of block 0 during block 1's.
        const bool blk1 = key0 + 32 < n_valid;                       // wave-uniform
        // Round 5: the S^T MFMA takes its 32 key rows in a PERMUTED order (bits 2 and 3 of the row index swapped), so accumulator
        // register r of lane half h is key (r & 7) + 8 h + 16 (r >> 3) of the block: the 8 k-slots a lane half feeds into a P.V MFMA
        // are 8 CONSECUTIVE keys = ONE 16-byte chunk of a V^T row (was: two 8-byte pieces 16 bytes apart, two ds_read_b64)
        const int kperm = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | (l31 & 16);
        if (wave_on) {
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(Ks + kperm * 128 + (((2 * ks + half) ^ ((kperm >> 1) & 7)) << 4));
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s0, 0, 0, 0);
        }
        if (blk1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int krow = 32 + kperm;
                const bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + (((2 * ks + half) ^ ((krow >> 1) & 7)) << 4));
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s1, 0, 0, 0);
            }
        }
        auto softmax_block = [&](f32x16& sv, int kbase, bf16x8 (&pf)[2]) {
            // causal > 1: causal inside aligned segments of `causal` rows (32 / 64) - short sequences packed back to back into one
            // 128-row block attend to their own segment only
            if (kbase + 32 > n_valid || (causal && kbase + 31 > q0) || causal > 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = kbase + (r & 7) + 8 * half + 16 * (r >> 3);
                    if (kidx >= n_valid || (causal && kidx > qrow) || (causal > 1 && kidx < (qrow & ~(causal - 1)))) sv[r] = -INFINITY;
                }
            }
            float mloc = sv[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sv[r]);
            {   // exchange with lane ^ 32 on the VALU (v_permlane32_swap) instead of ds_bpermute
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
                mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            // deferred rescale: keep the running max stale while it would grow by < 2^6 in the exponent domain
            // (p stays <= 64, harmless for the fp32 sums and the bf16 P operand).  The DECISION is per query (round 4): when some
            // lane of the wave needs it the rescale code runs, but a query that does not need it multiplies by exp2(0) = 1 exactly
            // and keeps its max - so a query's bits do not depend on which other queries share its wave (ragged vs padded rows)
            const float m_cand = fmaxf(m_run, mloc);
            const bool grow = (m_cand - m_run) * c > 6.0f;
            if (__any(grow)) {
                const float m_new = grow ? m_cand : m_run;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
                m_run = m_new;
            }
            const float mc = m_run * c;
            float psum = 0.f;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = __builtin_amdgcn_exp2f(fmaf(sv[r], c, -mc));
                psum += pv[r];
            }
            l_run += psum;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[s2][j] = (__bf16)pv[8 * s2 + j];
            if (DROP) {
                // Round 6: one hash word per FOUR probabilities (sc_common.h sc_drop8_*): a lane's 8 consecutive keys of a register group
                // are two quads = two words.  Bytes 0 / 2 of a word (quad positions 0, 1) are tested as one packed 16-bit pair, bytes
                // 1 / 3 (positions 2, 3) as the next: field - thr8 is negative exactly for the dropped fields, its sign smeared over the
                // half-word is the AND-NOT mask on the packed bf16 pair.  Four packed integer ops per PAIR, no compare -> VCC -> select.
                // (inline asm: written with vector builtins the compiler folds the sequence back into v_cmp + v_cndmask)
                const uint32_t thr2 = drop_thr | (drop_thr << 16), sh2 = 0x000f000fu, eight2 = 0x00080008u;
                const uint32_t quad0 = (drop_row + (uint32_t)(kbase + 8 * half)) >> 2;        // drop_row, kbase multiples of 8
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint4 w = __builtin_bit_cast(uint4, pf[s2]);
                    uint32_t* wp = (uint32_t*)&w;
#pragma unroll
                    for (int qd = 0; qd < 2; ++qd) {                                           // keys kbase + 16 s2 + 8 half + 4 qd .. + 3
                        const uint32_t hsh = sc_hash32((quad0 + (uint32_t)(4 * s2 + qd)) ^ drop_seed);
                        uint32_t f1, d0, d1, m0, m1;
                        const uint32_t f0 = hsh & 0x00ff00ffu;                                 // (byte 0, byte 2): positions 0, 1
                        asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(f1) : "s"(eight2), "v"(hsh)); // (byte 1, byte 3): positions 2, 3
                        asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d0) : "v"(f0), "s"(thr2));
                        asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d1) : "v"(f1), "s"(thr2));
                        asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m0) : "s"(sh2), "v"(d0));     // 0xffff where dropped
                        asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m1) : "s"(sh2), "v"(d1));
                        asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(wp[2 * qd]) : "v"(m0), "v"(wp[2 * qd]));           // ~m & w
                        asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(wp[2 * qd + 1]) : "v"(m1), "v"(wp[2 * qd + 1]));
                    }
                    pf[s2] = __builtin_bit_cast(bf16x8, w);
                }
            }
        };
        auto pv_block = [&](int kb, const bf16x8 (&pf)[2]) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int chunk = kb * 4 + 2 * s2 + half;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int vrow = dt * 32 + l31;
                    const bf16x8 vf = *(const bf16x8*)(Vs + vrow * 128 + ((chunk ^ ((vrow >> 1) & 7)) << 4));
                    if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], o0, 0, 0, 0);
                    else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], o1, 0, 0, 0);
                }
            }
        };
        bf16x8 pf0[2], pf1[2];
        SC_ST(1);
        softmax_block(s0, key0, pf0);
        SC_ST(2);
        pv_block(0, pf0);
        SC_ST(3);
        if (blk1) {
            softmax_block(s1, key0 + 32, pf1);
            SC_ST(4);
            pv_block(1, pf1);
            SC_ST(5);
        }
        }
        SC_ST(6);
    }
    if (STAMP && stamps && (blockIdx.x & 31) == 0 && lane == 0) {
        st_acc[7] = clock64() - st_first;
#pragma unroll
        for (int k = 0; k < 8; ++k) stamps[((blockIdx.x >> 5) * 4 + wave) * 8 + k] = st_acc[k];
    }
#undef SC_ST
    if (!wave_on) return;

    const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l_tot = __uint_as_float(lsw[0]) + __uint_as_float(lsw[1]);
    const float inv = DROP ? sc_drop8_scale(drop_thr) / l_tot : 1.0f / l_tot;
    if (qrow >= R) return;                                          // lanes past the pitch: the next utterance's rows
    if (lse2 && half == 0)
        lse2[row0 ? (int64_t)h * rows_total + r0 + qrow : ((int64_t)b * H + h) * R + qrow] = m_run * c + __builtin_amdgcn_logf(l_tot);   // log2 domain
    uint16_t* op = out + ((int64_t)r0 + q0 + l31) * ldo + h * 64 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint2 w0, w1;
        w0.x = pack2bf(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv);
        w0.y = pack2bf(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        w1.x = pack2bf(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv);
        w1.y = pack2bf(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        *(uint2*)(op + 8 * g) = w0;
        *(uint2*)(op + 32 + 8 * g) = w1;
    }
}

}  // namespace

extern "C" int sc_attn_fwd_bf16(const sc_bf16* qk, int64_t ldqk, const sc_bf16* vt, const int32_t* valid_len,
                                sc_bf16* out, int64_t ldo, int32_t B, int32_t R, int32_t H, int32_t D, float scale,
                                float* lse2, int32_t causal, float drop_p, uint32_t drop_seed, void* stream) {
    SC_CHECK(qk && vt && valid_len && out, "sc_attn_fwd_bf16: null pointer");
    // R % 128 != 0 (round 4): a last q-block of 32 / 64 / 96 rows; the K / V^T tiles of 64 keys may then read up to 32 rows / 64
    // elements past an utterance (into the next one, or - behind the last - into slack the caller provides: 64 rows of qk, 64
    // elements of vt); what they read there is masked
    SC_CHECK(B > 0 && H > 0 && R > 0 && R % 8 == 0, "sc_attn_fwd_bf16: R=%d must be a positive multiple of 8", R);
    SC_CHECK(D == H * 64, "sc_attn_fwd_bf16: head_dim must be 64 (D=%d, H=%d)", D, H);
    SC_CHECK(causal == 0 || causal == 1 || causal == 32 || causal == 64, "sc_attn_fwd_bf16: causal=%d (0, 1, or a segment of 32 / 64 rows)", causal);
    SC_CHECK(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || (int64_t)B * H * R * R < ((int64_t)1 << 32)),
             "sc_attn_fwd_bf16: drop_p=%f (needs B*H*R*R < 2^32)", (double)drop_p);
    SC_CHECK(ldqk % 8 == 0 && ldo % 4 == 0 && ldqk >= 2 * D && ldo >= D, "sc_attn_fwd_bf16: bad leading dims");
    SC_CHECK(((uintptr_t)qk % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)out % 8) == 0,
             "sc_attn_fwd_bf16: alignment");
    dim3 grid(((R + 127) / 128) * H * B);
    if (drop_p > 0.f)
        hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, qk, ldqk, vt, valid_len, out, ldo, R,
                           H, D, scale * 1.4426950408889634f, lse2, causal, drop_p, drop_seed, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, 0, 0);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, qk, ldqk, vt, valid_len, out, ldo, R,
                           H, D, scale * 1.4426950408889634f, lse2, causal, drop_p, drop_seed, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, 0, 0);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sc_attn_fwd_seg_bf16(const sc_bf16* qk, int64_t ldqk, const sc_bf16* vt, const int32_t* valid_len, sc_bf16* out, int64_t ldo,
                                    const sc_segments* seg, const int32_t* work, int32_t nwork, int32_t H, int32_t D, float scale,
                                    float* lse2, int32_t causal, float drop_p, uint32_t drop_seed, void* stream) {
    SC_CHECK(qk && vt && valid_len && out && seg && seg->row0, "sc_attn_fwd_seg_bf16: null pointer");
    SC_CHECK(!work || ((uintptr_t)work % 16) == 0, "sc_attn_fwd_seg_bf16: the work list must be 16-byte aligned");
    SC_CHECK(seg->B > 0 && seg->B < 65536 && H > 0 && seg->rows > 0 && seg->max_pitch > 0 && seg->max_pitch % SC_SEG_ROWS == 0 && (!work || nwork > 0),
             "sc_attn_fwd_seg_bf16: B=%d rows=%d max_pitch=%d", seg->B, seg->rows, seg->max_pitch);
    SC_CHECK(D == H * 64, "sc_attn_fwd_seg_bf16: head_dim must be 64 (D=%d, H=%d)", D, H);
    SC_CHECK(causal == 0 || causal == 1, "sc_attn_fwd_seg_bf16: causal=%d", causal);
    SC_CHECK(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || (int64_t)H * seg->rows * seg->max_pitch < ((int64_t)1 << 32)),
             "sc_attn_fwd_seg_bf16: drop_p=%f (needs H*rows*max_pitch < 2^32)", (double)drop_p);
    SC_CHECK(ldqk % 8 == 0 && ldo % 4 == 0 && ldqk >= 2 * D && ldo >= D, "sc_attn_fwd_seg_bf16: bad leading dims");
    SC_CHECK(((uintptr_t)qk % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)out % 8) == 0, "sc_attn_fwd_seg_bf16: alignment");
    const int nqb = (seg->max_pitch + 127) / 128;
    const int npairs = work ? nwork : seg->B * nqb;
    dim3 grid(npairs * H);
    if (drop_p > 0.f)
        hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, qk, ldqk, vt, valid_len, out, ldo, seg->max_pitch,
                           H, D, scale * 1.4426950408889634f, lse2, causal, drop_p, drop_seed, seg->row0, work, npairs, seg->rows, seg->max_pitch);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, qk, ldqk, vt, valid_len, out, ldo, seg->max_pitch,
                           H, D, scale * 1.4426950408889634f, lse2, causal, drop_p, drop_seed, seg->row0, work, npairs, seg->rows, seg->max_pitch);
    SC_LAUNCH_CHECK();
    return 0;
}

#ifdef SC_DIAG_BUILD
// diagnostics library only (not declared in include/speechclip_hip.h): the uniform-pitch forward with per-section cycle stamps
extern "C" int sc_diag_attn_fwd_stamps(const sc_bf16* qk, int64_t ldqk, const sc_bf16* vt, const int32_t* valid_len, sc_bf16* out, int64_t ldo,
                                       int32_t B, int32_t R, int32_t H, int32_t D, float scale, float drop_p, uint32_t drop_seed,
                                       long long* stamps /* [ceil(grid / 32)][4][8] */, void* stream) {
    SC_CHECK(qk && vt && valid_len && out && stamps && D == H * 64 && R % 8 == 0, "sc_diag_attn_fwd_stamps: bad arguments");
    dim3 grid(((R + 127) / 128) * H * B);
    if (drop_p > 0.f)
        hipLaunchKernelGGL((attn_fwd_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, qk, ldqk, vt, valid_len, out, ldo, R, H, D,
                           scale * 1.4426950408889634f, (float*)nullptr, 0, drop_p, drop_seed, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, 0, 0, stamps);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<0, 1>), grid, dim3(256), 0, (hipStream_t)stream, qk, ldqk, vt, valid_len, out, ldo, R, H, D,
                           scale * 1.4426950408889634f, (float*)nullptr, 0, drop_p, drop_seed, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, 0, 0, stamps);
    SC_LAUNCH_CHECK();
    return 0;
}
#endif
