"""Keyword -> CLIP sub-word vector quantiser and keyword BatchNorm of the cascaded+/hybrid+ branches on the library's kernels
(csrc/vq.hip; scope rows a11 / f3).

What the reference computes (avssl/model/kw_branches.py:158-197, avssl/module/speechclip_c_modules/my_vector_quantizer.py:64-165):
cosine of every projected keyword against every token embedding of the reduced vocabulary, the special tokens {0, 2, 3} masked
to -inf, ``argmax`` -> one-hot, and in training the straight-through estimator ``hard + soft - soft.detach()`` with
``soft = softmax(x / temp)``; then ``keywords = subword_prob @ token_embedding``.

How it runs here (``SimpleVectorQuantizer.quantize_keywords``, called by GeneralBranch.vq_audio_features):

    forward   kw -> normalise + transpose (sc_vq_prep_f32) -> cosine scores in exact fp32 on the matrix pipe (sc_sgemm_mfma_f32:
              the scores pick a discrete token, so no reduced-precision operand) -> one row pass (sc_vq_rowstats: mask, argmax,
              LSE(x / temp), LSE(x), entropy) -> perplexities (sc_vq_perplexity) -> keywords = table[argmax] (sc_vq_gather_f32:
              the value of ``hard @ table``; the dense (Nk, V) one-hot is never built)
    backward  t = g . table^T (bf16 MFMA GEMM, fp32 out) -> dx = soft (t - <soft, t>) / temp (sc_vq_soft_bwd, bf16) ->
              d cos-side = dx . normalised table (bf16 MFMA GEMM, split along the vocabulary) -> through the normalisation
              (sc_vq_norm_bwd_f32).  The reference trains under precision-16 autocast; gradients here are bf16 operands with
              fp32 accumulation.

``SimpleVectorQuantizer.forward(x)`` keeps the reference's module-level signature on a given score tensor (dense
``subword_prob`` with the same straight-through gradient, same kernels).  Same constructor keywords and result-dict keys.
There is no CPU path: these modules take device tensors only (the CPU restatement of this maths is oracle/cascaded_ref.py).
"""
import ast
import logging

import torch
import torch.nn as nn

from . import ops

logger = logging.getLogger(__name__)

__all__ = ["SimpleVectorQuantizer", "Kw_BatchNorm_dynamic", "VocabTables"]

SPECIAL_TOKENS = (0, 2, 3)        # my_vector_quantizer.py:64 default prob_msk
# the cosine scores on the fp32 matrix pipe (sc_sgemm_mfma_f32, rounds 2-5) instead of the three-way bf16 split product: A/B and tests
EXACT_FP32_SCORES = False


def _roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class VocabTables:
    """Derived copies of the frozen token table, built once per table version: the normalised table K-major in fp32 (the exact-fp32
    score GEMM of the debug / A-B path), its three-way bf16 split (the score GEMM of the product path), the table in bf16
    (``g . table^T``) and the normalised table K-major in bf16 (``dx . normalised table``)."""

    def __init__(self, weight: torch.Tensor):
        w = weight.detach().float().contiguous()
        V, Et = w.shape
        self.V, self.Et, self.Vp = V, Et, _roundup(V, 128)
        self.table = w
        wn = w / w.norm(dim=-1, keepdim=True).clamp_min(1e-8)                       # F.normalize(emb, dim=-1, eps=1e-8)
        self.norm_T = torch.zeros(Et, self.Vp, device=w.device, dtype=torch.float32)
        self.norm_T[:, :V] = wn.t()
        self.Etp = _roundup(Et, 64)                                                 # K of the bf16 GEMM g . table^T
        self.table_bf16 = torch.zeros(self.Vp, self.Etp, device=w.device, dtype=torch.bfloat16)
        self.table_bf16[:V, :Et] = w
        self.norm_T_bf16 = self.norm_T.to(torch.bfloat16)
        # round 6: the normalised table as three bf16 addends in the six-K-block layout (ops.split3_bf16): the cosine scores are then ONE
        # bf16 GEMM with fp32-accurate results (every bf16 x bf16 product is exact in fp32) instead of an fp32 GEMM on the fp32 matrix pipe
        self.norm_split = ops.split3_bf16(wn.contiguous(), 1, rows_pad=128)
        assert self.norm_split.shape[0] == self.Vp
        units = self.Vp // 64
        self.splits = max(s for s in range(1, 17) if units % s == 0)                # equal vocabulary slices, multiples of 64

    @staticmethod
    def of(weight: torch.Tensor, cache: dict) -> "VocabTables":
        key = (weight.data_ptr(), weight._version, tuple(weight.shape), weight.device)
        if cache.get("key") != key:
            cache["key"], cache["tables"] = key, VocabTables(weight)
        return cache["tables"]


class VQResult(dict):
    """Result dict of the fused path: ``subword_prob`` (a dense (B, N, V) one-hot nobody on the path reads - the keywords are
    gathered) is materialised on first access only."""

    def __missing__(self, key):
        if key == "subword_prob":
            bsz, tsz, V = self._shape
            self[key] = ops.vq_onehot(self["targets"].reshape(-1), V).view(bsz, tsz, V)
            return self[key]
        raise KeyError(key)


class _KeywordVQFn(torch.autograd.Function):
    """kw [Nk, Et] fp32 -> (keywords [Nk, Et] = table[argmax cos], idx [Nk], ent [Nk], ppl [2]); gradient to kw (training)."""

    @staticmethod
    def forward(ctx, kw, tb: VocabTables, temp: float, training: bool):
        Nk, Et = kw.shape
        kw = kw.detach().float().contiguous()
        kwn_T, rnorm = ops.vq_prep(kw)
        if EXACT_FP32_SCORES:
            cos = ops.sgemm_mfma(kwn_T, tb.norm_T, a_kmajor=True, b_kmajor=True)                               # [Nkp, Vp] fp32
        else:
            cos = ops.cosine_scores_split(kw, rnorm, tb.norm_split, tb.Vp)                                     # [Nkp, Vp] fp32, fp32-accurate
        idx, lse_t, lse_1, ent = ops.vq_rowstats(cos[:Nk], tb.V, temp, SPECIAL_TOKENS)
        ppl = ops.vq_perplexity(cos[:Nk], tb.V, idx, lse_1)
        out = ops.vq_gather(tb.table, idx)
        ctx.tb, ctx.temp, ctx.training = tb, temp, training
        if training:
            ctx.save_for_backward(kw, rnorm, cos, lse_t)
        ctx.mark_non_differentiable(idx, ent, ppl)
        return out, idx, ent, ppl

    @staticmethod
    def backward(ctx, g, *_):
        if not ctx.training:               # inference: subword_prob is the hard one-hot, nothing flows back
            return None, None, None, None
        kw, rnorm, cos, lse_t = ctx.saved_tensors
        tb, temp = ctx.tb, ctx.temp
        Nk, Et = kw.shape
        Nkp, Vp = cos.shape
        dev = kw.device
        gb = torch.zeros(Nkp, tb.Etp, device=dev, dtype=torch.bfloat16)
        gb[:Nk, :Et] = g
        t = torch.empty(Nkp, Vp, device=dev, dtype=torch.float32)
        ops.gemm_raw(gb, tb.Etp, tb.table_bf16, tb.Etp, t, Vp, Nkp, Vp, tb.Etp, out_f32=True)     # d subword_prob = g . table^T
        dx = torch.zeros(Nkp, Vp, device=dev, dtype=torch.bfloat16) if Nkp != Nk else None
        dxv = ops.vq_soft_bwd(cos[:Nk], lse_t, t[:Nk], tb.V, temp, out_bf16=True, Vpad=Vp)
        if dx is None:
            dx = dxv
        else:
            dx[:Nk] = dxv
        S = tb.splits
        Kc = Vp // S
        part = torch.empty(S, Nkp, Et, device=dev, dtype=torch.float32)
        ops.gemm_raw(dx, Vp, tb.norm_T_bf16, Vp, part, Et, Nkp, Et, Kc, out_f32=True, nb1=S, sA=(Kc, 0), sW=(Kc, 0), sC=(Nkp * Et, 0))
        dkwn = torch.empty(Nkp, Et, device=dev, dtype=torch.float32)
        ops.colsum(part, Nkp * Et, S, Nkp * Et, dkwn)
        return ops.vq_norm_bwd(kw, rnorm, dkwn[:Nk]), None, None, None


class _VQDenseFn(torch.autograd.Function):
    """module-level API on a given score matrix x [Nk, V] (masked in place): dense hard one-hot with the straight-through gradient."""

    @staticmethod
    def forward(ctx, x, temp: float, training: bool, prob_msk):
        Nk, V = x.shape
        idx, lse_t, lse_1, ent = ops.vq_rowstats(x, V, temp, prob_msk)
        ppl = ops.vq_perplexity(x, V, idx, lse_1)
        prob = ops.vq_onehot(idx, V)
        ctx.temp, ctx.training = temp, training
        if training:
            ctx.save_for_backward(x, lse_t)
        ctx.mark_non_differentiable(idx, ent, ppl)
        return prob, idx, ent, ppl

    @staticmethod
    def backward(ctx, g, *_):
        if not ctx.training:
            return None, None, None, None
        x, lse_t = ctx.saved_tensors
        return ops.vq_soft_bwd(x, lse_t, g.float().contiguous(), x.shape[1], ctx.temp, out_bf16=False), None, None, None


class SimpleVectorQuantizer(nn.Module):
    def __init__(self, temp, groundTruthPerplexity=None, time_first=True, use_gumbel=False, hard=True):
        super().__init__()
        self.time_first, self.use_gumbel, self.hard = time_first, use_gumbel, hard
        if use_gumbel or not hard:
            raise NotImplementedError("only the shipped quantiser is built: use_gumbel false, hard true "
                                      "(config/speechCLIP+/*: vq.args)")
        if isinstance(temp, str):
            if temp.startswith("learnable="):
                self.temp_type = "learnable"
                self.curr_temp = nn.parameter.Parameter(torch.FloatTensor([ast.literal_eval(temp[len("learnable="):])]))
            elif temp.startswith("fixed="):
                self.temp_type = "fixed"
                self._fixed_temp = float(ast.literal_eval(temp[len("fixed="):]))
                self.register_buffer("curr_temp", torch.FloatTensor([self._fixed_temp]))
            else:
                self.temp_type = "scheduled"
                sched = ast.literal_eval(temp)
                assert len(sched) == 3, f"{sched}, {len(sched)}"
                self.max_temp, self.min_temp, self.temp_decay = sched
                self.curr_temp = self.max_temp
        self.codebook_indices = None
        self.groundTruthPerplexity = groundTruthPerplexity
        self._tables = {}

    def set_num_updates(self, num_updates):
        if self.temp_type == "scheduled":
            self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)

    def _temp(self) -> float:
        if self.temp_type == "fixed":
            return self._fixed_temp
        if self.temp_type == "learnable":
            if self.training and torch.is_grad_enabled():
                raise NotImplementedError("temp='learnable=...': the gradient to the temperature is not built (no shipped config)")
            return float(self.curr_temp.item())
        return float(self.curr_temp)

    def _result(self, bsz, tsz, V, idx, ent, ppl, temp) -> VQResult:
        r = VQResult(num_vars=V)
        r._shape = (bsz, tsz, V)
        if bsz * tsz == 1:
            # my_vector_quantizer.py:90 squeezes the (1, V) one-hot: its "mean over rows" becomes the mean over the vocabulary
            p = torch.full((), 1.0 / V, device=idx.device)
            r["code_perplexity"] = torch.exp(-(p * torch.log(p + 1e-7)))
        else:
            r["code_perplexity"] = ppl[0]
        r["prob_perplexity"] = ppl[1]
        r["ent_per_t"] = ent.view(bsz, tsz).mean(dim=0)
        r["temp"] = temp
        if self.groundTruthPerplexity is not None:
            gt = float(self.groundTruthPerplexity)
            r["diversity_loss"] = (r["prob_perplexity"] - gt) ** 2 / (V - gt) ** 2
        else:
            r["diversity_loss"] = (V - r["prob_perplexity"]) / V
        r["targets"] = idx.view(bsz, tsz, 1)
        return r

    def forward(self, x, prob_msk=[0, 2, 3], produce_targets=True):
        """my_vector_quantizer.py:64-165 on a score tensor (B, T, V) (or (B, V, T) when not time_first); the masked columns of
        ``x`` are overwritten with -inf like the reference's in-place add."""
        if not x.is_cuda:
            raise RuntimeError("SimpleVectorQuantizer runs on the HIP kernels: device tensors only (CPU restatement: oracle/)")
        if not self.time_first:
            x = x.transpose(1, 2)
        bsz, tsz, fsz = x.shape
        x2 = x.reshape(bsz * tsz, fsz)
        if x2.dtype != torch.float32 or x2.stride(1) != 1:
            x2 = x2.float().contiguous()
        temp = self._temp()
        prob, idx, ent, ppl = _VQDenseFn.apply(x2, temp, self.training, tuple(prob_msk))
        r = self._result(bsz, tsz, fsz, idx, ent, ppl, temp)
        r["subword_prob"] = prob.view(bsz, tsz, fsz)
        if not produce_targets:
            del r["targets"]
        return r

    def quantize_keywords(self, keywords: torch.Tensor, token_table: torch.Tensor):
        """Fused cosine -> mask -> argmax / straight-through -> ``@ token_table`` (kw_branches.py:158-197):
        keywords (B, N, Et) projected keyword embeddings, token_table (V, Et) frozen -> (result dict, quantised keywords)."""
        if not keywords.is_cuda:
            raise RuntimeError("SimpleVectorQuantizer runs on the HIP kernels: device tensors only (CPU restatement: oracle/)")
        assert not token_table.requires_grad
        tb = VocabTables.of(token_table, self._tables)
        bsz, tsz, Et = keywords.shape
        temp = self._temp()
        out, idx, ent, ppl = _KeywordVQFn.apply(keywords.reshape(bsz * tsz, Et), tb, temp, self.training)
        return self._result(bsz, tsz, tb.V, idx, ent, ppl, temp), out.view(bsz, tsz, Et).to(keywords.dtype)


class _KwBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, training: bool, momentum: float, eps: float):
        x = x.detach().float().contiguous()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y, sm, sr = ops.bn_rows_fwd(x, g, b, run_mean, run_var, training, momentum, eps)
        ctx.training = training
        if training:
            ctx.save_for_backward(x, g, sm, sr)
        else:
            ctx.save_for_backward(g, run_var.detach().clone())
            ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.float().contiguous()
        if ctx.training:
            x, g, sm, sr = ctx.saved_tensors
            dx, dg, db = ops.bn_rows_bwd(x, dy, g, sm, sr)
            return dx, dg, db, None, None, None, None, None
        g, rv = ctx.saved_tensors                      # inference statistics are constants: a per-channel scale
        return dy * (g * torch.rsqrt(rv + ctx.eps)), None, None, None, None, None, None, None


class Kw_BatchNorm_dynamic(nn.Module):
    """BatchNorm1d over keyword embeddings, initialised from the CLIP token-embedding mean / std (kw_bn.py:167-228).
    ``bn_layer`` holds the parameters and running statistics under the reference's state-dict names; the arithmetic runs on
    sc_bn_rows_fwd / _bwd."""

    def __init__(self, kw_dim: int, init_bias: torch.Tensor, init_scale: torch.Tensor, std_scale: int = 1,
                 learnable: bool = True) -> None:
        super().__init__()
        assert std_scale > 0, f"std scale must > 0, but input std scale is {std_scale}"
        self.kw_dim, self.learnable, self.std_scale = kw_dim, learnable, std_scale
        self.bn_layer = nn.BatchNorm1d(kw_dim)
        self.bn_layer.weight.data.copy_(init_scale * std_scale)
        self.bn_layer.bias.data.copy_(init_bias)
        self.bn_layer.weight.requires_grad = learnable
        self.bn_layer.bias.requires_grad = learnable

    def forward(self, keywords: torch.Tensor) -> torch.Tensor:
        assert keywords.dim() == 3
        if not keywords.is_cuda:
            raise RuntimeError("Kw_BatchNorm_dynamic runs on the HIP kernels: device tensors only (CPU restatement: oracle/)")
        bn = self.bn_layer
        B, N, E = keywords.shape
        training = self.training
        if training:
            bn.num_batches_tracked.add_(1)
        y = _KwBNFn.apply(keywords.reshape(B * N, E), bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                          float(bn.momentum), float(bn.eps))
        return y.view(B, N, E).to(keywords.dtype)
