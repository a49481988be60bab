"""Frozen CLIP text tower as the cascaded branches use it: mirror of ClipModel.encode_keywords
(avssl/module/clip_official.py:222-279) over a restatement of openai/CLIP's text transformer (requirements.txt:4,
unpinned; the `clip` package and its weights are not available offline).

Architecture restated from the published model: token_embedding (V x W), learned positional_embedding (77 x W),
`layers` pre-LN residual blocks [ln_1 -> causal nn.MultiheadAttention(W, heads) ; ln_2 -> Linear(W, 4W) -> QuickGELU
(x * sigmoid(1.702 x)) -> Linear(4W, W)], ln_final, text_projection (W x E).  ViT-B/32: W = 512, 8 heads, 12 layers,
E = 512; ViT-L/14: W = 768, 12 heads, 12 layers, E = 768.  Parameter names follow openai/CLIP's state dict
(``model.token_embedding.weight``, ``model.transformer.resblocks.{i}.attn.in_proj_weight``, ...) so a real
checkpoint can be loaded with ``load_state_dict``.  The gradient flows THROUGH the frozen tower to the keyword embeddings:
the transformer blocks run on the library's kernels (clip_text_hip.py: bf16 GEMMs, causal attention forward and backward,
LayerNorm / QuickGELU forward and backward), ``ln_final`` and ``text_projection`` act on the B end-of-text rows only, in fp32
(row LayerNorm kernel + exact-fp32 matrix-pipe GEMM).  The nn modules below are PARAMETER CONTAINERS under openai/CLIP's names:
they have no arithmetic of their own - there is no stock-op or CPU path (a trainable text tower, which no shipped recipe uses,
raises).
"""
from collections import OrderedDict
from typing import Optional, Union

import torch
from torch import nn

CLIP_TEXT_ARCHS = {"ViT-B/32": dict(width=512, heads=8, layers=12, embed_dim=512),
                   "ViT-L/14": dict(width=768, heads=12, layers=12, embed_dim=768)}
SOT_TOKEN, EOT_TOKEN, CLIP_VOCAB, CONTEXT_LEN = 49406, 49407, 49408, 77


def _no_eager(what: str):
    raise RuntimeError(f"{what} holds the CLIP text tower's parameters only; the arithmetic runs on the HIP kernels through "
                       "ClipModel.encode_keywords (speechclip_plus_amd has no stock-op / CPU path)")


class QuickGELU(nn.Module):
    def forward(self, x: torch.Tensor):
        _no_eager("QuickGELU")


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)

    def forward(self, x: torch.Tensor, attn_mask: torch.Tensor = None):
        _no_eager("ResidualAttentionBlock")


class _TextTransformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x: torch.Tensor, attn_mask: torch.Tensor = None):
        _no_eager("the text transformer")


class _ClipTextCore(nn.Module):
    """Holds the text-side parameters under openai/CLIP's names."""

    def __init__(self, vocab: int, width: int, heads: int, layers: int, embed_dim: int, seed: int = 1234):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.empty(CONTEXT_LEN, width))
        self.transformer = _TextTransformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim))
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():            # CLIP's published init scales
            self.token_embedding.weight.copy_(torch.randn(vocab, width, generator=g) * 0.02)
            self.positional_embedding.copy_(torch.randn(CONTEXT_LEN, width, generator=g) * 0.01)
            self.text_projection.copy_(torch.randn(width, embed_dim, generator=g) * width ** -0.5)


class _EotHeadFn(torch.autograd.Function):
    """``ln_final`` + ``@ text_projection`` on the B end-of-text rows (clip_official.py:270-277), fp32, frozen parameters:
    sc_rowln_f32_fwd / _bwd and the exact-fp32 matrix-pipe GEMM; gradient w.r.t. the rows only."""

    @staticmethod
    def forward(ctx, rows, gamma, beta, eps, proj):
        from . import ops
        x = rows.detach().float().contiguous()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y, xhat, rstd = ops.rowln_fwd(x, None, 0, g, b, eps)
        P = proj.detach().float().contiguous()                       # [W, E]: K-major for out = y @ P
        ctx.save_for_backward(xhat, rstd, g, P)
        return ops.sgemm_mfma(y, P, b_kmajor=True)

    @staticmethod
    def backward(ctx, d_out):
        from . import ops
        xhat, rstd, g, P = ctx.saved_tensors
        dy = ops.sgemm_mfma(d_out.float().contiguous(), P)           # dy[b, w] = sum_e d_out[b, e] P[w, e]
        scratch = torch.zeros(2, g.numel(), device=g.device, dtype=torch.float32)      # the frozen affine's gradient: discarded
        return ops.rowln_bwd(dy, xhat, g, rstd, scratch[0], scratch[1]), None, None, None, None


class ClipModel(nn.Module):
    """Text half of avssl/module/clip_official.py:ClipModel (image tower: out of scope, embeddings are inputs)."""

    def __init__(self, name: str = "ViT-B/32", device: str = "cuda", image_encoder_trainable: bool = False,
                 text_encoder_trainable: bool = False, reduce_subword_embbedding: Optional[Union[str, torch.Tensor]] = None,
                 layers: Optional[int] = None, seed: int = 1234, **kwargs):
        super().__init__()
        assert name in CLIP_TEXT_ARCHS, name
        a = dict(CLIP_TEXT_ARCHS[name])
        if layers is not None:
            a["layers"] = layers
        self.name, self.device = name, device
        self.text_encoder_trainable = text_encoder_trainable
        self.model = _ClipTextCore(CLIP_VOCAB, a["width"], a["heads"], a["layers"], a["embed_dim"], seed)
        self.out_dim = a["width"]
        # keyword counts that pointed behind the keyword tensor and were clamped by encode_keywords (device counter, read on request)
        self.register_buffer("eot_clamped", torch.zeros((), dtype=torch.int64), persistent=False)
        self.selected_text_emb_ids = None
        if reduce_subword_embbedding is not None:
            # clip_official.py:63-108: keep only the sub-words seen in the captions.  Accepts the reference's .npy path
            # (columns id, frequency) or a 1-D tensor of original token ids.
            if isinstance(reduce_subword_embbedding, str):
                import numpy as np
                ids = torch.from_numpy(np.load(reduce_subword_embbedding)[:, 0].astype("int64"))
            else:
                ids = reduce_subword_embbedding.long().cpu()
            self.selected_text_emb_ids = ids
            self.model.token_embedding = nn.Embedding.from_pretrained(self.model.token_embedding.weight.detach()[ids])
            self.original2Reduced = {int(o): n for n, o in enumerate(ids.tolist())}
            self.reducedl2Original = {n: int(o) for n, o in enumerate(ids.tolist())}
            self.startOfTxt_reduced = self.original2Reduced[SOT_TOKEN]
            self.endOfTxt_reduced = self.original2Reduced[EOT_TOKEN]
        if not text_encoder_trainable:
            for p in self.model.parameters():
                p.requires_grad = False
        self.to(device)

    def _check_tower(self) -> None:
        core = self.model
        heads = core.transformer.resblocks[0].attn.num_heads
        if self.text_encoder_trainable:
            raise NotImplementedError("text_encoder_trainable: no shipped recipe trains the CLIP text tower; only the frozen tower "
                                      "(forward + input gradient) is built")
        if core.transformer.width != 64 * heads:
            raise NotImplementedError(f"text tower head_dim {core.transformer.width // heads}: the attention kernels are built for 64")

    def _tower_weights(self, dev):
        """bf16 (+ transposed) copies of the frozen tower's weights, (re)converted when a checkpoint is loaded or the module moves"""
        from .clip_text_hip import prepare_weights
        core = self.model
        key = (dev, tuple((p.data_ptr(), p._version) for p in core.transformer.parameters()))
        if getattr(self, "_hip_key", None) != key:
            self._hip_weights, self._hip_key = prepare_weights(core.transformer, dev), key
        return self._hip_weights

    def _transformer(self, x: torch.Tensor) -> torch.Tensor:
        """(B, T <= 77, W) -> (B, T, W) on the library's bf16 kernels, forward and input gradient (clip_text_hip.TextTowerFn).  The
        tower is frozen in every shipped recipe (clip_official.py:113-134) and both published towers have head_dim 64."""
        if not x.is_cuda:
            raise RuntimeError("the CLIP text tower runs on the HIP kernels: device tensors only")
        self._check_tower()
        from .clip_text_hip import TextTowerFn
        return TextTowerFn.apply(x, self._tower_weights(x.device), self.model.transformer.resblocks[0].attn.num_heads)

    def update_device(self, device):
        self.device = device

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.device = self.model.token_embedding.weight.device
        return self

    def _prompt_constants(self, dev):
        """(tok [3, W] = embeddings of SOT, EOT and token 0; pos [77, W]) as fp32 device tensors, per parameter version"""
        emb, posp = self.model.token_embedding.weight, self.model.positional_embedding
        key = (str(dev), emb.data_ptr(), emb._version, posp.data_ptr(), posp._version)
        if getattr(self, "_prompt_key", None) != key:
            sot, eot = (SOT_TOKEN, EOT_TOKEN) if self.selected_text_emb_ids is None else (self.startOfTxt_reduced, self.endOfTxt_reduced)
            tok = emb.detach()[torch.tensor([sot, eot, 0], device=emb.device)].to(device=dev, dtype=torch.float32).contiguous()
            self._prompt_const = (tok, posp.detach().to(device=dev, dtype=torch.float32).contiguous())
            self._prompt_key = key
        return self._prompt_const

    def encode_keywords(self, keywords: torch.Tensor, keyword_num: Union[int, torch.Tensor]) -> torch.Tensor:
        """clip_official.py:222-279: [SOT, kw_1 .. kw_n, EOT, 0 ...] -> text transformer -> EOT row @ text_projection."""
        if not isinstance(keywords, torch.Tensor):
            raise TypeError(f"Unknown keywords type {type(keywords)}")
        bsz = keywords.size(0)
        dev = keywords.device
        ln = self.model.ln_final
        if isinstance(keyword_num, torch.Tensor):
            # per-sample keyword counts (the CIF branches): prompt assembly, the tower's padding and the end-of-text gather are
            # single launches around the tower (clip_text_hip.KeywordTowerFn, csrc/prompt.hip).  The tower is causal and only the
            # end-of-text row is read: positions behind the LAST end-of-text token of the batch cannot influence any output, and
            # keywords.shape[1] is the batch's largest keyword count (sized on the host), so the prefix [SOT, kw_1 .. kw_N, EOT] =
            # N + 2 positions is all the transformer sees.  A count that points behind the keyword tensor (a caller error the
            # reference reports as a shape mismatch on the host) is clamped into the prefix and counted in ``eot_clamped`` instead of
            # indexing out of bounds on the device.
            if not keywords.is_cuda:
                raise RuntimeError("the CLIP text tower runs on the HIP kernels: device tensors only")
            self._check_tower()
            from .clip_text_hip import KeywordTowerFn
            n_pos = min(CONTEXT_LEN, keywords.shape[1] + 2)
            tok, pos = self._prompt_constants(dev)
            heads = self.model.transformer.resblocks[0].attn.num_heads
            rows = KeywordTowerFn.apply(keywords, keyword_num.to(dev), tok, pos, self._tower_weights(dev), heads, n_pos, self.eot_clamped)
            return _EotHeadFn.apply(rows, ln.weight, ln.bias, ln.eps, self.model.text_projection)
        sot, eot = (SOT_TOKEN, EOT_TOKEN) if self.selected_text_emb_ids is None else (self.startOfTxt_reduced, self.endOfTxt_reduced)
        text = torch.zeros([bsz, CONTEXT_LEN], device=dev, dtype=torch.long)
        text[:, 0] = sot
        text[:, keyword_num + 1] = eot
        x = self.model.token_embedding(text)
        x = torch.cat([x[:, :1], keywords, x[:, 1 + keyword_num:]], dim=1)
        x = x + self.model.positional_embedding
        n_pos = min(CONTEXT_LEN, int(keyword_num) + 2)          # the causal prefix up to the end-of-text position
        x = self._transformer(x[:, :n_pos])
        return _EotHeadFn.apply(x[:, 1 + keyword_num], ln.weight, ln.bias, ln.eps, self.model.text_projection)
