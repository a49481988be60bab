"""Frozen CLIP text tower as the cascaded branches use it: mirror of ClipModel.encode_keywords
(avssl/module/clip_official.py:222-279) over a restatement of openai/CLIP's text transformer (requirements.txt:4,
unpinned; the `clip` package and its weights are not available offline).

Architecture restated from the published model: token_embedding (V x W), learned positional_embedding (77 x W),
`layers` pre-LN residual blocks [ln_1 -> causal nn.MultiheadAttention(W, heads) ; ln_2 -> Linear(W, 4W) -> QuickGELU
(x * sigmoid(1.702 x)) -> Linear(4W, W)], ln_final, text_projection (W x E).  ViT-B/32: W = 512, 8 heads, 12 layers,
E = 512; ViT-L/14: W = 768, 12 heads, 12 layers, E = 768.  Parameter names follow openai/CLIP's state dict
(``model.token_embedding.weight``, ``model.transformer.resblocks.{i}.attn.in_proj_weight``, ...) so a real
checkpoint can be loaded with ``load_state_dict``.  The gradient flows THROUGH the frozen tower to the keyword embeddings:
on a GPU the transformer blocks run on the library's kernels (clip_text_hip.py: bf16 GEMMs, causal attention forward and
backward, LayerNorm / QuickGELU forward and backward); the stock-op modules below define the parameters, serve a trainable
tower and the CPU.
"""
from collections import OrderedDict
from typing import Optional, Union

import torch
from torch import nn

CLIP_TEXT_ARCHS = {"ViT-B/32": dict(width=512, heads=8, layers=12, embed_dim=512),
                   "ViT-L/14": dict(width=768, heads=12, layers=12, embed_dim=768)}
SOT_TOKEN, EOT_TOKEN, CLIP_VOCAB, CONTEXT_LEN = 49406, 49407, 49408, 77


class QuickGELU(nn.Module):
    def forward(self, x: torch.Tensor):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)

    def forward(self, x: torch.Tensor, attn_mask: torch.Tensor):
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False, attn_mask=attn_mask.to(dtype=x.dtype, device=x.device))[0]
        return x + self.mlp(self.ln_2(x))


class _TextTransformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x: torch.Tensor, attn_mask: torch.Tensor):
        for blk in self.resblocks:
            x = blk(x, attn_mask)
        return x


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
        mask = torch.full((CONTEXT_LEN, CONTEXT_LEN), float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)


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

    def _transformer(self, x: torch.Tensor) -> torch.Tensor:
        """(B, 77, W) -> (B, 77, W).  Frozen tower on a GPU with head_dim 64: the library's bf16 kernels, forward and input
        gradient (clip_text_hip.TextTowerFn); otherwise the stock-op blocks defined above."""
        core = self.model
        heads = core.transformer.resblocks[0].attn.num_heads
        if x.is_cuda and not self.text_encoder_trainable and core.transformer.width == 64 * heads:
            from .clip_text_hip import TextTowerFn, prepare_weights
            key = (x.device, tuple((p.data_ptr(), p._version) for p in core.transformer.parameters()))
            if getattr(self, "_hip_key", None) != key:          # (re)converted when a checkpoint is loaded or the module moves
                self._hip_weights, self._hip_key = prepare_weights(core.transformer, x.device), key
            return TextTowerFn.apply(x, self._hip_weights, heads)
        return core.transformer(x.permute(1, 0, 2), core.attn_mask).permute(1, 0, 2)

    def update_device(self, device):
        self.device = device

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.device = self.model.token_embedding.weight.device
        return self

    def encode_keywords(self, keywords: torch.Tensor, keyword_num: Union[int, torch.Tensor]) -> torch.Tensor:
        """clip_official.py:222-279: [SOT, kw_1 .. kw_n, EOT, 0 ...] -> text transformer -> EOT row @ text_projection."""
        if not isinstance(keywords, torch.Tensor):
            raise TypeError(f"Unknown keywords type {type(keywords)}")
        bsz = keywords.size(0)
        dev = keywords.device
        sot, eot = (SOT_TOKEN, EOT_TOKEN) if self.selected_text_emb_ids is None else (self.startOfTxt_reduced, self.endOfTxt_reduced)
        text = torch.zeros([bsz, CONTEXT_LEN], device=dev, dtype=torch.long)
        text[:, 0] = sot
        if isinstance(keyword_num, torch.Tensor):
            index = keyword_num.to(dev) + 1
            text = text.scatter(1, index.unsqueeze(1), eot)
        else:
            index = None
            text[:, keyword_num + 1] = eot
        x = self.model.token_embedding(text)
        if index is not None:
            pos = torch.arange(CONTEXT_LEN, device=dev).unsqueeze(0)
            is_kw = (pos >= 1) & (pos < index.unsqueeze(1))                        # rows 1 .. n of every sample
            n_kw = keywords.shape[1]
            src = torch.zeros(bsz, CONTEXT_LEN, x.shape[-1], device=dev, dtype=x.dtype)
            src[:, 1: 1 + n_kw] = keywords[:, : CONTEXT_LEN - 1]
            x = torch.where(is_kw.unsqueeze(-1), src, x)
        else:
            x = torch.cat([x[:, :1], keywords, x[:, 1 + keyword_num:]], dim=1)
        x = x + self.model.positional_embedding
        x = self._transformer(x)
        x = self.model.ln_final(x)
        if index is not None:
            return x[torch.arange(bsz, device=dev), index] @ self.model.text_projection
        return x[:, 1 + keyword_num] @ self.model.text_projection
