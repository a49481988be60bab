"""Trainable HuBERT transformer layers (scope row f2): ``audio_encoder.trainable`` with ``unfreeze_layers`` /
``reinit_layers`` (avssl/module/speech_encoder_plus.py:416-446: the listed layers train, everything else - conv
extractor, projection, pos_conv, the other layers - stays frozen).

Forward of an unfrozen layer keeps what its backward needs (row-major q | k | v, the attention output and its LSE, both
LayerNorm inputs, the FFN pre-activation and activation); the backward is a manual chain on the library's kernels, entered
from the head's backward (head_tail.ParallelHeadFn) with the gradient of the weighted-sum output:

    d hidden[n] = softmax(w)[n] * dX[:, 1 : T + 1]                     (the weighted sum reads every hidden state)
    per layer (post-LN):  LayerNorm' -> fc2 wgrad / dgrad -> GELU' -> fc1 wgrad / dgrad (+ residual) -> LayerNorm'
                          -> out_proj wgrad / dgrad -> attention backward -> qkv wgrad / dgrad (+ residual)

Weight gradients are written in fp32 straight into the parameters' ``.grad`` (views of the optimiser's flat buffer), so the
single flat all-reduce of parallel.GradAllReduce covers them.  Dropout / layerdrop inside the unfrozen layers
(fairseq trains them with p = 0.1) is not applied: the path is deterministic, as the rest of this build.
"""
from typing import Dict, List

import torch
from torch import nn

from . import ops

_PARAMS = ["self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight", "self_attn.k_proj.bias",
           "self_attn.v_proj.weight", "self_attn.v_proj.bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
           "self_attn_layer_norm.weight", "self_attn_layer_norm.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
           "final_layer_norm.weight", "final_layer_norm.bias"]


def _key(i: int, name: str) -> str:
    return f"layers_{i}_" + name.replace(".", "_")


def _gacc(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


class TrainableLayers(nn.Module):
    def __init__(self, arch, sd: Dict[str, torch.Tensor], layer_ids: List[int], device, reinit: bool = False, seed: int = 0):
        super().__init__()
        if arch.layer_norm_first:
            raise NotImplementedError("trainable layers are built for the post-LN (HuBERT-base) layer order only")
        self.arch = arch
        self.ids = sorted(int(i) for i in layer_ids)
        assert all(0 <= i < arch.layers for i in self.ids), layer_ids
        self.p = nn.ParameterDict()
        self.fairseq_names = {}                      # parameter key -> "encoder.layers.{i}.{name}" (checkpoint mapping)
        g = torch.Generator(device="cpu").manual_seed(seed)
        for i in self.ids:
            for name in _PARAMS:
                t = sd[f"encoder.layers.{i}.{name}"].detach().float().clone()
                if reinit:                           # fairseq init_bert_params: N(0, 0.02) weights, zero biases, unit LayerNorm
                    if "layer_norm" in name:
                        t = torch.ones_like(t) if name.endswith("weight") else torch.zeros_like(t)
                    elif name.endswith("weight"):
                        t = torch.randn(t.shape, generator=g) * 0.02
                    else:
                        t = torch.zeros_like(t)
                self.p[_key(i, name)] = nn.Parameter(t.to(device))
                self.fairseq_names[_key(i, name)] = f"encoder.layers.{i}.{name}"
        self._copies, self._versions = {}, None
        self.grad_ready_hook = None              # callable(layer_id): every gradient of that layer has been enqueued (train.py)

    def layer_parameters(self, i: int) -> List[nn.Parameter]:
        return [self.p[_key(i, name)] for name in _PARAMS]

    def get(self, i: int, name: str) -> nn.Parameter:
        return self.p[_key(i, name)]

    def refresh(self) -> None:
        """bf16 working copies (+ transposed ones for the dgrad products) of the fp32 masters, rebuilt after an optimiser step."""
        ver = tuple((p.data_ptr(), p._version) for p in self.p.values())
        if ver == self._versions:
            return
        bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
        for i in self.ids:
            c = {}
            qkv = torch.cat([self.get(i, f"self_attn.{n}.weight").detach() for n in ("q_proj", "k_proj", "v_proj")], 0)
            c["qkv_w"], c["qkv_wT"] = bf(qkv), bf(qkv.t())
            c["qkv_b"] = torch.cat([self.get(i, f"self_attn.{n}.bias").detach() for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous()
            for short, name in (("o", "self_attn.out_proj"), ("fc1", "fc1"), ("fc2", "fc2")):
                wt = self.get(i, name + ".weight").detach()
                c[short + "_w"], c[short + "_wT"], c[short + "_b"] = bf(wt), bf(wt.t()), self.get(i, name + ".bias").detach()
            c["ln1_g"], c["ln1_b"] = self.get(i, "self_attn_layer_norm.weight").detach(), self.get(i, "self_attn_layer_norm.bias").detach()
            c["ln2_g"], c["ln2_b"] = self.get(i, "final_layer_norm.weight").detach(), self.get(i, "final_layer_norm.bias").detach()
            self._copies[i] = c
        self._versions = ver

    # -------------------------------------------------------------------------------------------------- forward
    def layer_forward(self, i: int, x: torch.Tensor, out: torch.Tensor, pl, save: bool) -> None:
        """hidden[i] -> hidden[i + 1] (post-LN).  ``save``: keep the activations the backward needs in ``pl.train[i]``."""
        a, c = self.arch, self._copies[i]
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        D, F, H = a.embed_dim, a.ffn_dim, a.heads
        dev = x.device
        if save:
            s = pl.train.get(i)
            if s is None:
                z = lambda *sh, dtype=torch.bfloat16: torch.zeros(*sh, device=dev, dtype=dtype)
                s = pl.train[i] = dict(qkv=z(M, 3 * D), ctx=z(M, D), lse2=z(B, H, R, dtype=torch.float32), pre1=z(M, D), x1=z(M, D),
                                       u=z(M, F), f=z(M, F), pre2=z(M, D))
        else:
            s = dict(qkv=torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16), ctx=pl.ctx, lse2=None, pre1=pl.pre, x1=pl.x1,
                     u=pl.ffn, f=pl.ffn, pre2=pl.pre)
        ops.linear_bf16(x, c["qkv_w"], c["qkv_b"], out=s["qkv"], alg_rows=B * T)
        ops.head_transpose(s["qkv"][:, 2 * D:], B, R, H, out=pl.vt)
        ops.attn_fwd(s["qkv"][:, : 2 * D], pl.vt, pl.valid, s["ctx"], B, R, H, D, (D // H) ** -0.5, lse2=s["lse2"],
                     alg_flops=4.0 * B * T * T * D)
        ops.linear_bf16(s["ctx"], c["o_w"], c["o_b"], out=s["pre1"], residual=x, alg_rows=B * T)
        ops.layernorm_bf16(s["pre1"], c["ln1_g"], c["ln1_b"], out=s["x1"])
        if save:
            ops.linear_bf16(s["x1"], c["fc1_w"], c["fc1_b"], out=s["u"], alg_rows=B * T)
            ops.act_bf16(s["u"], 1, out=s["f"])
        else:
            ops.linear_bf16(s["x1"], c["fc1_w"], c["fc1_b"], out=s["f"], act=1, alg_rows=B * T)
        ops.linear_bf16(s["f"], c["fc2_w"], c["fc2_b"], out=s["pre2"], residual=s["x1"], alg_rows=B * T)
        ops.layernorm_bf16(s["pre2"], c["ln2_g"], c["ln2_b"], out=out)

    # -------------------------------------------------------------------------------------------------- backward
    def backward(self, pl, dX: torch.Tensor, w_soft: torch.Tensor) -> None:
        """dX [B, R, D] fp32: gradient of the weighted-sum output (row 0 = CLS slot, rows 1..T = frames 0..T-1)."""
        a = self.arch
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        D, F, H = a.embed_dim, a.ffn_dim, a.heads
        lo = self.ids[0]
        assert self.ids == list(range(lo, lo + len(self.ids))) or True     # gaps are fine: frozen layers in between still propagate
        dfeat = torch.zeros(B, R, D, device=dX.device, dtype=torch.float32)
        dfeat[:, :T] = dX[:, 1: T + 1]
        dfeat = dfeat.view(M, D)
        d_out = None
        for i in range(a.layers - 1, lo - 1, -1):
            g = (dfeat * w_soft[i + 1]).to(torch.bfloat16)                 # d hidden[i + 1] from the weighted sum
            d_out = g if d_out is None else d_out + g
            if i in self._copies and i in pl.train:
                d_out = self._layer_backward(i, pl, d_out, need_dx=i > lo)
                if self.grad_ready_hook is not None:
                    self.grad_ready_hook(i)
            else:
                raise NotImplementedError("a frozen layer above an unfrozen one needs the input-gradient-only backward; "
                                          "unfreeze a contiguous top block (the reference recipes unfreeze / reinit top layers)")

    def _layer_backward(self, i: int, pl, d_out: torch.Tensor, need_dx: bool):
        a, c, s = self.arch, self._copies[i], pl.train[i]
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        D, F, H = a.embed_dim, a.ffn_dim, a.heads
        x = pl.hidden[i]
        P = lambda name: _gacc(self.get(i, name))
        # LN2 -> fc2 -> GELU -> fc1 (+ residual)
        dpre2, dg, db = ops.layernorm_bwd(s["pre2"], d_out, c["ln2_g"], 1e-5, want_param_grads=True)
        P("final_layer_norm.weight").add_(dg)
        P("final_layer_norm.bias").add_(db)
        ops.wgrad_bf16(dpre2, s["f"], P("fc2.weight"), P("fc2.bias"))
        df = ops.linear_bf16(dpre2, c["fc2_wT"])
        du = ops.act_bf16(s["u"], 1, df=df, out=df)
        ops.wgrad_bf16(du, s["x1"], P("fc1.weight"), P("fc1.bias"))
        dx1 = ops.linear_bf16(du, c["fc1_wT"], residual=dpre2)
        # LN1 -> out_proj -> attention -> qkv (+ residual)
        dpre1, dg, db = ops.layernorm_bwd(s["pre1"], dx1, c["ln1_g"], 1e-5, want_param_grads=True)
        P("self_attn_layer_norm.weight").add_(dg)
        P("self_attn_layer_norm.bias").add_(db)
        ops.wgrad_bf16(dpre1, s["ctx"], P("self_attn.out_proj.weight"), P("self_attn.out_proj.bias"))
        dctx = ops.linear_bf16(dpre1, c["o_wT"])
        dqkv = torch.empty(M, 3 * D, device=x.device, dtype=torch.bfloat16)
        qkv = s["qkv"]
        ops.attn_bwd(qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:], s["ctx"], dctx, s["lse2"], pl.valid, dqkv[:, :D], dqkv[:, D: 2 * D],
                     dqkv[:, 2 * D:], B, R, H, (D // H) ** -0.5, q_rows=T)
        gW = torch.empty(3 * D, D, device=x.device, dtype=torch.float32)
        gb = torch.empty(3 * D, device=x.device, dtype=torch.float32)
        ops.wgrad_bf16(dqkv, x, gW, gb, beta=0.0)
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            P(f"self_attn.{n}.weight").add_(gW[j * D: (j + 1) * D])
            P(f"self_attn.{n}.bias").add_(gb[j * D: (j + 1) * D])
        if not need_dx:
            return None
        return ops.linear_bf16(dqkv, c["qkv_wT"], residual=dpre1)
