"""Trainable HuBERT transformer layers (scope row f2): ``audio_encoder.trainable`` with ``unfreeze_layers`` /
``reinit_layers`` (avssl/module/speech_encoder_plus.py:416-446: the listed layers train, everything else - conv
extractor, projection, pos_conv, the other layers - stays frozen).

Forward of an unfrozen layer keeps what its backward needs (row-major q | k | v, the attention output and its LSE, both
LayerNorm inputs, the FFN pre-activation and activation); the backward is a manual chain on the library's kernels, entered
from the head's backward (head_tail.ParallelHeadFn) with the gradient of the weighted-sum output:

    d hidden[n] = softmax(w)[n] * dX[:, 1 : T + 1]                     (the weighted sum reads every hidden state)
    per layer (post-LN):  LayerNorm' -> fc2 wgrad / dgrad -> GELU' -> fc1 wgrad / dgrad (+ residual) -> LayerNorm'
                          -> out_proj wgrad / dgrad -> attention backward -> qkv wgrad / dgrad (+ residual)
    pre-LN (large) order: the two LayerNorms sit in front of the attention and of the FFN; their backward is fused with the
                          residual add.  Frozen layers above the lowest unfrozen one run the same chain without weight gradients.

Weight gradients are written in fp32 straight into the parameters' ``.grad`` (views of the optimiser's flat buffer), so the
single flat all-reduce of parallel.GradAllReduce covers them.  Train-mode dropout (fairseq: p = 0.1 base, 0 large) runs in
these layers too: the forward applies the stateless hash masks in the kernels (attention probabilities, out_proj and fc2
epilogues) and the backward regenerates them from the same seeds - ops.dropout_bf16 on the two branch gradients, the
DROP variants of the attention backward kernels.  No shipped recipe unfreezes HuBERT layers.
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
        self.arch = arch
        self.ids = sorted(int(i) for i in layer_ids)
        assert all(0 <= i < arch.layers for i in self.ids), layer_ids
        # frozen layers ABOVE the lowest unfrozen one still carry the gradient down: they keep their activations and run the
        # input-gradient half of the backward (no weight gradients)
        self.pass_ids = [i for i in range(self.ids[0], arch.layers) if i not in self.ids]
        self._frozen = {i: {name: sd[f"encoder.layers.{i}.{name}"].detach().float().to(device) for name in _PARAMS}
                        for i in self.pass_ids}
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
        self.frontend = None                     # hubert_frontend_train.TrainableFrontend when the whole encoder trains
        self.grad_ready_hook = None              # callable(layer_id): every gradient of that layer has been enqueued (train.py)

    def invalidate(self) -> None:
        self._versions = None

    def layer_parameters(self, i: int) -> List[nn.Parameter]:
        return [self.p[_key(i, name)] for name in _PARAMS]

    def get(self, i: int, name: str) -> torch.Tensor:
        return self.p[_key(i, name)] if i in self.ids else self._frozen[i][name]

    def refresh(self) -> None:
        """bf16 working copies (+ transposed ones for the dgrad products) of the fp32 masters, rebuilt after an optimiser step.
        The fused Adam kernel writes the masters through a raw pointer (no ``_version`` bump), so the cache key also carries
        optim.param_generation(); ``invalidate()`` forces a rebuild after any other out-of-band write."""
        from .optim import param_generation
        ver = (param_generation(),) + tuple((p.data_ptr(), p._version) for p in self.p.values())
        if ver == self._versions:
            return
        bf = ops.bf16_copy          # one copy kernel whatever the view's strides
        # Trainable parameters managed by optim.FlatAdam are contiguous views of ONE flat fp32 buffer: cast the span they cover in a
        # single launch and hand out views of that bf16 mirror, instead of one cast launch per weight matrix (113 per step)
        mirror, flat32, lo = None, None, 0
        ps = [q for q in self.p.values()]
        if ps and all(q.is_contiguous() and q.dtype == torch.float32 and
                      q.data.untyped_storage().data_ptr() == ps[0].data.untyped_storage().data_ptr() for q in ps):
            lo = min(q.storage_offset() for q in ps)
            hi = max(q.storage_offset() + q.numel() for q in ps)
            if hi - lo <= 2 * sum(q.numel() for q in ps):          # a compact span (other modules' parameters may sit in between)
                flat = torch.empty(0, dtype=torch.float32, device=ps[0].device).set_(ps[0].data.untyped_storage(), lo, (hi - lo,))
                mirror, flat32 = flat.to(torch.bfloat16), flat

        def bfp(i, name):                                          # bf16 working copy of a layer parameter
            t = self.get(i, name)
            if mirror is not None and i in self.ids:
                off = t.storage_offset() - lo
                if off % 8 == 0:                                   # the kernels read operands in 16-byte pieces
                    return mirror[off: off + t.numel()].view(t.shape)
            return bf(t)

        # The trainable layers' matrices sit at ONE stride in the bf16 mirror (every layer registers the same parameters in the same
        # order): the fused QKV weight of all layers is gathered with three strided copies and every kind of transposed working copy
        # (QKV, out_proj, fc1, fc2) is ONE batched transpose launch over the layers - 7 launches per step where the per-layer
        # cat + .t().contiguous() took 60.
        batched = {}
        n = len(self.ids)
        if mirror is not None and n >= 2:
            offs = {nm: [self.get(i, nm).storage_offset() - lo for i in self.ids] for nm in _PARAMS if nm.endswith("weight") and "layer_norm" not in nm}
            stride = offs["fc1.weight"][1] - offs["fc1.weight"][0]
            if stride > 0 and stride % 8 == 0 and all(o[j] == o[0] + j * stride and o[0] % 8 == 0 for o in offs.values() for j in range(n)):
                D, F = self.arch.embed_dim, self.arch.ffn_dim
                lay = lambda nm, r, k: torch.as_strided(mirror, (n, r, k), (stride, k, 1), offs[nm][0])
                qkv = torch.empty(n, 3 * D, D, device=mirror.device, dtype=torch.bfloat16)
                for k3, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                    qkv[:, k3 * D: (k3 + 1) * D].copy_(lay(f"self_attn.{nm}.weight", D, D))
                batched["qkv"] = (qkv, ops.transpose_batched_bf16(qkv[0], 3 * D * D, n))
                for short, nm, r, k in (("o", "self_attn.out_proj.weight", D, D), ("fc1", "fc1.weight", F, D), ("fc2", "fc2.weight", D, F)):
                    w = lay(nm, r, k)
                    batched[short] = (w, ops.transpose_batched_bf16(w[0], stride, n))
                # the fused QKV bias (fp32, the GEMM's bias operand) of all layers: three strided gathers instead of a cat per layer
                boffs = [[self.get(i, f"self_attn.{nm}.bias").storage_offset() - lo for i in self.ids] for nm in ("q_proj", "k_proj", "v_proj")]
                if all(o[j] == o[0] + j * stride for o in boffs for j in range(n)):
                    qkv_b = torch.empty(n, 3, D, device=mirror.device, dtype=torch.float32)
                    for k3, o in enumerate(boffs):
                        qkv_b[:, k3].copy_(torch.as_strided(flat32, (n, D), (stride, 1), o[0]))
                    batched["qkv_b"] = qkv_b.view(n, 3 * D)

        for i in self.ids + [j for j in self.pass_ids if j not in self._copies]:
            c = {}
            jb = self.ids.index(i) if (batched and i in self.ids) else None
            if jb is not None:
                c["qkv_w"], c["qkv_wT"] = batched["qkv"][0][jb], batched["qkv"][1][jb]
            else:
                qkv_b16 = torch.cat([bfp(i, f"self_attn.{n_}.weight") for n_ in ("q_proj", "k_proj", "v_proj")], 0)
                c["qkv_w"], c["qkv_wT"] = qkv_b16, qkv_b16.t().contiguous()
            if jb is not None and "qkv_b" in batched:
                c["qkv_b"] = batched["qkv_b"][jb]
            else:
                c["qkv_b"] = torch.cat([self.get(i, f"self_attn.{n_}.bias").detach() for n_ in ("q_proj", "k_proj", "v_proj")], 0).contiguous()
            for short, name in (("o", "self_attn.out_proj"), ("fc1", "fc1"), ("fc2", "fc2")):
                if jb is not None:
                    c[short + "_w"], c[short + "_wT"] = batched[short][0][jb], batched[short][1][jb]
                else:
                    wb = bfp(i, name + ".weight")
                    c[short + "_w"], c[short + "_wT"] = wb, wb.t().contiguous()
                c[short + "_b"] = self.get(i, name + ".bias").detach()
            c["ln1_g"], c["ln1_b"] = self.get(i, "self_attn_layer_norm.weight").detach(), self.get(i, "self_attn_layer_norm.bias").detach()
            c["ln2_g"], c["ln2_b"] = self.get(i, "final_layer_norm.weight").detach(), self.get(i, "final_layer_norm.bias").detach()
            self._copies[i] = c
        self._versions = ver

    # -------------------------------------------------------------------------------------------------- forward
    def layer_forward(self, i: int, x: torch.Tensor, out: torch.Tensor, pl, save: bool, drops=None) -> None:
        """hidden[i] -> hidden[i + 1].  ``save``: keep the activations the backward needs in ``pl.train[i]``.
        ``drops`` = (p_residual, p_attention, seed_attention, seed_out_proj, seed_fc2) in train mode (fairseq's dropout1 /
        dropout3 in the GEMM epilogues, attention dropout in the attention kernel; the backward regenerates the same hash masks).
        post-LN (base):  pre1 = x + attn(x) ; x1 = LN1(pre1) ; pre2 = x1 + ffn(x1) ; out = LN2(pre2)
        pre-LN (large):  x1 = LN1(x) ; pre1 = x + attn(x1) ; x2 = LN2(pre1) ; out = pre1 + ffn(x2)"""
        a, c = self.arch, self._copies[i]
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        D, F, H = a.embed_dim, a.ffn_dim, a.heads
        dev = x.device
        pre_ln = a.layer_norm_first
        if save:
            s = pl.train.get(i)
            if s is None:
                z = lambda *sh, dtype=torch.bfloat16: torch.zeros(*sh, device=dev, dtype=dtype)
                s = pl.train[i] = dict(qkv=z(M, 3 * D), ctx=z(M, D), lse2=z(B, H, R, dtype=torch.float32), pre1=z(M, D), x1=z(M, D),
                                       u=z(M, F), f=z(M, F), pre2=z(M, D))       # pre-LN: x1 = LN1(x), pre2 = LN2(pre1)
        else:
            s = dict(qkv=torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16), ctx=pl.ctx, lse2=None,
                     pre1=torch.empty(M, D, device=dev, dtype=torch.bfloat16) if pre_ln else pl.pre, x1=pl.x1, u=pl.ffn, f=pl.ffn,
                     pre2=pl.pre)
        p_res, p_att, sd_a, sd_o, sd_f = drops if drops is not None else (0.0, 0.0, 0, 0, 0)
        s["drops"] = (p_res, p_att, sd_a, sd_o, sd_f)
        attn_in = x
        if pre_ln:
            ops.layernorm_bf16(x, c["ln1_g"], c["ln1_b"], out=s["x1"])
            attn_in = s["x1"]
        ops.linear_bf16(attn_in, c["qkv_w"], c["qkv_b"], out=s["qkv"], alg_rows=B * T)
        ops.head_transpose(s["qkv"][:, 2 * D:], B, R, H, out=pl.vt)
        ops.attn_fwd(s["qkv"][:, : 2 * D], pl.vt, pl.valid, s["ctx"], B, R, H, D, (D // H) ** -0.5, lse2=s["lse2"],
                     alg_flops=4.0 * B * T * T * D, drop_p=p_att, drop_seed=sd_a)
        ops.linear_bf16(s["ctx"], c["o_w"], c["o_b"], out=s["pre1"], residual=x, alg_rows=B * T, drop_p=p_res, drop_seed=sd_o)
        ffn_in = s["pre2"] if pre_ln else s["x1"]
        if pre_ln:
            ops.layernorm_bf16(s["pre1"], c["ln2_g"], c["ln2_b"], out=s["pre2"])
        else:
            ops.layernorm_bf16(s["pre1"], c["ln1_g"], c["ln1_b"], out=s["x1"])
        if save:      # one launch: u = fc1 pre-activation (kept for the backward) and f = gelu(u) (sc_gemm_args.aux_mode 1)
            ops.linear_bf16(ffn_in, c["fc1_w"], c["fc1_b"], out=s["f"], act=1, aux=s["u"], aux_mode=1, alg_rows=B * T)
        else:
            ops.linear_bf16(ffn_in, c["fc1_w"], c["fc1_b"], out=s["f"], act=1, alg_rows=B * T)
        if pre_ln:
            ops.linear_bf16(s["f"], c["fc2_w"], c["fc2_b"], out=out, residual=s["pre1"], alg_rows=B * T, drop_p=p_res, drop_seed=sd_f)
        else:
            ops.linear_bf16(s["f"], c["fc2_w"], c["fc2_b"], out=s["pre2"], residual=s["x1"], alg_rows=B * T, drop_p=p_res,
                            drop_seed=sd_f)
            ops.layernorm_bf16(s["pre2"], c["ln2_g"], c["ln2_b"], out=out)

    # -------------------------------------------------------------------------------------------------- backward
    def backward(self, pl, dX: torch.Tensor, w_soft: torch.Tensor, normalize: bool = False) -> None:
        """dX [B, R, D] fp32: gradient of the weighted-sum output (row 0 = CLS slot, rows 1..T = frames 0..T-1).
        ``normalize``: the weighted sum reads layer_norm(hidden[n]) without affine (weighted_sum.py:41-42)."""
        a = self.arch
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        D = a.embed_dim
        lo = self.ids[0]
        dX = dX.float().contiguous()
        ones = torch.ones(D, device=dX.device, dtype=torch.float32) if normalize else None

        def add_share(n: int, d_prev):
            # d hidden[n] = what came down from the layer above + this state's share of the weighted sum's gradient (frames of dX at
            # row offset 1), one pass (sc_wsum_share_bf16); with ``normalize`` the share passes the state's non-affine LayerNorm first
            # and the sum rides on that kernel's residual input
            if not normalize:
                return ops.wsum_share(dX, w_soft[n: n + 1], d_prev, B, R, T)
            g = ops.wsum_share(dX, w_soft[n: n + 1], None, B, R, T)
            return ops.layernorm_bwd(pl.hidden[n], g, ones, 1e-5, dres=d_prev)

        d_out = None
        for i in range(a.layers - 1, lo - 1, -1):
            d_out = add_share(i + 1, d_out)
            assert i in pl.train, "unfrozen layers ran without saved activations (forward in eval / no_grad mode?)"
            d_out = self._layer_backward(i, pl, d_out, need_dx=i > lo or self.frontend is not None, train=i in self.ids)
            if self.grad_ready_hook is not None and i in self.ids:
                self.grad_ready_hook(i)
        if self.frontend is not None:                      # lo == 0: on into the front end with d hidden[0]
            self.frontend.backward_frontend(pl, add_share(0, d_out))

    def _layer_backward(self, i: int, pl, d_out: torch.Tensor, need_dx: bool, train: bool = True):
        a, c, s = self.arch, self._copies[i], pl.train[i]
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        D, F, H = a.embed_dim, a.ffn_dim, a.heads
        x = pl.hidden[i]
        pre_ln = a.layer_norm_first
        P = lambda name: _gacc(self.get(i, name))

        def ln_bwd(xin, dy, which, dres=None, branch=None):                # LayerNorm backward (+ its parameter gradients)
            # ``branch`` = (linear layer, seed): the output is next read by that layer's residual branch - the same pass also writes
            # its dropped copy (the branch's F.dropout mask, regenerated) and adds the copy's column sums into the layer's bias gradient
            # (ops.layernorm_bwd drop / sum_acc): returns (dx, dx_dropped)
            gname = "self_attn_layer_norm" if which == 1 else "final_layer_norm"
            kw = {}
            if train:
                kw["acc"] = (P(gname + ".weight"), P(gname + ".bias"))
            if branch is None:
                return ops.layernorm_bwd(xin, dy, c[f"ln{which}_g"], 1e-5, dres=dres, **kw)
            name, seed = branch
            if train:
                kw["sum_acc"] = P(name + ".bias")
            if p_res > 0.0:
                return ops.layernorm_bwd(xin, dy, c[f"ln{which}_g"], 1e-5, dres=dres, drop=(p_res, seed), **kw)
            dx_ = ops.layernorm_bwd(xin, dy, c[f"ln{which}_g"], 1e-5, dres=dres, **kw)
            return dx_, dx_

        def wgrad(dy, xin, name, bias=True):
            if train:
                ops.wgrad_bf16(dy, xin, P(name + ".weight"), P(name + ".bias") if bias else None)

        p_res, p_att, sd_a, sd_o, sd_f = s.get("drops", (0.0, 0.0, 0, 0, 0))
        # ---- FFN half
        # gradient of the dropped branch = the same mask on the sum's gradient; the residual path keeps the un-masked one
        if pre_ln:                                                          # out = pre1 + drop(fc2(gelu(fc1(LN2(pre1)))))
            dffn_out, ffn_in = d_out, s["pre2"]
            dfc2 = ops.dropout_bf16(dffn_out, p_res, sd_f) if p_res > 0.0 else dffn_out
            wgrad(dfc2, s["f"], "fc2")
        else:                                                               # out = LN2(pre2), pre2 = x1 + drop(fc2(gelu(fc1(x1))))
            (dffn_out, dfc2), ffn_in = ln_bwd(s["pre2"], d_out, 2, branch=("fc2", sd_f)), s["x1"]
            wgrad(dfc2, s["f"], "fc2", bias=False)
        du = ops.linear_bf16(dfc2, c["fc2_wT"], act=1, aux=s["u"], aux_mode=2)          # (dfc2 W2) * gelu'(u) in the GEMM's epilogue
        wgrad(du, ffn_in, "fc1")
        if pre_ln:
            dx2n = ops.linear_bf16(du, c["fc1_wT"])
            dpre1, dop = ln_bwd(s["pre1"], dx2n, 2, dres=d_out, branch=("self_attn.out_proj", sd_o))      # through LN2 + the residual
        else:
            dx1 = ops.linear_bf16(du, c["fc1_wT"], residual=dffn_out)
            dpre1, dop = ln_bwd(s["pre1"], dx1, 1, branch=("self_attn.out_proj", sd_o))
        # ---- attention half:  pre1 = x + out_proj(attn(qkv(attn_in)))
        wgrad(dop, s["ctx"], "self_attn.out_proj", bias=False)
        dctx = ops.linear_bf16(dop, c["o_wT"])
        dqkv = torch.empty(M, 3 * D, device=x.device, dtype=torch.bfloat16)
        qkv = s["qkv"]
        ops.attn_bwd(qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:], s["ctx"], dctx, s["lse2"], pl.valid, dqkv[:, :D], dqkv[:, D: 2 * D],
                     dqkv[:, 2 * D:], B, R, H, (D // H) ** -0.5, q_rows=T, drop_p=p_att, drop_seed=sd_a)
        attn_in = s["x1"] if pre_ln else x
        if train:
            gb = torch.empty(3 * D, device=x.device, dtype=torch.float32)
            # the fused QKV product's slice reduction adds each D-row block straight into its projection's gradient
            ops.wgrad_bf16(dqkv, attn_in, [P(f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], None, beta=1.0)
            ops.colsum_bf16(dqkv, gb)
            tb = [P(f"self_attn.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")]
            step = (tb[1].data_ptr() - tb[0].data_ptr()) // 4
            if step > 0 and tb[2].data_ptr() == tb[0].data_ptr() + 8 * step and all(
                    t.is_contiguous() and t.untyped_storage().data_ptr() == tb[0].untyped_storage().data_ptr() for t in tb):
                torch.as_strided(tb[0], (3, D), (step, 1)).add_(gb.view(3, D))      # q / k / v biases sit at one stride in the flat buffer
            else:
                for j, t in enumerate(tb):
                    t.add_(gb[j * D: (j + 1) * D])
        if pre_ln:                       # LN1 sits in front of the attention: its parameters need the gradient even at the lowest layer
            if not (need_dx or train):
                return None
            dx1n = ops.linear_bf16(dqkv, c["qkv_wT"])
            return ln_bwd(x, dx1n, 1, dres=dpre1)
        if not need_dx:
            return None
        return ops.linear_bf16(dqkv, c["qkv_wT"], residual=dpre1)
