"""One contrastive train step (frozen-HuBERT recipe of every shipped config):
forward -> packed all-gather -> global-batch InfoNCE -> backward (head + weighted-sum weights [+ temperature])
-> flat gradient all-reduce -> clip + Adam.  Mirrors training_step / training_step_end / configure_optimizers
of avssl/model/kwClip.py:145-193,646-674 without PyTorch-Lightning.

Stream schedule: the gradient all-reduce, the clip + Adam launch and the zeroing of the gradient buffer are enqueued on a
SIDE stream after the backward.  HuBERT is frozen in every shipped recipe, so the next step's encoder forward (12 of the
13 ms) touches no trainable parameter: the main stream joins the side stream only right before the first trainable module
of the next step (the weighted sum at the end of the encoder).  The collective and the optimiser therefore run under the next
encoder forward instead of extending the step.  With unfrozen HuBERT layers (hubert_train.py) that premise does not hold - the
encoder reads trainable parameters from its first unfrozen layer on - so the encoder joins at the top of its forward instead
(speech_encoder.forward) and only the per-layer gradient all-reduces overlap (with the backward of the layers below)."""
import os
from typing import Optional

import torch
import torch.distributed as dist

from .optim import FlatAdam, linear_warmup_decay
from .parallel import AccumulationSchedule, GradAllReduce, comm_span, dp_world, gather_loss_feats, scale_replicated_grads


class ContrastiveTrainer:
    def __init__(self, model, group: Optional[dist.ProcessGroup] = None, check_every: int = 200):
        self.model, self.group = model, group
        # every ``check_every`` steps the CIF modules' device-side consistency counters are read (one host synchronisation): the
        # reference asserts on every call; here the training path never reads the device, so the assertion is deferred, not dropped
        self.check_every = int(check_every)
        self._cif = [m for m in model.modules() if hasattr(m, "check_flags") and hasattr(m, "consistency_flags")]
        cfg = model.config
        oc = cfg.audio_encoder.optim
        assert oc.name == "Adam", "the shipped recipes use Adam"
        self.base_lr = float(oc.args.lr)
        self.sched = cfg.audio_encoder.get("scheduler", None)
        self.opt = FlatAdam(model.getTrainableParams(), lr=self.base_lr, weight_decay=float(oc.args.get("weight_decay", 0.0)),
                            max_grad_norm=float(cfg.trainer.get("gradient_clip_val", 0.0)))
        # trainer.accumulate_grad_batches (spchclip_h+.yaml:138): gradients of n micro-steps add up in the flat buffer, collectives and
        # optimiser on the boundary micro-step only
        self.accum = AccumulationSchedule(int(cfg.trainer.get("accumulate_grad_batches", 1) or 1))
        self._boundary = self.accum.n == 1          # is the micro-step in flight the one that ends with the optimiser step?
        self.allreduce = GradAllReduce(self.opt.flat_g, group)
        self.replicated = [p for p in model.criterion.parameters() if p.requires_grad]   # evaluated on the full batch by every rank
        from .ops import shared_stream
        self.side = shared_stream("optimiser", self.opt.flat_g.device) if self.opt.flat_g.is_cuda else None
        self._pending = False
        self._one = None
        # unfrozen HuBERT layers (hubert_train.py): the slice of the flat gradient buffer that belongs to a layer is all-reduced
        # on the side stream as soon as that layer's backward has been enqueued, i.e. under the backward of the layers below it
        self._reduced = []                         # [start, end) ranges already summed across ranks in this step
        tl = getattr(model.audio_encoder, "train_layers", None)
        if tl is not None:
            self._layer_span = {i: self.opt.span(tl.layer_parameters(i)) for i in tl.ids}
            tl.grad_ready_hook = self._layer_ready
        # join point: first use of a trainable parameter in a step
        model.audio_encoder.before_trainable = self.join
        for m in (model.parallel_branch, model.cascaded_branch, model.criterion):
            if m is not None:
                m.register_forward_pre_hook(lambda *_: self.join())

    def _world(self) -> int:
        return dp_world(self.group)

    def _layer_ready(self, i: int) -> None:
        if self._world() == 1 or not self._boundary:       # (a micro-step inside an accumulation window: nothing to exchange yet)
            return
        lo, hi = self._layer_span[i]
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side), comm_span("all_reduce", True):
                dist.all_reduce(self.opt.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        else:
            with comm_span("all_reduce", False):
                dist.all_reduce(self.opt.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        self._reduced.append((lo, hi))

    def join(self) -> None:
        """Make the current stream wait for the optimiser work of the previous step (no host synchronisation)."""
        if self._pending:
            if self.side is not None:
                # bracket = how long the main stream sits blocked here: the part of the side stream's work (all-reduce, clip + Adam)
                # that the next step's frozen encoder forward did NOT cover
                with comm_span("join_wait", True):
                    torch.cuda.current_stream().wait_stream(self.side)
            self.opt.zero_grad()                      # gradients of the finished step stay readable until the next step needs the buffer
        self._pending = False

    def lr_at(self, step: int) -> float:
        s = self.sched
        if s is None or s.name != "linear_warmup_decay":
            return self.base_lr
        return self.base_lr * linear_warmup_decay(step, int(s.warmup), int(s.max_step), self.base_lr, float(s.final_lr))

    def step(self, batch: dict) -> torch.Tensor:
        model = self.model
        loss_feats = model.training_step(batch)["loss_feats"]
        keys = [k for k in ("parallel_audio_feat", "cascaded_audio_feat") if k in loss_feats]
        feats, i, ids = gather_loss_feats([loss_feats[k] for k in keys], loss_feats["image_feat"], loss_feats["id"], self.group)
        gathered = {"image_feat": i, "id": ids, **dict(zip(keys, feats))}
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if "cif_quantity_out" in loss_feats:
            # the L1 quantity loss is a per-sample mean: each rank contributes its local mean / world so that the
            # SUM all-reduce of the gradients yields the global-batch mean (kwClip.py:1030-1038 runs on the gathered batch)
            gathered["cif_quantity_out"] = loss_feats["cif_quantity_out"]
            gathered["cif_target_len"] = loss_feats["cif_target_len"]
        losses = model.compute_loss(gathered)
        loss = losses["loss"]
        if world > 1 and "quantity_loss" in losses:
            loss = loss - model.quantity_loss_weight * losses["quantity_loss"] * (1.0 - 1.0 / world)
        if self._one is None or self._one.device != loss.device:
            self._one = torch.full((), self.accum.loss_scale, device=loss.device, dtype=loss.dtype)
        self._boundary = self.accum.micro + 1 >= self.accum.n          # read by the per-layer hooks during this backward
        loss.backward(gradient=self._one)               # (a cached 1 / accumulate_grad_batches: no fill launch per step)
        if not self.accum.advance():
            return loss.detach()                        # inside an accumulation window: no collective, no optimiser, gradients stay
        lr = self.lr_at(model.global_step)
        if self.side is None:
            self._finish(lr)
        else:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self._finish(lr)
        self._pending = True
        model.global_step += 1
        if self._cif and self.check_every > 0 and model.global_step % self.check_every == 0:
            self.check_consistency()
        return loss.detach()

    def check_consistency(self) -> None:
        """Reads the CIF counters (synchronises).  Raises if a call saw no positive weight sum at all (the reference's assert) or
        if a keyword count disagreed with the host-side target the output buffer was sized from."""
        for m in self._cif:
            flags = m.check_flags()
            if flags["count_mismatches"] > 0:
                raise RuntimeError(f"CIF: {flags['count_mismatches']} keyword counts differed from clip(target_len, 1, 75) while the "
                                   f"output was sized from the host-side targets ({flags})")

    def _finish(self, lr: float) -> None:
        scale_replicated_grads(self.replicated, self.group)
        if not self._reduced:
            self.allreduce.launch()
            self.allreduce.wait()
        else:                                          # everything the per-layer collectives have not covered yet
            pos = 0
            for lo, hi in sorted(self._reduced) + [(self.opt.size, self.opt.size)]:
                if lo > pos and self._world() > 1:
                    with comm_span("all_reduce", self.opt.flat_g.is_cuda):
                        dist.all_reduce(self.opt.flat_g[pos:lo], op=dist.ReduceOp.SUM, group=self.group)
                pos = max(pos, hi)
            self._reduced = []
        self.opt.step(lr=lr)
        # parameter-only pieces of the next forward (the CLS query of the parallel head) go here too: on the side stream, right
        # behind the optimiser, under the next step's encoder forward
        pre = getattr(self.model, "prefetch_after_step", None)
        if pre is not None and os.environ.get("SC_PREFETCH", "1") == "1":
            pre()
