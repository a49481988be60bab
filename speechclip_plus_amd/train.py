"""One contrastive train step (frozen-HuBERT recipe of every shipped config):
forward -> packed all-gather -> global-batch InfoNCE -> backward (head + weighted-sum weights [+ temperature])
-> flat gradient all-reduce -> clip + Adam.  Mirrors training_step / training_step_end / configure_optimizers
of avssl/model/kwClip.py:145-193,646-674 without PyTorch-Lightning."""
from typing import Optional

import torch
import torch.distributed as dist

from .optim import FlatAdam, linear_warmup_decay
from .parallel import GradAllReduce, gather_loss_feats


class ContrastiveTrainer:
    def __init__(self, model, group: Optional[dist.ProcessGroup] = None):
        self.model, self.group = model, group
        cfg = model.config
        oc = cfg.audio_encoder.optim
        assert oc.name == "Adam", "the shipped recipes use Adam"
        self.base_lr = float(oc.args.lr)
        self.sched = cfg.audio_encoder.get("scheduler", None)
        self.opt = FlatAdam(model.getTrainableParams(), lr=self.base_lr, weight_decay=float(oc.args.get("weight_decay", 0.0)),
                            max_grad_norm=float(cfg.trainer.get("gradient_clip_val", 0.0)))
        self.allreduce = GradAllReduce(self.opt.flat_g, group)

    def lr_at(self, step: int) -> float:
        s = self.sched
        if s is None or s.name != "linear_warmup_decay":
            return self.base_lr
        return self.base_lr * linear_warmup_decay(step, int(s.warmup), int(s.max_step), self.base_lr, float(s.final_lr))

    def step(self, batch: dict) -> torch.Tensor:
        model = self.model
        self.opt.zero_grad()
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
        loss.backward()
        self.allreduce.launch()
        self.allreduce.wait()
        self.opt.step(lr=self.lr_at(model.global_step))
        model.global_step += 1
        return loss.detach()
