"""KWClip_GeneralTransformer: host-side mirror of avssl/model/kwClip.py for the contrastive hot path.

Kept API: ``forward(batch) -> (losses, log_metrics, others)`` with the reference's keys
(kwClip.py:898-963), ``compute_loss(dict)`` (:999-1040), ``encode_speech(wav)`` (:1042-1091),
``feature_extractor_s3prl(wav)`` (:965-997), ``processWavs`` (:600-615), ``getTrainableParams``,
``training_step`` / ``training_step_end`` semantics (:145-193) so it drops into the existing
PyTorch-Lightning loop (it is a plain ``nn.Module``; pytorch_lightning is not installed here).

Out of scope here: the frozen CLIP towers (the ``clip`` package and its weights are not available offline).
``batch["image"]`` may already hold the CLIP image embeddings (B, E); for pixel input pass an
``image_encoder`` callable (frozen, no grad) to the constructor.
"""
import logging
from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import nn

from . import losses
from .clip_text import ClipModel
from .head_tail import unit_rows
from .kw_branches import KW_CascadedBranchPlus, KW_HybridBranchPlus, KW_ParallelBranch
from .speech_encoder import FairseqSpeechEncoder_Hubert

logger = logging.getLogger(__name__)


from .config import Config, load_config, synthetic_reduced_vocab  # noqa: E402,F401


def _cif_args(d: int) -> dict:
    return {"quantity_loss_weight": 0.25, "using_gt_len": False, "cif_output_dim": d, "encoder_embed_dim": d,
            "produce_weight_type": "conv", "cif_threshold": 1.0, "conv_cif_layer_num": 1, "conv_cif_width": 3,
            "conv_cif_dropout": 0.1, "apply_scaling": True, "scaling_step": 5000, "apply_tail_handling": True,
            "tail_handling_firing_threshold": 0.5, "add_cif_ctxt_layers": False}


def cascaded_plus_base_config(**overrides) -> Config:
    """config/speechCLIP+/model_base/spchclip_c+.yaml (BASELINE configs[2]) restricted to the keys the path reads
    (tests/test_host_cpu.py::test_builtin_configs_equal_the_reference_yamls holds every one of them against the yaml)."""
    cfg = base_parallel_config()
    cfg.audio_encoder.name = "hubert_base"
    ms = cfg.model_settings
    ms.cascaded_objective_weight, ms.parallel_objective_weight = 1.0, 0.0
    ms.cascaded_branch = Config({
        "type": "CascadedBranch_dynamic",
        "vq": {"activation": "gelu", "type": "SimpleVectorQuantizer",
               "args": {"temp": "fixed=0.1", "time_first": True, "use_gumbel": False, "hard": True}},
        "downsampling": {"type": "cif", "cif": _cif_args(768)},
        "keyword": {"detokenized_K_neighbors": 5, "retrieve_method": "cosine",
                    "batchnorms": {"type": "eachKw", "std_scale": 1.0, "learnable": True, "parallel": True}},
        "transformer_args": {"type": "MultiheadAttentionAndNorm", "n_layers": 1, "d_model": 768, "nhead": 1,
                             "dim_feedforward": 3072, "dropout": 0.1, "activation": "gelu", "layer_norm_eps": 1.0e-5,
                             "batch_first": True, "norm_first": False}})
    cfg.cl_loss.args.temperature_trainable = True
    cfg.retrieval.audio_feat_src = "cascaded"
    cfg.clip = Config({"name": "ViT-B/32", "embed_dim": 512, "reduce_subword_embbedding": synthetic_reduced_vocab(8112)})
    for k, v in overrides.items():
        cfg[k] = v
    return cfg


def hybrid_plus_large_config(**overrides) -> Config:
    """config/speechCLIP+/model_large/coco/spchclip_h+.yaml (BASELINE configs[4]) restricted to the keys the path reads.  Unlike the
    PARALLEL large recipe this yaml has no ``normalize_hiddenstates`` key: the reference default applies (False,
    speech_encoder_plus.py:350) - round 4 inherited True from large_parallel_config; and the recipe accumulates
    two micro-batches per optimiser step (``trainer.accumulate_grad_batches: 2``, :138)."""
    cfg = large_parallel_config()
    cfg.audio_encoder.normalize_hiddenstates = False
    cfg.trainer.accumulate_grad_batches = 2
    ms = cfg.model_settings
    ms.cascaded_objective_weight, ms.parallel_objective_weight = 1.0, 1.0
    ms.cascaded_branch = Config({
        "type": "HybridBranch_dynamic",
        "vq": {"activation": "gelu", "type": "SimpleVectorQuantizer",
               "args": {"temp": "fixed=0.1", "time_first": True, "use_gumbel": False, "hard": True}},
        "downsampling": {"type": "cif", "cif": _cif_args(1024)},
        "keyword": {"detokenized_K_neighbors": 5, "retrieve_method": "cosine",
                    "batchnorms": {"type": "eachKw", "std_scale": 1.0, "learnable": True, "parallel": True},
                    "kw_projection": {"dropout": 0.1, "dimensions": [1024, 1024, 768]}},
        "transformer_args": {"type": "MultiheadAttentionAndNorm", "n_layers": 1, "d_model": 1024, "nhead": 8,
                             "dim_feedforward": 4096, "dropout": 0.1, "activation": "gelu", "layer_norm_eps": 1.0e-5,
                             "batch_first": True, "norm_first": False}})
    cfg.cl_loss.args.temperature_trainable = True
    cfg.clip = Config({"name": "ViT-L/14", "embed_dim": 768, "reduce_subword_embbedding": synthetic_reduced_vocab(19787)})
    for k, v in overrides.items():
        cfg[k] = v
    return cfg


def large_parallel_config(**overrides) -> Config:
    """config/speechCLIP/model_large/flickr/spchclp_p.yaml restricted to the keys the hot path reads
    (HuBERT-large ll60k, normalised hidden states, 1024-wide head, CLIP ViT-L/14 joint width 768)."""
    cfg = base_parallel_config()
    cfg.model_settings.parallel_branch.transformer_args.d_model = 1024
    cfg.model_settings.parallel_branch.transformer_args.dim_feedforward = 4096
    cfg.clip = Config({"name": "ViT-L/14", "embed_dim": 768})
    cfg.audio_encoder.name = "hubert_large_ll60k"
    cfg.audio_encoder.normalize_hiddenstates = True
    cfg.cl_loss.args.temperature_trainable = True
    for k, v in overrides.items():
        cfg[k] = v
    return cfg


def base_parallel_config(**overrides) -> Config:
    """config/speechCLIP/model_base/spchclp_p.yaml restricted to the keys the hot path reads."""
    cfg = Config({
        "model_settings": {
            "cascaded_objective_weight": 0.0,
            "parallel_objective_weight": 1.0,
            "parallel_branch": {
                "transformer_type": "TransformerEncoder",
                "transformer_args": {"n_layers": 1, "d_model": 768, "nhead": 8, "dim_feedforward": 3072, "dropout": 0.1,
                                     "activation": "gelu", "layer_norm_eps": 1.0e-5, "batch_first": True,
                                     "norm_first": False},
                "need_projection": True,
            },
        },
        "cl_loss": {"type": "MaskedContrastiveLoss",
                    "args": {"temperature": 0.07, "temperature_trainable": False, "margin": 0.0, "dcl": False,
                             "a2b": True, "b2a": True}},
        "retrieval": {"audio_feat_src": "parallel", "recall_at": [1, 5, 10]},
        "clip": {"name": "ViT-B/32", "embed_dim": 512},
        "audio_encoder": {"type": "FairseqHubert", "name": "hubert", "pretrained": True, "trainable": False,
                          "feat_select_idx": "weighted_sum", "layer_drop": 0.0, "max_audio_len": 102400,
                          "normalize_hiddenstates": False,
                          "optim": {"name": "Adam", "args": {"lr": 1.0e-4, "weight_decay": 1.0e-6}},
                          "scheduler": {"name": "linear_warmup_decay", "warmup": 5000, "max_step": 50000,
                                        "final_lr": 1.0e-8}},
        "trainer": {"gradient_clip_val": 4, "accumulate_grad_batches": 1},
    })
    for k, v in overrides.items():
        cfg[k] = v
    return cfg


def set_dropout(model: nn.Module, enabled: bool) -> nn.Module:
    """Switch every train-mode dropout site of ``model`` off (or back on): the frozen HuBERT's (speech_encoder.HubertArch), the
    parallel head's four sites (head_tail.ParallelHeadFn), nn.MultiheadAttention's and every nn.Dropout.  The reference has no
    such switch (its train step always runs them); the deterministic form is what the parity tests compare with the oracle.
    The original probabilities are remembered on the modules, so ``set_dropout(model, True)`` restores them."""
    from .transformer_models import TransformerEncoder
    for m in model.modules():
        if isinstance(m, FairseqSpeechEncoder_Hubert):
            m.hubert_dropout = bool(enabled)
        elif isinstance(m, nn.Dropout):
            if not hasattr(m, "_sc_p"):
                m._sc_p = m.p
            m.p = m._sc_p if enabled else 0.0
        elif isinstance(m, (nn.MultiheadAttention, TransformerEncoder)):
            if not hasattr(m, "_sc_p"):
                m._sc_p = m.dropout
            m.dropout = m._sc_p if enabled else 0.0
    return model


class KWClip_GeneralTransformer(nn.Module):
    def __init__(self, config, image_encoder: Optional[Callable] = None, device: str = "cuda", hubert_state_dict=None,
                 hubert_arch=None):
        super().__init__()
        if isinstance(config, str):                   # path of a reference yaml recipe (config/**/*.yaml parse unchanged)
            config = load_config(config)
        self.config = config if isinstance(config, Config) else load_config(config)
        config = self.config
        self._device = torch.device(device)
        self.audio_encoder_type = config.audio_encoder.type
        if self.audio_encoder_type != "FairseqHubert":
            raise NotImplementedError(f"audio_encoder.type = {self.audio_encoder_type}: only FairseqHubert is built "
                                      "(every shipped config uses it)")
        enc_args = {k: v for k, v in config.audio_encoder.items() if k not in ("type", "optim", "scheduler", "device")}
        self.audio_encoder = FairseqSpeechEncoder_Hubert(device=device, state_dict=hubert_state_dict, arch=hubert_arch,
                                                         **enc_args)
        self.audio_embd_dim = self.audio_encoder.out_dim
        self.image_encoder = image_encoder
        # CLIP joint embedding width (512 ViT-B/32, 768 ViT-L/14): width of image_feat and of the branch projection
        self.subword_embd_dim = int(config.clip.get("embed_dim", 512))
        self.recall_at = config.retrieval.recall_at
        self.criterion = getattr(losses, config.cl_loss.type)(**config.cl_loss.args)
        self.keyword_num = None
        self.cascaded_branch = None
        self.parallel_branch = None
        ms = config.model_settings
        self.clip = None
        if ms.cascaded_objective_weight > 0:
            # kwClip.py:684-745.  The CLIP text tower is frozen; without a checkpoint offline it is seeded random
            # (load real weights with self.clip.load_state_dict).
            clip_args = {k: v for k, v in config.clip.items() if k not in ("embed_dim", "device")}
            self.clip = ClipModel(device=device, **clip_args)
            text_dim = self.clip.model.token_embedding.weight.size(-1)
            cBranchType = ms.cascaded_branch.type.replace("KW_", "").replace("dynamic", "plus")
            if cBranchType == "CascadedBranch_plus":
                self.cascaded_branch = KW_CascadedBranchPlus(config=config, audio_dim=self.audio_embd_dim,
                                                             text_dim=text_dim, clip=self.clip)
            elif cBranchType == "HybridBranch_plus":
                assert ms.parallel_objective_weight > 0, ms.parallel_objective_weight
                self.cascaded_branch = KW_HybridBranchPlus(config=config, audio_dim=self.audio_embd_dim,
                                                           text_dim=text_dim, out_dim=self.subword_embd_dim,
                                                           clip=self.clip)
            else:
                raise NotImplementedError(f"{cBranchType}: the non-plus cascaded / hybrid branches are out of scope "
                                          "(not in BASELINE configs; SURVEY section 2)")
            ds = ms.cascaded_branch.get("downsampling", None)
            if ds is not None and ds.type == "cif":
                self.quantity_loss_weight = ds.cif.get("quantity_loss_weight", 1.0)
                self.quantity_loss_criteria = nn.L1Loss()
        if ms.parallel_objective_weight > 0 and self.cascaded_branch is None:
            logger.info("Create Parallel Branch")
            self.parallel_branch = KW_ParallelBranch(config=config, audio_dim=self.audio_embd_dim,
                                                     text_dim=self.subword_embd_dim)
        # frames behind feat_len the encoder still has to compute (speech_encoder.segment_pitches): the CIF weight conv of the plus
        # branches looks conv_cif_width // 2 frames past the last valid one (avssl/module/cif.py:44-52); the CLS head reads none
        if self.cascaded_branch is None:
            self.audio_encoder.tail_rows = 0
        else:
            ds = ms.cascaded_branch.get("downsampling", None)
            width = ds.cif.get("conv_cif_width", 5) if (ds is not None and ds.type == "cif") else 5
            self.audio_encoder.tail_rows = max(1, int(width) // 2)
            self.audio_encoder.branch_inplace = True      # the branch's attention block reads the encoder's output rows in place
        self.img_enc_proj_net = None
        self.p_branch_proj_net = None
        self.c_branch_proj_net = None
        self.global_step = 0
        self.to(self._device)

    @property
    def device(self):
        return self._device

    # ---------------------------------------------------------------------------------------------
    def prefetch_after_step(self) -> None:
        """Called by the trainer on its side stream right after the optimiser step: computes what the next forward needs from the
        parameters alone (head_tail.cls_query, cached per parameter version), off the critical path."""
        pb = self.parallel_branch
        if pb is not None and getattr(pb, "self_att", None) is not None and hasattr(pb.self_att, "cls_forward"):
            from .head_tail import cls_query
            with torch.no_grad():
                cls_query(pb.self_att, pb.cls)

    def getTrainableParams(self) -> list:
        """kwClip.py:620-644 + :812-837."""
        _params = []
        _params += self.audio_encoder.trainable_params()
        _params += list(self.criterion.parameters())
        if self.cascaded_branch is not None:
            _params += [p for p in self.cascaded_branch.parameters() if p.requires_grad]
        if self.parallel_branch is not None:
            _params += list(self.parallel_branch.parameters())
        return _params

    def forward_audio(self, wav, wav_len=[], return_hidden_states: bool = False):
        return self.audio_encoder(wav, wav_len, return_hidden_states=return_hidden_states)

    def forward_image(self, images: Union[list, torch.Tensor]) -> torch.Tensor:
        if isinstance(images, torch.Tensor) and images.dim() == 2:
            return images.to(self._device)                     # pre-computed (frozen) CLIP image embeddings
        if self.image_encoder is None:
            raise RuntimeError("pixel input needs an image_encoder callable (the CLIP image tower is frozen and out of "
                               "scope); pass CLIP embeddings (B, E) in batch['image'] instead")
        if isinstance(images, torch.Tensor) and (images.dim() != 4 or images.shape[1] != 3):
            raise ValueError(f"Incorrect image tensor shape {images.shape}")
        with torch.no_grad():
            return self.image_encoder(images)

    def processWavs(self, wav):
        wav_len = [len(x) for x in wav]
        return wav, wav_len

    # ---------------------------------------------------------------------------------------------
    def forward(self, batch: dict) -> tuple:
        """kwClip.py:839-963."""
        wav, wav_len, image, id = batch["wav"], batch["wav_len"], batch["image"], batch["id"]
        audio_feat, audio_feat_len = self.forward_audio(wav, wav_len, return_hidden_states=False)
        image_feat = self.forward_image(image)
        image_feat = unit_rows(image_feat.float())
        if self.cascaded_branch is not None:                                       # kwClip.py:859-880
            otherInputs = {"global_step": self.global_step}
            if getattr(self.cascaded_branch, "using_gt_len", False):
                assert "text" in batch, f"Text captions are required, {batch.keys()}"
                target_len = torch.LongTensor([(t.squeeze().tolist().index(49407) - 1) for t in batch["text"]]).to(self._device)
            else:
                target_len = getattr(audio_feat_len, "_sc_target20", None)     # uploaded with the batch's other integers (speech_encoder)
                if target_len is None:
                    target_len = (audio_feat_len / 20).round().long()
                    lens_host = getattr(audio_feat_len, "_sc_host", None)
                    if lens_host is not None:          # host twin of the targets: the CIF output is sized without a device read
                        from .kw_branches import target_len_host
                        target_len._sc_host = target_len_host(lens_host)
            otherInputs["target_len"] = target_len
            output = self.cascaded_branch(audio_feat=audio_feat, audio_feat_len=audio_feat_len, otherInputs=otherInputs)
        if self.parallel_branch is not None:
            output = self.parallel_branch(audio_feat=audio_feat, audio_feat_len=audio_feat_len)
        parallel_audio_feat = output["parallel_audio_feat"]
        cascaded_audio_feat = output["cascaded_audio_feat"]
        vq_results, keywords, dsample_results = output["vq_results"], output["keywords"], output["dsample_results"]
        keywords_len = dsample_results["dsample_feats_length"] if dsample_results is not None else None
        id = id.to(self._device)
        losses_ = {"id": id, "image_feat": image_feat}
        if cascaded_audio_feat is not None:
            cascaded_audio_feat = unit_rows(cascaded_audio_feat.float())
            losses_["cascaded_audio_feat"] = cascaded_audio_feat
        if parallel_audio_feat is not None:
            parallel_audio_feat = unit_rows(parallel_audio_feat.float())
            losses_["parallel_audio_feat"] = parallel_audio_feat
        if self.cascaded_branch is not None and getattr(self.cascaded_branch, "downsampling_type", None) == "cif":
            assert "target_len" in dsample_results and "quantity_out" in dsample_results, f"{dsample_results.keys()}"
            losses_["cif_quantity_out"] = dsample_results["quantity_out"]
            losses_["cif_target_len"] = dsample_results["target_len"]
        log_metrics = {"cl_temp": self.criterion.temperature_for_logging}   # kwClip.py: .item() per step; here no host sync
        if vq_results is not None:
            log_metrics["softmax_temp"] = vq_results["temp"]
        if self.cascaded_branch is not None:
            if dsample_results is not None and "dsample_len_diff" in dsample_results:
                log_metrics["dsample_len_diff"] = dsample_results["dsample_len_diff"]
            log_metrics.update({k: vq_results[k] for k in ["temp", "code_perplexity", "prob_perplexity", "ent_per_t"]})
        return (losses_, log_metrics,
                {"id": id, "image_feat": image_feat, "parallel_audio_feat": parallel_audio_feat,
                 "cascaded_audio_feat": cascaded_audio_feat, "vq_results": vq_results, "keywords": keywords,
                 "dsample_results": dsample_results, "keywords_len": keywords_len})

    def compute_loss(self, inputDict: dict):
        """kwClip.py:999-1040."""
        assert isinstance(inputDict, dict)
        required_keys = {"id", "image_feat"}
        assert required_keys.issubset(set(inputDict.keys())), f"required: {required_keys}, input: {inputDict.keys()}"
        losses_ = {}
        terms = []                                  # (weight, loss term): summed below without 0 + / 1.0 * launches
        image_feat = inputDict["image_feat"].float()
        id = inputDict["id"]
        for branchType in ["cascaded", "parallel"]:
            loss_weight = self.config.model_settings.get(f"{branchType}_objective_weight", 0.0)
            if loss_weight > 0.0:
                feats_key = f"{branchType}_audio_feat"
                assert feats_key in inputDict, f"{inputDict.keys()}"
                losses_[f"{branchType[0]}_cl_loss"] = self.criterion(feat_A=inputDict[feats_key].float(),
                                                                     feat_B=image_feat, index=id)
                terms.append((float(loss_weight), losses_[f"{branchType[0]}_cl_loss"]))
        if "cif_quantity_out" in inputDict and "cif_target_len" in inputDict and hasattr(self, "quantity_loss_criteria"):
            losses_["quantity_loss"] = self.quantity_loss_criteria(inputDict["cif_quantity_out"], inputDict["cif_target_len"])
            terms.append((float(self.quantity_loss_weight), losses_["quantity_loss"]))
        total = 0
        for w, t in terms:
            t = t if w == 1.0 else w * t
            total = t if isinstance(total, int) else total + t
        losses_["loss"] = total
        return losses_

    def transfer_batch_to_device(self, batch: dict, device=None, dataloader_idx: int = 0) -> dict:
        """The LightningModule hook of the same name (what hands kwClip.py:145-147 its device batch): see
        data.transfer_batch_to_device - host twin of ``wav_len``, the waveform copied on its own stream with a completion event."""
        from .data import transfer_batch_to_device
        return transfer_batch_to_device(batch, self._device if device is None else device)

    def training_step(self, batch: dict) -> dict:
        losses_, log_metrics = self.forward(batch)[:2]
        return {"loss_feats": losses_, "log_metrics": log_metrics}

    def training_step_end(self, outputs: dict) -> dict:
        """kwClip.py:149-193: the loss is computed on the gathered batch."""
        if "loss" in outputs:
            return {"loss": torch.mean(outputs["loss"])}
        losses_ = self.compute_loss(outputs["loss_feats"])
        return {"loss": losses_["loss"], **{f"train_{k}": v for k, v in losses_.items()}}

    # --------------------------------------------------------------------------------------------- Lightning-shaped hooks
    def configure_optimizers(self):
        """kwClip.py:646-674: ``([optimizer], [{"scheduler": ..., "interval": "step"}])`` over getTrainableParams().  The optimiser
        is the flat fused clip + Adam (optim.FlatAdamOptimizer, a torch.optim.Optimizer), the schedule a LambdaLR."""
        from .optim import FlatAdamOptimizer, get_scheduler
        oc = self.config.audio_encoder.optim
        if oc.name != "Adam":
            raise NotImplementedError(f"optimizer {oc.name}: the shipped recipes use Adam")
        opt = FlatAdamOptimizer(self.getTrainableParams(), **{k: v for k, v in dict(oc.args).items()})
        sched = self.config.audio_encoder.get("scheduler", None)
        if sched is None:
            return [opt], []
        return [opt], [{"scheduler": get_scheduler(opt, **dict(sched)), "interval": "step"}]

    def log_dict(self, metrics: dict, **kwargs) -> None:
        """Stand-in for LightningModule.log_dict (pytorch_lightning is not installed here): keeps the last values."""
        self.logged = getattr(self, "logged", {})
        self.logged.update(metrics)

    # --------------------------------------------------------------------------------------------- validation (f4)
    def validation_step(self, batch: dict, batch_idx: int = 0) -> dict:
        """kwClip.py:195-246: ``{"loss_feats", "log_metrics", "others"}`` (the reference's keys)."""
        with torch.no_grad():
            losses_, log_metrics, others = self.forward(batch)
        return {"loss_feats": losses_, "log_metrics": log_metrics, "others": others}

    def validation_step_end(self, outputs: dict) -> dict:
        """kwClip.py:248-285: loss on the gathered features, logged as ``val_*``; returns ``others``.  The reference moves every
        tensor of ``others`` to the CPU here (a synchronisation per validation step); they stay on the device - the epoch-end
        retrieval runs there."""
        assert isinstance(outputs, dict)
        with torch.no_grad():
            losses_ = self.compute_loss(outputs["loss_feats"])
        result = {f"val_{k}": v for k, v in losses_.items()}
        result.update({f"val_{k}": (v.float().mean() if isinstance(v, torch.Tensor) else v) for k, v in outputs["log_metrics"].items()})
        self.log_dict(result, on_step=True, on_epoch=True, prog_bar=True, logger=True, sync_dist=True)
        return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in outputs["others"].items()}

    def validation_epoch_end(self, outputs: list) -> dict:
        """kwClip.py:447-482: one image embedding per id (the last seen, as the reference's dict does), audio x image scores
        and recall@k in both directions - everything stays on the device (retrieval.mutualRetrieval).  ``outputs``: the
        ``others`` dicts returned by validation_step_end (or validation_step's full dicts)."""
        from .retrieval import mutualRetrieval
        outputs = [o["others"] if "others" in o else o for o in outputs]
        src = self.config.retrieval.get("audio_feat_src", "parallel")
        key = "cascaded_audio_feat" if src == "cascaded" else "parallel_audio_feat"
        ids = torch.cat([o["id"] for o in outputs], dim=0)
        audio = torch.cat([o["audio_feat"] if "audio_feat" in o else o[key] for o in outputs], dim=0).float()
        imgs = torch.cat([o["image_feat"] for o in outputs], dim=0).float()
        # last occurrence of every id, ids kept in first-seen order (python dict semantics of the reference)
        uniq, inv = torch.unique(ids, return_inverse=True)
        n = ids.shape[0]
        pos = torch.arange(n, device=ids.device)
        last = torch.zeros(uniq.shape[0], dtype=torch.long, device=ids.device).scatter_reduce(0, inv, pos, "amax", include_self=False)
        first = torch.full((uniq.shape[0],), n, dtype=torch.long, device=ids.device).scatter_reduce(0, inv, pos, "amin", include_self=False)
        order = torch.argsort(first)
        img_ids, img_feats = uniq[order], imgs[last[order]]
        score_per_audio = audio @ img_feats.t()
        return mutualRetrieval(score_per_A=score_per_audio, score_per_B=score_per_audio.t(), AB_answers=ids, BA_answers=img_ids,
                               recall_at=self.recall_at)

    # --------------------------------------------------------------------------------------------- checkpoints (f4)
    @staticmethod
    def split_reference_state_dict(sd: dict):
        """A PyTorch-Lightning checkpoint of the reference (``ckpt["state_dict"]``) -> (hubert_state_dict in fairseq naming,
        rest).  Reference keys: ``audio_encoder.encoder.<fairseq HubertModel key>``, ``audio_encoder.weightedsum_layer.weights``,
        ``parallel_branch.* / cascaded_branch.*`` (same module names here), ``criterion.temperature``, ``clip.model.*`` (the text
        side is kept, ``clip.model.visual.*`` is dropped: image embeddings are inputs)."""
        hubert, rest = {}, {}
        for k, v in sd.items():
            if k.startswith("audio_encoder.encoder."):
                hubert[k[len("audio_encoder.encoder."):]] = v
            elif k.startswith("clip.model.visual.") or k in ("clip.model.logit_scale",):
                continue
            else:
                rest[k] = v
        return hubert, rest

    @classmethod
    def from_reference_checkpoint(cls, config, state_dict: dict, **kw):
        """Build the model on the reference's weights: the HuBERT part is converted by the encoder's loader (fairseq key names,
        weight-normed pos_conv accepted), everything else loads by name (non-strict for keys this build does not hold)."""
        hubert, rest = cls.split_reference_state_dict(state_dict)
        model = cls(config, hubert_state_dict=hubert if hubert else None, **kw)
        own = model.state_dict()
        loadable = {k: v for k, v in rest.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
        missing = [k for k in own if k not in loadable and not k.startswith("audio_encoder.train_layers.")]
        model.load_state_dict(loadable, strict=False)
        model._reference_load_report = {"loaded": sorted(loadable), "not_in_checkpoint": missing,
                                        "ignored": sorted(k for k in rest if k not in loadable)}
        return model

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, config=None, **kw):
        """example.py:9-10 ``KWClip_GeneralTransformer.load_from_checkpoint(path)``: a PyTorch-Lightning checkpoint file holds
        ``state_dict`` and (via save_hyperparameters, avssl/base/base_model.py:14) the config under ``hyper_parameters``."""
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        if config is None:
            hp = ckpt.get("hyper_parameters", {})
            config = hp.get("config", hp)
        return cls.from_reference_checkpoint(config, ckpt["state_dict"], **kw)

    # ---------------------------------------------------------------------------------------------
    def encode_speech(self, wav) -> dict:
        """kwClip.py:1042-1091 (un-normalised branch output)."""
        wav, wav_len = self.processWavs(wav)
        audio_feat, audio_feat_len = self.forward_audio(wav, wav_len)
        if self.cascaded_branch is not None:
            output = self.cascaded_branch(audio_feat=audio_feat, audio_feat_len=audio_feat_len)
        if self.parallel_branch is not None:
            output = self.parallel_branch(audio_feat=audio_feat, audio_feat_len=audio_feat_len)
        return {"cascaded_audio_feat": output["cascaded_audio_feat"], "parallel_audio_feat": output["parallel_audio_feat"],
                "vq_results": output["vq_results"], "keywords": output["keywords"]}

    def feature_extractor_s3prl(self, wav) -> Tuple[torch.Tensor, Tuple]:
        """kwClip.py:965-997: the HuBERT states (fresh tensors: the encoder clones what it hands out) followed by the branch
        layer's hidden states (full-sequence layer on the library's kernels)."""
        wav, wav_len = self.processWavs(wav)
        audio_feat, audio_len, hidden_states = self.forward_audio(wav, wav_len, return_hidden_states=True)
        assert isinstance(hidden_states, tuple)
        with torch.no_grad():
            if self.cascaded_branch is not None:
                hidden_states = hidden_states + tuple(self.cascaded_branch.extract_hidden_states(audio_feat, audio_len)[1:])
            if self.parallel_branch is not None:
                hidden_states = hidden_states + tuple(self.parallel_branch.extract_hidden_states(audio_feat, audio_len)[1:])
        return hidden_states[-1], hidden_states
