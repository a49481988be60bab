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
from .kw_branches import KW_ParallelBranch
from .speech_encoder import FairseqSpeechEncoder_Hubert

logger = logging.getLogger(__name__)


class Config(dict):
    """Minimal attribute-style nested dict (stands in for avssl/base/ordered_namespace.py)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def large_parallel_config(**overrides) -> Config:
    """config/speechCLIP/model_large/flickr/spchclp_p.yaml restricted to the keys the hot path reads
    (HuBERT-large ll60k, normalised hidden states, 1024-wide head, CLIP ViT-L/14 joint width 768)."""
    cfg = base_parallel_config()
    cfg.model_settings.parallel_branch.transformer_args.d_model = 1024
    cfg.model_settings.parallel_branch.transformer_args.dim_feedforward = 4096
    cfg.clip = Config({"name": "ViT-L/14", "embed_dim": 768})
    cfg.audio_encoder.name = "hubert_large_ll60k"
    cfg.audio_encoder.normalize_hiddenstates = True
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


class KWClip_GeneralTransformer(nn.Module):
    def __init__(self, config, image_encoder: Optional[Callable] = None, device: str = "cuda", hubert_state_dict=None,
                 hubert_arch=None):
        super().__init__()
        self.config = config if isinstance(config, Config) else Config(config)
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
        if ms.cascaded_objective_weight > 0:
            raise NotImplementedError("cascaded / hybrid(+) branches are scope row f3: not built yet")
        if ms.parallel_objective_weight > 0:
            logger.info("Create Parallel Branch")
            self.parallel_branch = KW_ParallelBranch(config=config, audio_dim=self.audio_embd_dim,
                                                     text_dim=self.subword_embd_dim)
        self.img_enc_proj_net = None
        self.p_branch_proj_net = None
        self.c_branch_proj_net = None
        self.global_step = 0
        self.to(self._device)

    @property
    def device(self):
        return self._device

    # ---------------------------------------------------------------------------------------------
    def getTrainableParams(self) -> list:
        """kwClip.py:620-644 + :812-837."""
        _params = []
        _params += self.audio_encoder.trainable_params()
        _params += list(self.criterion.parameters())
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
        image_feat = image_feat / image_feat.norm(dim=-1, keepdim=True)
        output = self.parallel_branch(audio_feat=audio_feat, audio_feat_len=audio_feat_len)
        parallel_audio_feat = output["parallel_audio_feat"]
        cascaded_audio_feat = output["cascaded_audio_feat"]
        vq_results, keywords, dsample_results = output["vq_results"], output["keywords"], output["dsample_results"]
        keywords_len = None
        id = id.to(self._device)
        losses_ = {"id": id, "image_feat": image_feat}
        if parallel_audio_feat is not None:
            parallel_audio_feat = parallel_audio_feat / parallel_audio_feat.norm(dim=-1, keepdim=True)
            losses_["parallel_audio_feat"] = parallel_audio_feat
        log_metrics = {"cl_temp": self.criterion.current_temperature}
        return (losses_, log_metrics,
                {"id": id, "image_feat": image_feat, "parallel_audio_feat": parallel_audio_feat,
                 "cascaded_audio_feat": cascaded_audio_feat, "vq_results": vq_results, "keywords": keywords,
                 "dsample_results": dsample_results, "keywords_len": keywords_len})

    def compute_loss(self, inputDict: dict):
        """kwClip.py:999-1040."""
        assert isinstance(inputDict, dict)
        required_keys = {"id", "image_feat"}
        assert required_keys.issubset(set(inputDict.keys())), f"required: {required_keys}, input: {inputDict.keys()}"
        losses_ = {"loss": 0}
        image_feat = inputDict["image_feat"].float()
        id = inputDict["id"]
        for branchType in ["cascaded", "parallel"]:
            loss_weight = self.config.model_settings.get(f"{branchType}_objective_weight", 0.0)
            if loss_weight > 0.0:
                feats_key = f"{branchType}_audio_feat"
                assert feats_key in inputDict, f"{inputDict.keys()}"
                losses_[f"{branchType[0]}_cl_loss"] = self.criterion(feat_A=inputDict[feats_key].float(),
                                                                     feat_B=image_feat, index=id)
                losses_["loss"] += loss_weight * losses_[f"{branchType[0]}_cl_loss"]
        return losses_

    def training_step(self, batch: dict) -> dict:
        losses_, log_metrics = self.forward(batch)[:2]
        return {"loss_feats": losses_, "log_metrics": log_metrics}

    def training_step_end(self, outputs: dict) -> dict:
        """kwClip.py:149-193: the loss is computed on the gathered batch."""
        if "loss" in outputs:
            return {"loss": torch.mean(outputs["loss"])}
        losses_ = self.compute_loss(outputs["loss_feats"])
        return {"loss": losses_["loss"], **{f"train_{k}": v for k, v in losses_.items()}}

    # ---------------------------------------------------------------------------------------------
    def encode_speech(self, wav) -> dict:
        """kwClip.py:1042-1091 (un-normalised branch output)."""
        wav, wav_len = self.processWavs(wav)
        audio_feat, audio_feat_len = self.forward_audio(wav, wav_len)
        output = self.parallel_branch(audio_feat=audio_feat, audio_feat_len=audio_feat_len)
        return {"cascaded_audio_feat": output["cascaded_audio_feat"], "parallel_audio_feat": output["parallel_audio_feat"],
                "vq_results": output["vq_results"], "keywords": output["keywords"]}

    def feature_extractor_s3prl(self, wav) -> Tuple[torch.Tensor, Tuple]:
        """kwClip.py:965-997.  The parallel branch's own hidden states need the full-sequence head layer
        (scope row f3); the 13 HuBERT states are returned (cloned: they live in a reused workspace)."""
        wav, wav_len = self.processWavs(wav)
        audio_feat, audio_len, hidden_states = self.forward_audio(wav, wav_len, return_hidden_states=True)
        assert isinstance(hidden_states, tuple)
        hidden_states = tuple(h.clone() for h in hidden_states)
        return hidden_states[-1], hidden_states
