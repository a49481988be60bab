"""Reference yaml recipes -> the attribute-style config the model reads (avssl/task/base_task.py:80-82 merges argparse + yaml into an
OrderedNamespace; here: ``yaml.safe_load`` into ``Config``).  The yaml files under /root/reference/config/** parse unchanged; only
the keys on the hot path are interpreted, everything else (data, trainer, logger, log_setting) is carried along untouched.

Normalisations applied on load:
  * ``clip.embed_dim`` from ``clip.name`` (512 for ViT-B/32, 768 for ViT-L/14) when absent;
  * ``clip.reduce_subword_embbedding``: the reference points at avssl/data/{flickr,coco}_stat/text_clip_vocab_usage_byfreq.npy
    (8112 x 2 / 19787 x 2 int64).  When that file is not reachable (it does not travel with this package) a synthetic table of
    the same size replaces it, with a warning;
  * ``audio_encoder.name`` aliases (``hubert_base`` -> the same architecture as ``hubert``).
"""
import logging
import os
from typing import Union

import yaml

logger = logging.getLogger(__name__)

CLIP_EMBED_DIM = {"ViT-B/32": 512, "ViT-B/16": 512, "ViT-L/14": 768}
REDUCED_VOCAB_SIZE = {"flickr": 8112, "coco": 19787}


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


def synthetic_reduced_vocab(n: int = 8112, seed: int = 0):
    """Stand-in for avssl/data/flickr_stat/text_clip_vocab_usage_byfreq.npy (8112 sub-words; 19787 for coco), which
    does not travel to the GPU box: n distinct CLIP token ids that contain <|startoftext|> and <|endoftext|>."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.randperm(49406, generator=g)[: n - 2]
    return torch.cat([ids, torch.tensor([49406, 49407])])


def load_config(src: Union[str, dict], reference_root: str = ".", allow_synthetic_vocab: bool = False) -> Config:
    """``src``: path of a reference yaml recipe, yaml text, or an already parsed dict.  A recipe's reduced-vocabulary table
    (``clip.reduce_subword_embbedding``, a .npy of CLIP token ids under the reference's ``avssl/data/*_stat/``) must exist under
    ``reference_root``: with a real checkpoint a made-up reduced-index -> token-id table would silently decode wrong keywords.
    ``allow_synthetic_vocab=True`` (benchmarks and tests on random weights) substitutes a synthetic table of the same size."""
    if isinstance(src, dict):
        cfg = Config(src)
    else:
        text = open(src).read() if os.path.exists(src) else src
        cfg = Config(yaml.safe_load(text))
    for key in ("model_settings", "cl_loss", "audio_encoder"):
        if key not in cfg:
            raise KeyError(f"config has no '{key}' section")
    cfg.setdefault("retrieval", Config({"audio_feat_src": "parallel", "recall_at": [1, 5, 10]}))
    cfg.setdefault("trainer", Config({"gradient_clip_val": 0.0, "accumulate_grad_batches": 1}))
    clip = cfg.setdefault("clip", Config({"name": "ViT-B/32"}))
    if "embed_dim" not in clip:
        clip["embed_dim"] = CLIP_EMBED_DIM[clip.get("name", "ViT-B/32")]
    vocab = clip.get("reduce_subword_embbedding", None)
    if isinstance(vocab, str):
        path = vocab if os.path.isabs(vocab) else os.path.join(reference_root, vocab)
        if os.path.exists(path):
            clip["reduce_subword_embbedding"] = path
        elif not allow_synthetic_vocab:
            raise FileNotFoundError(f"reduced-vocabulary table {path} not found (clip.reduce_subword_embbedding = {vocab!r}): point "
                                    "reference_root at the reference checkout, or pass allow_synthetic_vocab=True when running "
                                    "on random weights")
        else:
            stat = "coco" if "coco" in vocab else "flickr"
            logger.warning("reduced-vocabulary table %s not found: using a synthetic table of %d sub-words", vocab,
                           REDUCED_VOCAB_SIZE[stat])
            clip["reduce_subword_embbedding"] = synthetic_reduced_vocab(REDUCED_VOCAB_SIZE[stat])
    ms = cfg.model_settings
    ms.setdefault("cascaded_objective_weight", 0.0)
    ms.setdefault("parallel_objective_weight", 0.0)
    return cfg
