"""Recall@k parity on the Flickr8k-test-shaped synthetic eval set (1000 image ids x 5 utterances), shared by
tests/test_gpu_recall.py, bench.py's ``recall`` field and the fixture generator tests/golden/make_recall_fixture.py.

Everything here is seeded and oracle-free: waveforms and weights come from CPU torch.Generator streams (identical on every host
with this torch build), so the only thing that has to travel from the build container is tests/golden/recall_eval.npz (image
embeddings + the oracle's ranks / recalls).  The weight generators draw the same streams as oracle.init_hubert_weights /
oracle.init_parallel_branch_weights (tests/test_host_cpu.py checks that), without importing the oracle.
"""
import os
from typing import Dict

import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
FIXTURE = os.path.join(ROOT, "tests", "golden", "recall_eval.npz")
SEED_W, SEED_HEAD, SEED_DATA = 7122, 7123, 20260
PER_ID = 5
GALLERY = 3      # captions 0..2 of every id build its image; captions 3, 4 are held out of the construction
WS_WEIGHTS = torch.linspace(-1, 1, 13)


def utterance(k: int, j: int) -> torch.Tensor:
    """Caption j of image id k: a per-id base signal (1.5 - 2.5 s) plus per-caption noise, ragged length."""
    g = torch.Generator().manual_seed(SEED_DATA + k)
    L = int(torch.randint(24000, 40001, (1,), generator=g))
    base = torch.randn(L, generator=g)
    gj = torch.Generator().manual_seed(SEED_DATA * 7 + k * PER_ID + j)
    lj = L - int(torch.randint(0, 4001, (1,), generator=gj))
    return 0.6 * base[:lj] + 0.5 * torch.randn(lj, generator=gj)


def eval_set(n_ids: int):
    """-> (list of n_ids * 5 waveforms, ids [n_ids * 5])"""
    wavs = [utterance(k, j) for k in range(n_ids) for j in range(PER_ID)]
    return wavs, torch.arange(n_ids).repeat_interleave(PER_ID)


def head_weights(d_model: int = 768, ffn: int = 3072, out_dim: int = 512, seed: int = SEED_HEAD) -> Dict[str, torch.Tensor]:
    """Seeded parallel-branch weights under the reference's state-dict names (1 post-LN layer).  The CLS token is scaled to 0.1: at
    unit scale the residual path of the CLS row (a constant) dominates the pooled output and the embeddings of all utterances
    collapse onto one direction (|mean| = 0.999), which makes every rank a near-tie; a trained head does not do that."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def randn(*shape, s):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * s

    W = {"cls": randn(1, 1, d_model, s=1.0)}
    p = "self_att.model.layers.0."
    W[p + "self_attn.in_proj_weight"] = randn(3 * d_model, d_model, s=d_model ** -0.5)
    W[p + "self_attn.in_proj_bias"] = randn(3 * d_model, s=0.05)
    W[p + "self_attn.out_proj.weight"] = randn(d_model, d_model, s=d_model ** -0.5)
    W[p + "self_attn.out_proj.bias"] = randn(d_model, s=0.05)
    W[p + "linear1.weight"] = randn(ffn, d_model, s=d_model ** -0.5)
    W[p + "linear1.bias"] = randn(ffn, s=0.05)
    W[p + "linear2.weight"] = randn(d_model, ffn, s=ffn ** -0.5)
    W[p + "linear2.bias"] = randn(d_model, s=0.05)
    for n in ("norm1", "norm2"):
        W[p + n + ".weight"] = 1.0 + randn(d_model, s=0.1)
        W[p + n + ".bias"] = randn(d_model, s=0.1)
    W["self_att.model.norm.weight"] = 1.0 + randn(d_model, s=0.1)
    W["self_att.model.norm.bias"] = randn(d_model, s=0.1)
    W["linear_proj.weight"] = randn(out_dim, d_model, s=d_model ** -0.5)
    W["linear_proj.bias"] = randn(out_dim, s=0.05)
    W["cls"] = W["cls"] * 0.1
    return W


def hubert_weights() -> Dict[str, torch.Tensor]:
    from speechclip_plus_amd import HubertArch, random_hubert_state_dict
    return random_hubert_state_dict(HubertArch(), seed=SEED_W)


def images_from(a_o: torch.Tensor, ids: torch.Tensor, n_ids: int, sigma: float) -> torch.Tensor:
    """Image of id k = centred mean of the (oracle) embeddings of its first GALLERY captions + seeded noise.  The held-out captions
    never enter an image: on them the oracle has no home advantage (its own rounding noise is not part of the target), so the
    HIP-vs-oracle comparison on the held-out queries is unbiased."""
    mu = a_o.mean(0, keepdim=True)
    cap = torch.arange(len(ids)) % PER_ID
    c = torch.stack([(a_o[(ids == k) & (cap < GALLERY)] - mu).mean(0) for k in range(n_ids)])
    c = c / c.norm(dim=-1, keepdim=True)
    g = torch.Generator().manual_seed(SEED_DATA + 99)
    noise = torch.randn(n_ids, a_o.shape[1], generator=g) / a_o.shape[1] ** 0.5
    img = c + sigma * noise
    return img / img.norm(dim=-1, keepdim=True)


def correct_rank(score: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """number of images scoring strictly above the utterance's own image"""
    own = score.gather(1, ids.unsqueeze(1))
    return (score > own).sum(1)


def build_model(device: str = "cuda:0"):
    """The product model (Parallel SpeechCLIP base) on the eval set's seeded weights."""
    from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    model = KWClip_GeneralTransformer(cfg, device=device, hubert_state_dict=hubert_weights()).eval()
    model.parallel_branch.load_state_dict(head_weights(), strict=True)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(WS_WEIGHTS)
    return model


def hip_recall(model, fixture: dict, batch: int = 40, dump: str = None) -> dict:
    """Embeds the 5000 utterances with ``model`` and compares recall@{1,5,10} with the oracle numbers held by the fixture."""
    import numpy as np
    from speechclip_plus_amd import mutualRetrieval
    n_ids = int(fixture["n_ids"])
    image = torch.from_numpy(np.asarray(fixture["image"]))
    dev = next(model.parameters()).device
    wavs, ids = eval_set(n_ids)
    order = sorted(range(len(wavs)), key=lambda i: len(wavs[i]))
    emb = torch.zeros(len(wavs), image.shape[1])
    with torch.no_grad():
        for s in range(0, len(order), batch):
            sel = order[s: s + batch]
            emb[sel] = model.encode_speech([wavs[i].to(dev) for i in sel])["parallel_audio_feat"].float().cpu()
    if dump:
        import numpy as _np
        _np.save(dump, emb.numpy())
    a = F.normalize(emb, dim=-1)
    score = a.to(dev) @ image.to(dev).t()
    img_ids = torch.arange(n_ids, device=dev)
    AB, BA, mean = mutualRetrieval(score, score.t(), ids.to(dev), img_ids, [1, 5, 10])
    held = ((torch.arange(len(ids)) % PER_ID) >= GALLERY).to(dev)
    AB_h, _, _ = mutualRetrieval(score[held], score[held].t(), ids.to(dev)[held], img_ids, [1, 5, 10])
    rank = correct_rank(score.cpu(), ids)
    rank_o = torch.from_numpy(np.asarray(fixture["rank"]).astype("int64"))
    margin = torch.from_numpy(np.asarray(fixture["margin"]))
    flip1 = (rank == 0) != (rank_o == 0)
    ks = [1, 5, 10]
    cos = F.cosine_similarity(emb[:64], torch.from_numpy(np.asarray(fixture["emb_head"])), dim=-1)
    return {
        "queries": len(ids), "images": n_ids,
        "audio_to_image": {"hip": [round(AB[f"recall@{k}"], 2) for k in ks], "oracle": [round(float(v), 2) for v in fixture["AB"]]},
        "image_to_audio": {"hip": [round(BA[f"recall@{k}"], 2) for k in ks], "oracle": [round(float(v), 2) for v in fixture["BA"]]},
        "audio_to_image_heldout": {"hip": [round(AB_h[f"recall@{k}"], 2) for k in ks],
                                   "oracle": [round(float(v), 2) for v in fixture["AB_heldout"]], "queries": int(held.sum())},
        "recall_at": ks, "rank1_flips": int(flip1.sum()),
        "rank1_flips_heldout": int(flip1[held.cpu()].sum()),
        "worst_oracle_margin_of_a_flip": round(float(margin[flip1].abs().max()) if flip1.any() else 0.0, 5),
        "embedding_cosine_min": round(float(cos.min()), 6),
    }
