#!/usr/bin/env python3
"""Which bf16 STORAGE SITE moves the embeddings / the recall decisions?  (CPU only, oracle only.)

The bf16-storage-emulated oracle (oracle/hubert_ref.py: bf16_weights + store hooks) sits 5.6e-3 of the unit norm from the fp32 oracle on
the 5000-utterance eval set, exactly like the HIP path; on the natural-margin galleries that moves 79 / 66 / 61 rank-1 / 5 / 10 decisions.
This tool switches the rounding on for ONE site class at a time (oracle.SitedStore) and tables, per class, against the fp32 oracle:
  mean embedding distance, its part common to all utterances, the per-utterance rest, and the rank flips on gallery A
  (tests/golden/recall_eval_natural.npz) and gallery B (recall_eval_natural_b.npz), audio -> image and image -> audio.
Configs: fp32 | all (= the emulation) | weights | conv | ln | proj | residual | qkv | p | ctx | ffn_act | wsum, and any "+"-joined set;
weight GROUPS (round 6, after "weights" turned out to carry 0.0052 of the 0.0057): w_conv (conv layers 1-6) | w_proj (post_extract_proj,
pos_conv) | w_attn (q / k / v / out projections) | w_ffn (fc1, fc2) | w_lo (encoder layers 0-5) | w_hi (layers 6-11).

    python tools/storage_ablation.py --cache /tmp/ablation --threads 6 [--configs weights conv ...] [--out profiles/r06_storage_ablation.json]

Embeddings are cached per config (resumable); the JSON is rewritten after every config.  ~10 min per config on 8 cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from recall_eval import BATCH, WS_WEIGHTS, embed_all, eval_set, head_weights, hubert_weights, rank_stats, recalls  # noqa: E402

SITES = ("conv", "ln", "proj", "residual", "qkv", "p", "ctx", "ffn_act")
_layer = lambda k: int(k.split(".")[2]) if k.startswith("encoder.layers.") else -1
WGROUPS = {"w_conv": lambda k: k.startswith("feature_extractor."), "w_proj": lambda k: k.startswith(("post_extract_proj.", "encoder.pos_conv.")),
           "w_attn": lambda k: ".self_attn." in k, "w_ffn": lambda k: ".fc1." in k or ".fc2." in k,
           "w_lo": lambda k: 0 <= _layer(k) <= 5, "w_hi": lambda k: _layer(k) >= 6}
DEFAULT = ["fp32", "all", "weights", "conv", "ln", "proj", "residual", "qkv", "p", "ctx", "ffn_act", "wsum"]


def embeddings(config: str, wavs, cache: str) -> torch.Tensor:
    import oracle
    path = os.path.join(cache, f"{config}.npy")
    if os.path.exists(path):
        return torch.from_numpy(np.load(path))
    on = set(SITES) | {"weights", "wsum"} if config == "all" else set() if config == "fp32" else set(config.split("+"))
    assert on <= set(SITES) | {"weights", "wsum"} | set(WGROUPS), on
    Wh, Whead, arch = hubert_weights(), head_weights(), oracle.HubertArch.base()
    if "weights" in on:
        Wh = oracle.bf16_weights(Wh)
    for grp in on & set(WGROUPS):            # the bf16 rounding on ONE group of GEMM weights
        rounded = oracle.bf16_weights(Wh)
        Wh = {k: (rounded[k] if WGROUPS[grp](k) else v) for k, v in Wh.items()}
    sites = on & set(SITES)
    store = oracle.SitedStore(sites) if sites else None
    t0, done = time.time(), [0]

    def encode(ws):
        hs, fl = oracle.speech_encoder_forward(Wh, arch, ws, store=store)
        f = oracle.weighted_sum(WS_WEIGHTS, hs)
        if "wsum" in on:
            f = oracle.bf16_store(f)
        done[0] += len(ws)
        if done[0] % 1000 < len(ws):
            print(f"{config}: {done[0]} utterances, {time.time() - t0:.0f} s", flush=True)
        return oracle.parallel_branch_forward(Whead, f, fl, nhead=8)

    with torch.no_grad():
        emb = embed_all(encode, wavs, 512, BATCH)
    np.save(path, emb.numpy())
    return emb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cache", default="/tmp/ablation")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--configs", nargs="*", default=DEFAULT)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_storage_ablation.json"))
    args = ap.parse_args()
    os.makedirs(args.cache, exist_ok=True)
    torch.set_num_threads(args.threads)
    wavs, ids = eval_set(1000)
    unit = lambda x: x / x.norm(dim=-1, keepdim=True)
    galleries = {}
    for tag, name in (("A", "recall_eval_natural.npz"), ("B", "recall_eval_natural_b.npz")):
        fx = np.load(os.path.join(ROOT, "tests", "golden", name))
        galleries[tag] = torch.from_numpy(fx["image"])
    a32 = unit(embeddings("fp32", wavs, args.cache))
    base = {g: rank_stats(a32, img, ids) for g, img in galleries.items()}
    flips = lambda a, b: [int(((a < k) != (b < k)).sum()) for k in (1, 5, 10)]
    report = {"eval_set": "1000 ids x 5 utterances, tools/recall_eval.py; batches of %d length-sorted" % BATCH,
              "fp32": {g: {"audio_to_image": recalls(base[g]["rank_ai"]), "image_to_audio": recalls(base[g]["rank_ia"])} for g in galleries},
              "configs": {}}
    for cfg in args.configs:
        if cfg == "fp32":
            continue
        a = unit(embeddings(cfg, wavs, args.cache))
        d = a - a32
        row = {"embedding_distance_mean": float(d.norm(dim=-1).mean()), "common_shift_norm": float(d.mean(0).norm()),
               "per_utterance_rest_mean": float((d - d.mean(0)).norm(dim=-1).mean())}
        for g, img in galleries.items():
            st = rank_stats(a, img, ids)
            noise = (st["own"].unsqueeze(1) - st["scores"].gather(1, base[g]["kth_idx"])) - base[g]["margin_ai"]
            row["gallery_" + g] = {"audio_to_image": recalls(st["rank_ai"]), "image_to_audio": recalls(st["rank_ia"]),
                                   "flips_audio_to_image": flips(base[g]["rank_ai"], st["rank_ai"]),
                                   "flips_image_to_audio": flips(base[g]["rank_ia"], st["rank_ia"]),
                                   "margin_noise_sigma": float(noise.std())}
        report["configs"][cfg] = row
        print(cfg, json.dumps(row), flush=True)
        json.dump(report, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
