#!/usr/bin/env python3
"""Second recall@k fixture (round 4, VERDICT r03 item 3 / ADVICE r03): the SAME 5000 utterances, weights and protocol as
make_recall_fixture.py, but an image gallery with NATURAL margins - nothing planted.

    image_k = unit( P (c_k / |c_k| + s * eps_k) ),   c_k = centre of id k's three gallery captions (fp32 oracle embeddings, centred),
    eps_k ~ N(0, I / E) seeded, P = projection orthogonal to the mean embedding, s = the noise level at which the fp32 oracle's
    audio -> image recall@1 is closest to 50 % (bisection, deterministic; stored)

so the own-minus-best-other margins are continuously distributed through zero (round 2's construction) and a fraction of the
decisions lies inside the noise of ANY bf16-storage implementation.  On such a set "recall@1 within 0.1" is not a meaningful bar;
what can be required (tests/test_gpu_recall.py::test_recall_on_natural_margins) is that
  * every rank decision the HIP model takes differently from the fp32 oracle has an oracle margin below 4 sigma of the margin noise
    that the bf16-storage-EMULATED oracle shows against fp32 (a flip at a resolvable margin would be a defect), and
  * the HIP model flips no more decisions against the emulated oracle than the emulated oracle flips against fp32 (x 1.25).
Stored (tests/golden/recall_eval_natural.npz): images, fp32 / emulated ranks, the fp32 margins + k-th-best indices at the rank
boundaries 1 / 5 / 10 (both directions), the emulation's margin-noise sigma, its flip counts, the first 64 embeddings of both references.

    python tests/golden/make_recall_natural_fixture.py [--threads 8] [--cache DIR]     (the oracle embeddings: ~10 min per reference on 8 cores)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "tools")))
from make_recall_fixture import oracle_embeddings  # noqa: E402
from recall_eval import BATCH, GALLERY, PER_ID, SEED_DATA, eval_set, natural_gallery, rank_stats, recalls  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--cache", default=None, help="directory with / for new_fp32.npy, new_emu.npy (raw embeddings of the two references)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=None, help="seed of the gallery's isotropic noise (default: recall_eval.natural_gallery's own, "
                    "gallery A = recall_eval_natural.npz).  Round 5: --seed 20261005 = gallery B (recall_eval_natural_b.npz), drawn AFTER the "
                    "acceptance criterion of tests/test_gpu_recall.py::test_recall_on_natural_margins was written down")
    args = ap.parse_args()
    suffix = "" if args.seed is None else "_b"
    if args.out is None:
        args.out = os.path.join(HERE, f"recall_eval_natural{suffix}.npz")
    gkw = {} if args.seed is None else {"seed": args.seed}
    torch.set_num_threads(args.threads)
    wavs, ids = eval_set(args.ids)
    e32 = oracle_embeddings("fp32", wavs, args.cache)
    eem = oracle_embeddings("emu", wavs, args.cache)
    n = lambda x: x / x.norm(dim=-1, keepdim=True)
    a32, aem = n(e32), n(eem)
    # noise level: bisection on the fp32 oracle's recall@1 (monotone in s up to sampling noise; 24 steps, deterministic)
    lo, hi = 0.0, 8.0
    for _ in range(24):
        s = 0.5 * (lo + hi)
        r1 = recalls(rank_stats(a32, natural_gallery(a32, ids, args.ids, s, **gkw), ids)["rank_ai"])[0]
        lo, hi = (s, hi) if r1 > 50.0 else (lo, s)
    s = 0.5 * (lo + hi)
    image = natural_gallery(a32, ids, args.ids, s, **gkw)
    s32, sem = rank_stats(a32, image, ids), rank_stats(aem, image, ids)
    held = (torch.arange(len(ids)) % PER_ID) >= GALLERY
    noise_ai = (sem["own"].unsqueeze(1) - sem["scores"].gather(1, s32["kth_idx"])) - s32["margin_ai"]
    sigma = float(noise_ai.std())
    sigma_ia = float((sem["margin_ia"] - s32["margin_ia"]).std())      # image -> audio: own-best minus k-th foreign, each at its own ranks
    flips = lambda a, b: [int(((a < k) != (b < k)).sum()) for k in (1, 5, 10)]
    summary = {
        "protocol": {"ids": args.ids, "captions_per_id": PER_ID, "gallery_captions": GALLERY, "batch": BATCH, "noise_level": s,
                     "gallery": "class centre + isotropic noise, orthogonal to the mean embedding; nothing planted"},
        "fp32": {"audio_to_image": recalls(s32["rank_ai"]), "audio_to_image_heldout": recalls(s32["rank_ai"], held),
                 "image_to_audio": recalls(s32["rank_ia"])},
        "bf16emu": {"audio_to_image": recalls(sem["rank_ai"]), "audio_to_image_heldout": recalls(sem["rank_ai"], held),
                    "image_to_audio": recalls(sem["rank_ia"]),
                    "rank_flips_vs_fp32_audio_to_image": flips(s32["rank_ai"], sem["rank_ai"]),
                    "rank_flips_vs_fp32_image_to_audio": flips(s32["rank_ia"], sem["rank_ia"]),
                    "margin_noise_sigma": sigma, "margin_noise_max": float(noise_ai.abs().max()),
                    "margin_noise_sigma_image_to_audio": sigma_ia},
        "fraction_of_queries_within_3_sigma_at_1_5_10": [float((s32["margin_ai"][:, i].abs() < 3 * sigma).float().mean()) for i in range(3)],
        "largest_fp32_margin_of_an_emulation_flip_in_sigma": [
            float((s32["margin_ai"][:, i].abs()[(s32["rank_ai"] < k) != (sem["rank_ai"] < k)].max() / sigma)
                  if ((s32["rank_ai"] < k) != (sem["rank_ai"] < k)).any() else 0.0) for i, k in enumerate((1, 5, 10))],
        "largest_fp32_margin_of_an_emulation_flip_in_sigma_image_to_audio": [
            float((s32["margin_ia"][:, i].abs()[(s32["rank_ia"] < k) != (sem["rank_ia"] < k)].max() / sigma_ia)
                  if ((s32["rank_ia"] < k) != (sem["rank_ia"] < k)).any() else 0.0) for i, k in enumerate((1, 5, 10))],
    }
    print(json.dumps(summary, indent=1))
    json.dump(summary, open(os.path.join(HERE, f"recall_eval_natural{suffix}_margins.json"), "w"), indent=1)
    np.savez_compressed(
        args.out, image=image.numpy(), n_ids=np.int64(args.ids), batch=np.int64(BATCH), gallery=np.int64(GALLERY), noise_level=np.float64(s),
        rank_ai_fp32=s32["rank_ai"].numpy().astype(np.int16), rank_ia_fp32=s32["rank_ia"].numpy().astype(np.int16),
        rank_ai_bf16emu=sem["rank_ai"].numpy().astype(np.int16), rank_ia_bf16emu=sem["rank_ia"].numpy().astype(np.int16),
        margin_ai_fp32=s32["margin_ai"].numpy().astype(np.float32), kth_idx_fp32=s32["kth_idx"].numpy().astype(np.int16),
        margin_ia_fp32=s32["margin_ia"].numpy().astype(np.float32), sigma_bf16emu=np.float64(sigma), sigma_ia_bf16emu=np.float64(sigma_ia),
        emb_head_fp32=e32[:64].numpy(), emb_head_bf16emu=eem[:64].numpy())
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
