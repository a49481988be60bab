#!/usr/bin/env python3
"""Recall@k parity fixture (Flickr8k-test-shaped synthetic eval set, 1000 image ids x 5 utterances).

Runs in the build container on the CPU ORACLE (torch fp32 restatement of the reference maths): HuBERT-base with seeded weights ->
weighted sum -> parallel branch -> 5000 audio embeddings; image embeddings are built FROM the oracle's audio embeddings (class
mean of the centred first three captions + seeded noise; captions 3 and 4 are held out of the construction) so that recall@1
is neither 0 nor 100; recall@{1,5,10} in both directions with the oracle's mutual_retrieval, over all 5000 queries (the
reference's protocol) and over the 2000 held-out queries (unbiased HIP-vs-oracle comparison).  Stored (tests/golden/recall_eval.npz, ~2 MB): the image embeddings, the oracle's recalls, the rank of
the correct image for every utterance (rank-flip accounting), the first 64 oracle audio embeddings (cosine spot check) and the
generation parameters.  Waveforms and weights are NOT stored: `eval_set()` regenerates them from the seeds (CPU torch.Generator
streams: identical on every host with this torch build), for the GPU test and for bench.py's `recall` field.

    python tests/golden/make_recall_fixture.py [--ids 1000] [--threads 8]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "tools")))
from recall_eval import (GALLERY, PER_ID, WS_WEIGHTS, correct_rank, eval_set, head_weights, hubert_weights,  # noqa: E402
                         images_from)


def weights():
    return hubert_weights(), head_weights()


def centred_scores(a: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """audio x image scores as the validation epoch computes them (kwClip.py:467-471): plain dot products of unit vectors."""
    return a @ img.t()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--sigma", type=float, default=None, help="image noise; default: tuned so that A->I recall@1 is 35-65 %%")
    ap.add_argument("--out", default=os.path.join(HERE, "recall_eval.npz"))
    args = ap.parse_args()
    import oracle
    torch.set_num_threads(args.threads)
    Wh, Whead = weights()
    arch = oracle.HubertArch.base()
    wavs, ids = eval_set(args.ids)
    order = sorted(range(len(wavs)), key=lambda i: len(wavs[i]))            # length-sorted batches: little padding
    emb = torch.zeros(len(wavs), 512)
    t0 = time.time()
    with torch.no_grad():
        for s in range(0, len(order), 40):
            sel = order[s: s + 40]
            hs, fl = oracle.speech_encoder_forward(Wh, arch, [wavs[i] for i in sel])
            e = oracle.parallel_branch_forward(Whead, oracle.weighted_sum(WS_WEIGHTS, hs), fl, nhead=8)
            emb[sel] = e
            if (s // 40) % 10 == 0:
                print(f"{s + len(sel)} / {len(order)} utterances, {time.time() - t0:.0f} s", flush=True)
    a_o = emb / emb.norm(dim=-1, keepdim=True)
    sigma = args.sigma
    if sigma is None:                                # bisection on the noise level for a useful operating point
        lo, hi = 0.0, 8.0
        for _ in range(14):
            sigma = 0.5 * (lo + hi)
            sc = centred_scores(a_o, images_from(a_o, ids, args.ids, sigma))
            r1 = float((correct_rank(sc, ids) == 0).float().mean())
            lo, hi = (sigma, hi) if r1 > 0.5 else (lo, sigma)
    img = images_from(a_o, ids, args.ids, sigma)
    score = centred_scores(a_o, img)
    img_ids = torch.arange(args.ids)
    AB, BA, mean = oracle.mutual_retrieval(score, score.t(), ids, img_ids, [1, 5, 10])
    held = (torch.arange(len(ids)) % PER_ID) >= GALLERY
    AB_h, _, _ = oracle.mutual_retrieval(score[held], score[held].t(), ids[held], img_ids, [1, 5, 10])
    print("sigma", sigma, "A->I", AB, "I->A", BA, "held-out A->I", AB_h)
    ks = [1, 5, 10]
    np.savez_compressed(args.out, image=img.numpy(), n_ids=np.int64(args.ids), sigma=np.float64(sigma),
                        rank=correct_rank(score, ids).numpy().astype(np.int32),
                        AB=np.array([AB[f"recall@{k}"] for k in ks]), BA=np.array([BA[f"recall@{k}"] for k in ks]),
                        mean=np.array([mean[f"recall@{k}"] for k in ks]), emb_head=emb[:64].numpy(),
                        AB_heldout=np.array([AB_h[f"recall@{k}"] for k in ks]), gallery=np.int64(GALLERY),
                        margin=(score.gather(1, ids.unsqueeze(1)).squeeze(1) - score.masked_fill(
                            torch.nn.functional.one_hot(ids, args.ids).bool(), -9.0).max(1).values).numpy().astype(np.float32))
    print("wrote", args.out, os.path.getsize(args.out), "bytes;", f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
