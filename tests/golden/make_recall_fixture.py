#!/usr/bin/env python3
"""Recall@k parity fixture (Flickr8k-test-shaped synthetic eval set, 1000 image ids x 5 utterances).

Runs in the build container on the CPU ORACLE (torch fp32 restatement of the reference maths): HuBERT-base with seeded weights ->
weighted sum -> parallel branch -> 5000 audio embeddings; image embeddings are built FROM the oracle's audio embeddings (class
mean of the centred captions + seeded noise) so that recall@1 is neither 0 nor 100; recall@{1,5,10} in both directions with the
oracle's mutual_retrieval.  Stored (tests/golden/recall_eval.npz, ~2 MB): the image embeddings, the oracle's recalls, the rank of
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

SEED_W, SEED_HEAD, SEED_DATA = 7122, 7123, 20260
PER_ID = 5
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


def weights():
    """Seeded HuBERT-base + parallel-branch weights.  The CLS token is scaled to 0.1: at unit scale the residual path of the CLS
    row (a constant) dominates the pooled output and the embeddings of all utterances collapse onto one direction (|mean| =
    0.999), which makes every rank a near-tie; a trained head does not do that."""
    import oracle
    head = oracle.init_parallel_branch_weights(seed=SEED_HEAD)
    head["cls"] = head["cls"] * 0.1
    return oracle.init_hubert_weights(oracle.HubertArch.base(), seed=SEED_W), head


def images_from(a_o: torch.Tensor, ids: torch.Tensor, n_ids: int, sigma: float) -> torch.Tensor:
    mu = a_o.mean(0, keepdim=True)
    c = torch.stack([(a_o[ids == k] - mu).mean(0) for k in range(n_ids)])
    c = c / c.norm(dim=-1, keepdim=True)
    g = torch.Generator().manual_seed(SEED_DATA + 99)
    noise = torch.randn(n_ids, a_o.shape[1], generator=g) / a_o.shape[1] ** 0.5
    img = c + sigma * noise
    return img / img.norm(dim=-1, keepdim=True)


def centred_scores(a: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """audio x image scores as the validation epoch computes them (kwClip.py:467-471): plain dot products of unit vectors."""
    return a @ img.t()


def correct_rank(score: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    own = score.gather(1, ids.unsqueeze(1))
    return (score > own).sum(1)


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
    print("sigma", sigma, "A->I", AB, "I->A", BA)
    ks = [1, 5, 10]
    np.savez_compressed(args.out, image=img.numpy(), n_ids=np.int64(args.ids), sigma=np.float64(sigma),
                        rank=correct_rank(score, ids).numpy().astype(np.int32),
                        AB=np.array([AB[f"recall@{k}"] for k in ks]), BA=np.array([BA[f"recall@{k}"] for k in ks]),
                        mean=np.array([mean[f"recall@{k}"] for k in ks]), emb_head=emb[:64].numpy(),
                        margin=(score.gather(1, ids.unsqueeze(1)).squeeze(1) - score.masked_fill(
                            torch.nn.functional.one_hot(ids, args.ids).bool(), -9.0).max(1).values).numpy().astype(np.float32))
    print("wrote", args.out, os.path.getsize(args.out), "bytes;", f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
