#!/usr/bin/env python3
"""Recall@k parity fixture (Flickr8k-test-shaped synthetic eval set, 1000 image ids x 5 utterances).  Round 3 version.

Runs in the build container on the CPU ORACLE, twice:
  fp32      the torch fp32 restatement of the reference maths (HuBERT-base with seeded weights -> weighted sum -> parallel branch)
  bf16emu   the same code with bf16 rounding at exactly the tensors the HIP path stores in bf16 and on the GEMM weights it holds in
            bf16 (oracle.bf16_store / oracle.bf16_weights) - the CONTROL that separates "what bf16 storage does to recall" from
            "what a kernel defect does" 
and builds the image gallery from the fp32 embeddings (tools/recall_eval.build_gallery: planted, discrete confusions).

Stored (tests/golden/recall_eval.npz): images, roles, per-utterance / per-image ranks of both references, the fp32 margins at the
rank boundaries 1 / 5 / 10 with the index of the k-th best other image (for the HIP-vs-oracle margin noise), the first 64 embeddings
of both references, the protocol's batch size.  tests/golden/recall_eval_margins.json: the margin histograms and the near-tie
accounting in readable form.  Waveforms and weights are NOT stored: tools/recall_eval.eval_set / hubert_weights / head_weights
regenerate them from seeds (CPU torch.Generator streams).

    python tests/golden/make_recall_fixture.py [--threads 8] [--cache DIR]     (~10 min per reference on 8 cores)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "tools")))
from recall_eval import (BATCH, GALLERY, PER_ID, WS_WEIGHTS, build_gallery, embed_all, eval_set, head_weights,  # noqa: E402
                         hubert_weights, rank_stats, recalls)


def oracle_embeddings(mode: str, wavs, cache: str = None) -> torch.Tensor:
    import oracle
    path = os.path.join(cache, f"new_{mode}.npy") if cache else None
    if path and os.path.exists(path):
        print("re-using", path)
        return torch.from_numpy(np.load(path))
    Wh, Whead, arch = hubert_weights(), head_weights(), oracle.HubertArch.base()
    store = None
    if mode == "emu":
        Wh, store = oracle.bf16_weights(Wh), oracle.bf16_store
    t0 = time.time()
    done = [0]

    def encode(ws):
        hs, fl = oracle.speech_encoder_forward(Wh, arch, ws, store=store)
        f = oracle.weighted_sum(WS_WEIGHTS, hs)
        if store is not None:
            f = store(f)                    # the weighted sum is stored in bf16 (csrc/rowops.hip wsum_fwd); the head's row tail is fp32
        done[0] += len(ws)
        if done[0] % 1000 < len(ws):
            print(f"{mode}: {done[0]} utterances, {time.time() - t0:.0f} s", flush=True)
        return oracle.parallel_branch_forward(Whead, f, fl, nhead=8)

    with torch.no_grad():
        emb = embed_all(encode, wavs, 512, BATCH)
    if path:
        np.save(path, emb.numpy())
    return emb


def histogram(x: torch.Tensor, edges) -> list:
    return [int(((x >= lo) & (x < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--cache", default=None, help="directory with / for new_fp32.npy, new_emu.npy (raw embeddings of the two references)")
    ap.add_argument("--out", default=os.path.join(HERE, "recall_eval.npz"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    wavs, ids = eval_set(args.ids)
    e32 = oracle_embeddings("fp32", wavs, args.cache)
    eem = oracle_embeddings("emu", wavs, args.cache)
    n = lambda x: x / x.norm(dim=-1, keepdim=True)
    a32, aem = n(e32), n(eem)
    image, role = build_gallery(a32, ids, args.ids)
    s32, sem = rank_stats(a32, image, ids), rank_stats(aem, image, ids)
    held = (torch.arange(len(ids)) % PER_ID) >= GALLERY
    # margin noise of the emulation against fp32 (the HIP path's is measured by the GPU test the same way)
    m_e = sem["own"].unsqueeze(1) - sem["scores"].gather(1, s32["kth_idx"])
    noise = m_e - s32["margin_ai"]
    sigma = float(noise.std())
    edges = [-1.0, -0.03, -0.01, -0.003, -0.0015, 0.0, 0.0015, 0.003, 0.01, 0.03, 1.0]
    summary = {
        "protocol": {"ids": args.ids, "captions_per_id": PER_ID, "gallery_captions": GALLERY, "batch": BATCH,
                     "scores": "dot products of unit vectors (kwClip.py:467-471)"},
        "roles": {"pure": int((role == 0).sum()), "carrier": int((role == 1).sum()),
                  "victims_by_competitors": {str(m): int((role == 1 + m).sum()) for m in (1, 3, 7, 14)}},
        "fp32": {"audio_to_image": recalls(s32["rank_ai"]), "audio_to_image_heldout": recalls(s32["rank_ai"], held),
                 "image_to_audio": recalls(s32["rank_ia"])},
        "bf16emu": {"audio_to_image": recalls(sem["rank_ai"]), "audio_to_image_heldout": recalls(sem["rank_ai"], held),
                    "image_to_audio": recalls(sem["rank_ia"]),
                    "rank_flips_vs_fp32_audio_to_image": [int(((s32["rank_ai"] < k) != (sem["rank_ai"] < k)).sum()) for k in (1, 5, 10)],
                    "rank_flips_vs_fp32_image_to_audio": [int(((s32["rank_ia"] < k) != (sem["rank_ia"] < k)).sum()) for k in (1, 5, 10)],
                    "embedding_distance_from_fp32_mean": float((aem - a32).norm(dim=-1).mean()),
                    "common_shift_norm": float((aem - a32).mean(0).norm()),
                    "margin_noise_sigma": sigma, "margin_noise_max": float(noise.abs().max())},
        "margin_histogram_edges": edges,
        "margins_fp32": {},
    }
    for i, k in enumerate((1, 5, 10)):
        m_ai, m_ia = s32["margin_ai"][:, i], s32["margin_ia"][:, i]
        summary["margins_fp32"][f"@{k}"] = {
            "audio_to_image_all": histogram(m_ai, edges), "audio_to_image_heldout": histogram(m_ai[held], edges),
            "image_to_audio": histogram(m_ia, edges),
            "smallest_abs_audio_to_image": float(m_ai.abs().min()), "smallest_abs_image_to_audio": float(m_ia.abs().min()),
            "heldout_fraction_within_3_sigma_of_emulation_noise": float((m_ai[held].abs() < 3 * sigma).float().mean()),
            "all_fraction_within_3_sigma_of_emulation_noise": float((m_ai.abs() < 3 * sigma).float().mean())}
    print(json.dumps(summary, indent=1))
    json.dump(summary, open(os.path.join(HERE, "recall_eval_margins.json"), "w"), indent=1)
    np.savez_compressed(
        args.out, image=image.numpy(), role=role.numpy().astype(np.int8), n_ids=np.int64(args.ids), batch=np.int64(BATCH),
        gallery=np.int64(GALLERY),
        rank_ai_fp32=s32["rank_ai"].numpy().astype(np.int16), rank_ia_fp32=s32["rank_ia"].numpy().astype(np.int16),
        rank_ai_bf16emu=sem["rank_ai"].numpy().astype(np.int16), rank_ia_bf16emu=sem["rank_ia"].numpy().astype(np.int16),
        margin_ai_fp32=s32["margin_ai"].numpy().astype(np.float32), kth_idx_fp32=s32["kth_idx"].numpy().astype(np.int16),
        margin_ia_fp32=s32["margin_ia"].numpy().astype(np.float32),
        emb_head_fp32=e32[:64].numpy(), emb_head_bf16emu=eem[:64].numpy())
    # the control as its own file too (VERDICT r02 names it): the emulated oracle's ranks / recalls / embeddings
    np.savez_compressed(os.path.join(HERE, "recall_eval_bf16emu.npz"), rank_ai=sem["rank_ai"].numpy().astype(np.int16),
                        rank_ia=sem["rank_ia"].numpy().astype(np.int16), emb_head=eem[:64].numpy(),
                        AB=np.array(recalls(sem["rank_ai"])), BA=np.array(recalls(sem["rank_ia"])),
                        AB_heldout=np.array(recalls(sem["rank_ai"], held)))
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
