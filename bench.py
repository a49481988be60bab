#!/usr/bin/env python3
"""bench.py - utterances/s of one contrastive train step (frozen-HuBERT parallel-base recipe) on N MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment), or started plainly - then this process, BEFORE any GPU call, starts the N ranks as a child
`python -m torch.distributed.run` (127.0.0.1 rendezvous), relays rank 0's JSON line and exits with the child's code.
The parent never touches the GPU.

Workload (BASELINE.json configs[1]): Parallel SpeechCLIP base, bf16, batch 64 per GPU x 10 s synthetic audio
(L = 160000 -> T = 499 frames), frozen HuBERT-base forward + weighted sum + CLS attention-pooling head
forward/backward + global-batch InfoNCE forward/backward + clip + Adam; weak scaling (per-GPU batch fixed),
RCCL all-gather of the pooled embeddings + all-reduce of the flat gradient for N > 1.
Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     the dominant kernel (the bf16 MFMA GEMM variant with the largest share of the step): algorithmic
                  TFLOP/s from a HIP-event pair around every launch (recorded on the launch stream) over the same K
                  steps, against the 2.5 PFLOP/s dense bf16 peak
  "cpu_baseline": the CPU oracle (torch fp32 restatement of the reference maths) timed on the host cores of
                  rank 0 on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# before the HIP runtime initialises (inherited by the ranks torch.distributed.run starts): see speechclip_plus_amd/__init__.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: ~2.5 PF dense bf16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--ragged", action="store_true", help="SURVEY 8d's variable-length run: utterance lengths U{32000 .. --seconds * 16000} "
                    "(seeded) instead of all-equal; algorithmic flops then count every utterance at ITS OWN length (padding is not work)")
    ap.add_argument("--cpu-utts", type=int, default=8, help="utterances in the CPU-baseline sample: BASELINE configs[0] = batch 8 (0 = skip)")
    ap.add_argument("--cpu-iters", type=int, default=10, help="timed CPU iterations (after 3 warm-up): SURVEY 8d's 3 + 10")
    ap.add_argument("--cpu-full", action="store_true", help="also time the CPU baseline at torch's default thread count and at 8 threads "
                    "(several minutes more; the default is SURVEY 8d's 3 + 10 at 32 threads, the fastest setting on every host measured)")
    ap.add_argument("--no-recall", action="store_true", help="skip the recall@k parity field (5000-utterance synthetic eval set)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-dropout", action="store_true", help="deterministic train step: every dropout site off (A/B only; the "
                    "reference's train step runs HuBERT's and the head's dropout, which is the default here)")
    ap.add_argument("--gemm-shapes", action="store_true", help="print a per-shape GEMM timing table to stderr")
    ap.add_argument("--unfreeze", type=int, default=0, help="train the top K HuBERT transformer layers too (audio_encoder.trainable "
                    "+ unfreeze_layers; NOT the headline configuration, which freezes HuBERT like every shipped recipe)")
    ap.add_argument("--trainable", action="store_true", help="audio_encoder.trainable: true - the WHOLE HuBERT trains (conv extractor, "
                    "projection, pos_conv, every layer: fwd + bwd + Adam); NOT the headline configuration")
    ap.add_argument("--model", choices=["base", "large", "cascaded_plus", "hybrid_plus_large"], default="base",
                    help="base = BASELINE configs[1] (headline); large = Parallel large; cascaded_plus = configs[2]; "
                         "hybrid_plus_large = configs[4] recipe on one GPU")
    ap.add_argument("--no-recipes", action="store_true", help="skip the `recipes` object of the default N = 1 run (BASELINE configs[2] "
                    "cascaded+ base and configs[4] hybrid+ large on one GPU, the ragged batch and the 6.4 s training crop: 10 steps each)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live HBM-traffic counters of the default N = 1 run (two short child "
                    "runs of this script's train step under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`, started before this process "
                    "touches the GPU); roofline.traffic then falls back to the newest committed profiles/*_traffic.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)     # the counter passes' target: warm-up + K steps, no output
    ap.add_argument("--rehearse-launch", action="store_true", help="launch plumbing only, no GPU: the ranks rendezvous, issue the "
                    "step's two collectives (packed all-gather, flat all-reduce) on CPU tensors and rank 0 prints the JSON line "
                    "with value null (tests/test_dp_gloo.py runs this with SC_DIST_BACKEND=gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse_launch:
        return rehearse_launch(args, world)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    default_run = (world == 1 and args.model == "base" and not args.ragged and not args.unfreeze and not args.trainable
                   and not args.no_dropout and abs(args.seconds - 10.0) < 1e-9 and args.batch == 64)
    live_traffic = rccl_rehearsal = None
    if default_run and not args.no_recipes and not args.pmc_child and os.environ.get("SC_FORCE_COLLECTIVES", "0") != "1":
        # child processes, run to completion BEFORE this process makes its first GPU call
        if not args.no_pmc:
            live_traffic = live_pmc_traffic()
        rccl_rehearsal = rccl_one_rank_rehearsal(args)
    if world > 1 or os.environ.get("SC_FORCE_COLLECTIVES", "0") == "1":   # the latter: one-rank RCCL rehearsal (parallel.dp_world)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if int(os.environ.get("SC_SHARE_GPU", "0")):               # rehearsal: all ranks on one device
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("SC_DIST_BACKEND", "nccl")      # "nccl" = RCCL; "gloo" only to rehearse on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    dev = torch.device("cuda", local_rank)
    rccl_ranks = count_ranks(dist, dev)

    from speechclip_plus_amd import ops

    B, L = args.batch, int(round(args.seconds * 16000))
    model, trainer, batch, sd, wav_len = make_workload(args.model, B, L, args.ragged, rank, dev, unfreeze=args.unfreeze,
                                                       trainable=args.trainable, no_dropout=args.no_dropout)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(batch)
    if args.pmc_child:                              # target of the counter passes (live_pmc_traffic): K more steps and out
        for _ in range(args.steps):
            trainer.step(batch)
        torch.cuda.synchronize()
        return
    # ---- timed region: exactly K steps, un-instrumented (timing events would put a marker packet between
    # back-to-back kernels and cost ~20 % of the step) ------------------------------------------------------
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    # ---- the same K steps again with a HIP-event pair around every GEMM / attention launch (recorded on the
    # launch stream): per-launch durations of the dominant kernel for the roofline object -------------------
    timer = None
    if not args.no_kernel_timer and rank == 0:
        timer = ops.KernelTimer()
    ops.set_timer(timer)
    for _ in range(args.steps):
        trainer.step(batch)
    torch.cuda.synchronize()
    ops.set_timer(None)
    # ---- the one-stream schedule beside the overlapped one: 2 + 5 more steps with the switch off, same process --
    one_stream_ms = None
    if getattr(model.audio_encoder, "enc_overlap", False) and not args.trainable:
        model.audio_encoder.enc_overlap = False
        for _ in range(4):                               # (the first steps after the switch absorb one-off costs: with a process group a 20 ms outlier was seen once)
            trainer.step(batch)
        sync()
        t1 = time.perf_counter()
        for _ in range(5):
            trainer.step(batch)
        sync()
        one_stream_ms = (time.perf_counter() - t1) / 5 * 1e3
        model.audio_encoder.enc_overlap = True
        trainer.step(batch)
        torch.cuda.synchronize()
    # ---- N > 1: K more steps with brackets around the collectives (every rank), so that a scaling record explains itself ---------
    collectives = None
    if dist is not None:
        collectives = collective_report(dist, dev, args.steps, elapsed / args.steps * 1e3, lambda: trainer.step(batch))
    # ---- forward only (north_star: fraction of the MFMA bf16 peak on the HuBERT + attention-pool forward) ---------
    fwd_ms = fwd_train_ms = conv_ms = conv_train_ms = None
    if rank == 0 and args.model == "base":
        fwd_train_ms, conv_train_ms = time_forward(model, batch, args.steps)     # as the train step runs it: HuBERT + head dropout live
        model.eval()                                                             # inference forward: no dropout
        fwd_ms, conv_ms = time_forward(model, batch, args.steps)
        model.train()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())

    result = None
    if rank == 0:
        T = model.audio_encoder._plan(B, L).T
        roof = None
        extra = {}
        if timer is not None:
            summ = timer.summary()
            if args.gemm_shapes:                                 # per-shape table on stderr (diagnostics; not part of the JSON line)
                import sys as _sys
                for tag, r in sorted(timer.summary_by_tag().items(), key=lambda kv: -kv[1]["ms"]):
                    print(f"[gemm] {tag:58s} {r['launches'] // args.steps:3d}/step  {r['ms'] * 1e3 / r['launches']:8.1f} us  "
                          f"{r['work'] / (r['ms'] * 1e-3) / 1e12:7.1f} TF/s  {r['ms'] / args.steps:6.3f} ms/step", file=_sys.stderr)
            dom_name = max((k for k in summ if k.startswith("gemm_bf16_")), key=lambda k: summ[k]["ms"])
            dom = summ[dom_name]
            tflops = dom["work"] / (dom["ms"] * 1e-3) / 1e12
            kname = {"gemm_bf16_256x256": "gemm256_kernel<0,BN> (256 x {256|192} x 64 tiles)", "gemm_bf16_128x128": "gemm_bf16_kernel<128,128>",
                     "gemm_bf16_128x64": "gemm_bf16_kernel<128,64>"}[dom_name]
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(tflops, 2),
                    "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                    "launches_per_step": dom["launches"] // args.steps,
                    "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                    "alg_gflop_per_launch": round(dom["work"] / dom["launches"] / 1e9, 3)}
            ksub = "gemm256_kernel<0," if dom_name == "gemm_bf16_256x256" else "gemm_bf16_kernel<128, 128>"
            roof.update(traffic_from_live(live_traffic, ksub) or pmc_traffic(ksub))
            if isinstance(live_traffic, str):
                roof["traffic_live_skipped"] = live_traffic
            if dom_name == "gemm_bf16_256x256" and world == 1:
                roof["k_loop"] = k_loop_clock(dev)
            for k, v in summ.items():
                if not k.startswith("gemm_bf16_"):
                    continue          # event brackets are validated against rocprofv3 for the GEMM launches only
                extra[k] = {"ms_per_step": round(v["ms"] / args.steps, 3),
                            "alg_tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else None}
        cpu = None
        if world == 1 and args.cpu_utts > 0 and args.model == "base":
            cpu = cpu_baseline(sd, model, args.cpu_utts, L, args.cpu_iters, args.cpu_full)
        recall = None
        if world == 1 and args.model == "base" and not args.no_recall:
            recall = recall_parity(dev)
        result = {
            "metric": "utterances/sec (train step)", "value": round(B * world * args.steps / elapsed, 2),
            "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "one_stream_ms_per_step": None if one_stream_ms is None else round(one_stream_ms, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"[top {args.unfreeze} HuBERT layers unfrozen: fwd + bwd + Adam] " if args.unfreeze else "") +
                                   ("[whole HuBERT trainable: fwd + bwd + Adam] " if args.trainable else "") +
                                   {"base": "Parallel SpeechCLIP base train step (HuBERT-base frozen fwd + weighted sum + CLS "
                                            "attention-pool head fwd/bwd + InfoNCE fwd/bwd + Adam)",
                                    "large": "Parallel SpeechCLIP large train step (HuBERT-large frozen fwd + normalised weighted sum + "
                                             "CLS attention-pool head fwd/bwd + InfoNCE fwd/bwd + Adam)",
                                    "cascaded_plus": "Cascaded+ base train step (HuBERT-base frozen fwd + weighted sum + attention block + "
                                                     "CIF + keyword projection / BatchNorm / VQ + frozen CLIP text tower, fwd/bwd + InfoNCE "
                                                     "+ quantity loss + Adam)",
                                    "hybrid_plus_large": "Hybrid+ large train step (HuBERT-large frozen fwd + weighted sum + shared attention "
                                                         "block: CLS row -> parallel embedding, frames -> CIF / VQ / frozen CLIP ViT-L/14 text "
                                                         "tower, fwd/bwd + both InfoNCE losses + quantity loss + Adam)"}[args.model] +
                                   f", {B} utt/GPU x {args.seconds:g} s (L={L}, T={T})" +
                                   (f", RAGGED lengths U{{32000..{L}}} (mean {float(wav_len.float().mean()) / 16000:.2f} s)" if args.ragged else "") +
                                   ", CLIP image embeddings given",
                       "global_batch": B * world, "per_gpu_batch": B, "audio_samples": L, "frames": T,
                       "parallelism": f"dp{world}", "dropout": ("off (--no-dropout)" if args.no_dropout else
                                   "on, as the reference's train step: HuBERT in train mode (base: input / residual / attention "
                                   "p=0.1; large: 0) + head p=0.1; masks = stateless hash inside the GEMM / attention kernels"),
                       "schedule": ("frozen encoder of step N + 1 enqueued on its own HIP stream under step N's head / loss / backward / "
                                    "optimiser kernels (two alternating sets of resident buffers; every step's full work inside the "
                                    "timed region; SC_ENC_OVERLAP=0 = one stream)"
                                    if getattr(model.audio_encoder, "enc_overlap", False) and not args.trainable else "one stream per step"),
                       "hw_queues": __import__("speechclip_plus_amd").hw_queue_status()},
            "rccl_ranks": rccl_ranks, "dist_backend": (os.environ.get("SC_DIST_BACKEND", "nccl") if dist is not None else None),
            "collectives": collectives,
            "loss": round(loss_val, 5), "roofline": roof, "kernels": extra, "cpu_baseline": cpu,
            "forward": None if fwd_ms is None else forward_summary(fwd_ms, B, L, T, wav_len.tolist() if args.ragged else None, conv_ms),
            "forward_train_mode": None if fwd_train_ms is None else forward_summary(fwd_train_ms, B, L, T, wav_len.tolist() if args.ragged else None,
                                                                                    conv_train_ms),
            "recall": recall,
        }
        if default_run and not args.no_recipes:
            del trainer, model, batch
            torch.cuda.empty_cache()
            result["recipes"] = run_recipes(args, dev)
            result["recipes"]["rccl_one_rank_rehearsal"] = rccl_rehearsal
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def make_workload(kind, B, L, ragged, rank, dev, unfreeze=0, trainable=False, no_dropout=False, max_audio_len=-1):
    """model (random-init weights of the named architecture, seeded), trainer and one synthetic batch resident in HBM:
    B utterances of L samples N(0, 1) (``ragged``: lengths U{32000 .. L}, zero behind), unit-norm "CLIP image embeddings", 5 captions per id."""
    from speechclip_plus_amd import (KWClip_GeneralTransformer, base_parallel_config, cascaded_plus_base_config,
                                     hybrid_plus_large_config, large_parallel_config, random_hubert_state_dict)
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    torch.manual_seed(7122)
    large = kind in ("large", "hybrid_plus_large")
    sd = random_hubert_state_dict(ARCHS["hubert_large_ll60k" if large else "hubert"], seed=7122)
    cfg = {"base": base_parallel_config, "large": large_parallel_config, "cascaded_plus": cascaded_plus_base_config,
           "hybrid_plus_large": hybrid_plus_large_config}[kind]()
    E = int(cfg.clip.embed_dim)
    # every timed "step" here is one optimiser step over B utterances (hybrid+ large's yaml accumulates 2 micro-batches of 128 per
    # optimiser step: tests/test_gpu_recipes.py::test_hybrid_plus_large_at_the_recipe_batch_of_128_with_accumulation)
    cfg.trainer.accumulate_grad_batches = 1
    # -1: the batch is given at its length (no second crop inside the model); 102400 (every shipped yaml): the reference's training
    # entry - full utterances in, random 6.4 s crop inside the encoder forward (recipes.train_crop_in_forward)
    cfg.audio_encoder.max_audio_len = max_audio_len
    if unfreeze > 0:
        nl = 24 if large else 12
        cfg.audio_encoder.trainable = True
        cfg.audio_encoder.unfreeze_layers = list(range(nl - unfreeze, nl))
    if trainable:
        cfg.audio_encoder.trainable = True
    model = KWClip_GeneralTransformer(cfg, device=str(dev), hubert_state_dict=sd)
    model.train()
    if no_dropout:
        from speechclip_plus_amd import set_dropout
        set_dropout(model, False)
    trainer = ContrastiveTrainer(model)
    g = torch.Generator(device="cpu").manual_seed(7122 + rank)
    wav = torch.randn(B, L, generator=g).to(dev)
    wav_len = torch.full((B,), L, dtype=torch.long)
    if ragged:
        wav_len = torch.randint(min(32000, L), L + 1, (B,), generator=g)
        wav_len[0] = L                                   # the batch is padded to its longest utterance: keep the geometry
        wav = wav * (torch.arange(L).unsqueeze(0) < wav_len.unsqueeze(1)).to(dev)
    # the synthetic batch is RESIDENT (the metric's contract: inputs in HBM before the timed region): marked ready, so that the
    # encoder stream need not wait for the caller's stream before reading it (speech_encoder._encode_overlapped)
    wav._sc_ready = True
    img = torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=-1).to(dev)
    ids = (torch.arange(B) + rank * B) // 5            # Flickr8k shape: 5 captions per image id
    batch = {"wav": wav, "wav_len": wav_len, "image": img, "id": ids.to(dev)}
    return model, trainer, batch, sd, wav_len


def time_forward(model, batch, steps):
    """-> (ms per forward, ms of it inside the conv feature extractor): the model's forward under no_grad, wall clock over ``steps``
    back-to-back calls; the conv share from three event records per call (speech_encoder._section_ev: start, conv stack done, end)."""
    enc = model.audio_encoder
    with torch.no_grad():
        model(batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            model(batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / steps * 1e3
        conv = []
        for _ in range(min(steps, 5)):
            enc._section_ev = {}
            model(batch)
            torch.cuda.synchronize()
            ev = enc._section_ev
            if "start" in ev and "conv_done" in ev:
                conv.append(ev["start"].elapsed_time(ev["conv_done"]))
        enc._section_ev = None
    return ms, (sum(conv) / len(conv) if conv else None)


def crop_entry_recipes(args, dev, out):
    """The reference's own training entry on the driver's clock: whole 10 s utterances in, ``max_audio_len:
    102400`` as in every shipped yaml, the random 6.4 s crop INSIDE the encoder forward (speech_encoder_plus.py:548-552), fresh batches
    alternating.  Three ways the batch arrives:
      train_crop_in_forward         two batches resident in HBM (the metric's contract), lengths on the device with their host twin
      train_crop_in_forward_h2d     two pinned host batches, each step through model.transfer_batch_to_device (the Lightning hook: H2D on
                                    a copy stream + completion event, host twin of the lengths) - the H2D copy is inside the timed region
      train_crop_in_forward_ragged  as the first, full lengths U{2 .. 10 s} (cropped lengths then U{2 .. 6.4 s}, the rows follow them)
    Each: 3 warm-up + 10 timed steps.  Target: the first within 3 % of crop_6p4s_base (the same work on a pre-cropped resident batch)."""
    import gc
    import numpy as np
    from speechclip_plus_amd.data import attach_host_lengths
    B = args.batch
    L = int(round(args.seconds * 16000))
    model, trainer, batch0, _, _ = make_workload("base", B, L, False, 0, dev, max_audio_len=102400)
    E = batch0["image"].shape[1]
    g = torch.Generator(device="cpu").manual_seed(99)

    def host_batch(ragged):
        wl = torch.full((B,), L, dtype=torch.long)
        if ragged:
            wl = torch.randint(32000, L + 1, (B,), generator=g)
            wl[0] = L
        wav = torch.empty(B, L, pin_memory=True)
        wav.copy_(torch.randn(B, L, generator=g) * (torch.arange(L).unsqueeze(0) < wl.unsqueeze(1)))
        return {"wav": wav, "wav_len": attach_host_lengths(wl), "image": torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=-1),
                "id": torch.arange(B) // 5}

    def resident(hb):
        d = {k: v.to(dev) for k, v in hb.items()}
        attach_host_lengths(d["wav_len"], hb["wav_len"]._sc_host)
        d["wav"]._sc_ready = True                       # resident before the timed region starts; nothing writes it afterwards
        return d

    for name, ragged, via_hook in (("train_crop_in_forward", False, False), ("train_crop_in_forward_h2d", False, True),
                                   ("train_crop_in_forward_ragged", True, False)):
        hosts = [host_batch(ragged), host_batch(ragged)]
        feeds = hosts if via_hook else [resident(h) for h in hosts]
        torch.cuda.synchronize()
        np.random.seed(7122)
        nxt = (lambda i: model.transfer_batch_to_device(feeds[i % 2])) if via_hook else (lambda i: feeds[i % 2])
        for i in range(3):
            trainer.step(nxt(i))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            loss = trainer.step(nxt(i + 1))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        pl = next(reversed(model.audio_encoder._plans.values()))
        out[name] = {"ms_per_step": round(ms, 3), "utterances_per_s": round(B / ms * 1e3, 1), "steps": 10, "warmup": 3, "batch": B,
                     "seconds_in": args.seconds, "max_audio_len": 102400, "loss": round(float(loss.item()), 5),
                     "input": "pinned host batch -> transfer_batch_to_device every step (H2D inside the timed region)" if via_hook
                              else "two resident device batches alternating; wav_len on the device + host twin",
                     "rows": {"layout": int(pl.M), "frames_T": int(pl.T), "device_side_crop": pl.wav_off is not None}}
        if ragged:
            rec_len = [min(int(v), 102400) for h in hosts for v in h["wav_len"]._sc_host]
            out[name]["mean_cropped_seconds"] = round(sum(rec_len) / len(rec_len) / 16000, 2)
        del feeds, hosts
    del model, trainer, batch0
    gc.collect()
    torch.cuda.empty_cache()


def run_recipes(args, dev):
    """The other shapes and recipes on the driver's clock: 3 warm-up + 10 timed train steps each, same protocol as
    the headline (barrier-free at N = 1: synchronize on both sides), models built and released one at a time."""
    import gc
    out = {}
    B = args.batch
    for name, kind, seconds, ragged, extra in (("ragged_base", "base", args.seconds, True, {}), ("crop_6p4s_base", "base", 6.4, False, {}),
                                               ("cascaded_plus_base", "cascaded_plus", args.seconds, False, {}),
                                               ("hybrid_plus_large", "hybrid_plus_large", args.seconds, False, {}),
                                               ("unfreeze_top2_base", "base", args.seconds, False, {"unfreeze": 2}),
                                               ("trainable_base", "base", args.seconds, False, {"trainable": True})):
        L = int(round(seconds * 16000))
        model, trainer, batch, _, wav_len = make_workload(kind, B, L, ragged, 0, dev, **extra)
        for _ in range(3):
            trainer.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            loss = trainer.step(batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        rec = {"ms_per_step": round(ms, 3), "utterances_per_s": round(B / ms * 1e3, 1), "steps": 10, "warmup": 3, "batch": B,
               "seconds": seconds, "loss": round(float(loss.item()), 5)}
        if not ragged:                                   # whole-step algorithmic flops and their fraction of the dense bf16 peak
            alg = step_alg_gflop(kind, L, unfreeze=extra.get("unfreeze", 0), trainable=bool(extra.get("trainable")))
            rec.update(alg)
            rec["step_frac_of_mfma_bf16_peak"] = round(alg["alg_gflop_per_utt_step"] * B / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)      # GFLOP / ms = TFLOP/s
        if extra:
            rec["mode"] = ("top 2 HuBERT layers unfrozen (audio_encoder.trainable + unfreeze_layers [10, 11]): fwd + bwd + Adam" if "unfreeze" in extra
                           else "audio_encoder.trainable: true - the whole HuBERT trains (speech_encoder_plus.py:556-562): fwd + bwd + Adam")
        if ragged:
            rec["mean_seconds"] = round(float(wav_len.float().mean()) / 16000, 2)
        if kind == "base" and not extra:
            T = model.audio_encoder._plan(B, L).T
            model.eval()
            fwd_ms, conv_ms = time_forward(model, batch, 10)
            rec["forward"] = forward_summary(fwd_ms, B, L, T, wav_len.tolist() if ragged else None, conv_ms)
            rec["rows"] = {"layout": int(model.audio_encoder._plan(B, L).M), "padded_batch": B * T}
        out[name] = rec
        del model, trainer, batch
        gc.collect()
        torch.cuda.empty_cache()
        if name == "crop_6p4s_base":
            crop_entry_recipes(args, dev, out)
            out["train_crop_in_forward"]["vs_crop_6p4s_base"] = round(out["train_crop_in_forward"]["ms_per_step"] / rec["ms_per_step"], 4)
    return out


def launch_ranks(n: int) -> int:
    """Parent of a plain `bench.py --gpus N` call: start the N ranks as ONE child process tree (torch.distributed.run) and
    return its exit code.  Nothing here touches the GPU (no HIP call, no torch.cuda.* query): the children initialise it."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: the only mode this pool's driver supports
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def collective_report(dist, dev, steps: int, own_ms_per_step: float, step_fn) -> dict:
    """N > 1 diagnostics (every rank calls this; the numbers of all ranks come back on each).  ``steps`` more steps run with
    brackets (speechclip_plus_amd.parallel.CommTimer) around the packed all-gather (main stream: all of it is exposed, and it
    absorbs the arrival skew between ranks), the gradient all-reduce (side stream) and the main stream's wait for the side stream
    at the next step's join point (= the part of all-reduce + clip + Adam that the frozen encoder forward did not hide).
    ``own_ms_per_step``: this rank's own clock over the TIMED region (before the max over ranks that ``value`` uses)."""
    from speechclip_plus_amd.parallel import CommTimer, set_comm_timer
    world = dist.get_world_size()
    timer = CommTimer()
    set_comm_timer(timer)
    for _ in range(steps):
        step_fn()
    summ = timer.summary(steps)
    set_comm_timer(None)
    names = ("all_gather", "all_reduce", "join_wait")
    mine = [own_ms_per_step] + [summ.get(n, {}).get("us_per_step", 0.0) for n in names]
    on_dev = os.environ.get("SC_DIST_BACKEND", "nccl") == "nccl"
    t = torch.tensor(mine, dtype=torch.float64, device=dev if on_dev else "cpu")
    allt = torch.empty(world * len(mine), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(allt, t)
    allt = allt.view(world, len(mine)).cpu()
    out = {"measured_in": f"{steps} extra steps with event brackets on the issuing streams (not the timed region)",
           "ms_per_step_by_rank": {"min": round(float(allt[:, 0].min()), 3), "max": round(float(allt[:, 0].max()), 3),
                                   "all": [round(float(v), 3) for v in allt[:, 0]]}}
    for i, n in enumerate(names):
        col = allt[:, 1 + i]
        out[n + "_us"] = {"rank0": round(float(col[0]), 1), "min": round(float(col.min()), 1), "max": round(float(col.max()), 1),
                          "calls_per_step": summ.get(n, {}).get("calls_per_step", 0)}
    exposed = allt[:, 1] + allt[:, 3]
    out["exposed_us_per_step"] = {"rank0": round(float(exposed[0]), 1), "max": round(float(exposed.max()), 1),
                                  "is": "all_gather (on the main stream) + join_wait (main stream blocked on the side stream)"}
    return out


def count_ranks(dist, dev) -> int:
    """Number of ranks that really took part: an all-reduce (SUM) of ones through the backend in use."""
    if dist is None:
        return 1
    one = torch.ones(1, device=dev if os.environ.get("SC_DIST_BACKEND", "nccl") == "nccl" else "cpu")
    dist.all_reduce(one)
    return int(round(float(one.item())))


def rehearse_launch(args, world: int) -> None:
    """--rehearse-launch: everything of the N-rank run except the GPU work (this container has no GPU)."""
    import torch.distributed as dist
    from speechclip_plus_amd.parallel import GradAllReduce, gather_loss_feats
    rank = int(os.environ.get("RANK", "0"))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(os.environ.get("SC_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
    ranks = count_ranks(dist if world > 1 else None, "cpu")
    B, E = args.batch, 512
    g = torch.Generator().manual_seed(rank)
    a = torch.randn(B, E, generator=g, requires_grad=True)
    a_all, i_all, id_all = gather_loss_feats(a, torch.randn(B, E, generator=g), torch.arange(B) + rank * B)
    assert a_all.shape[0] == B * world and id_all.tolist() == list(range(B * world))
    a_all.sum().backward()
    flat = torch.full((1000,), float(rank + 1))
    ar = GradAllReduce(flat)
    ar.launch()
    ar.wait()
    assert float(flat[0]) == world * (world + 1) / 2
    collectives = None
    if world > 1:                                   # the N > 1 diagnostics field, here with wall-clock brackets on CPU tensors
        def step():
            x = torch.randn(B, E, requires_grad=True)
            gather_loss_feats(x, torch.randn(B, E), torch.arange(B) + rank * B)[0].sum().backward()
            f = torch.ones(1000)
            r = GradAllReduce(f)
            r.launch()
            r.wait()
        collectives = collective_report(dist, "cpu", 3, 1.0 + rank, step)
    if rank == 0:
        print(json.dumps({"metric": "utterances/sec (train step)", "value": None, "unit": "utterances/s", "n_gpus": world,
                          "steps": 0, "warmup": 0, "rehearsal": "launch plumbing only (no GPU work)", "rccl_ranks": ranks, "collectives": collectives,
                          "dist_backend": os.environ.get("SC_DIST_BACKEND", "gloo") if world > 1 else None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def forward_summary(fwd_ms, B, L, T, lens=None, conv_ms=None):
    """Algorithmic flops of the HuBERT-base + CLS-pool forward (SURVEY 8d formulae, conv extractor included; the
    collapsed CLS head is ~0.013 GFLOP/utt) over the measured forward time.  ``lens``: per-utterance sample counts of a ragged
    batch - every utterance then counts at its own length (mean reported).  ``conv_ms``: the part of the forward spent in the conv
    feature extractor (event records around it): SURVEY 8d asks for the "transformer + head only" fraction next to the one that
    includes the extractor - its flops (projection, pos_conv, 12 layers, head) over the forward time without the extractor."""
    def one(Lb):
        t, cin, conv = Lb, 1, 0.0
        ks, ss = (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)
        for k, s_ in zip(ks, ss):
            t = (t - k) // s_ + 1
            conv += 2.0 * t * 512 * cin * k
            cin = 512
        Tb = t
        D, F, NL = 768, 3072, 12
        rest = 2.0 * Tb * 512 * D                                   # post_extract_proj
        rest += 2.0 * Tb * D * (D // 16) * 128                      # pos_conv
        rest += NL * Tb * 2.0 * (4 * D * D + 2 * D * F)             # linear layers
        rest += NL * 4.0 * Tb * Tb * D                              # attention
        return conv, rest + 0.013e9
    pairs = [one(L)] if lens is None else [one(int(l)) for l in lens]
    conv = sum(p[0] for p in pairs) / len(pairs)
    rest = sum(p[1] for p in pairs) / len(pairs)
    flop = conv + rest
    tfl = flop * B / (fwd_ms * 1e-3) / 1e12
    out = {"ms": round(fwd_ms, 3), "utterances_per_s": round(B / fwd_ms * 1e3, 1), "alg_gflop_per_utt": round(flop / 1e9, 2),
           "alg_tflops": round(tfl, 1), "frac_of_mfma_bf16_peak": round(tfl / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    if conv_ms is not None and 0 < conv_ms < fwd_ms:
        t_rest = rest * B / ((fwd_ms - conv_ms) * 1e-3) / 1e12
        out.update({"conv_extractor_ms": round(conv_ms, 3), "alg_gflop_per_utt_transformer_head_only": round(rest / 1e9, 2),
                    "frac_transformer_head_only": round(t_rest / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                    "frac_conv_extractor_only": round(conv * B / (conv_ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)})
    return out


def step_alg_gflop(kind: str, L: int, unfreeze: int = 0, trainable: bool = False) -> dict:
    """ALGORITHMIC flops of one TRAIN step per utterance of a recipe (SURVEY 8d conventions: an utterance at its own length, the parallel
    head as the CLS row only, backward of a trainable block = 2 x its forward, of a frozen block that gradients cross = 1 x, nothing
    for frozen blocks in front of the first trainable one).  Terms, with T frames, D / F / NL of the encoder, S = T + 1 rows of a
    branch's attention block, N = round(T / 20) keywords (kwClip.py:876), E_t / V the text width / reduced vocabulary:
      hubert_fwd   conv stack + projection + pos_conv + NL (linear + attention)                         (forward_summary's formulae)
      head         parallel: 1.19 (base) / 2.12 (large) GFLOP forward, CLS row only
      block        MultiheadAttentionAndNorm over S rows: 8 S D^2 + 4 S^2 D
      cif_conv     Conv1d(D, D, 3) of the CIF weight generator: 6 T D^2
      kw           keyword projection(s) + cosine against the vocabulary + the embedding product: 2 N (D E_t [+ D D]) + 4 N E_t V
      text         the frozen CLIP text tower on the causal PREFIX the end-of-text pooling reads (N + 2 of the 77 tokens; the
                   reference runs all 77): 12 (24 P W^2 + 4 P^2 W)"""
    large = kind in ("large", "hybrid_plus_large")
    D, F, NL, W, V = (1024, 4096, 24, 768, 19787) if large else (768, 3072, 12, 512, 8112)
    t, cin, conv = L, 1, 0.0
    for k, s_ in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        t = (t - k) // s_ + 1
        conv += 2.0 * t * 512 * cin * k
        cin = 512
    T = t
    layer = T * 2.0 * (4 * D * D + 2 * D * F) + 4.0 * T * T * D
    front = conv + 2.0 * T * 512 * D + 2.0 * T * D * (D // 16) * 128
    hubert = front + NL * layer
    terms = {"hubert_fwd": hubert}
    if trainable:
        terms["hubert_bwd"] = 2.0 * hubert
    elif unfreeze:
        terms["hubert_bwd_top_layers"] = 2.0 * unfreeze * layer
    if kind in ("base", "large"):
        terms["head_fwd_bwd"] = 3.0 * (2.12e9 if large else 1.19e9)
    else:
        S, N = T + 1, round(T / 20)
        block = 8.0 * S * D * D + 4.0 * S * S * D
        cif = 6.0 * T * D * D
        kw = 2.0 * N * (D * W + (D * D if large else 0)) + 4.0 * N * W * V
        P = N + 2
        text = 12 * (24.0 * P * W * W + 4.0 * P * P * W)
        terms.update({"branch_block_cif_kw_fwd_bwd": 3.0 * (block + cif + kw), "text_tower_fwd_dgrad": 2.0 * text})
        if kind == "hybrid_plus_large":
            terms["parallel_projection_fwd_bwd"] = 3.0 * 2.0 * D * W
    total = sum(terms.values())
    return {"alg_gflop_per_utt_step": round(total / 1e9, 1), "terms_gflop": {k: round(v / 1e9, 2) for k, v in terms.items()}}


def rccl_one_rank_rehearsal(args):
    """The only part of SURVEY 8(e) a one-GPU box can watch: the headline step with a ONE-rank RCCL process group
    (SC_FORCE_COLLECTIVES=1: parallel.dp_world() = 1 but the packed all-gather with autograd, the side-stream flat all-reduce and the
    join are issued for real), as a child run of this script that finishes before this process touches the GPU.  -> the child's
    ms_per_step (overlapped and one-stream schedules), its `collectives` object (all_gather_us / all_reduce_us / join_wait_us under the
    overlapped encoder) and the hardware-queue setting, or {"skipped": why}."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SC_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--cpu-utts", "0", "--no-recall", "--no-recipes",
           "--no-pmc", "--no-kernel-timer", "--batch", str(args.batch), "--seconds", str(args.seconds)]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        line = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"skipped": f"child exit {r.returncode}: " + r.stderr.decode(errors="replace")[-300:]}
        d = json.loads(line[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"skipped": f"{type(e).__name__}: {e}"[:300]}
    return {"what": "headline train step in a child process with a one-rank RCCL group (SC_FORCE_COLLECTIVES=1): every collective of the "
                    "N > 1 step is issued; 10 timed steps after 3 warm-up",
            "ms_per_step": d.get("ms_per_step"),
            "rccl_ranks": d.get("rccl_ranks"), "collectives": d.get("collectives"), "hw_queues": d.get("config", {}).get("hw_queues")}


def live_pmc_traffic():
    """HBM-side bytes per kernel launch of THIS run's train step, from hardware counters collected now: two child runs of this script
    (`--pmc-child`: 2 warm-up + 3 train steps, nothing else) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` - separate passes,
    `--kernel-trace` only beside them, exactly the recipe of MI355X_MICROARCH.md (units KiB; gfx950 reports half of a wide streaming
    read, hence FETCH x 2; WRITE exact) - started BEFORE this process makes its first GPU call (a child that profiles is an ordinary child
    process, not an exec of a GPU-initialised one).  -> {kernel name: {"fetch_kib": [...], "write_kib": [...]}} or a string saying why not."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return "this process is itself being profiled"
    out = {}
    tmp = tempfile.mkdtemp(prefix="sc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter, key in (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib")):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "c", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "3", "--warmup", "2"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            files = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
            if r.returncode != 0 or not files:
                return f"rocprofv3 --pmc {counter}: exit {r.returncode}, {len(files)} csv; " + r.stderr.decode(errors="replace")[-200:]
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter:
                    out.setdefault(row["Kernel_Name"], {"fetch_kib": [], "write_kib": []})[key].append(float(row["Counter_Value"]))
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return f"counter pass failed: {type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def traffic_from_live(live, kernel_substr):
    """roofline.traffic fields from live_pmc_traffic()'s table: the average over every launch of every instantiation of the dominant
    kernel in the child runs' train steps (the first launch of each instantiation, which fills the caches and loads the code, dropped)."""
    if not isinstance(live, dict):
        return None
    tot_f = tot_w = n = 0
    for name, v in live.items():
        if kernel_substr in name and len(v["fetch_kib"]) == len(v["write_kib"]) and len(v["fetch_kib"]) > 1:
            tot_f += sum(v["fetch_kib"][1:])
            tot_w += sum(v["write_kib"][1:])
            n += len(v["fetch_kib"]) - 1
    if n == 0:
        return None
    return {"traffic": round((2.0 * tot_f + tot_w) * 1024.0 / n), "traffic_unit": "bytes/launch (PMC, avg over launches of all instantiations)",
            "traffic_source": f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this bench.py invocation ({n} launches; FETCH x 2 gfx950 correction)",
            "traffic_fetch_bytes": round(2.0 * tot_f * 1024.0 / n), "traffic_write_bytes": round(tot_w * 1024.0 / n)}


def pmc_traffic(kernel_substr):
    """Fallback of traffic_from_live (no rocprofv3, a nested profiler, --no-pmc): HBM bytes per launch of the dominant kernel from the
    newest committed rocprofv3 PMC summary of this same command (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH x2 gfx950
    correction; tools/summarize_pmc.py), reported only while the kernel source it was measured on is the tree's."""
    import glob
    import re

    def version(path):                      # profiles/rNN_vMM_*: numeric order (r01_v10 after r01_v9)
        m = re.search(r"r(\d+)_v(\d+)", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=version)
    if not files:
        return {"traffic": None}
    data = json.load(open(files[-1]))
    src = source_sha(("gemm256_bf16.hip", "gemm_epilogue.inc") if "gemm256" in kernel_substr else ("gemm_bf16.hip",))
    meta = data.get("_meta", {})
    hits = [v for name, v in data.items() if kernel_substr in name and name != "_meta"]
    if not hits:
        return {"traffic": None}
    if meta.get("kernel_source_sha256") != src:
        # the counters were collected on another version of the kernel: stale, not reported
        return {"traffic": None, "traffic_source": os.path.relpath(files[-1], ROOT) + " (stale: kernel source changed since; "
                f"profile {str(meta.get('kernel_source_sha256'))[:12]} vs tree {src[:12]})"}
    n = sum(v["launches_sampled"] for v in hits)
    avg = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in hits) / n
    return {"traffic": round(avg), "traffic_unit": "bytes/launch (PMC, avg over launches of all instantiations)",
            "traffic_source": os.path.relpath(files[-1], ROOT) + f" (stored profile; kernel source sha256 {src[:12]} = this tree's)"}


def k_loop_clock(dev):
    """Shader clock and cycles per K-tile INSIDE the dominant kernel's K loop, from its stamped diagnostic build (tile 34: s_memtime
    and s_memrealtime around every tile's K loop; same code otherwise, its launch time equals the production kernel's) on the
    step's QKV and FC1 shapes.  `peak` above is the 2.4 GHz figure; under this load the chip holds the clock reported here, and a
    K-tile cannot take fewer than 2048 cycles (2 waves x 64 MFMA x 16 cycles per SIMD)."""
    from speechclip_plus_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    Bq, R, D, F = 64, 512, 768, 3072
    out = {"mfma_paced_cycles_per_k_tile": 2048, "nominal_clock_ghz": 2.4}
    for name, n, act in (("qkv", 3 * D, 0), ("fc1", F, 1)):
        A = torch.randn(Bq * R, D, generator=g).to(torch.bfloat16).to(dev)
        W = (torch.randn(n, D, generator=g) * D ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        C = torch.empty(Bq * R, n, device=dev, dtype=torch.bfloat16)
        dbg = torch.zeros(4 * 8 * 8 * 8, device=dev, dtype=torch.int64)
        for _ in range(6):
            ops.gemm_raw(A, D, W, D, C, n, Bq * R, n, D, bias=bias, act=act, tile=34, Ct=dbg.view(torch.bfloat16))
        torch.cuda.synchronize()
        raw = dbg.view(4, 8, 8, 8).cpu().double()
        cyc = raw[..., 7] - raw[..., 6]
        wall_ns = (raw[..., 2] - raw[..., 1]) * 10.0
        ok = (raw[..., 0] != 0) & (wall_ns > 0)
        out[name] = {"shader_clock_ghz": round(float((cyc[ok] / wall_ns[ok]).median()), 3),
                     "cycles_per_k_tile": round(float((cyc[ok] / (D // 64)).median()))}
    return out


def source_sha(names) -> str:
    import hashlib
    h = hashlib.sha256()
    for name in names:
        h.update(open(os.path.join(ROOT, "speechclip_plus_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def recall_parity(dev):
    """recall@{1,5,10} of the HIP model on the 5000-utterance synthetic eval set against the two references held by
    tests/golden/recall_eval.npz - the fp32 oracle and the bf16-storage-emulated oracle (tools/recall_eval.py; the oracle itself is
    not run here)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import recall_eval
    if not os.path.exists(recall_eval.FIXTURE):
        return None
    fx = dict(np.load(recall_eval.FIXTURE))
    model = recall_eval.build_model(str(dev))
    emb = recall_eval.hip_embeddings(model, int(fx["n_ids"]), int(fx["batch"]))
    out = recall_eval.hip_recall(model, fx, emb=emb)
    out["set"] = "planted margins (every confusion >= 30 sigma of the bf16 margin noise): parity holds by construction for any bf16-storage implementation"
    if os.path.exists(recall_eval.FIXTURE_NATURAL):       # the same utterances against a gallery with natural margins (nothing planted)
        out["natural_margins"] = recall_eval.natural_margin_report(emb, dict(np.load(recall_eval.FIXTURE_NATURAL)))
        second = recall_eval.FIXTURE_NATURAL.replace(".npz", "_b.npz")      # gallery B: same construction, another noise seed (round 5)
        if os.path.exists(second):
            out["natural_margins_gallery_b"] = recall_eval.natural_margin_report(emb, dict(np.load(second)))
    return out


def physical_cores():
    """Physical cores of the host from /proc/cpuinfo (distinct (physical id, core id) pairs); None if it cannot be told."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        return len(pairs) or None
    except OSError:
        return None


def cpu_baseline(sd, model, n_utts, L, iters, full=False):
    """The oracle (kind "port": torch-CPU fp32 restatement of the reference maths) on the host cores, BASELINE configs[0]:
    Parallel SpeechCLIP base, batch 8 (the reference's CPU-runnable case), same utterance length as the GPU workload, SURVEY 8d's
    protocol: 3 warm-up + 10 timed train steps.  Forward (frozen HuBERT under no_grad + weighted sum + parallel head + loss) and
    forward + backward are timed separately inside each iteration.  Threads: min(32, torch's default) - torch's CPU kernels scale
    negatively past a few dozen threads (a 128-thread host: 0.83 utt/s at 128 threads, 3.5 at 32, 3.1 at 8; rounds 3-5), so the
    default no longer spends time on the slow settings; ``full`` adds torch's default count and 8 for the record."""
    import oracle
    torch.manual_seed(0)
    head_W = {k: v.detach().cpu().float().clone().requires_grad_(True) for k, v in model.parallel_branch.state_dict().items()}
    ws = torch.zeros(13, requires_grad=True)
    wavs = [torch.randn(L) for _ in range(n_utts)]
    img = torch.nn.functional.normalize(torch.randn(n_utts, 512), dim=-1)
    ids = torch.arange(n_utts) // 5
    arch = oracle.HubertArch.base()
    warm, timed = 3, iters

    def step():
        t0 = time.perf_counter()
        with torch.no_grad():
            hs, fl = oracle.speech_encoder_forward(sd, arch, wavs)
        feat = oracle.weighted_sum(ws, hs)
        e = oracle.parallel_branch_forward(head_W, feat, fl, nhead=8)
        loss = oracle.masked_contrastive_loss(e / e.norm(dim=-1, keepdim=True), img, ids)
        t1 = time.perf_counter()
        loss.backward()
        return t1 - t0, time.perf_counter() - t0

    default_threads = torch.get_num_threads()
    runs = {}
    for threads in ((min(32, default_threads), default_threads, 8) if full else (min(32, default_threads),)):
        if threads in runs or threads > default_threads:
            continue
        torch.set_num_threads(threads)
        for _ in range(warm):
            step()
        ts = [step() for _ in range(timed)]
        fwd = sum(t[0] for t in ts) / len(ts)
        tot = sum(t[1] for t in ts) / len(ts)
        runs[threads] = {"threads": threads, "forward_s": round(fwd, 3), "forward_backward_s": round(tot, 3),
                         "forward_utt_per_s": round(n_utts / fwd, 3), "train_step_utt_per_s": round(n_utts / tot, 3)}
    torch.set_num_threads(default_threads)
    best = max(runs.values(), key=lambda r: r["train_step_utt_per_s"])
    return {"value": best["train_step_utt_per_s"], "unit": "utterances/s", "cores": best["threads"], "kind": "port",
            "sample": f"SURVEY 8d protocol: BASELINE configs[0], batch {n_utts} x {L} samples, {warm} warm-up + {timed} timed train steps at "
                      f"{best['threads']} threads, torch fp32 on {os.cpu_count()} logical CPUs ({physical_cores()} physical cores)",
            "host": {"logical_cpus": os.cpu_count(), "physical_cores": physical_cores(), "torch_default_threads": default_threads},
            "s_per_step": best["forward_backward_s"], "forward_utt_per_s": best["forward_utt_per_s"],
            "by_threads": list(runs.values())}


if __name__ == "__main__":
    main()
