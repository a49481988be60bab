"""Data-parallel train steps on the GPU: two ranks (gloo rendezvous, both on cuda:0 - a 1-GPU box cannot host an RCCL world) each
take half of a batch through train.ContrastiveTrainer; the parameters after two steps must equal those of one process that
takes the whole batch.  Exercises the packed all-gather with autograd, the replicated-parameter scaling (trainable
temperature), the side-stream all-reduce + Adam schedule and, in the second case, unfrozen HuBERT layers with their per-layer
gradient all-reduce."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(unfreeze, dropout=False):
    import dataclasses
    sys.path.insert(0, ROOT)
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    arch = dataclasses.replace(ARCHS["hubert"], layers=3)
    sd = random_hubert_state_dict(arch, seed=4)
    torch.manual_seed(4)
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    cfg.cl_loss.args.temperature_trainable = True
    cfg.audio_encoder.optim.args.lr = 1e-2                      # large steps so that a wrong gradient scale shows
    cfg.audio_encoder.scheduler.warmup = 1
    if unfreeze:
        cfg.audio_encoder.trainable = True
        cfg.audio_encoder.unfreeze_layers = [1, 2]
    model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), dropout)
    return model, ContrastiveTrainer(model)


def _batch(lo, hi):
    g = torch.Generator().manual_seed(31)
    B, L = 8, 9000
    wav = torch.randn(B, L, generator=g) * 0.4
    lens = torch.tensor([9000, 7000, 9000, 5200, 8000, 9000, 6100, 9000])
    img = torch.randn(B, 512, generator=g)
    ids = torch.tensor([0, 1, 1, 2, 3, 4, 4, 5])
    return {"wav": wav[lo:hi].cuda(), "wav_len": lens[lo:hi], "image": img[lo:hi].cuda(), "id": ids[lo:hi].cuda()}


def _worker(rank, world, port, unfreeze, q, backend="gloo"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    if backend == "nccl":                                        # RCCL rehearsal: one rank, every collective forced through the backend
        os.environ["SC_FORCE_COLLECTIVES"] = "1"
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    model, trainer = _make(unfreeze)
    n = 8 // world
    batch = _batch(rank * n, (rank + 1) * n)
    losses = [float(trainer.step(batch)) for _ in range(2)]
    torch.cuda.synchronize()
    if rank == 0:
        q.put((losses, trainer.opt.flat_p.detach().cpu().numpy()))      # by value: torch tensors travel as fds the exiting rank may close
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("unfreeze", [False, True])
def test_two_ranks_equal_one_process(unfreeze):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, unfreeze, q)) for r in range(2)]
    for p in procs:
        p.start()
    losses2, flat2 = q.get(timeout=240)
    flat2 = torch.from_numpy(flat2)
    for p in procs:
        p.join(timeout=60)
    model, trainer = _make(unfreeze)
    batch = _batch(0, 8)
    losses1 = [float(trainer.step(batch)) for _ in range(2)]
    torch.cuda.synchronize()
    flat1 = trainer.opt.flat_p.detach().cpu()
    assert abs(losses1[0] - losses2[0]) < 1e-4 and abs(losses1[1] - losses2[1]) < 2e-3, (losses1, losses2)
    rel = float((flat1 - flat2).norm() / flat1.norm())
    moved = float((flat1 - _initial_flat(unfreeze)).norm() / flat1.norm())
    print("two ranks vs one process: |dp - single| / |single| = %.3g, parameter movement %.3g" % (rel, moved))
    assert rel < 0.05 * moved + 1e-7, (rel, moved)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("unfreeze", [False, True])
def test_rccl_backend_one_rank_rehearsal(unfreeze):
    """The collectives of the step through the backend the multi-GPU runs use ("nccl" = RCCL), as far as a one-GPU box allows:
    a one-rank RCCL group with SC_FORCE_COLLECTIVES=1 issues the packed all-gather (with autograd), the side-stream flat
    all-reduce and the per-layer slice all-reduces; one-rank collectives are identities, so the result must equal the plain
    single-process run bit for bit in the loss and closely in the parameters."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_worker, args=(0, 1, _free_port(), unfreeze, q, "nccl"))
    proc.start()
    losses_r, flat_r = q.get(timeout=240)
    proc.join(timeout=60)
    assert proc.exitcode == 0
    model, trainer = _make(unfreeze)
    batch = _batch(0, 8)
    losses1 = [float(trainer.step(batch)) for _ in range(2)]
    torch.cuda.synchronize()
    flat1 = trainer.opt.flat_p.detach().cpu()
    assert abs(losses1[0] - losses_r[0]) < 1e-5 and abs(losses1[1] - losses_r[1]) < 1e-3, (losses1, losses_r)
    assert float((flat1 - torch.from_numpy(flat_r)).norm() / flat1.norm()) < 1e-5


def _worker_dropout(rank, world, port, q, unfreeze=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(123)                                      # same torch seed on both ranks: the rank enters the mask seeds itself
    model, trainer = _make(unfreeze, dropout=True)
    n = 8 // world
    batch = _batch(rank * n, (rank + 1) * n)
    enc = model.audio_encoder
    seeds = enc._dropout_seeds()
    enc._drop_calls = 0
    losses = [float(trainer.step(batch)) for _ in range(2)]
    torch.cuda.synchronize()
    q.put((rank, losses, trainer.opt.flat_p.detach().cpu().numpy(), seeds(0)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("unfreeze", [False, True])
def test_two_ranks_with_train_mode_dropout_stay_in_sync(unfreeze):
    """The reference's train step has dropout live: each rank draws its OWN masks (the rank enters the seeds), the all-reduced
    gradient is the same everywhere, so the replicas' parameters must remain bit-identical after the optimiser steps."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dropout, args=(r, 2, port, q, unfreeze)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, losses, flat, seed0 = q.get(timeout=240)
        got[rank] = (losses, torch.from_numpy(flat), seed0)
    for p in procs:
        p.join(timeout=60)
    assert got[0][2] != got[1][2]                                # different masks per rank
    assert all(l == l for l in got[0][0] + got[1][0])
    assert got[0][0] == got[1][0]                                # the loss is evaluated on the gathered global batch by every rank
    assert torch.equal(got[0][1], got[1][1])


def _initial_flat(unfreeze):
    _, trainer = _make(unfreeze)
    return trainer.opt.flat_p.detach().cpu()
