"""world_size-2 gloo test of the data-parallel pieces (speechclip_plus_amd.parallel): the packed all-gather
with autograd and the flat gradient all-reduce reproduce the single-process global-batch gradients.
The loss here is the CPU oracle (the HIP loss needs a GPU); what is under test is the collective plumbing."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from speechclip_plus_amd.parallel import GradAllReduce, gather_loss_feats, scale_replicated_grads
    torch.manual_seed(0)
    Bg, E, D = 8, 16, 12
    X = torch.randn(Bg, D)
    img = torch.nn.functional.normalize(torch.randn(Bg, E), dim=-1)
    ids = torch.tensor([0, 0, 1, 2, 3, 3, 4, 5])
    W = torch.nn.Parameter(torch.randn(E, D) * 0.3)
    logt = torch.nn.Parameter(torch.tensor(2.0))              # trainable log(1 / temperature): a parameter of the loss itself
    # single-process reference on the global batch
    a = torch.nn.functional.normalize(X @ W.t(), dim=-1)
    ref_loss = oracle.masked_contrastive_loss(a, img, ids, inv_temperature=logt.exp())
    ref_grad, ref_gt = torch.autograd.grad(ref_loss, [W, logt])
    # data parallel: each rank owns Bg / world rows
    n = Bg // world
    sl = slice(rank * n, (rank + 1) * n)
    a_loc = torch.nn.functional.normalize(X[sl] @ W.t(), dim=-1)
    a_all, i_all, id_all = gather_loss_feats(a_loc, img[sl], ids[sl])
    assert torch.equal(id_all, ids) and torch.allclose(i_all, img)
    loss = oracle.masked_contrastive_loss(a_all, i_all, id_all, inv_temperature=logt.exp())
    loss.backward()
    scale_replicated_grads([logt])                            # every rank holds the full temperature gradient
    flat = torch.cat([W.grad.reshape(-1), logt.grad.reshape(1)]).clone()
    ar = GradAllReduce(flat)
    ar.launch()
    ar.wait()
    gW, gt = flat[:-1].view_as(W), flat[-1]
    ok = (abs(loss.item() - ref_loss.item()) < 1e-6 and torch.allclose(gW, ref_grad, atol=1e-6)
          and abs(float(gt) - float(ref_gt)) < 1e-6)
    q.put((rank, bool(ok), float((gW - ref_grad).abs().max()), float(gt), float(ref_gt)))
    dist.destroy_process_group()


def _worker_accumulate(rank, world, port, q):
    """accumulate_grad_batches = 2 over two ranks (spchclip_h+.yaml:138): each rank back-propagates loss / 2 of two micro-batches into
    ONE flat buffer and the SUM all-reduce runs on the boundary micro-step only; the result must be the mean over the micro-batches of
    the single-process global-batch gradients, and exactly one collective per window may be issued."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from speechclip_plus_amd import parallel
    from speechclip_plus_amd.parallel import AccumulationSchedule, GradAllReduce, gather_loss_feats
    torch.manual_seed(0)
    Bg, E, D, n_acc = 8, 16, 12, 2
    W = torch.nn.Parameter(torch.randn(E, D) * 0.3)
    data = [(torch.randn(Bg, D), torch.nn.functional.normalize(torch.randn(Bg, E), dim=-1), torch.arange(Bg) // 2) for _ in range(2 * n_acc)]
    calls = [0]
    real = dist.all_reduce

    def counted(*a, **k):
        calls[0] += 1
        return real(*a, **k)
    parallel.dist.all_reduce = counted
    sched = AccumulationSchedule(n_acc)
    flat = torch.zeros(W.numel())
    ar = GradAllReduce(flat)
    n = Bg // world
    sl = slice(rank * n, (rank + 1) * n)
    ok, worst = True, 0.0
    for step, (X, img, ids) in enumerate(data):
        a_loc = torch.nn.functional.normalize(X[sl] @ W.t(), dim=-1)
        a_all, i_all, id_all = gather_loss_feats(a_loc, img[sl], ids[sl])
        loss = oracle.masked_contrastive_loss(a_all, i_all, id_all)
        (g,) = torch.autograd.grad(loss * sched.loss_scale, [W])
        flat += g.reshape(-1)
        before = calls[0]
        if sched.advance():
            ar.launch()
            ar.wait()
            ref = torch.zeros_like(W)
            for Xr, imr, idr in data[step + 1 - n_acc: step + 1]:
                ar_ = torch.nn.functional.normalize(Xr @ W.t(), dim=-1)
                ref += torch.autograd.grad(oracle.masked_contrastive_loss(ar_, imr, idr), [W])[0] / n_acc
            worst = max(worst, float((flat.view_as(W) - ref).abs().max()))
            ok = ok and calls[0] == before + 1
            flat.zero_()
        else:
            ok = ok and calls[0] == before                   # inside the window: no gradient collective
    parallel.dist.all_reduce = real
    q.put((rank, bool(ok and worst < 1e-6 and calls[0] == 2), worst, calls[0]))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_accumulate_grad_batches_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_accumulate, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] for r in res), res


@pytest.mark.timeout(120)
def test_gather_and_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] for r in res), res


@pytest.mark.timeout(180)
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` as the driver calls it (no WORLD_SIZE in the environment): the parent starts the two ranks
    itself, they rendezvous on 127.0.0.1, run the step's collectives (here on CPU tensors over gloo: --rehearse-launch, the
    container has no GPU) and exactly one JSON line with n_gpus = rccl_ranks = 2 comes back with exit code 0."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SC_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch", "--batch", "4"],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dist_backend"] == "gloo"
    # the N > 1 diagnostics: per-collective time, exposed time, per-rank step time - present and sane
    c = rec["collectives"]
    assert c["ms_per_step_by_rank"]["all"] == [1.0, 2.0] and c["ms_per_step_by_rank"]["max"] == 2.0
    for key in ("all_gather_us", "all_reduce_us", "join_wait_us"):
        assert set(c[key]) >= {"rank0", "min", "max", "calls_per_step"} and c[key]["min"] <= c[key]["max"]
    assert c["all_gather_us"]["calls_per_step"] == 1 and c["all_reduce_us"]["calls_per_step"] == 1 and c["all_gather_us"]["rank0"] > 0
    assert abs(c["exposed_us_per_step"]["rank0"] - c["all_gather_us"]["rank0"] - c["join_wait_us"]["rank0"]) < 0.2
    # a failing rank must surface as a non-zero exit code of the parent
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch", "--batch", "-1"],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode != 0
