"""Batch format of the reference's loaders (scope row f4): ``collate_general`` builds the dict that
``KWClip_GeneralTransformer.forward`` consumes - mirror of avssl/data/collate_function.py:7-36.

Semantics kept: a ``wav_len`` key is derived from the un-padded waveforms, ``wav`` is zero-padded to the longest
utterance of the batch (batch first), other tensors are stacked, non-tensor fields (ids, lengths) become int64 tensors.

Round 5 - the way a batch reaches the device (avssl/model/kwClip.py:145-147 receives what Lightning's transfer produced):
``transfer_batch_to_device`` is the body of the LightningModule hook of the same name.  It keeps what a plain ``.to(device)`` of
every tensor loses: the lengths stay readable on the HOST (``wav_len._sc_host`` - the ragged row layout and the crop offsets are
host decisions, and reading a device tensor back would synchronise every step) and the waveform's H2D copy runs on a copy stream
of its own with a completion event (``wav._sc_ready``), so the encoder - a step ahead of the caller's stream on the overlapped
schedule - starts when the DATA is there, not when the caller's stream gets there.  ``collate_general(pin_memory=True)`` builds the
padded waveform in pinned memory so that copy is asynchronous.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.utils.data


def attach_host_lengths(wav_len: torch.Tensor, host: Optional[Sequence[int]] = None) -> torch.Tensor:
    """``wav_len`` with its host twin attached (a python attribute: it does not survive ``.to()`` - attach it to the tensor the
    model receives).  ``host`` defaults to the tensor's own values, which must then live on the CPU."""
    if host is None:
        if wav_len.is_cuda:
            raise ValueError("attach_host_lengths: a device tensor needs the host values passed in (reading it back would synchronise)")
        host = wav_len.tolist()
    wav_len._sc_host = [int(v) for v in host]
    return wav_len


def transfer_batch_to_device(batch: dict, device, copy_stream: Optional["torch.cuda.Stream"] = None) -> dict:
    """Body of ``LightningModule.transfer_batch_to_device(batch, device, dataloader_idx)`` for the batch dict of ``collate_general``.
    Every tensor goes to ``device``; ``wav_len`` keeps a host twin; EVERY host -> device copy runs on ``copy_stream`` (default: the
    package's shared "h2d" stream) from pinned memory, and one completion event behind them is (a) attached to ``wav`` - the encoder
    stream waits for it, a step ahead of the caller's stream - and (b) waited for by the caller's stream, which reads the others.

    Why not ``.to(device, non_blocking=True)`` on the caller's stream: measured on MI355X / ROCm 7.2, a 128 KiB copy - even from pinned
    memory - queued on a stream that holds a step's worth of kernels blocks the HOST for 1.2-1.8 ms per step (a pageable one likewise),
    on an idle copy stream it returns in 20 us (tools/h2d_probe3.py).  No host synchronisation in here when the batch is pinned
    (``collate_general(pin_memory=True)`` / a DataLoader with ``pin_memory=True``); pageable tensors are pinned first (a host copy)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:        # torch.device("cuda") != cuda:0: compare (and skip) on the resolved index
        device = torch.device("cuda", torch.cuda.current_device())
    if device.type != "cuda":
        out = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        if isinstance(batch.get("wav_len"), torch.Tensor):
            host = getattr(batch["wav_len"], "_sc_host", None) or (batch["wav_len"].tolist() if not batch["wav_len"].is_cuda else None)
            if host is not None:
                attach_host_lengths(out["wav_len"], host)
        return out
    from . import ops
    cs = copy_stream if copy_stream is not None else ops.shared_stream("h2d", device, priority=-1)
    main = torch.cuda.current_stream(device)
    out, moved = {}, []
    with torch.cuda.stream(cs):
        for k, v in batch.items():
            if not isinstance(v, torch.Tensor) or v.device == device:
                out[k] = v
                continue
            if v.is_cuda:                      # device -> device (another GPU): ordered on the CALLER's stream like any torch op - its
                with torch.cuda.stream(main):  # producer may still be queued there, and the copy-stream event must not stand in for it
                    out[k] = v.to(device, non_blocking=True)
                if k == "wav_len" and getattr(v, "_sc_host", None) is not None:
                    attach_host_lengths(out[k], v._sc_host)
                continue
            src = v if v.is_pinned() else v.pin_memory()
            d = src.to(device, non_blocking=True)
            if k == "wav_len":
                attach_host_lengths(d, getattr(v, "_sc_host", None) or v.tolist())
            out[k] = d
            moved.append(d)                    # host -> device copies only: these are what the event stands for
        ev = torch.cuda.Event()
        ev.record(cs)
    if moved:
        main.wait_event(ev)                    # (enqueued behind the previous step's kernels: the copies are long done by then)
        for d in moved:
            d.record_stream(main)
        if isinstance(out.get("wav"), torch.Tensor) and any(d is out["wav"] for d in moved):
            out["wav"]._sc_ready = ev
    return out


def collate_general(batch: Sequence[dict], pin_memory: bool = False) -> Dict[str, torch.Tensor]:
    if len(batch) == 0:
        raise ValueError("empty batch")
    keys: List[str] = list(batch[0].keys())
    # pin_memory inside a DataLoader WORKER (collate_fn runs there when num_workers > 0, as in the reference's loaders) would initialise
    # CUDA in a forked process ("Cannot re-initialize CUDA in forked subprocess") or create a GPU context per worker, and pinned-ness
    # does not survive the worker -> main hand-over anyway: there, leave it to DataLoader(pin_memory=True) + transfer_batch_to_device
    in_worker = torch.utils.data.get_worker_info() is not None
    pin = lambda t: bool(pin_memory) and not in_worker and torch.cuda.is_available() and (t is None or not t.is_cuda)     # every tensor
    derive_len = "wav" in keys and isinstance(batch[0]["wav"], torch.Tensor)
    out: Dict[str, torch.Tensor] = {}
    for k in keys:
        vals = [row[k] for row in batch]
        if isinstance(vals[0], torch.Tensor):
            if k == "wav":
                L = max(int(v.shape[0]) for v in vals)
                padded = torch.zeros((len(vals), L) + tuple(vals[0].shape[1:]), dtype=vals[0].dtype, pin_memory=pin(vals[0]),
                                     device=vals[0].device)
                for i, v in enumerate(vals):
                    padded[i, : v.shape[0]] = v
                out[k] = padded
            else:
                out[k] = torch.stack(vals, dim=0)
                if pin(out[k]):
                    out[k] = out[k].pin_memory()
        else:
            out[k] = torch.tensor(vals, dtype=torch.long, pin_memory=pin(None))
    if derive_len:
        out["wav_len"] = attach_host_lengths(torch.tensor([int(row["wav"].shape[0]) for row in batch], dtype=torch.long,
                                                          pin_memory=pin(None)))
    return out
