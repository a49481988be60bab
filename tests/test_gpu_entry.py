"""GPU: the reference's own TRAINING ENTRY as the fast path.

Every shipped recipe feeds whole utterances and crops them to ``max_audio_len: 102400`` samples INSIDE the encoder forward in train
mode (avssl/module/speech_encoder_plus.py:548-552 -> avssl/data/audio_transforms.py:5-23), on a batch Lightning has already moved to
the device (avssl/model/kwClip.py:145-147; ``wav_len`` from avssl/data/collate_function.py:7-36).  Here that route is: crop offsets
drawn on the host (the reference's stream of draws), uploaded with the step's other integers, and the two kernels that read the
caller's batch (sc_wav_prep*_crop, sc_conv0_stats_len_crop) read utterance b from ``wav[b, off_b : off_b + len_b]`` in place - no
list of slices, no re-padded copy; lengths keep a host twin through the transfer (data.transfer_batch_to_device), so the rows stay
ragged and nothing reads the device back.

What must hold: the cropped forward equals the forward of the PRE-CROPPED batch bit for bit, for every way a batch can arrive; a
train step through the Lightning-shaped transfer raises nothing under torch's sync debug mode; and the overlapped schedule stays
correct with two forwards outstanding (f1 f2 b1 b2 f3)."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CAP = 16000
LENS = [40000, 16000, 9000, 16001, 33000, 12345, 27000, 40000]      # at the cap (no draw), one over it (randint(1)), under, well over


def _rows(lens, seed=5):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(l, generator=g) for l in lens]


@pytest.fixture(scope="module")
def setup():
    from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict, set_dropout
    from speechclip_plus_amd.speech_encoder import ARCHS
    arch = dataclasses.replace(ARCHS["hubert"], layers=2)
    sd = random_hubert_state_dict(arch, seed=11)

    def make(max_audio_len):
        torch.manual_seed(11)
        cfg = base_parallel_config()
        cfg.audio_encoder.max_audio_len = max_audio_len
        m = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch)
        with torch.no_grad():
            m.audio_encoder.weightedsum_layer.weights.copy_(torch.linspace(-1, 1, arch.layers + 1))
        return set_dropout(m.train(), False)        # train mode (the crop is a train-mode step), deterministic

    return make


def _precropped(rows, offs, outl):
    L = max(outl)
    wav = torch.zeros(len(rows), L)
    for b, (r, o, n) in enumerate(zip(rows, offs, outl)):
        wav[b, :n] = r[o: o + n]
    return wav


@pytest.mark.parametrize("arrives", ["device_batch_host_lengths", "pinned_host_batch", "lightning_transfer", "device_lengths_only",
                                     "list_of_waveforms"])
def test_in_forward_crop_equals_the_precropped_batch_bit_for_bit(setup, arrives):
    from speechclip_plus_amd.data import collate_general
    from speechclip_plus_amd.speech_encoder import crop_windows
    rows = _rows(LENS)
    batch = collate_general([{"wav": r} for r in rows], pin_memory=True)
    crop_model, plain_model = setup(CAP), setup(-1)
    enc_c, enc_p = crop_model.audio_encoder, plain_model.audio_encoder
    # what the reference would cut under this seed
    np.random.seed(77)
    if arrives == "device_lengths_only":
        u = np.random.random_sample(len(LENS))
        room = np.maximum(np.asarray(LENS) - CAP, 0)
        offs = np.minimum((u * room).astype(np.int64), np.maximum(room - 1, 0)).tolist()
        outl = [min(l, CAP) for l in LENS]
    else:
        offs, outl = crop_windows(LENS, CAP)
    assert offs[1] == 0 and offs[2] == 0 and offs[3] == 0 and max(offs) > 0 and outl == [min(l, CAP) for l in LENS]
    ref_wav = _precropped(rows, offs, outl).cuda()
    with torch.no_grad():
        if arrives == "device_lengths_only":
            f_ref, l_ref = enc_p(ref_wav, torch.tensor(outl).cuda())
        else:
            f_ref, l_ref = enc_p(ref_wav, torch.tensor(outl))
        f_ref, l_ref = f_ref.float().clone(), l_ref.clone()
        np.random.seed(77)
        if arrives == "device_batch_host_lengths":
            f, l = enc_c(batch["wav"].cuda(), batch["wav_len"])
        elif arrives == "pinned_host_batch":
            assert batch["wav"].is_pinned()
            f, l = enc_c(batch["wav"], batch["wav_len"])
        elif arrives == "lightning_transfer":
            moved = crop_model.transfer_batch_to_device(batch, torch.device("cuda:0"))
            assert moved["wav"].is_cuda and isinstance(moved["wav"]._sc_ready, torch.cuda.Event)
            assert moved["wav_len"].is_cuda and moved["wav_len"]._sc_host == LENS
            f, l = enc_c(moved["wav"], moved["wav_len"])
        elif arrives == "device_lengths_only":
            wd, ld = batch["wav"].cuda(), batch["wav_len"].cuda().clone()
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("error")
            try:
                f, l = enc_c(wd, ld)
            finally:
                torch.cuda.set_sync_debug_mode("default")
        else:
            f, l = enc_c([r.cuda() for r in rows], [])
    assert torch.equal(l.cpu(), l_ref.cpu())
    assert f.shape == f_ref.shape and torch.equal(f.float(), f_ref)
    assert float(f_ref.abs().sum()) > 0


def test_train_steps_with_the_crop_equal_train_steps_on_precropped_batches(setup):
    """Whole train steps (encoder ahead on its own stream, head, loss, backward, optimiser), four different batches through the
    Lightning-shaped transfer with the crop in the forward, against the same steps on host-cropped batches: losses and parameters
    bit-identical."""
    from speechclip_plus_amd.data import collate_general
    from speechclip_plus_amd.speech_encoder import crop_windows
    from speechclip_plus_amd.train import ContrastiveTrainer
    g = torch.Generator().manual_seed(9)
    all_lens = [LENS, LENS[::-1], [30000, 8000, 16000, 24000, 40000, 17000, 5000, 16500], [16000] * 8]
    B = len(LENS)
    data = [(_rows(l, seed=20 + i), torch.randn(B, 512, generator=g), torch.arange(B)) for i, l in enumerate(all_lens)]
    finals = []
    for cropped_in_forward in (True, False):
        model = setup(CAP if cropped_in_forward else -1)
        trainer = ContrastiveTrainer(model)
        np.random.seed(123)
        losses = []
        for (rows, img, ids), lens in zip(data, all_lens):
            if cropped_in_forward:
                batch = collate_general([{"wav": r, "image": im, "id": int(i)} for r, im, i in zip(rows, img, ids)], pin_memory=True)
                batch = model.transfer_batch_to_device(batch)
            else:
                offs, outl = crop_windows(lens, CAP)
                batch = {"wav": _precropped(rows, offs, outl).cuda(), "wav_len": torch.tensor(outl), "image": img.cuda(), "id": ids.cuda()}
            losses.append(float(trainer.step(batch)))
        torch.cuda.synchronize()
        finals.append((losses, trainer.opt.flat_p.clone()))
    assert finals[0][0] == finals[1][0], finals
    assert torch.equal(finals[0][1], finals[1][1])
    assert len(set(finals[0][0])) == len(data)


def test_train_step_through_the_lightning_transfer_needs_no_host_read(setup):
    """collate (pinned) -> transfer_batch_to_device (copy stream + event, host twin of the lengths) -> train step with the in-forward
    crop, ragged rows and the encoder a step ahead: no call in it may synchronise the host with the device."""
    from speechclip_plus_amd.data import collate_general
    from speechclip_plus_amd.train import ContrastiveTrainer
    model = setup(CAP)
    trainer = ContrastiveTrainer(model)
    g = torch.Generator().manual_seed(4)
    B = len(LENS)
    mk = lambda i: collate_general([{"wav": r, "image": torch.randn(512, generator=g), "id": b}
                                    for b, r in enumerate(_rows(LENS[i:] + LENS[:i], seed=40 + i))], pin_memory=True)
    host_batches = [mk(i) for i in range(4)]
    trainer.step(model.transfer_batch_to_device(host_batches[0]))          # builds plans / streams / workspaces
    trainer.step(model.transfer_batch_to_device(host_batches[1]))
    torch.cuda.synchronize()
    enc = model.audio_encoder
    torch.cuda.set_sync_debug_mode("error")
    try:
        losses = [trainer.step(model.transfer_batch_to_device(hb)) for hb in host_batches[2:] + host_batches[:2]]
    finally:
        torch.cuda.set_sync_debug_mode("default")
    vals = [float(l) for l in losses]
    assert all(v == v for v in vals)
    pl = next(reversed(enc._plans.values()))
    assert pl.seg is not None and pl.M < B * pl.R and pl.wav_off is not None       # ragged rows + a device-side crop were in play
    assert enc._enc_stream is not None


def test_two_outstanding_forwards_then_their_backwards_on_the_overlapped_schedule(setup):
    """f1 f2 b1 b2 f3 in train mode.  f3 re-uses f1's plan while - before the fix - the encoder stream only
    waited for the ENTRY of f2, which b1 was enqueued after: f3's encoder could overwrite the hidden states b1's weighted-sum backward
    still reads.  Now b1 records an event on the plan and f3's encoder waits for it.  The gradients of b1 / b2 must equal those of the
    strictly sequential single-stream schedule; a stall between b1's launch and its reads (a long kernel queued in front of it) makes
    the race reproducible if it exists."""
    from speechclip_plus_amd import ops
    g = torch.Generator().manual_seed(31)
    lens = [[24000, 17000, 24000, 9000], [24000, 24000, 12000, 20000], [16000, 24000, 24000, 7000]]
    mk = lambda l: {"wav": torch.randn(4, 24000, generator=g).cuda(), "wav_len": torch.tensor(l), "image": torch.randn(4, 512, generator=g).cuda(),
                    "id": torch.arange(4).cuda()}
    b = [mk(l) for l in lens]
    torch.cuda.synchronize()
    for x in b:
        x["wav"]._sc_ready = True
    spin = torch.empty(1 << 28, device="cuda")        # 1 GiB: fills of it keep the caller's stream busy for a while
    results = []
    for overlap in (False, True):
        model = setup(-1)
        model.audio_encoder.enc_overlap = overlap
        params = [p for p in model.getTrainableParams()]
        grads = []

        def fwd(x):
            return model.compute_loss(model(x)[0])["loss"]

        def bwd(loss):
            for p in params:
                p.grad = None
            for _ in range(6):
                spin.fill_(1.0)                        # the backward's kernels sit behind ~milliseconds of queued work
            loss.backward()
            return [p.grad.detach().clone() for p in params if p.grad is not None]

        if overlap:
            l1, l2 = fwd(b[0]), fwd(b[1])
            grads.append(bwd(l1))
            grads.append(bwd(l2))
            l3 = fwd(b[2])                              # re-uses f1's plan, enqueued while b1 / b2 are still queued behind the fills
            grads.append(bwd(l3))
        else:
            for x in b:
                grads.append(bwd(fwd(x)))
        torch.cuda.synchronize()
        results.append(grads)
    assert len(results[0][0]) > 3
    for step in range(3):
        for ga, gb in zip(results[0][step], results[1][step]):
            assert torch.equal(ga, gb), step


def test_a_third_forward_before_the_first_backward_still_fails_loudly(setup):
    """Two outstanding forwards are the limit of the overlapped schedule (two alternating plans): a third one overwrites the first
    one's hidden states, and its late backward must raise instead of differentiating against them."""
    g = torch.Generator().manual_seed(2)
    mk = lambda: {"wav": torch.randn(3, 9000, generator=g).cuda(), "wav_len": torch.tensor([9000, 6000, 7000]),
                  "image": torch.randn(3, 512, generator=g).cuda(), "id": torch.arange(3).cuda()}
    model = setup(-1)
    l1 = model.compute_loss(model(mk())[0])["loss"]
    l2 = model.compute_loss(model(mk())[0])["loss"]
    l3 = model.compute_loss(model(mk())[0])["loss"]
    with pytest.raises(RuntimeError, match="another forward"):
        l1.backward()
    l2.backward()
    l3.backward()
    torch.cuda.synchronize()
