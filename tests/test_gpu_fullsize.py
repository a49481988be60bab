"""GPU, BASELINE configs[1] full size (B = 64 x 10 s, HuBERT-base): the oracle cannot run this in seconds, so the path
is checked through size-independent properties of the computation:
  * batch-permutation equivariance: every utterance's embedding is BITWISE independent of its row in the batch
    (row-independent kernels, fixed k order, padded layout) - catches any cross-utterance leak through the padded
    row layout, the overlapping conv rows, attention key bounds or tile edges;
  * padding invariance at the fairseq frame mask: appended zero samples beyond wav_len do not change the embedding;
  * the loss is invariant under a joint permutation of the batch and symmetric under swapping the two modalities;
  * recall@k of a query set against itself is 100."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config
    torch.manual_seed(7122)
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    return KWClip_GeneralTransformer(cfg, device="cuda:0").eval()


def test_full_size_properties(model):
    from speechclip_plus_amd import mutualRetrieval
    B, L = 64, 160000
    g = torch.Generator().manual_seed(1)
    wav = torch.randn(B, L, generator=g)
    lens = torch.randint(32000, L + 1, (B,), generator=g)
    lens[0] = L                                   # keep the batch geometry (max length) fixed
    wav = wav * (torch.arange(L)[None] < lens[:, None])
    img = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    ids = torch.arange(B) // 5
    batch = {"wav": wav.cuda(), "wav_len": lens, "image": img.cuda(), "id": ids.cuda()}
    with torch.no_grad():
        l1, _, o1 = model(batch)
        e1 = o1["parallel_audio_feat"].clone()
        assert e1.shape == (B, 512) and torch.isfinite(e1).all()
        loss1 = model.compute_loss(l1)["loss"].item()
        # ---- permutation equivariance (utterance 0 stays first so that max(wav_len) sits in the same place)
        perm = torch.cat([torch.tensor([0]), 1 + torch.randperm(B - 1, generator=g)])
        pb = {"wav": batch["wav"][perm.cuda()], "wav_len": lens[perm], "image": batch["image"][perm.cuda()],
              "id": batch["id"][perm.cuda()]}
        l2, _, o2 = model(pb)
        assert torch.equal(o2["parallel_audio_feat"], e1[perm.cuda()])
        loss2 = model.compute_loss(l2)["loss"].item()
        assert abs(loss1 - loss2) < 2e-5 * max(1.0, abs(loss1))
        # ---- modality swap: the masked InfoNCE is symmetric
        ls = model.criterion(l1["image_feat"].float(), l1["parallel_audio_feat"].float(), l1["id"]).item()
        assert abs(ls - loss1) < 2e-5 * max(1.0, abs(loss1))
        # ---- junk beyond wav_len is ignored (the kernels re-apply the zero padding from the length vector)
        wav3 = batch["wav"].clone()
        for b in range(1, B):
            wav3[b, int(lens[b]):] = 3.0
        _, _, o3 = model({**batch, "wav": wav3})
        assert torch.equal(o3["parallel_audio_feat"], e1)
        # ---- retrieval of the set against itself
        a = torch.nn.functional.normalize(e1.float(), dim=-1)
        r = mutualRetrieval(a @ a.T, a @ a.T, torch.arange(B), torch.arange(B), [1, 5])
        assert r[0]["recall@1"] == 100.0 and r[1]["recall@1"] == 100.0


def test_ten_second_utterances_vs_oracle():
    """Full-length utterances (10 s -> T = 499 frames, and a ragged 7.7 s one) through all 12 layers against the oracle: the
    13 hidden states over the valid frames (rel-L2 <= 2e-2) and the pooled embedding (cosine >= 0.999).  Two utterances keep
    the CPU oracle at a few seconds; the B = 64 geometry is covered by the invariants above."""
    import oracle
    from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict, HubertArch
    sd = random_hubert_state_dict(HubertArch(), seed=7122)
    torch.manual_seed(7122)
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd).eval()
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(torch.linspace(-1, 1, 13))
    g = torch.Generator().manual_seed(3)
    wavs = [torch.randn(160000, generator=g) * 0.5, torch.randn(123456, generator=g) * 0.5]
    with torch.no_grad():
        out = model.encode_speech(wavs)["parallel_audio_feat"].float().cpu()
        _, fl_m, hs_m = model.forward_audio(*model.processWavs(wavs), return_hidden_states=True)
        hs_m = [h.float().cpu() for h in hs_m]
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    hs_o, fl = oracle.speech_encoder_forward(sd, oracle.HubertArch.base(), wavs)
    assert fl_m.cpu().tolist() == fl.tolist() == [499, 386]
    worst = 0.0
    for n in range(13):
        for b, nv in enumerate(fl.tolist()):
            e = float((hs_m[n][b, :nv] - hs_o[n][b, :nv]).norm() / hs_o[n][b, :nv].norm())
            worst = max(worst, e)
    assert worst < 2e-2, worst
    head_W = {k: v.detach().cpu().float() for k, v in model.parallel_branch.state_dict().items()}
    feat = oracle.weighted_sum(model.audio_encoder.weightedsum_layer.weights.detach().cpu(), list(hs_o), False)
    e = oracle.parallel_branch_forward(head_W, feat, fl, nhead=8)
    assert float(torch.nn.functional.cosine_similarity(out, e, dim=-1).min()) > 0.999
    print("10 s utterances: worst hidden-state rel-L2 %.3g" % worst)
