"""Recall@k parity on the Flickr8k-test-shaped synthetic eval set (1000 image ids x 5 utterances), shared by
tests/test_gpu_recall.py, bench.py's ``recall`` field and the fixture generator tests/golden/make_recall_fixture.py.

Everything here is seeded and oracle-free: waveforms and weights come from CPU torch.Generator streams (identical on every host
with this torch build), so the only thing that has to travel from the build container is tests/golden/recall_eval.npz (image
embeddings + the oracle's ranks / recalls).  The weight generators draw the same streams as oracle.init_hubert_weights /
oracle.init_parallel_branch_weights (tests/test_host_cpu.py checks that), without importing the oracle.
"""
import math
import os
from typing import Dict

import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
FIXTURE = os.path.join(ROOT, "tests", "golden", "recall_eval.npz")
FIXTURE_NATURAL = os.path.join(ROOT, "tests", "golden", "recall_eval_natural.npz")    # the same utterances against a gallery with natural margins
SEED_W, SEED_HEAD, SEED_DATA = 7122, 7123, 20261
PER_ID = 5
GALLERY = 3      # captions 0..2 of every id build its image; captions 3, 4 are held out of the construction
WS_WEIGHTS = torch.linspace(-1, 1, 13)


SR = 16000.0
FRAME = 320          # the encoder's hop (5 * 2^6 samples): caption shifts are whole frames, see utterance()
_sig_cache: Dict[int, tuple] = {}


def signature(k: int):
    """The sound of image id k: 1.9 - 2.5 s of a 4-band mixture - per band an id-specific centre frequency (150 Hz .. 5.9 kHz,
    log-uniform), 6 partials inside an id-specific bandwidth, an id-specific amplitude and a squared-sine AM envelope at an
    id-specific rate (1.5 - 9.5 Hz).  Unit variance.  Every id therefore differs from every other in spectrum AND in envelope
    (round 2 used white noise with identical statistics for all ids, which made every rank a near-tie)."""
    if k not in _sig_cache:
        g = torch.Generator().manual_seed(SEED_DATA + k)
        L = int(torch.randint(30400 // FRAME, 40000 // FRAME + 1, (1,), generator=g)) * FRAME
        t = torch.arange(L + 8 * FRAME).double() / SR
        nb = 4
        fc = 150.0 * (2.0 ** (torch.rand(nb, generator=g).double() * 5.3))
        bw = 30.0 + 200.0 * torch.rand(nb, generator=g).double()
        am = 1.5 + 8.0 * torch.rand(nb, generator=g).double()
        amp = 0.4 + torch.rand(nb, generator=g).double()
        x = torch.zeros_like(t)
        for c in range(nb):
            f = fc[c] + bw[c] * (torch.rand(6, generator=g).double() - 0.5)
            ph = 2 * math.pi * torch.rand(6, generator=g).double()
            car = torch.sin(2 * math.pi * f[:, None] * t[None] + ph[:, None]).sum(0) / 6 ** 0.5
            env = 0.5 * (1.0 + torch.sin(2 * math.pi * am[c] * t + 2 * math.pi * torch.rand(1, generator=g).double()))
            x += amp[c] * env ** 2 * car
        _sig_cache[k] = ((x / x.std()).float(), L)
    return _sig_cache[k]


def utterance(k: int, j: int) -> torch.Tensor:
    """Caption j of image id k = the id's signature, shifted in time by 0 .. 7 whole encoder frames, cut 0 .. 9 frames short,
    at a gain of 0.8 .. 1.2, plus white noise at -22 dB.  Shifts are multiples of the 320-sample hop on purpose: a random-weight
    conv stack (stride 5 x 2^6, no anti-aliasing learnt) is not invariant to sub-frame shifts - measured: centred embedding cosine
    between two captions of one id 0.39 with arbitrary shifts vs 0.98 with whole-frame shifts - and a caption that decorrelates
    from its own image says nothing about arithmetic parity."""
    x, L = signature(k)
    gj = torch.Generator().manual_seed(SEED_DATA * 7 + k * PER_ID + j)
    shift = int(torch.randint(0, 8, (1,), generator=gj)) * FRAME
    lj = L - int(torch.randint(0, 10, (1,), generator=gj)) * FRAME
    gain = 0.8 + 0.4 * float(torch.rand(1, generator=gj))
    return gain * x[shift: shift + lj] + 0.08 * torch.randn(lj, generator=gj)


def eval_set(n_ids: int):
    """-> (list of n_ids * 5 waveforms, ids [n_ids * 5])"""
    wavs = [utterance(k, j) for k in range(n_ids) for j in range(PER_ID)]
    return wavs, torch.arange(n_ids).repeat_interleave(PER_ID)


def head_weights(d_model: int = 768, ffn: int = 3072, out_dim: int = 512, seed: int = SEED_HEAD) -> Dict[str, torch.Tensor]:
    """Seeded parallel-branch weights under the reference's state-dict names (1 post-LN layer).  The CLS token is scaled to 0.1: at
    unit scale the residual path of the CLS row (a constant) dominates the pooled output and the embeddings of all utterances
    collapse onto one direction (|mean| = 0.999), which makes every rank a near-tie; a trained head does not do that."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def randn(*shape, s):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * s

    W = {"cls": randn(1, 1, d_model, s=1.0)}
    p = "self_att.model.layers.0."
    W[p + "self_attn.in_proj_weight"] = randn(3 * d_model, d_model, s=d_model ** -0.5)
    W[p + "self_attn.in_proj_bias"] = randn(3 * d_model, s=0.05)
    W[p + "self_attn.out_proj.weight"] = randn(d_model, d_model, s=d_model ** -0.5)
    W[p + "self_attn.out_proj.bias"] = randn(d_model, s=0.05)
    W[p + "linear1.weight"] = randn(ffn, d_model, s=d_model ** -0.5)
    W[p + "linear1.bias"] = randn(ffn, s=0.05)
    W[p + "linear2.weight"] = randn(d_model, ffn, s=ffn ** -0.5)
    W[p + "linear2.bias"] = randn(d_model, s=0.05)
    for n in ("norm1", "norm2"):
        W[p + n + ".weight"] = 1.0 + randn(d_model, s=0.1)
        W[p + n + ".bias"] = randn(d_model, s=0.1)
    W["self_att.model.norm.weight"] = 1.0 + randn(d_model, s=0.1)
    W["self_att.model.norm.bias"] = randn(d_model, s=0.1)
    W["linear_proj.weight"] = randn(out_dim, d_model, s=d_model ** -0.5)
    W["linear_proj.bias"] = randn(out_dim, s=0.05)
    W["cls"] = W["cls"] * 0.1
    return W


def hubert_weights() -> Dict[str, torch.Tensor]:
    from speechclip_plus_amd import HubertArch, random_hubert_state_dict
    return random_hubert_state_dict(HubertArch(), seed=SEED_W)


PLAN = ((1, 200), (3, 50), (7, 20), (14, 6))     # (images ranked above the own image, number of such ids)


def build_gallery(a_o: torch.Tensor, ids: torch.Tensor, n_ids: int, seed: int = SEED_DATA + 99, reg: float = 1.0, r_sub: int = 128,
                  safety: float = 1.45, vic_frac: float = 0.78, plan=PLAN):
    """The 1000 "frozen CLIP" image embeddings of the eval set, built from the ORACLE's unit audio embeddings ``a_o`` [5 n_ids, E]
    (gallery captions 0..2 of every id only; captions 3, 4 never enter an image) -> (images [n_ids, E] unit, role [n_ids]).

    Round 2 drew image = class mean + isotropic noise tuned to recall@1 = 50 %: margins (own score - best other score) were then
    continuously distributed through zero and ~1 % of the queries sat inside the bf16 noise - no implementation other than the
    oracle itself can reproduce such a set to +-0.1.  This gallery has its confusions PLANTED at discrete score levels, far apart
    compared with the noise (the fixture script commits the resulting margin histogram):

    * every id gets a "pure" direction P_k = whitened class centre (regularised inverse covariance of the centred embeddings:
      cross-talk between ids drops from 0.54 to 0.32 of the own score on average), made orthogonal to the mean embedding so that
      score = (a - mean) . image and the dominant common direction of random-weight embeddings drops out of every rank;
    * "victim" ids (PLAN: 200 / 50 / 20 / 6 ids with 1 / 3 / 7 / 14 competitors) have a weakened own image t P_k + sqrt(1 - t^2) r_k
      (r_k from the 128 lowest-variance directions of the embedding covariance), t chosen so that the own score is vic_frac of the
      weakest competitor's;
    * every competitor is carried by ONE other id's image ("carrier": x P_j + y P_v, solved on the class centres), with the
      victim's score rho = 1.5 .. 2.2 times the carrier's own - rho from the spread of the carrier's / victim's gallery captions, so
      that on image j all five captions of v outrank the best caption of j (image -> audio ranks are then discrete too);
    * the remaining ids keep their pure image.
    Expected A->I recall@1/5/10 = 72.4 / 97.4 / 99.4, I->A = 42.6 / 42.6 / 100 by construction."""
    N, E = a_o.shape
    cap = torch.arange(N) % PER_ID
    gal = cap < GALLERY
    mu = a_o.mean(0)
    d = a_o - mu
    muh = mu / mu.norm()
    C = torch.stack([d[(ids == k) & gal].mean(0) for k in range(n_ids)])
    ev, V = torch.linalg.eigh((d.t() @ d / N).double())
    Winv = (V @ torch.diag(1.0 / (ev + ev.mean() * reg)) @ V.t()).float()
    P = C @ Winv
    P = P - (P @ muh)[:, None] * muh
    P = P / P.norm(dim=-1, keepdim=True)
    pure = (C * P).sum(-1)
    own_g = (d * P[ids]).sum(-1) / pure[ids]            # a caption's score with its own pure image, relative to the centre's
    hi = torch.stack([own_g[(ids == k) & gal].max() for k in range(n_ids)])
    lo = torch.stack([own_g[(ids == k) & gal].min() for k in range(n_ids)])
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(n_ids, generator=g).tolist()
    victims = []
    for m, cnt in plan:
        victims += [(perm.pop(), m) for _ in range(cnt)]
    slots = [v for v, m in victims for _ in range(m)]
    carriers = [perm.pop() for _ in range(len(slots))]
    order = torch.randperm(len(slots), generator=g).tolist()
    img = P.clone()
    role = torch.zeros(n_ids, dtype=torch.long)          # 0 pure, 1 carrier, 1 + m victim with m competitors
    weakest = {}
    for j, si in zip(carriers, order):
        v = slots[si]
        rho = min(max(safety * float(hi[j] / lo[v]), 1.5), 2.2)
        G = torch.tensor([[float(C[j] @ P[j]), float(C[j] @ P[v])], [float(C[v] @ P[j]), float(C[v] @ P[v])]])
        x = torch.linalg.solve(G, torch.tensor([1.0, rho]))
        w = x[0] * P[j] + x[1] * P[v]
        img[j] = w / w.norm()
        role[j] = 1
        weakest[v] = min(weakest.get(v, 9.0), float(C[v] @ img[j]))
    for v, m in victims:
        t = min(vic_frac * weakest[v] / float(pure[v]), 0.95)
        r = V[:, 1: 1 + r_sub].float() @ torch.randn(r_sub, generator=g)
        r = r - (r @ muh) * muh
        r = r - (r @ P[v]) * P[v]
        r = r / r.norm()
        img[v] = t * P[v] + (1 - t * t) ** 0.5 * r
        role[v] = 1 + m
    return img, role


def natural_gallery(a_o: torch.Tensor, ids: torch.Tensor, n_ids: int, noise: float, seed: int = SEED_DATA + 177) -> torch.Tensor:
    """Image embeddings with NATURAL margins (the second fixture, tests/golden/make_recall_natural_fixture.py): class centre of the
    three gallery captions + isotropic noise of relative size ``noise``, orthogonal to the mean embedding, unit norm.  Nothing is
    planted: the own-minus-best-other margins pass continuously through zero, as they did in round 2's set."""
    N, E = a_o.shape
    cap = torch.arange(N) % PER_ID
    gal = cap < GALLERY
    mu = a_o.mean(0)
    d = a_o - mu
    muh = mu / mu.norm()
    C = torch.stack([d[(ids == k) & gal].mean(0) for k in range(n_ids)])
    C = C / C.norm(dim=-1, keepdim=True)
    g = torch.Generator().manual_seed(seed)
    eps = torch.randn(n_ids, E, generator=g) / E ** 0.5
    img = C + noise * eps
    img = img - (img @ muh)[:, None] * muh
    return img / img.norm(dim=-1, keepdim=True)


def rank_stats(a_unit: torch.Tensor, image: torch.Tensor, ids: torch.Tensor) -> dict:
    """Scores as the validation epoch computes them (kwClip.py:467-471: plain dot products of unit vectors) -> for every utterance
    the number of images above its own (A->I rank), for every image the number of foreign captions above its best own caption
    (I->A rank), and the margins at the rank boundaries 1 / 5 / 10 (own score - k-th best other score)."""
    n_ids = image.shape[0]
    S = a_unit @ image.t()
    own = S.gather(1, ids.unsqueeze(1)).squeeze(1)
    other = S.scatter(1, ids.unsqueeze(1), -9.0)
    top, top_idx = other.topk(10, dim=1)
    St = S.t()
    corr = ids.unsqueeze(0) == torch.arange(n_ids, device=ids.device).unsqueeze(1)
    best = St.masked_fill(~corr, -9.0).amax(1)
    topf = St.masked_fill(corr, -9.0).topk(10, dim=1).values
    return {"rank_ai": (S > own.unsqueeze(1)).sum(1), "rank_ia": (St > best.unsqueeze(1)).sum(1),
            "margin_ai": torch.stack([own - top[:, k - 1] for k in (1, 5, 10)], 1),
            "margin_ia": torch.stack([best - topf[:, k - 1] for k in (1, 5, 10)], 1),
            "own": own, "kth_idx": torch.stack([top_idx[:, k - 1] for k in (1, 5, 10)], 1), "scores": S}


def recalls(rank: torch.Tensor, mask=None) -> list:
    r = rank if mask is None else rank[mask]
    return [round(100.0 * float((r < k).float().mean()), 2) for k in (1, 5, 10)]


def correct_rank(score: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """number of images scoring strictly above the utterance's own image"""
    own = score.gather(1, ids.unsqueeze(1))
    return (score > own).sum(1)


def build_model(device: str = "cuda:0"):
    """The product model (Parallel SpeechCLIP base) on the eval set's seeded weights."""
    from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    model = KWClip_GeneralTransformer(cfg, device=device, hubert_state_dict=hubert_weights()).eval()
    model.parallel_branch.load_state_dict(head_weights(), strict=True)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(WS_WEIGHTS)
    return model


BATCH = 40          # utterances per encoder batch, length-sorted.  PART OF THE PROTOCOL: conv layer 0's GroupNorm takes its statistics
#                     over the PADDED length of the batch (fairseq semantics), so an embedding depends on the batch's longest utterance.
#                     Round 2 embedded the oracle in batches of 40 and the HIP model in batches of 125 and read the difference
#                     (embedding cosine 0.99973 instead of 0.99998, -0.56 points of recall@1) as bf16 noise; it was the protocol.


def embed_all(encode, wavs, dim: int, batch: int = BATCH) -> torch.Tensor:
    """``encode(list of waveforms) -> [n, dim]`` over length-sorted batches of ``batch`` (the same composition for every implementation)."""
    order = sorted(range(len(wavs)), key=lambda i: len(wavs[i]))
    emb = torch.zeros(len(wavs), dim)
    for s0 in range(0, len(order), batch):
        sel = order[s0: s0 + batch]
        emb[sel] = encode([wavs[i] for i in sel]).float().cpu()
    return emb


def hip_embeddings(model, n_ids: int, batch: int = BATCH) -> torch.Tensor:
    """the product model's [5 n_ids, E] embeddings of the eval set (length-sorted batches of ``batch``: part of the protocol)"""
    dev = next(model.parameters()).device
    wavs, _ = eval_set(n_ids)
    with torch.no_grad():
        return embed_all(lambda ws: model.encode_speech([w.to(dev) for w in ws])["parallel_audio_feat"], wavs,
                         model.parallel_branch.linear_proj.out_features if hasattr(model.parallel_branch, "linear_proj") else 512, batch)


def natural_margin_report(emb: torch.Tensor, fixture: dict) -> dict:
    """The HIP embeddings on the NATURAL-margin gallery (tests/golden/recall_eval_natural.npz: class centre + isotropic noise, nothing
    planted, fp32 recall@1 = 50 %): recalls, rank flips against both references and - per boundary k and direction - the largest fp32
    margin among the decisions the HIP model takes differently from the fp32 oracle, in units of the bf16-emulation's margin noise."""
    import numpy as np
    T = lambda k: torch.from_numpy(np.asarray(fixture[k]))
    n_ids = int(fixture["n_ids"])
    image = T("image")
    ids = torch.arange(n_ids).repeat_interleave(PER_ID)
    held = (torch.arange(len(ids)) % PER_ID) >= GALLERY
    st = rank_stats(F.normalize(emb, dim=-1), image, ids)
    sig_ai, sig_ia = float(fixture["sigma_bf16emu"]), float(fixture["sigma_ia_bf16emu"])
    out = {"queries": len(ids), "images": n_ids, "recall_at": [1, 5, 10], "noise_level": float(fixture["noise_level"]),
           "hip": {"audio_to_image": recalls(st["rank_ai"]), "audio_to_image_heldout": recalls(st["rank_ai"], held),
                   "image_to_audio": recalls(st["rank_ia"])},
           "emulation_margin_noise_sigma": {"audio_to_image": sig_ai, "image_to_audio": sig_ia}}
    for ref in ("fp32", "bf16emu"):
        r_ai, r_ia = T(f"rank_ai_{ref}").long(), T(f"rank_ia_{ref}").long()
        out["oracle_" + ref] = {"audio_to_image": recalls(r_ai), "audio_to_image_heldout": recalls(r_ai, held), "image_to_audio": recalls(r_ia),
                                "rank_flips_audio_to_image": [int(((st["rank_ai"] < k) != (r_ai < k)).sum()) for k in (1, 5, 10)],
                                "rank_flips_image_to_audio": [int(((st["rank_ia"] < k) != (r_ia < k)).sum()) for k in (1, 5, 10)]}
    out["emulation_flips_vs_fp32"] = {
        "audio_to_image": [int(((T("rank_ai_bf16emu").long() < k) != (T("rank_ai_fp32").long() < k)).sum()) for k in (1, 5, 10)],
        "image_to_audio": [int(((T("rank_ia_bf16emu").long() < k) != (T("rank_ia_fp32").long() < k)).sum()) for k in (1, 5, 10)]}
    worst = {}
    for name, rank, ref, margin, sig in (("audio_to_image", st["rank_ai"], T("rank_ai_fp32").long(), T("margin_ai_fp32"), sig_ai),
                                         ("image_to_audio", st["rank_ia"], T("rank_ia_fp32").long(), T("margin_ia_fp32"), sig_ia)):
        w = []
        for i, k in enumerate((1, 5, 10)):
            flip = (rank < k) != (ref < k)
            w.append(round(float(margin[:, i].abs()[flip].max() / sig), 3) if flip.any() else 0.0)
        worst[name] = w
    out["largest_fp32_margin_of_a_flip_in_sigma"] = worst
    return out


def hip_recall(model, fixture: dict, dump: str = None, emb: torch.Tensor = None) -> dict:
    """Embeds the 5000 utterances with ``model`` (the product, on the GPU) and compares recall@{1,5,10}, both directions, all
    queries and the 2000 held-out ones, with BOTH references held by the fixture: the fp32 oracle and the bf16-storage-emulated
    oracle (oracle.bf16_store / bf16_weights: the control that separates storage precision from kernel defects)."""
    import numpy as np
    from speechclip_plus_amd import mutualRetrieval
    n_ids = int(fixture["n_ids"])
    image = torch.from_numpy(np.asarray(fixture["image"]))
    dev = next(model.parameters()).device
    ids = torch.arange(n_ids).repeat_interleave(PER_ID)
    if emb is None:
        emb = hip_embeddings(model, n_ids, int(fixture["batch"]))
    if dump:
        np.save(dump, emb.numpy())
    a = F.normalize(emb, dim=-1)
    st = rank_stats(a, image, ids)
    held = (torch.arange(len(ids)) % PER_ID) >= GALLERY
    # the product's own metric code on the same scores (must agree with the rank counts)
    score = st["scores"].to(dev)
    AB, BA, _ = mutualRetrieval(score, score.t(), ids.to(dev), torch.arange(n_ids, device=dev), [1, 5, 10])
    assert [round(AB[f"recall@{k}"], 2) for k in (1, 5, 10)] == recalls(st["rank_ai"]), (AB, recalls(st["rank_ai"]))
    assert [round(BA[f"recall@{k}"], 2) for k in (1, 5, 10)] == recalls(st["rank_ia"]), (BA, recalls(st["rank_ia"]))
    T = lambda k: torch.from_numpy(np.asarray(fixture[k]))
    out = {"queries": len(ids), "images": n_ids, "recall_at": [1, 5, 10], "batch": int(fixture["batch"]),
           "hip": {"audio_to_image": recalls(st["rank_ai"]), "audio_to_image_heldout": recalls(st["rank_ai"], held),
                   "image_to_audio": recalls(st["rank_ia"])}}
    for ref in ("fp32", "bf16emu"):
        r_ai, r_ia = T(f"rank_ai_{ref}").long(), T(f"rank_ia_{ref}").long()
        flips_ai = [int(((st["rank_ai"] < k) != (r_ai < k)).sum()) for k in (1, 5, 10)]
        flips_ia = [int(((st["rank_ia"] < k) != (r_ia < k)).sum()) for k in (1, 5, 10)]
        e_ref = F.normalize(T(f"emb_head_{ref}"), dim=-1)
        n_head = e_ref.shape[0]
        dist = (a[:n_head] - e_ref).norm(dim=-1)
        out["oracle_" + ref] = {"audio_to_image": recalls(r_ai), "audio_to_image_heldout": recalls(r_ai, held),
                                "image_to_audio": recalls(r_ia), "rank_flips_audio_to_image": flips_ai,
                                "rank_flips_image_to_audio": flips_ia,
                                "embedding_distance_mean": round(float(dist.mean()), 5),
                                "embedding_cosine_min": round(float((a[:n_head] * e_ref).sum(-1).min()), 6)}
    # margin noise against the fp32 oracle: this implementation's (own score - score of the oracle's k-th best other image) minus
    # the oracle's, per query; near-ties = held-out queries whose oracle margin lies inside 3 sigma of that noise
    kth = T("kth_idx_fp32").long()
    m_o = T("margin_ai_fp32")
    m_h = st["own"].unsqueeze(1) - st["scores"].gather(1, kth)
    noise = (m_h - m_o)
    sigma = float(noise.std())
    near = (m_o[held].abs() < 3 * sigma).float().mean(0)
    out["margin_noise_vs_fp32"] = {"sigma": round(sigma, 6), "max": round(float(noise.abs().max()), 6),
                                   "heldout_fraction_within_3_sigma_at_1_5_10": [round(float(v), 5) for v in near],
                                   "smallest_heldout_oracle_margin": round(float(m_o[held].abs().min()), 6)}
    e_f, e_e = F.normalize(T("emb_head_fp32"), dim=-1), F.normalize(T("emb_head_bf16emu"), dim=-1)
    out["emulation_distance_from_fp32"] = round(float((e_e - e_f).norm(dim=-1).mean()), 5)
    return out
