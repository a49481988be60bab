"""GPU: recall@k parity on the Flickr8k-test-shaped synthetic eval set of SURVEY 8d (1000 image ids x 5 utterances = 5000 audio
queries), round-3 form.  North star: recall@1 within +-0.1 of the reference, recall@k identical.

The fixture (tests/golden/make_recall_fixture.py, run in the build container on the CPU oracle) holds TWO references:
  * fp32      the oracle as it restates the reference's arithmetic;
  * bf16emu   the same oracle with bf16 rounding at exactly the tensors the HIP path stores in bf16 and on the GEMM weights it
              holds in bf16 (oracle/hubert_ref.py: bf16_store, bf16_weights) - what ANY implementation with this storage format
              computes, up to summation order.
and an image gallery whose confusions are planted at discrete score levels (tools/recall_eval.build_gallery), so that the eval
set's own margins resolve +-0.1: tests/golden/recall_eval_margins.json lists the oracle's margin histogram; 0.1 % of the held-out
queries lie within 3 sigma of the bf16 margin noise at the rank-1 boundary, none at the rank-5 / rank-10 boundaries.

What round 2 measured as "-0.56 points, bf16 noise" was neither: the oracle had been embedded in length-sorted batches of 40 and
the HIP model in batches of 125, and conv layer 0's GroupNorm takes its statistics over the PADDED batch length (fairseq
semantics, reproduced by both).  With the batch composition fixed (recall_eval.BATCH, part of the protocol now) the HIP embeddings
sit at the same distance from the fp32 oracle as the emulation does (0.0056, of which 0.0050 is ONE common shift caused by the
bf16 weights) and 0.0028 from the emulation itself with no common component - storage precision, no kernel bias.

The test requires, against BOTH references: recall@1 within 0.1 (all queries, the 2000 held-out ones, and image -> audio), recall@5
and recall@10 identical; and as evidence that the remaining difference is storage precision: the HIP embeddings are closer to the
emulated oracle than the emulated oracle is to fp32, and the measured HIP-vs-fp32 margin noise puts < 0.2 % of the held-out
queries inside 3 sigma."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def hip_emb():
    """the product model's embeddings of the 5000 utterances (one pass, shared by the two galleries)"""
    import recall_eval
    model = recall_eval.build_model()
    return model, recall_eval.hip_embeddings(model, 1000, recall_eval.BATCH)


def _same_rate(n_a: int, n_b: int, z: float = 3.0) -> bool:
    """Two counts of rare events as two draws from Poisson distributions with ONE rate (the conditional test: given n_a + n_b = N,
    n_a ~ Binomial(N, 1/2), standard deviation sqrt(N) / 2, so |n_a - n_b| = 2 |n_a - N / 2| <= z sqrt(N))."""
    return abs(n_a - n_b) <= z * (n_a + n_b) ** 0.5


@pytest.mark.parametrize("gallery", ["recall_eval_natural.npz", "recall_eval_natural_b.npz"])
def test_recall_on_natural_margins(golden, hip_emb, gallery):
    """Round 4: the same utterances against a gallery with NATURAL margins - class centre +
    isotropic noise, nothing planted, fp32 recall@1 = 50.02 %, 14.8 % of the rank-1 decisions within 3 sigma of the margin noise that
    bf16 STORAGE alone causes (the bf16-emulated oracle flips 79 / 66 / 61 of the 5000 rank-1 / 5 / 10 decisions against fp32).  On such
    a set no bf16 implementation can match recall@1 to 0.1.

    Acceptance criterion, PRE-REGISTERED in round 5.  History, so that nobody has to dig for it: round 4's first
    bound on the HIP-vs-fp32 flip counts, max(1.25 emu, emu + 3), failed on the GPU at image -> audio @5 (16 flips, emulation 12) and
    was replaced two minutes later by 1.25 emu + 2 sqrt(emu), a bound fitted to that one observation.  Both are gone.  The model:
      * a rank decision flips when the implementation's margin noise exceeds the fp32 margin; over thousands of decisions with
        independent margins the number of flips of ONE bf16-storage implementation against fp32 is Poisson with a rate lambda that
        depends on the storage format (where values are rounded), not on the summation order;
      * the bf16-storage-emulated oracle and the HIP model are two such implementations: their flip counts against fp32, n_emu and
        n_hip, are two draws with one rate.  Whether two Poisson counts share a rate is the conditional binomial test:
        |n_hip - n_emu| <= z sqrt(n_hip + n_emu); z = 3 (two-sided p = 0.0027 per comparison, 12 comparisons per gallery);
      * a flip at a RESOLVABLE margin is a defect whatever the counts say: every decision the HIP model takes differently from fp32
        must sit at an fp32 margin below 4 sigma of the emulation's margin noise (the emulation's own flips reach 2.3 sigma);
      * against the EMULATED oracle the HIP model differs by summation order only - fewer flips than the emulation has against
        fp32: n_hip_vs_emu <= n_emu (+ the same z sqrt allowance);
      * the recalls themselves stay within the band the flip counts allow.
    Gallery A (recall_eval_natural.npz) is the one round 4 looked at: (16, 12) passes the test with |4| <= 15.9, which is no
    evidence by itself.  Gallery B (recall_eval_natural_b.npz: the same construction, noise seed 20261005) was generated and this
    criterion committed BEFORE the HIP model was ever scored on it (git history: the fixture and this text precede the first GPU
    run).  A future red run is answered in the code or in a written derivation committed before any new bound - not by editing z."""
    import recall_eval
    model, emb = hip_emb
    fx = golden(gallery)
    r = recall_eval.natural_margin_report(emb, fx)
    print("natural-margin recall", gallery, r)
    assert r["queries"] == 5000 and r["images"] == 1000
    n = {"audio_to_image": 5000, "image_to_audio": 1000}
    for d in ("audio_to_image", "image_to_audio"):
        assert max(r["largest_fp32_margin_of_a_flip_in_sigma"][d]) < 4.0, r["largest_fp32_margin_of_a_flip_in_sigma"]
        for k in range(3):
            emu = r["emulation_flips_vs_fp32"][d][k]
            vs_fp32 = r["oracle_fp32"]["rank_flips_" + d][k]
            vs_emu = r["oracle_bf16emu"]["rank_flips_" + d][k]
            assert _same_rate(vs_fp32, emu), (gallery, d, k, vs_fp32, emu)
            assert vs_emu <= emu + 3.0 * (vs_emu + emu) ** 0.5, (gallery, d, k, vs_emu, emu)
            band = 100.0 * vs_fp32 / n[d]
            assert abs(r["hip"][d][k] - r["oracle_fp32"][d][k]) <= band + 1e-9


def test_recall_at_k_matches_the_oracle_on_5000_utterances(golden, hip_emb):
    import recall_eval
    fx = golden("recall_eval.npz")
    r = recall_eval.hip_recall(hip_emb[0], fx, emb=hip_emb[1])
    print("recall parity:", r)
    assert r["queries"] == 5000 and r["images"] == 1000 and r["batch"] == recall_eval.BATCH
    hip = r["hip"]
    for ref in ("oracle_fp32", "oracle_bf16emu"):
        o = r[ref]
        for key in ("audio_to_image", "audio_to_image_heldout", "image_to_audio"):
            assert abs(hip[key][0] - o[key][0]) <= 0.1, (ref, key, hip[key], o[key])          # recall@1: north star's +-0.1
            assert hip[key][1:] == o[key][1:], (ref, key, hip[key], o[key])                   # recall@5, @10: identical
        assert o["embedding_cosine_min"] > 0.9999, (ref, o)
    # precision, not defect: no further from the emulated oracle than the emulation is from fp32 ...
    assert r["oracle_bf16emu"]["embedding_distance_mean"] <= r["emulation_distance_from_fp32"], r
    # ... and as far from fp32 as the emulation itself (same storage format), within 25 %
    assert r["oracle_fp32"]["embedding_distance_mean"] <= 1.25 * r["emulation_distance_from_fp32"], r
    # the eval set resolves the target: < 0.2 % of the held-out queries within 3 sigma of the measured margin noise
    assert max(r["margin_noise_vs_fp32"]["heldout_fraction_within_3_sigma_at_1_5_10"]) < 0.002, r["margin_noise_vs_fp32"]
