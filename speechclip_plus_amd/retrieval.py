"""Recall@k of the validation epoch - same signature and result dictionaries as the reference's ``mutualRetrieval``
(avssl/module/retrieval.py:6-121; scores built at avssl/model/kwClip.py:447-482).

The reference sorts every score row and looks the answers up in rank order.  Here no row is sorted: a query hits at k iff fewer
than k candidates score strictly above its best-scoring correct candidate, so one masked row maximum and one comparison count per
direction give every k at once (on the device the scores live on).

Degenerate scores: the reference leaves exact float ties between a correct and a wrong candidate to the order of an
unstable ``argsort`` - collapsed embeddings (all scores equal) then score about chance - and sorts NaN first.  Here a tie counts
AGAINST the query (a wrong candidate with the same score as the best correct one is ahead of it), a NaN candidate is ahead of
everything (where the reference's sort puts it) and a query whose best correct score is not finite is a miss at every k: a
diverged or collapsed model reports recall near 0, never 100, so a "keep the best validation recall" monitor cannot latch onto it."""
from typing import Dict, Sequence, Tuple

import torch


def _recall(score: torch.Tensor, query_ids: torch.Tensor, cand_ids: torch.Tensor, ks: Sequence[int], cand_title: str) -> Dict[str, float]:
    """score [Q, C]; a candidate c is correct for query q iff cand_ids[c] == query_ids[q] (an image with several captions has
    several correct candidates; one of them inside the top k is a hit)."""
    correct = cand_ids.unsqueeze(0) == query_ids.unsqueeze(1)
    nan = torch.isnan(score)
    best = score.masked_fill(~correct | nan, float("-inf")).amax(dim=1, keepdim=True)
    # wrong candidates ahead of the best correct one: higher score, equal score (ties count against the query), or NaN
    ahead = (score > best).sum(dim=1) + (((score == best) | nan) & ~correct).sum(dim=1)
    # a query without a correct candidate, or whose best correct score is not finite (-inf: none / all NaN; +inf: overflow): never a hit
    ok = correct.any(dim=1) & torch.isfinite(best.squeeze(1))
    ahead = torch.where(ok, ahead, torch.full_like(ahead, score.shape[1]))
    out = {}
    for k in ks:
        if k > score.shape[1]:
            print(f"recall@{k} asks for more than the {score.shape[1]} {cand_title} candidates there are; counted over all of them")
        out[f"recall@{k}"] = 100.0 * (ahead < k).float().mean().item()
    return out


def mutualRetrieval(score_per_A: torch.Tensor, score_per_B: torch.Tensor, AB_answers: torch.Tensor,
                    BA_answers: torch.Tensor, recall_at: list, modality_A_title: str = "audio",
                    modality_B_title: str = "image") -> Tuple[dict, dict, dict]:
    nA, nB = AB_answers.numel(), BA_answers.numel()
    if AB_answers.dim() != 1 or BA_answers.dim() != 1:
        raise AssertionError("answers must be 1-d id vectors")
    if tuple(score_per_A.shape) != (nA, nB) or tuple(score_per_B.shape) != (nB, nA):
        raise AssertionError(f"score shapes {tuple(score_per_A.shape)} / {tuple(score_per_B.shape)} do not match {nA} x {nB} answers")
    a_ids = AB_answers.to(score_per_A.device)
    b_ids = BA_answers.to(score_per_A.device)
    results_AB = _recall(score_per_A, a_ids, b_ids, recall_at, modality_B_title)
    results_BA = _recall(score_per_B.to(score_per_A.device), b_ids, a_ids, recall_at, modality_A_title)
    results_mean = {key: 0.5 * (results_AB[key] + results_BA[key]) for key in results_AB}
    return results_AB, results_BA, results_mean
