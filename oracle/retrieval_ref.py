"""Restatement of mutualRetrieval (recall@k both directions).

ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows avssl/module/retrieval.py:6-121; score construction and image de-duplication follow
avssl/model/kwClip.py:447-482.  Ties are broken by torch.argsort(descending=True) order, as in
the reference.
"""
from typing import Dict, List, Tuple

import torch


def dedupe_images(all_ids: torch.Tensor, all_imgs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """kwClip.py:448-458: dict keyed by id (last occurrence wins, first-seen order kept)."""
    pairs = {}
    for _id, _img in zip(all_ids.tolist(), all_imgs):
        pairs[_id] = _img
    return torch.stack(list(pairs.values()), dim=0), torch.tensor(list(pairs.keys()), dtype=torch.long)


def mutual_retrieval(score_per_A: torch.Tensor, score_per_B: torch.Tensor, AB_answers: torch.Tensor,
                     BA_answers: torch.Tensor, recall_at: List[int]) -> Tuple[Dict, Dict, Dict]:
    assert score_per_A.shape == (len(AB_answers), len(BA_answers))
    assert score_per_B.shape == (len(BA_answers), len(AB_answers))
    order_A = torch.argsort(score_per_A, dim=1, descending=True)
    order_B = torch.argsort(score_per_B, dim=1, descending=True)
    rank_AB = BA_answers[order_A] == AB_answers.unsqueeze(-1)      # (nA, nB) hit matrix in rank order
    rank_BA = AB_answers[order_B] == BA_answers.unsqueeze(-1)
    res_AB, res_BA, res_mean = {}, {}, {}
    for k in recall_at:
        key = "recall@{}".format(k)
        kk = min(k, rank_AB.shape[1])
        res_AB[key] = (rank_AB[:, :kk].any(dim=1).sum() / rank_AB.shape[0]).item() * 100
        kk = min(k, rank_BA.shape[1])
        res_BA[key] = (rank_BA[:, :kk].any(dim=1).sum() / rank_BA.shape[0]).item() * 100
        res_mean[key] = (res_AB[key] + res_BA[key]) / 2.0
    return res_AB, res_BA, res_mean
