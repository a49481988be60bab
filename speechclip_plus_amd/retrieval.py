"""mutualRetrieval (avssl/module/retrieval.py:6-121): recall@k in both directions from a score matrix.
Metric code, not a kernel: argsort + gathers on whatever device the scores live on."""
from typing import Tuple

import torch


def mutualRetrieval(score_per_A: torch.Tensor, score_per_B: torch.Tensor, AB_answers: torch.Tensor,
                    BA_answers: torch.Tensor, recall_at: list, modality_A_title: str = "audio",
                    modality_B_title: str = "image") -> Tuple[dict, dict, dict]:
    assert len(score_per_A.shape) == 2 and len(score_per_B.shape) == 2
    assert len(AB_answers.shape) == 1 and len(BA_answers.shape) == 1
    assert score_per_A.shape == (len(AB_answers), len(BA_answers)), "{} , {}".format(
        score_per_A.shape, (len(AB_answers), len(BA_answers)))
    assert score_per_B.shape == (len(BA_answers), len(AB_answers)), "{} , {}".format(
        score_per_B.shape, (len(BA_answers), len(AB_answers)))
    dev = score_per_A.device
    AB_answers, BA_answers = AB_answers.to(dev), BA_answers.to(dev)
    order_A = torch.argsort(score_per_A, dim=1, descending=True)
    order_B = torch.argsort(score_per_B, dim=1, descending=True)
    rank_AB = BA_answers[order_A] == AB_answers.unsqueeze(-1)
    rank_BA = AB_answers[order_B] == BA_answers.unsqueeze(-1)
    recall_results_AB, recall_results_BA, recall_results_mean = {}, {}, {}
    for k in recall_at:
        if k > rank_AB.shape[1]:
            print("recall@{} is not eligible for #{} {} samples".format(k, rank_AB.shape[1], modality_B_title))
        recall_results_AB["recall@{}".format(k)] = (
            rank_AB[:, : min(k, rank_AB.shape[1])].any(dim=1).sum() / rank_AB.shape[0]).item()
    for k in recall_at:
        if k > rank_BA.shape[1]:
            print("recall@{} is not eligible for #{} {} samples".format(k, rank_BA.shape[1], modality_A_title))
        recall_results_BA["recall@{}".format(k)] = (
            rank_BA[:, : min(k, rank_BA.shape[1])].any(dim=1).sum() / rank_BA.shape[0]).item()
    for _k in ["recall@{}".format(r) for r in recall_at]:
        recall_results_BA[_k] *= 100
        recall_results_AB[_k] *= 100
        recall_results_mean[_k] = (recall_results_BA[_k] + recall_results_AB[_k]) / 2.0
    return recall_results_AB, recall_results_BA, recall_results_mean
