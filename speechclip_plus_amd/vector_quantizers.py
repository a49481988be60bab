"""SimpleVectorQuantizer (avssl/module/speechclip_c_modules/my_vector_quantizer.py:12-165) and the keyword
BatchNorm (kw_bn.py:167-228) of the cascaded(+)/hybrid(+) branches.  Scope row a11, stock device-side torch ops."""
import ast
import logging

import torch
import torch.nn as nn
import torch.nn.functional as F

logger = logging.getLogger(__name__)

__all__ = ["SimpleVectorQuantizer", "Kw_BatchNorm_dynamic"]


class SimpleVectorQuantizer(nn.Module):
    """Straight-through one-hot selection of a CLIP sub-word per keyword from its cosine scores."""

    def __init__(self, temp, groundTruthPerplexity=None, time_first=True, use_gumbel=False, hard=True):
        super().__init__()
        self.time_first, self.use_gumbel, self.hard = time_first, use_gumbel, hard
        if isinstance(temp, str):
            if temp.startswith("learnable="):
                self.temp_type = "learnable"
                self.curr_temp = nn.parameter.Parameter(torch.FloatTensor([ast.literal_eval(temp[len("learnable="):])]))
            elif temp.startswith("fixed="):
                self.temp_type = "fixed"
                self.register_buffer("curr_temp", torch.FloatTensor([ast.literal_eval(temp[len("fixed="):])]))
                self._fixed_temp = float(ast.literal_eval(temp[len("fixed="):]))     # host copy: reporting it needs no device read
            else:
                self.temp_type = "scheduled"
                sched = ast.literal_eval(temp)
                assert len(sched) == 3, f"{sched}, {len(sched)}"
                self.max_temp, self.min_temp, self.temp_decay = sched
                self.curr_temp = self.max_temp
        self.codebook_indices = None
        self.groundTruthPerplexity = groundTruthPerplexity
        if groundTruthPerplexity is not None:
            self.perplexity_criteria = nn.MSELoss()

    def set_num_updates(self, num_updates):
        if self.temp_type == "scheduled":
            self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)

    def forward(self, x, prob_msk=[0, 2, 3], produce_targets=True):
        if not self.time_first:
            x = x.transpose(1, 2)
        result = {"num_vars": x.shape[-1]}
        bsz, tsz, fsz = x.shape
        x = x.reshape(bsz * tsz, fsz)
        for i in prob_msk:                         # special tokens can never be selected (in place, like the reference)
            x[:, i] += float("-inf")
        k = x.argmax(-1)
        hard_x = x.new_zeros(*x.shape).scatter_(-1, k.view(-1, 1), 1.0)
        if bsz * tsz == 1:
            hard_x = hard_x.squeeze()              # my_vector_quantizer.py:90 squeezes unconditionally
        hard_probs = torch.mean(hard_x.float(), dim=0)
        result["code_perplexity"] = torch.exp(-torch.sum(hard_probs * torch.log(hard_probs + 1e-7), dim=-1)).sum()
        avg_probs = torch.softmax(x.view(bsz * tsz, 1, -1).float(), dim=-1).mean(dim=0)
        probs_per_t = torch.softmax(x.view(bsz, tsz, -1), dim=-1).permute(1, 0, 2)
        result["ent_per_t"] = (-torch.sum(probs_per_t * torch.log(probs_per_t + 1e-9), dim=-1)).mean(dim=-1)
        result["prob_perplexity"] = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-7), dim=-1)).sum()
        if self.temp_type == "fixed":
            result["temp"] = self._fixed_temp
        else:
            result["temp"] = self.curr_temp.item() if isinstance(self.curr_temp, torch.Tensor) else float(self.curr_temp)
        if self.training:
            if self.use_gumbel:
                x = F.gumbel_softmax(x.float(), tau=self.curr_temp, hard=self.hard).type_as(x)
            else:
                x = F.softmax(x / self.curr_temp, dim=-1).type_as(x)
                if self.hard:
                    x = hard_x + x - x.detach()
        else:
            x = hard_x
        x = x.view(bsz * tsz, -1)
        result["subword_prob"] = x.view(bsz, tsz, -1)
        if self.groundTruthPerplexity is not None:
            result["diversity_loss"] = self.perplexity_criteria(
                result["prob_perplexity"], torch.tensor(self.groundTruthPerplexity).type_as(x)
            ) / (result["num_vars"] - self.groundTruthPerplexity) ** 2
        else:
            result["diversity_loss"] = (result["num_vars"] - result["prob_perplexity"]) / result["num_vars"]
        if produce_targets:
            result["targets"] = x.argmax(dim=-1).view(bsz, tsz, 1).detach()
        return result


class Kw_BatchNorm_dynamic(nn.Module):
    """BatchNorm1d over keyword embeddings, initialised from the CLIP token-embedding mean / std."""

    def __init__(self, kw_dim: int, init_bias: torch.Tensor, init_scale: torch.Tensor, std_scale: int = 1,
                 learnable: bool = True) -> None:
        super().__init__()
        assert std_scale > 0, f"std scale must > 0, but input std scale is {std_scale}"
        self.kw_dim, self.learnable, self.std_scale = kw_dim, learnable, std_scale
        self.bn_layer = nn.BatchNorm1d(kw_dim)
        self.bn_layer.weight.data.copy_(init_scale * std_scale)
        self.bn_layer.bias.data.copy_(init_bias)
        self.bn_layer.weight.requires_grad = learnable
        self.bn_layer.bias.requires_grad = learnable

    def forward(self, keywords: torch.Tensor) -> torch.Tensor:
        assert keywords.dim() == 3
        return self.bn_layer(keywords.permute(0, 2, 1)).permute(0, 2, 1)
