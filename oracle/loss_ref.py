"""fp32 torch-CPU restatement of MaskedContrastiveLoss.

ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows avssl/module/losses.py:185-245 (ctor :130-168).  The reference indexes pre-built
``eye`` buffers of size MAX_EYE = 256 (losses.py:126,165), so it cannot run B > 256;
the restatement builds the masks at the requested size, which is the same arithmetic for B <= 256.
"""
from typing import Optional

import torch


def masked_contrastive_loss(feat_A: torch.Tensor, feat_B: torch.Tensor, index: Optional[torch.Tensor] = None,
                            inv_temperature=1.0 / 0.07, margin: float = 0.0, dcl: bool = False,
                            a2b: bool = True, b2a: bool = True) -> torch.Tensor:
    """``inv_temperature`` is what the reference calls ``temperature`` after its ctor:
    1/0.07 (fixed) or exp(log-parameter) (trainable); may be a tensor requiring grad."""
    assert feat_A.shape == feat_B.shape, (feat_A.shape, feat_B.shape)
    assert a2b or b2a
    B = feat_A.shape[0]
    with torch.no_grad():
        eye = torch.eye(B, dtype=torch.bool)
        if index is not None:
            assert index.shape[0] == B
            idx = index.unsqueeze(1)
            neg_mask = idx != idx.t()
        else:
            neg_mask = ~eye
        if not dcl:
            neg_mask = neg_mask | eye
        neg_mask_fl = neg_mask.type(feat_A.dtype)
    logits = feat_A @ feat_B.t() * inv_temperature
    if margin > 0.0:
        logits = logits - margin * eye.type(logits.dtype)
    pos_logits = logits[eye]
    exp_logits = logits.exp() * neg_mask_fl
    loss = 0
    if a2b:
        loss = loss + (-pos_logits + torch.log(exp_logits.sum(1))).mean()
    if b2a:
        loss = loss + (-pos_logits + torch.log(exp_logits.sum(0))).mean()
    if a2b and b2a:
        loss = loss / 2
    return loss
