"""MLPLayers (avssl/module/projections.py:6-29): Linear / ReLU / Dropout stack without the trailing ReLU + Dropout (the keyword
projection 1024 -> 1024 -> 768 of the hybrid+ large recipe).  The Linear layers run in exact fp32 on the matrix pipe
(linear_fn.LinearF32Fn): what they produce is compared against the whole vocabulary by an argmax."""
from torch import nn

from .linear_fn import linear_f32_autograd

__all__ = ["MLPLayers"]


class MLPLayers(nn.Module):
    def __init__(self, units=[512, 512, 512], nonlin=nn.ReLU(), dropout=0.1):
        super().__init__()
        self.nonlin, self.dropout = nonlin, dropout
        seq = []
        for u0, u1 in zip(units[:-1], units[1:]):
            seq += [nn.Linear(u0, u1), self.nonlin, nn.Dropout(self.dropout)]
        self.sequential = nn.Sequential(*seq[:-2])

    def forward(self, X):
        for m in self.sequential:
            # exact forward; in train steps the two gradient products of a layer run on the bf16 GEMMs (linear_fn.LinearF32Fn)
            X = linear_f32_autograd(X, m.weight, m.bias, bf16_backward=self.training) if isinstance(m, nn.Linear) else m(X)
        return X
