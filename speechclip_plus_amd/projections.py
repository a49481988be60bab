"""MLPLayers (avssl/module/projections.py:6-29): Linear / ReLU / Dropout stack without the trailing ReLU+Dropout."""
from torch import nn

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
        return self.sequential(X)
