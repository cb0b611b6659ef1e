"""Classification / ordinal / uncertainty heads on the HIP path.  Mirrors /root/reference/models/heads.py
(:7-22, :25-77, :80-112): same constructors, sub-module and parameter names."""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from rovit_hip.functions import MLPHeadFn

LIN_CLAMP10 = 2


def dropout_mask(drop: nn.Dropout, training: bool, shape, device) -> Optional[torch.Tensor]:
    """Scaled keep-mask for the fused Linear+ReLU+Dropout kernel (None in eval mode / p == 0)."""
    if not training or drop.p <= 0.0:
        return None
    keep = 1.0 - drop.p
    return torch.empty(shape, device=device, dtype=torch.float32).bernoulli_(keep).mul_(1.0 / keep)      # two launches


class _Head(nn.Module):
    def _hidden_mask(self, x):
        return dropout_mask(self.dropout, self.training, (x.shape[0], self.fc1.out_features), x.device)


class ClassificationHead(_Head):
    def __init__(self, embed_dim: int = 384, hidden_dim: int = 128, num_classes: int = 4, dropout: float = 0.3):
        super().__init__()
        self.fc1 = nn.Linear(embed_dim, hidden_dim)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(dropout)
        self.fc2 = nn.Linear(hidden_dim, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return MLPHeadFn.apply(x, self._hidden_mask(x), (0,), self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)[0]


class OrdinalHead(_Head):
    def __init__(self, embed_dim: int = 384, hidden_dim: int = 128, num_classes: int = 4, dropout: float = 0.3):
        super().__init__()
        self.num_classes = num_classes
        self.num_thresholds = num_classes - 1
        self.fc1 = nn.Linear(embed_dim, hidden_dim)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(dropout)
        self.fc2 = nn.Linear(hidden_dim, self.num_thresholds)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return MLPHeadFn.apply(x, self._hidden_mask(x), (0,), self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)[0]

    @staticmethod
    def probabilities_from_logits(cum_logits: torch.Tensor) -> torch.Tensor:
        cp = torch.sigmoid(cum_logits)
        return torch.cat([cp[:, :1], cp[:, 1:] - cp[:, :-1], 1.0 - cp[:, -1:]], dim=1)

    def predict_probabilities(self, x: torch.Tensor) -> torch.Tensor:
        return self.probabilities_from_logits(self.forward(x))

    def predict_severity(self, x: torch.Tensor) -> torch.Tensor:
        probs = self.predict_probabilities(x)
        levels = torch.arange(self.num_classes, dtype=torch.float32, device=probs.device)
        return (probs * levels).sum(dim=1, keepdim=True)


class UncertaintyHead(_Head):
    def __init__(self, embed_dim: int = 384, hidden_dim: int = 128, dropout: float = 0.3):
        super().__init__()
        self.fc1 = nn.Linear(embed_dim, hidden_dim)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(dropout)
        self.fc_mu = nn.Linear(hidden_dim, 1)
        self.fc_logvar = nn.Linear(hidden_dim, 1)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        mu, log_var = MLPHeadFn.apply(x, self._hidden_mask(x), (0, LIN_CLAMP10), self.fc1.weight, self.fc1.bias,
                                      self.fc_mu.weight, self.fc_mu.bias, self.fc_logvar.weight, self.fc_logvar.bias)
        return mu, log_var

    def sample(self, x: torch.Tensor, num_samples: int = 100) -> torch.Tensor:
        mu, log_var = self.forward(x)
        eps = torch.randn(x.size(0), num_samples, device=x.device)
        return mu + torch.exp(0.5 * log_var) * eps
