"""Loss module of the trainer surface (reference: MarT/lit_models/utils.py:30-66), HIP-backed."""
import torch.nn as nn

from .. import functional as Fn


class LabelSmoothSoftmaxCEV1(nn.Module):
    """Label-smoothed CE: target eps/C everywhere, 1-eps at the label.  As the reference: rows whose label == ``ignore_index`` contribute
    nothing and 'mean' divides by the number of the others (utils.py:49-52,58-60); reduction 'mean' / 'sum', anything else returns the
    per-row losses (utils.py:59-64)."""

    def __init__(self, lb_smooth=0.1, reduction="mean", ignore_index=-100):
        super().__init__()
        self.lb_smooth = lb_smooth
        self.reduction = reduction
        self.lb_ignore = ignore_index

    def forward(self, logits, label):
        return Fn.label_smooth_ce(logits, label, self.lb_smooth, ignore_index=self.lb_ignore, reduction=self.reduction)
