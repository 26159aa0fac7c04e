"""Loss module of the trainer surface (reference: MarT/lit_models/utils.py:30-66), HIP-backed."""
import torch.nn as nn

from .. import functional as Fn


class LabelSmoothSoftmaxCEV1(nn.Module):
    """Label-smoothed CE: target eps/C everywhere, 1-eps at the label; reduction 'mean' (the only mode MarT uses)."""

    def __init__(self, lb_smooth=0.1, reduction="mean", ignore_index=-100):
        super().__init__()
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' (lit_models/transformer.py:22-23) is implemented on the HIP path")
        self.lb_smooth = lb_smooth
        self.reduction = reduction
        self.lb_ignore = ignore_index

    def forward(self, logits, label):
        return Fn.label_smooth_ce(logits, label, self.lb_smooth)
