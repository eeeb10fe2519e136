"""adaptExtraDimensionLoss -- the reference's alegnn/modules/loss.py:23-91: wraps a torch loss so that regression losses
accept the B x 1 output of an architecture against a B target (the examples build every Model's loss through it,
sourceLocGNN.py:782, movieGNN.py:817)."""
import torch.nn as nn


class adaptExtraDimensionLoss(nn.modules.loss._Loss):
    def __init__(self, lossFunction, *args):
        super().__init__()
        self.loss = lossFunction(*args)                             # the loss class is instantiated here (loss.py:62-65)

    def forward(self, estimate, target):
        kind = repr(self.loss)
        if 'CrossEntropyLoss' in kind:
            assert len(estimate.shape) == 2                         # B x nClasses
        elif 'SmoothL1Loss' in kind or 'MSELoss' in kind or 'L1Loss' in kind:
            if len(estimate.shape) == 2:                            # B x 1 -> B
                assert estimate.shape[1] == 1
                estimate = estimate.squeeze(1)
            assert len(estimate.shape) == 1
        return self.loss(estimate, target)
