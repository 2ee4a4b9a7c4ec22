"""Loss of the training step on the HIP path.

``CrossEntropyLoss`` is a drop-in for the ``nn.CrossEntropyLoss()`` the reference's loop builds
(spectre_vit/repl/train.py:196, used at :226 and :258): class-index targets, mean over the batch.  On GPU fp32 logits it is
one launch forward and one backward (csrc/spv_head.hip); the options the reference never sets are refused, not emulated.
"""
import torch
import torch.nn as nn

from . import hip_ops


class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean", label_smoothing=0.0):
        super().__init__()
        if weight is not None or reduction != "mean" or label_smoothing != 0.0 or size_average is not None or reduce is not None:
            raise NotImplementedError("spectre_vit.loss.CrossEntropyLoss implements nn.CrossEntropyLoss() with its defaults only")
        # ignore_index: targets outside [0, classes) -- including torch's -100 -- are NOT skipped: they make the loss NaN
        self.ignore_index = ignore_index

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if not input.is_cuda:
            raise RuntimeError("spectre_vit.loss.CrossEntropyLoss runs on the GPU (libspv_hip.so); there is no CPU path")
        return hip_ops.cross_entropy(input.float() if input.dtype != torch.float32 else input, target)
