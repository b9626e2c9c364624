"""WeightSmoothL1Loss(beta): mean Smooth-L1, the data ("margin") loss (reference losses/weights_loss.py:12-20).

On HIP tensors of shape [N,6] the forward and backward run in libdpn_hip.so (dpn_smooth_l1); device tensors of other shapes use the
torch expression (the same formula); host tensors raise unless reference math is switched on (_lib.enable_cpu_reference_math).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class WeightSmoothL1Loss(nn.Module):
    def __init__(self, beta=0.1):
        super().__init__()
        self.beta = float(beta)

    def forward(self, input, target):
        if input.is_cuda and input.dim() == 2 and input.shape[1] == 6 and input.dtype == torch.float32:
            from ..point_path import smooth_l1_data_loss
            return smooth_l1_data_loss(input, target, beta=self.beta, factor=1.0)
        from .._lib import host_math_or_raise
        host_math_or_raise(input, 'WeightSmoothL1Loss')
        return F.smooth_l1_loss(input, target, beta=self.beta, reduction='none').mean()
