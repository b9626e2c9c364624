"""name -> loss-class factory with the reference's surface (losses/builder.py:8-20)."""
import torch.nn as nn

from .weights_loss import WeightSmoothL1Loss

losses_dict = {
    'CrossEntropyLoss': nn.CrossEntropyLoss,
    'L1Loss': nn.L1Loss,
    'MSELoss': nn.MSELoss,
    'WeightSmoothL1Loss': WeightSmoothL1Loss,
}


def builder_loss(name='CrossEntropyLoss', **kwargs):
    if name in losses_dict:
        return losses_dict[name](**kwargs)
    raise NotImplementedError('{0} not in availables values.'.format(name))
