from .builder import builder_loss, losses_dict
from .weights_loss import WeightSmoothL1Loss

__all__ = ['builder_loss', 'losses_dict', 'WeightSmoothL1Loss']
