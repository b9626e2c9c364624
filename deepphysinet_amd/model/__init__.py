from .meta_net import MetaNet, TransformerNet
from .physics_net import PhysicsNet
from .variable_net import ResMLP, VariableNet

__all__ = ['MetaNet', 'TransformerNet', 'PhysicsNet', 'VariableNet', 'ResMLP']
