from .build import builder_models, interface_dict
from .interface_physics import InterfacePhysics

__all__ = ['builder_models', 'interface_dict', 'InterfacePhysics']
