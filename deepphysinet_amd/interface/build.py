"""name -> interface-class factory with the reference's surface (interface/build.py:10-20)."""
from .interface_physics import InterfacePhysics

interface_dict = {'InterfacePhysics': InterfacePhysics}


def builder_models(name='InterfaceDownScale', **kwargs):
    if name in interface_dict:
        return interface_dict[name](**kwargs)
    raise NotImplementedError('{0} not in availables values.'.format(name))
