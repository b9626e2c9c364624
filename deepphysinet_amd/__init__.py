"""deepphysinet_amd -- MI355X-native physics-informed training step for DeepPhysiNet-style models.

Drop-in surfaces (same names / arguments as the reference, /root/reference/DeepPhysiNet):
    deepphysinet_amd.model.PhysicsNet, VariableNet, MetaNet            (model/physics_net.py, variable_net.py, meta_net.py)
    deepphysinet_amd.utils.position_encoding.SineCosPE                 (utils/position_encoding.py)
    deepphysinet_amd.losses.builder_loss, WeightSmoothL1Loss           (losses/builder.py, weights_loss.py)
    deepphysinet_amd.interface.InterfacePhysics, builder_models        (interface/interface_physics.py, build.py)
    deepphysinet_amd.sampler.CollocationSampler                        (dataset/physics_dataset.py:323-587, on the device)
    deepphysinet_amd.optim.FusedClipAdam, distributed.GradientAllReduce (clip_grad_norm_ + Adam, DDP gradient averaging)
The per-point arithmetic runs in libdpn_hip.so (hand-written HIP for gfx950, C ABI in include/dpn_hip.h).
"""
from . import _lib
from .point_path import PointConfig, pde_losses, pde_losses_batch, point_fields, pde_fields_and_jacobian, smooth_l1_data_loss

__all__ = ['PointConfig', 'pde_losses', 'pde_losses_batch', 'point_fields', 'pde_fields_and_jacobian', 'smooth_l1_data_loss', '_lib']
