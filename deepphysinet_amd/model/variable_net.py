"""VariableNet / ResMLP with the reference's constructor, parameters and forward signature (model/variable_net.py:13-87).

The module only owns parameters and the tiny per-FIELD computations (hyper-network heads, lead-time embedding);
everything per-POINT is delegated to the HIP point path, which evaluates all six VariableNets of a PhysicsNet in
one launch.  Calling a single VariableNet directly is supported through the same kernels (the other five slots
reuse this net's weights and their outputs are discarded).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.position_encoding import SineCosPE


class ResMLP(nn.Module):
    """fc.0 -> ReLU -> fc.2, plus the input (variable_net.py:13-24).  Inside a VariableNet the arithmetic happens in the fused HIP point
    kernels (which read these parameters directly); called on its own it is the reference's per-row expression on the library's
    fp32 GEMMs."""

    def __init__(self, in_channels):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(in_channels, in_channels), nn.ReLU(inplace=True), nn.Linear(in_channels, in_channels))

    def forward(self, x):
        from ..linear import linear
        return linear(torch.relu(linear(x, self.fc[0].weight, self.fc[0].bias)), self.fc[2].weight, self.fc[2].bias) + x


class VariableNet(nn.Module):
    def __init__(self, token_num, in_channels, hidden_channels):
        super().__init__()
        if in_channels != 192 or hidden_channels != 256 or token_num != 256:
            raise NotImplementedError('the HIP point kernels are specialised for token_num=256, in_channels=192, '
                                      'hidden_channels=256 (configs/DeepPhysiNet_NCEP_cfg.py:25-32)')
        self.in_channels, self.hidden_channels, self.token_num = in_channels, hidden_channels, token_num
        self.coord_input_fc = nn.Linear(token_num, in_channels + 1)
        self.coord_hidden_fc = nn.Linear(token_num, hidden_channels + 1)
        self.data_input_fc = nn.Linear(in_channels, hidden_channels)
        self.fore_h_fc = nn.Linear(in_channels, hidden_channels)
        self.cat_fc1 = ResMLP(hidden_channels)
        self.out_fc = nn.Linear(hidden_channels, 1)
        self.pe = SineCosPE(6, N_freqs=in_channels // 2 // 6, include_input=False)
        self.pe_fore_h = SineCosPE(1, N_freqs=in_channels // 2, include_input=False)
        self.relu = nn.ReLU(inplace=True)

    # ---- per-field pieces (host, torch autograd) -------------------------------------------------
    def hyper_weights(self, meta_out, fore_h):
        """(w1b1 [256,193], w2b2 [256,257], evec [256]) -- variable_net.py:57-65,75-78."""
        m = torch.squeeze(meta_out, dim=0)[0:self.token_num]
        w1b1 = self.coord_input_fc(m.T)
        w2b2 = self.coord_hidden_fc(m.T)
        evec = self.fore_h_fc(self.pe_fore_h(fore_h.squeeze(dim=-1)))[0]
        return w1b1, w2b2, evec

    def static_params(self):
        """The eight parameter tensors the point kernels read directly (point_path.STATIC_NAMES order)."""
        return [self.data_input_fc.weight, self.data_input_fc.bias, self.cat_fc1.fc[0].weight, self.cat_fc1.fc[0].bias,
                self.cat_fc1.fc[2].weight, self.cat_fc1.fc[2].bias, self.out_fc.weight, self.out_fc.bias]

    def forward(self, meta_out, coord, coord_data, ref_data, fore_h):
        """Reference signature: coord is the [N,192] encoded coordinate tensor, ref_data [N,1] is added to the output."""
        from ..point_path import PointConfig, point_fields
        w1b1, w2b2, evec = self.hyper_weights(meta_out, fore_h)
        import copy
        cfg = copy.copy(getattr(self, '_point_cfg', None) or PointConfig())
        heads = torch.cat([w1b1] * 6 + [w2b2] * 6, dim=1)
        # the point kernels evaluate six nets per launch: this one fills all six slots, and an inference call (nothing requires a gradient)
        # launches the first slot only (cfg.n_nets = 1 -> dpn_fwd_ref_nets); a differentiated call keeps the six-slot launch, whose saved
        # state the backward kernels expect.  ref_data is added inside the kernel (dpn_fwd_ref): the output is the kernel's own sum, not
        # rebuilt by subtraction.  ref_data enters as a constant; where the caller needs d out / d ref_data (= 1) it is added outside
        ref = ref_data.reshape(-1, 1)
        cfg.ref6 = (torch.zeros_like(ref) if ref.requires_grad else ref.detach()).expand(-1, 6).contiguous()
        needs_grad = torch.is_grad_enabled() and any(v.requires_grad for v in (meta_out, coord, coord_data, ref_data) + tuple(self.parameters()))
        cfg.n_nets = 6 if needs_grad else 1
        out = point_fields(cfg, coord_data, heads, evec.unsqueeze(0).expand(6, -1).contiguous(), self.static_params() * 6, pe_in=coord)
        return out[:, 0:1] + ref if ref.requires_grad else out[:, 0:1]
