"""SineCosPE with the reference's constructor and output layout (utils/position_encoding.py:11-50).

Used on the host side for the few per-FIELD encodings (lead time); the per-POINT encodings of the hot path
are generated inside the HIP kernels (csrc/dpn_kernels.hip: build_pe3 / build_pe6) with the same layout
[frequency, (sin, cos), channel].
"""
import torch
import torch.nn as nn


class SineCosPE(nn.Module):
    def __init__(self, input_dim, N_freqs=32, max_freq=4, periodic_fns=(torch.sin, torch.cos),
                 log_sampling=True, include_input=True, trainable=False):
        super().__init__()
        self.periodic_fns = tuple(periodic_fns)
        self.include_input = include_input or len(self.periodic_fns) == 0
        self.out_dim = len(self.periodic_fns) * input_dim * N_freqs + (input_dim if self.include_input else 0)
        if log_sampling:
            bands = 2.0 ** torch.linspace(0.0, max_freq, steps=N_freqs)          # fp32 on purpose (reference :27)
        else:
            bands = torch.linspace(2.0 ** 0.0, 2.0 ** max_freq, steps=N_freqs)
        if trainable:
            self.freq_bands = nn.Parameter(bands, requires_grad=True)
        else:
            self.register_buffer('freq_bands', bands, persistent=False)

    def forward(self, inputs):
        # [..., C] -> [..., F, C] per function, interleaved as [..., F, n_fns, C], flattened
        scaled = inputs.unsqueeze(-2) * self.freq_bands.unsqueeze(-1)
        feats = torch.stack([fn(scaled) for fn in self.periodic_fns], dim=-2)
        feats = feats.flatten(start_dim=-3)
        if self.include_input:
            feats = torch.cat([inputs, feats], dim=-1)
        return feats
