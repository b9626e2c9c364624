"""TEST INFRASTRUCTURE -- closed-form, RNG-free parameter / input fills.

Both the golden-vector generator (which fills the *reference* model) and the
tests (which fill the oracle and the HIP-backed modules) call these functions,
so 22 MB of weights never have to be committed.  Every value is an exact fp32
number derived from an integer hash of (tensor name, flat index).
"""
import zlib

import numpy as np
import torch


def _hash_u32(seed: int, n: int) -> np.ndarray:
    """n 32-bit hashes of (seed, 0..n-1): a xorshift-multiply mixer on uint64."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        h = i * np.uint64(0x9E3779B97F4A7C15) + np.uint64((int(seed) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(31)
        h = h * np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(29)
        h = h * np.uint64(0xD6E8FEB86659FD93)
        h ^= h >> np.uint64(32)
    return (h & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def unit_uniform(name: str, n: int, salt: int = 0) -> np.ndarray:
    """n exact-fp32 values in [-1, 1) (multiples of 2**-15) keyed by `name`."""
    seed = zlib.crc32(name.encode()) + 0x51ED27 * salt
    h = _hash_u32(seed, n)
    return ((h >> np.uint32(16)).astype(np.float32) - np.float32(32768.0)) / np.float32(32768.0)


def unit_normalish(name: str, n: int, salt: int = 0) -> np.ndarray:
    """Approximately N(0,1) (Irwin-Hall of 4 uniforms), exact fp32, keyed by `name`."""
    acc = np.zeros(n, dtype=np.float32)
    for k in range(4):
        acc += unit_uniform(name, n, salt=salt * 7 + k + 1)
    return acc * np.float32(np.sqrt(3.0 / 4.0))


def _scale_for(name: str, shape) -> float:
    """~1/sqrt(fan_in) for weights, small for biases, so activations stay O(1)."""
    leaf = name.split('.')[-1]
    if name.endswith('learnable_token'):
        return 0.5
    if leaf == 'bias':
        return 0.05
    if leaf == 'weight':
        if len(shape) == 1:          # LayerNorm gain, handled separately
            return 0.1
        fan_in = int(np.prod(shape[1:]))
        if 'coord_input_fc' in name or 'coord_hidden_fc' in name:
            # hyper-network heads: their OUTPUT is a weight matrix, keep it ~1/sqrt(fan_in of that matrix)
            return float(1.7 / np.sqrt(fan_in) / 8.0)
        return float(1.7 / np.sqrt(fan_in))
    return 0.1


def fill_state_dict_(state: dict, gain: float = 1.0) -> dict:
    """Overwrite every floating tensor of `state` (name -> tensor) in place.

    Non-persistent / constant buffers (the sinusoid `pe` table) are left alone.
    LayerNorm gains are 1 + small so the encoder stays well conditioned.
    `gain` > 1 widens the VariableNet outputs so more points hit the clip bounds.
    """
    for name, t in state.items():
        if not torch.is_floating_point(t):
            continue
        if name.endswith('position_embedding.pe'):
            continue
        n = t.numel()
        vals = unit_uniform(name, n) * np.float32(_scale_for(name, tuple(t.shape)))
        leaf = name.split('.')[-1]
        if leaf == 'weight' and t.dim() == 1:
            vals = vals + np.float32(1.0)
        if gain != 1.0 and ('_net.' in name and 'meta_net' not in name) and leaf == 'weight' and 'out_fc' in name:
            vals = vals * np.float32(gain)
        with torch.no_grad():
            t.copy_(torch.from_numpy(vals.reshape(tuple(t.shape))).to(t.dtype))
    return state


# ----------------------------------------------------------------------------
# synthetic inputs (shapes / distributions follow physics_dataset.py)
# ----------------------------------------------------------------------------
def synthetic_inputs(n_points: int, lon: int = 257, lat: int = 145, dx: float = 27000.0, dy: float = 27000.0,
                     tag: str = 'inter', margin: bool = False, forecast_h: float = 24.0 / 360.0):
    """Closed-form stand-ins for one dataset sample (fp32 torch tensors on CPU).

    field_data [1,159,2405]; forecast_h [1,1,1]; x,y,t,f [N,1]; coord_data [N,6]; labels [N,6].
    Interior points are continuous in the domain, margin points sit on grid nodes
    (reference: dataset/physics_dataset.py:431-499 and :323-429).
    """
    field = unit_normalish('field_data', 159 * 2405).reshape(1, 159, 2405).copy()
    field[:, 155:, :] = (unit_uniform('field_const', 4 * 2405).reshape(1, 4, 2405) + 1.0) * 0.5
    ux = (unit_uniform(tag + '.x', n_points) + 1.0) * 0.5
    uy = (unit_uniform(tag + '.y', n_points) + 1.0) * 0.5
    ut = (unit_uniform(tag + '.t', n_points) + 1.0) * 0.5
    if margin:
        x = np.floor(ux * lon).clip(0, lon - 1).astype(np.float32) * np.float32(dx)
        y = np.floor(uy * lat).clip(0, lat - 1).astype(np.float32) * np.float32(dy)
    else:
        x = (ux * np.float32(lon - 1) * np.float32(dx)).astype(np.float32)
        y = (uy * np.float32(lat - 1) * np.float32(dy)).astype(np.float32)
    t = np.floor(ut * 25).clip(0, 24).astype(np.float32) * np.float32(3600.0)
    lat_deg = 18.0 + y.astype(np.float64) / dy * 0.25
    f = (2.0 * 7.29e-5 * np.sin(lat_deg * np.pi / 180.0)).astype(np.float32)
    coord_data = unit_normalish(tag + '.coord_data', n_points * 6).reshape(n_points, 6)
    labels = unit_normalish(tag + '.labels', n_points * 6).reshape(n_points, 6)
    tt = torch.from_numpy
    return dict(
        field_data=tt(field), forecast_h=torch.full((1, 1, 1), float(forecast_h), dtype=torch.float32),
        x=tt(x).reshape(-1, 1), y=tt(y).reshape(-1, 1), t=tt(t).reshape(-1, 1), f=tt(f).reshape(-1, 1),
        coord_data=tt(np.ascontiguousarray(coord_data)), labels=tt(np.ascontiguousarray(labels)),
    )
