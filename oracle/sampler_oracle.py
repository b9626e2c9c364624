"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's collocation-point generators
(dataset/physics_dataset.py:323-429 get_item_label_data, :431-499 get_inter_data, :528-587 get_margin_grid,
:521-526 get_coriolis) for given draws.

PARITY UNPINNED against the reference itself: those functions need `xarray` (absent from this image) and GeoTIFF
inputs, so the reference cannot be run here and no golden vectors exist.  `DataArray.interp` with its default
method='linear' on a regular grid is scipy.interpolate.interpn(method='linear'); this oracle calls scipy's
RegularGridInterpolator (the same code) in float64 and casts to float32 as the reference's `.float()` does.
"""
import numpy as np
from scipy.interpolate import RegularGridInterpolator


def coriolis(lat_deg):
    """physics_dataset.py:521-526"""
    omega = 7.29e-5
    f = 2 * omega * np.sin(lat_deg / 180 * np.pi)
    return np.expand_dims(f, axis=1) if f.ndim == 1 else f


def points_from_draws(cube, x_rand, y_rand, t_rand, begin_lon, begin_lat, in_lon, in_lat, input_time_step, dx, dy, out_res=0.25):
    """cube [6, y, x, t] (float32); draws in fine-grid index units / hours (float64 or int).
    -> x, y, t (float32 [n]), coord_data float32 [n,6], f float32 [n,1]        (physics_dataset.py:442-499 / :334-428)"""
    x_rand, y_rand, t_rand = (np.asarray(v, dtype=np.float64) for v in (x_rand, y_rand, t_rand))
    lon = begin_lon + x_rand * out_res
    lat = begin_lat + y_rand * out_res
    coord_t = np.arange(cube.shape[3]) * input_time_step
    cols = []
    for k in range(6):
        interp = RegularGridInterpolator((np.asarray(in_lat, dtype=np.float64), np.asarray(in_lon, dtype=np.float64), coord_t.astype(np.float64)),
                                         cube[k].astype(np.float64), method='linear', bounds_error=False, fill_value=np.nan)
        cols.append(interp(np.stack([lat, lon, t_rand], axis=1)))
    data = np.stack(cols, axis=-1)
    return ((x_rand * dx).astype(np.float32), (y_rand * dy).astype(np.float32), (t_rand * 3600.0).astype(np.float32),
            data.astype(np.float32), coriolis(lat).astype(np.float32))


def labels_at(labels, x_idx, y_idx, t_idx):
    """labels [hours, 6, lat, lon]: the read_point loop of physics_dataset.py:347-365 -> [n,6]"""
    x_idx, y_idx, t_idx = (np.asarray(v).astype(np.int64) for v in (x_idx, y_idx, t_idx))
    return labels[t_idx, :, y_idx, x_idx].astype(np.float32)


def grid_maps(out_n, lon, lat, mean, std, clip_lo, clip_hi, with_clip):
    """interface_physics.py:559-591: inverse_norm + scatter of node-ordered (x outer, y inner) fields into [6, lat, lon] maps."""
    v = out_n.astype(np.float32) * np.asarray(std, np.float32)[None] + np.asarray(mean, np.float32)[None]
    if with_clip:
        for k in range(2, 6):
            v[:, k] = np.clip(v[:, k], np.float32(clip_lo[k]), np.float32(clip_hi[k]))
    return np.ascontiguousarray(v.reshape(lon, lat, 6).transpose(2, 1, 0))
