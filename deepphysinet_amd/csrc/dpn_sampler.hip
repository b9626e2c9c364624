// Collocation sampler + trilinear interpolation of the coarse forecast cube, and the full-grid de-normalise/scatter
// (SURVEY.md section 8 rows f1 / f3).  HBM-bound gather kernels: one thread per point, all index / weight arithmetic in
// fp64 exactly as the reference's numpy/xarray code does it (dataset/physics_dataset.py:334-338, 383-415, 442-446,
// 454-499, 521-526, 528-587), outputs cast to fp32 at the end like the reference's `.float()`.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dpn_hip.h"

namespace {

// Philox-4x32-10 (Salmon et al. 2011), counter = (point index, stream id), key = seed.
__device__ inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ inline double u53(uint32_t hi, uint32_t lo) {               // uniform double in [0,1), 53 random bits (np.random.rand)
    return (double)((((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11)) & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}
__device__ inline uint32_t below(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }   // randint(0, n)

struct SampleArgs {
    DpnSampler s;
    const float* cube;
    const float* labels;
    const int32_t *xi, *yi, *ti;
    int64_t n;
    uint64_t seed, offset;
    const int32_t* step_dev;      // optional device-side step counter: the draw counter starts at offset + *step_dev * stride
    uint64_t stride;
    int mode;
    float *x, *y, *t, *f, *coord_data, *label_out;
    double* raw;
};

__global__ __launch_bounds__(256) void dpn_sample_kernel(SampleArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const DpnSampler& s = a.s;
    double xr, yr, tr;
    if (a.mode == DPN_SAMPLE_EXPLICIT) {
        xr = a.xi[i]; yr = a.yi[i]; tr = a.ti[i];
    } else {
        uint32_t w[4], v[4];
        const uint64_t ctr = a.offset + (a.step_dev ? (uint64_t)(uint32_t)(*a.step_dev) * a.stride : 0ull) + (uint64_t)i;
        philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
        philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), v);
        if (a.mode == DPN_SAMPLE_INTERIOR) {                            // np.random.rand(n) * (size - 1)   (:442-443)
            xr = u53(w[0], w[1]) * (double)(s.lon - 1);
            yr = u53(w[2], w[3]) * (double)(s.lat - 1);
        } else {                                                        // np.random.randint(0, size)        (:334-335)
            xr = below(w[0], (uint32_t)s.lon);
            yr = below(w[2], (uint32_t)s.lat);
        }
        tr = below(v[0], (uint32_t)(s.t_hours + 1));                    // randint(0, step * nums + 1)       (:338, :446)
    }
    // trilinear interpolation of the coarse cube [6][lat_in][lon_in][t_in] at (lat, lon, hour)               (:405-411)
    const double fx = xr * s.cells_x, fy = yr * s.cells_y, ft = tr / s.t_step_hours;
    int ix = (int)floor(fx), iy = (int)floor(fy), it = (int)floor(ft);
    ix = ix < 0 ? 0 : (ix > s.lon_in - 2 ? s.lon_in - 2 : ix);
    iy = iy < 0 ? 0 : (iy > s.lat_in - 2 ? s.lat_in - 2 : iy);
    it = it < 0 ? 0 : (it > s.t_in - 2 ? s.t_in - 2 : it);
    const double wx = fx - ix, wy = fy - iy, wt = ft - it;
    const bool inside = wx >= 0.0 && wx <= 1.0 && wy >= 0.0 && wy <= 1.0 && wt >= 0.0 && wt <= 1.0;   // outside -> NaN (xarray)
    const int64_t sy = (int64_t)s.lon_in * s.t_in, sx = s.t_in, sk = (int64_t)s.lat_in * sy;
    const float* c0 = a.cube + (int64_t)iy * sy + (int64_t)ix * sx + it;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float* c = c0 + k * sk;
        const double v000 = c[0], v001 = c[1], v010 = c[sx], v011 = c[sx + 1];
        const double v100 = c[sy], v101 = c[sy + 1], v110 = c[sy + sx], v111 = c[sy + sx + 1];
        const double lo = (v000 * (1 - wt) + v001 * wt) * (1 - wx) + (v010 * (1 - wt) + v011 * wt) * wx;
        const double hi = (v100 * (1 - wt) + v101 * wt) * (1 - wx) + (v110 * (1 - wt) + v111 * wt) * wx;
        a.coord_data[i * 6 + k] = inside ? (float)(lo * (1 - wy) + hi * wy) : __builtin_nanf("");
    }
    const double lat_deg = s.begin_lat + yr * s.dlat;
    a.x[i] = (float)(xr * (double)s.dx);
    a.y[i] = (float)(yr * (double)s.dy);
    a.t[i] = (float)(tr * 3600.0);
    a.f[i] = (float)(2.0 * 7.29e-5 * sin(lat_deg / 180.0 * 3.141592653589793));                           // get_coriolis (:521-526)
    if (a.raw) { a.raw[i * 3] = xr; a.raw[i * 3 + 1] = yr; a.raw[i * 3 + 2] = tr; }
    if (a.labels && a.label_out) {                                      // read_point of the ERA5 label at (x, y, t)  (:347-365)
        const int X = (int)xr, Y = (int)yr, T = (int)tr;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            a.label_out[i * 6 + k] = a.labels[(((int64_t)T * 6 + k) * s.lat + Y) * s.lon + X];
    }
}

struct MapArgs {
    const float* out_n;
    int lon, lat;
    DpnPhysics ph;
    int with_clip;
    float* maps;
};
// out_n [lon*lat, 6] in the reference's node order (x outer, y inner; interface_physics.py:538-543) -> maps [6][lat][lon]
// de-normalised (inverse_norm, :232-262).  Reads are strided by node order, writes are coalesced along lon.
__global__ __launch_bounds__(256) void dpn_grid_maps_kernel(MapArgs a) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nodes = (int64_t)a.lon * a.lat;
    if (j >= nodes * 6) return;
    const int k = (int)(j / nodes);
    const int64_t r = j - k * nodes;
    const int yy = (int)(r / a.lon), xx = (int)(r - (int64_t)yy * a.lon);
    // two roundings (mul, then add) like the reference's `v * std + mean` in torch -- not contracted into one fma
    float v;
    {
#pragma clang fp contract(off)
        const float prod = a.out_n[((int64_t)xx * a.lat + yy) * 6 + k] * a.ph.std[k];
        v = prod + a.ph.mean[k];
        if (a.ph.sq_on[k]) { const float sq = v * v; v = sq + a.ph.sq_add[k]; }       // three-factor min_max (interface_physics.py:244-247)
    }
    if (a.with_clip && k >= 2) v = v != v ? v : fminf(fmaxf(v, a.ph.clip_lo[k]), a.ph.clip_hi[k]);      // NaN passes through like torch.clip
    a.maps[j] = v;
}

}  // namespace

extern "C" {

int dpn_sample_points(const DpnSampler* s, const float* cube, const float* labels, int mode, const int32_t* xi, const int32_t* yi,
                      const int32_t* ti, int64_t n, uint64_t seed, uint64_t offset, float* x, float* y, float* t, float* f,
                      float* coord_data, float* label_out, double* raw, void* stream) {
    return dpn_sample_points_replay(s, cube, labels, mode, xi, yi, ti, n, seed, offset, nullptr, 0, x, y, t, f, coord_data, label_out, raw, stream);
}

int dpn_sample_points_replay(const DpnSampler* s, const float* cube, const float* labels, int mode, const int32_t* xi, const int32_t* yi,
                             const int32_t* ti, int64_t n, uint64_t seed, uint64_t offset, const int32_t* step_dev, uint64_t stride,
                             float* x, float* y, float* t, float* f, float* coord_data, float* label_out, double* raw, void* stream) {
    if (!s || !cube || !x || !y || !t || !f || !coord_data || n <= 0) return -1;
    if (mode != DPN_SAMPLE_INTERIOR && mode != DPN_SAMPLE_MARGIN && mode != DPN_SAMPLE_EXPLICIT) return -1;
    if (mode == DPN_SAMPLE_EXPLICIT && (!xi || !yi || !ti)) return -1;
    if (s->lon_in < 2 || s->lat_in < 2 || s->t_in < 2 || s->lon < 2 || s->lat < 2) return -1;
    if (label_out && (!labels || mode == DPN_SAMPLE_INTERIOR)) return -1;          // labels exist at grid nodes only
    SampleArgs a{*s, cube, labels, xi, yi, ti, n, seed, offset, step_dev, stride, mode, x, y, t, f, coord_data, label_out, raw};
    hipLaunchKernelGGL(dpn_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int dpn_grid_maps(const float* out_n, int lon, int lat, const DpnPhysics* phys, int with_clip, float* maps, void* stream) {
    if (!out_n || !phys || !maps || lon <= 0 || lat <= 0) return -1;
    MapArgs a{out_n, lon, lat, *phys, with_clip, maps};
    const int64_t total = (int64_t)lon * lat * 6;
    hipLaunchKernelGGL(dpn_grid_maps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
