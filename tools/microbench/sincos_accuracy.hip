// Accuracy of candidate sin/cos evaluations for the positional features (angles up to ~16 rad for coordinates, ~70 rad for the data
// features), against double-precision libm on the host.   hipcc --offload-arch=gfx950 -O3 sincos_accuracy.hip -o sincos_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void sincos_precise(float th, float& s, float& c) {      // the kernel's current routine (dpn_kernels.hip)
    const float k = rintf(th * 0.63661977236758134f);
    float r = fmaf(k, -1.5707963705062866f, th);
    r = fmaf(k, 4.371138828673793e-08f, r);
    const float r2 = r * r;
    float sp = fmaf(r2, 2.7183114939898219064e-6f, -1.9839334836096632576e-4f);
    sp = fmaf(sp, r2, 8.3333293858894631756e-3f);
    sp = fmaf(sp, r2, -1.6666666641626524100e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.4433157826443582e-5f, -1.3887316255057415e-3f);
    cp = fmaf(cp, r2, 4.1666645683529456e-2f);
    cp = fmaf(cp, r2, -0.5f);
    cp = fmaf(cp, r2, 1.0f);
    const int q = ((int)k) & 3;
    const float ss = (q & 1) ? cp : sp;
    const float cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
// hardware v_sin_f32 / v_cos_f32 (argument in revolutions) behind a two-term reduction: t = th / 2pi with the product's rounding error
// recovered by an fma, the integer part removed exactly
__device__ __forceinline__ void sincos_hw2(float th, float& s, float& c) {
    const float inv_hi = 0.15915494309189535f;                   // float(1 / 2pi)
    const float inv_lo = -6.8927500628e-09f;                     // 1 / 2pi - float(1 / 2pi)  (approx.)
    const float t = th * inv_hi;
    const float e = fmaf(th, inv_hi, -t);                        // exact rounding error of the product
    const float k = rintf(t);
    float r = (t - k) + fmaf(th, inv_lo, e);                     // |r| <= 0.5 revolutions
    asm("v_sin_f32 %0, %1" : "=v"(s) : "v"(r));
    asm("v_cos_f32 %0, %1" : "=v"(c) : "v"(r));
}
__device__ __forceinline__ void sincos_fast(float th, float& s, float& c) { s = __sinf(th); c = __cosf(th); }

__global__ void run(const float* th, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_precise(th[i], s, c); out[6 * i + 0] = s; out[6 * i + 1] = c;
    sincos_hw2(th[i], s, c);     out[6 * i + 2] = s; out[6 * i + 3] = c;
    sincos_fast(th[i], s, c);    out[6 * i + 4] = s; out[6 * i + 5] = c;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> th(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;
        th[i] = (float)((u * 2.0 - 1.0) * ((i & 1) ? 16.0 : 70.0));
    }
    float *d_th, *d_out;
    hipMalloc(&d_th, n * 4); hipMalloc(&d_out, (size_t)n * 24);
    hipMemcpy(d_th, th.data(), n * 4, hipMemcpyHostToDevice);
    run<<<(n + 255) / 256, 256>>>(d_th, d_out, n);
    std::vector<float> out((size_t)n * 6);
    hipMemcpy(out.data(), d_out, (size_t)n * 24, hipMemcpyDeviceToHost);
    double err[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    for (int i = 0; i < n; ++i) {
        const double s = sin((double)th[i]), c = cos((double)th[i]);
        const int range = (i & 1) ? 0 : 1;
        for (int v = 0; v < 3; ++v) {
            const double e = fmax(fabs(out[6 * (size_t)i + 2 * v] - s), fabs(out[6 * (size_t)i + 2 * v + 1] - c));
            if (e > err[v][range]) err[v][range] = e;
        }
    }
    const char* names[3] = {"sincos_precise (Cody-Waite + minimax)", "v_sin/v_cos + two-term reduction", "__sinf/__cosf"};
    for (int v = 0; v < 3; ++v) printf("%-40s max abs error  |th| <= 16: %.3e   |th| <= 70: %.3e\n", names[v], err[v][0], err[v][1]);
    return 0;
}
