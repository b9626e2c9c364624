// Micro-benchmark (round 6): is the stage-1 backward kernel's store ORDER what keeps it at 4.4-4.9 TB/s when a plain fill writes 6.8 TB/s?
// dpn_bwd_tiles_kernel writes its operands as K-layout images [plane][tile32][kk 2][ct][64 lanes x 16 B]: every store instruction of a wave is ONE
// contiguous 1-KB piece, and consecutive instructions of a wave go to pieces that are 6-8 KB (kk), hundreds of MB (plane) or 16 KB (tile) apart.
// Here: the Z1-like image (2 planes, 8 column tiles), 3 498 workgroups x 4 waves x 16 stores of 1 KB, in the kernel's order (A), with every wave writing
// four ADJACENT pieces back to back (B), and as one linear fill (C); non-temporal and plain stores.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int ORDER, bool NT>
__global__ __launch_bounds__(256) void k(char* base, long tiles32) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long b = blockIdx.x;
    const u32x4 v = {(unsigned)b, (unsigned)w, (unsigned)lane, 7u};
    auto st = [&](int s, long tile, int kk, int ct) __attribute__((always_inline)) {
        char* p = base + ((s * tiles32 + tile) * 16384) + ((kk * 8 + ct) * 64 + lane) * 16;
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = v;
    };
    if (ORDER == 0) {
        for (int t = 0; t < 2; ++t) for (int p = 0; p < 2; ++p) for (int s = 0; s < 2; ++s) for (int kk = 0; kk < 2; ++kk) st(s, 2 * b + p, kk, 2 * w + t);
    } else if (ORDER == 1) {
        const int p = w & 1, c0 = 4 * (w >> 1);
        for (int s = 0; s < 2; ++s) for (int kk = 0; kk < 2; ++kk) for (int c = 0; c < 4; ++c) st(s, 2 * b + p, kk, c0 + c);
    } else {
        char* p = base + (b * 4 + w) * 16384 + lane * 16;      // 16 KB contiguous per wave
        for (int i = 0; i < 16; ++i) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + i * 1024)); else *reinterpret_cast<u32x4*>(p + i * 1024) = v; }
    }
}
template <int ORDER, bool NT> void run(char* buf, long wgs, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<ORDER, NT>), dim3(wgs), dim3(256), 0, 0, buf, 2 * wgs);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<ORDER, NT>), dim3(wgs), dim3(256), 0, 0, buf, 2 * wgs);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * 65536;
    printf("%-44s %7.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const long wgs = 3498 * 2;                        // 458 MB: the stage-1 launch's volume
    char* buf; hipMalloc(&buf, wgs * 65536);
    run<0, true>(buf, wgs, "kernel's order, non-temporal");
    run<1, true>(buf, wgs, "four adjacent pieces per wave, non-temporal");
    run<2, true>(buf, wgs, "16 KB contiguous per wave, non-temporal");
    run<0, false>(buf, wgs, "kernel's order, plain");
    run<1, false>(buf, wgs, "four adjacent pieces per wave, plain");
    run<2, false>(buf, wgs, "16 KB contiguous per wave, plain");
    return 0;
}
