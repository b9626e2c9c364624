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
    } else if (ORDER == 3) {                                   // the kernel's order cut into four workgroups: four stores per wave instead of sixteen
        const long bb = b >> 2; const int q = (int)(b & 3), t = q >> 1, p = q & 1;
        for (int s = 0; s < 2; ++s) for (int kk = 0; kk < 2; ++kk) st(s, 2 * bb + p, kk, 2 * w + t);
    } else if (ORDER == 4) {                                   // ... into sixteen: one store per wave
        const long bb = b >> 4; const int q = (int)(b & 15), t = q >> 3, p = (q >> 2) & 1, s = (q >> 1) & 1, kk = q & 1;
        st(s, 2 * bb + p, kk, 2 * w + t);
    } else {
        char* p = base + (b * 4 + w) * 16384 + lane * 16;      // 16 KB contiguous per wave
        for (int i = 0; i < 16; ++i) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + i * 1024)); else *reinterpret_cast<u32x4*>(p + i * 1024) = v; }
    }
}
// other shapes of a plain linear fill: LPT 16-byte stores per lane per iteration (adjacent), persistent workgroups walking the buffer with a grid stride
template <int LPT, bool NT, bool CONST_V>
__global__ __launch_bounds__(256) void fill(char* base, long bytes) {
    const long stride = (long)gridDim.x * 256 * 16 * LPT;
    for (long off = ((long)blockIdx.x * 256 + threadIdx.x) * 16 * LPT; off < bytes; off += stride) {
        const u32x4 v = CONST_V ? u32x4{1u, 1u, 1u, 1u} : u32x4{(unsigned)off, (unsigned)threadIdx.x, 3u, 7u};
#pragma unroll
        for (int i = 0; i < LPT; ++i) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + off + 16 * i)); else *reinterpret_cast<u32x4*>(base + off + 16 * i) = v; }
    }
}
template <int LPT, bool NT, bool CONST_V> void run_fill(char* buf, long bytes, int grid, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill<LPT, NT, CONST_V>), dim3(grid), dim3(256), 0, 0, buf, bytes);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((fill<LPT, NT, CONST_V>), dim3(grid), dim3(256), 0, 0, buf, bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s grid %6d  %7.1f us  %.2f TB/s\n", name, grid, ms / 20 * 1e3, (double)bytes / (ms / 20 * 1e-3) / 1e12);
}
template <int ORDER, bool NT> void run(char* buf, long wgs0, const char* name) {
    const long wgs = ORDER == 3 ? wgs0 * 4 : ORDER == 4 ? wgs0 * 16 : wgs0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<ORDER, NT>), dim3(wgs), dim3(256), 0, 0, buf, 2 * wgs0);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<ORDER, NT>), dim3(wgs), dim3(256), 0, 0, buf, 2 * wgs0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs0 * 65536;
    printf("%-44s %7.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const long wgs = 3498 * 2;                        // 458 MB: the stage-1 launch's volume
    char* buf; hipMalloc(&buf, wgs * 65536);
    run<0, true>(buf, wgs, "kernel's order, non-temporal");
    run<1, true>(buf, wgs, "four adjacent pieces per wave, non-temporal");
    run<2, true>(buf, wgs, "16 KB contiguous per wave, non-temporal");
    run<3, true>(buf, wgs, "kernel's order, 4 stores per wave, NT");
    run<4, true>(buf, wgs, "kernel's order, 1 store per wave, NT");
    run<4, false>(buf, wgs, "kernel's order, 1 store per wave, plain");
    run<0, false>(buf, wgs, "kernel's order, plain");
    run<1, false>(buf, wgs, "four adjacent pieces per wave, plain");
    run<2, false>(buf, wgs, "16 KB contiguous per wave, plain");
    const long bytes = wgs * 65536;
    run_fill<1, false, false>(buf, bytes, (int)(bytes / 4096), "fill 16 B/lane, one pass per workgroup");
    run_fill<4, false, false>(buf, bytes, (int)(bytes / 16384), "fill 64 B/lane, one pass per workgroup");
    run_fill<4, false, true>(buf, bytes, (int)(bytes / 16384), "fill 64 B/lane, constant value");
    run_fill<1, false, false>(buf, bytes, 2048, "fill 16 B/lane, 2048 persistent workgroups");
    run_fill<4, false, false>(buf, bytes, 2048, "fill 64 B/lane, 2048 persistent workgroups");
    run_fill<4, true, false>(buf, bytes, 2048, "fill 64 B/lane, 2048 persistent, non-temporal");
    run_fill<1, true, false>(buf, bytes, 2048, "fill 16 B/lane, 2048 persistent, non-temporal");
    return 0;
}
