// The tile-split forward kernel's k-step in isolation: per wave and step 4 x 1 KB weight fragments through a buffer descriptor (two tiles x
// hi, lo), 4 x 1 KB activation fragments from LDS (two column tiles x hi, lo), 12 MFMAs.  256-thread workgroups, two per CU, nothing else
// in the kernel: what does the loop reach on its own, and which of its three streams costs what?
//   hipcc --offload-arch=gfx950 -O3 -o kstep_loop tools/microbench/kstep_loop.hip && ./kstep_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// GL: weight fragments from global (else registers), LD: activation fragments from LDS (else registers), PF: k-steps of weights in flight
template <bool GL, bool LD, int PF, int PRIO>
__global__ __launch_bounds__(256, 2) void k(const char* buf, long region_bytes, int regions, int steps, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const char* base = buf + (long)(blockIdx.x % regions) * region_bytes + wave * (region_bytes / 4);
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(base)[i & 1023];
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(region_bytes / 4), 0x00020000);
    const int voff = lane * 16;
    const char* xl = lds + lane * 16;
    f32x16 acc[2][2] = {{(f32x16)0.f, (f32x16)0.f}, {(f32x16)0.f, (f32x16)0.f}};
    u32x4 A[PF][4];
    u32x4 B[2][4];
    int so = 0;
    const int wrap = (int)(region_bytes / 4) - 4096;
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int f = 0; f < 4; ++f) A[d][f] = (u32x4){0x3c003c00u + d, 0x3c003c00u, 0x3c003c00u + f, 0x3c003c00u};
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int f = 0; f < 4; ++f) B[d][f] = (u32x4){0x3c003c00u, 0x3c003c00u + d, 0x3c003c00u, 0x3c003c00u + f};
    for (int st = 0; st < steps; st += PF * 2) {
#pragma unroll
        for (int u = 0; u < PF * 2; ++u) {
            if constexpr (GL) {
#pragma unroll
                for (int f = 0; f < 4; ++f) A[u % PF][f] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (f & 1) * 1024, so + (f >> 1) * 2048, 0));
                so += 4096; if (so > wrap) so = 0;
                asm volatile("" : "+s"(so));
            }
            if constexpr (LD) {
#pragma unroll
                for (int f = 0; f < 4; ++f) B[(u + 1) & 1][f] = *reinterpret_cast<const u32x4*>(xl + (((st + u) & 15) * 4 + f) * 1024);
            }
            if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
            const int a = (u + 1) % PF, b = u & 1;        // the oldest weight slot, the activation slot read one step ago
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    acc[t][p] = MF(A[a][2 * t], B[b][2 * p + 1], acc[t][p]);
                    acc[t][p] = MF(A[a][2 * t + 1], B[b][2 * p], acc[t][p]);
                    acc[t][p] = MF(A[a][2 * t], B[b][2 * p], acc[t][p]);
                }
            if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    if (s == 12345.678f) sink[0] = s;
}

template <bool GL, bool LD, int PF, int PRIO>
static void run(const char* buf, long region, int regions, float* sink, int cus, const char* what) {
    const int steps = 1664, grid = cus * 2 * 4;       // 1664 k-steps = 16 layers' worth per workgroup, four rounds of two workgroups per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<GL, LD, PF, PRIO>), dim3(grid), dim3(256), 0, 0, buf, region, regions, 64, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<GL, LD, PF, PRIO>), dim3(grid), dim3(256), 0, 0, buf, region, regions, steps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mfma = (double)grid * 4 * steps * 12;
    printf("%-44s weights in flight %d k-steps: %7.3f ms  mfma %5.1f %% of 2.5 PF  (%5.1f B/clk/CU from L2 at 2.1 GHz)\n", what, PF, best,
           mfma * 32768.0 / (best * 1e-3) / 2.5e15 * 100.0, GL ? (double)grid * 4 * steps * 4096 / (best * 1e-3) / cus / 2.1e9 : 0.0);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const long region = 1638400;
    const int regions = 6;
    char* buf; float* sink;
    hipMalloc(&buf, region * regions);
    hipMalloc(&sink, 64);
    std::vector<unsigned short> h(region * regions / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00u + (i * 2654435761u >> 22 & 0x3ffu));
    hipMemcpy(buf, h.data(), region * regions, hipMemcpyHostToDevice);
    run<false, false, 3, 0>(buf, region, regions, sink, cus, "MFMAs only");
    run<false, true, 3, 0>(buf, region, regions, sink, cus, "+ activation fragments from LDS");
    run<true, false, 3, 0>(buf, region, regions, sink, cus, "+ weight fragments from L2");
    run<true, true, 2, 0>(buf, region, regions, sink, cus, "both (the kernel's k-step)");
    run<true, true, 3, 0>(buf, region, regions, sink, cus, "both (the kernel's k-step)");
    run<true, true, 4, 0>(buf, region, regions, sink, cus, "both (the kernel's k-step)");
    run<true, true, 3, 1>(buf, region, regions, sink, cus, "both, s_setprio 1 around the MFMAs");
    return 0;
}
