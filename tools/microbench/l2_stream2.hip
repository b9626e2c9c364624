// Follow-up to l2_stream.hip: WHO pays for a 1-KB global load issued next to MFMAs?
//   mode 0  every wave loads and multiplies (the forward kernel's shape), 8 waves per CU (2 per SIMD)
//   mode 1  wave-specialised: workgroups of 512 threads, waves 0-3 only multiply, waves 4-7 only load (same loads and MFMAs per SIMD as
//           mode 0 with the same ratio) -- if the multiplying waves run at full rate here, the cost of a load is local to the issuing wave
//   mode 2  mode 1 without the loader waves' loads (multiplying waves alone + idle partners): the reference rate
//   mode 3  as mode 0, loads through a buffer descriptor with 32-bit offsets (buffer_load_dwordx4 ... offen)
//   mode 4  as mode 0, s_setprio 1 around the MFMAs
//   mode 5  as mode 0, the loaded fragments come from LDS instead (ds_read_b128 of a 64 KB image): what an LDS-fed operand costs
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream2 tools/microbench/l2_stream2.hip && ./l2_stream2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int MPK>
__global__ __launch_bounds__(MODE == 1 || MODE == 2 ? 512 : 256, 2) void k(const char* buf, long region_bytes, int regions, int passes, float* sink) {
    constexpr int DEPTH = 8;
    __shared__ __attribute__((aligned(16))) char lds[MODE == 5 ? 65536 : 16];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long quarter = region_bytes / 4;
    const char* base = buf + (long)(blockIdx.x % regions) * region_bytes + (wave & 3) * quarter;
    const u32x4* p = reinterpret_cast<const u32x4*>(base) + lane;
    const int n_kb = (int)(quarter / 1024);
    f32x16 acc[2] = {(f32x16)0.f, (f32x16)0.f};
    u32x4 x = (u32x4)0u;
    const bf16x8 bfrag = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    if constexpr (MODE == 5) {
        for (int i = threadIdx.x; i < 65536 / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = p[i & 1023];
        __syncthreads();
    }
    const bool loader = (MODE == 1 || MODE == 2) ? (wave >= 4) : true;
    const bool mult = (MODE == 1 || MODE == 2) ? (wave < 4) : true;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)quarter, 0x00020000);
    const int my_passes = ((MODE == 1) && wave >= 4) ? passes * (MPK / 3) : passes;      // loader waves: as many bytes as keeps them busy as long as their partners
    for (int pass = 0; pass < my_passes; ++pass) {
        u32x4 ring[DEPTH];
        if (loader && MODE != 2 && MODE != 5) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) ring[d] = p[d * 64];
        } else {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) ring[d] = (u32x4){0x3c003c00u + d, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        }
        for (int kb = 0; kb < n_kb; kb += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const u32x4 v = ring[d];
                const int nxt = kb + DEPTH + d;
                const int idx = (nxt < n_kb ? nxt : d);
                if (loader) {
                    if constexpr (MODE == 0 || MODE == 1 || MODE == 4) ring[d] = p[idx * 64];
                    else if constexpr (MODE == 3) ring[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + idx * 1024, 0, 0));
                    else if constexpr (MODE == 5) ring[d] = *reinterpret_cast<const u32x4*>(lds + ((idx & 63) * 1024 + lane * 16));
                }
                if (mult) {
                    if constexpr (MODE == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int m = 0; m < MPK; ++m)
                        acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), bfrag, acc[m & 1], 0, 0, 0);
                    if constexpr (MODE == 4) __builtin_amdgcn_s_setprio(0);
                } else {
                    x ^= v;
                }
            }
        }
    }
    float s = __builtin_bit_cast(float, x[0] ^ x[1] ^ x[2] ^ x[3]);
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE, int MPK>
static void run(const char* buf, long region, int regions, float* sink, int cus, const char* what) {
    const bool spec = (MODE == 1 || MODE == 2);
    const int passes = 4, grid = spec ? cus * 4 : cus * 2 * 4;       // specialised: one 512-thread workgroup per CU (2 waves per SIMD either way)
    const int threads = spec ? 512 : 256;
    // in the specialised form a multiplying wave executes the MFMAs of TWO mode-0 waves' loads?  No: keep per-wave work equal -- each
    // multiplying wave runs MPK MFMAs per KB its loader partner loads, so per SIMD: same loads, HALF the MFMAs of mode 0 at equal MPK.
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, MPK>), dim3(grid), dim3(threads), 0, 0, buf, region, regions, 1, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, MPK>), dim3(grid), dim3(threads), 0, 0, buf, region, regions, passes, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mwaves = spec ? 4.0 : 4.0;                          // multiplying waves per workgroup
    const double lwaves = (MODE == 2) ? 0.0 : 4.0;
    const double bytes = (double)grid * lwaves * (region / 4) * passes * (MODE == 1 ? MPK / 3 : 1);
    const double mfma = (double)grid * mwaves * (region / 4 / 1024) * passes * MPK;
    printf("mode %d (%-34s) mfma/KB %2d : %7.3f ms  %6.1f GB/s per CU (%5.1f B/clk at 2.1 GHz)  mfma %5.1f %% of 2.5 PF\n", MODE, what, MPK, best,
           bytes / best * 1e-6 / cus, bytes / best * 1e-6 / cus / 2.1, mfma * 32768.0 / (best * 1e-3) / 2.5e15 * 100.0);
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
    printf("%d CUs\n", cus);
    run<0, 3>(buf, region, regions, sink, cus, "all waves load + multiply");
    run<4, 3>(buf, region, regions, sink, cus, "same, s_setprio 1 around MFMAs");
    run<3, 3>(buf, region, regions, sink, cus, "same, buffer_load offen");
    run<5, 3>(buf, region, regions, sink, cus, "same, operand from LDS (ds_read_b128)");
    run<1, 6>(buf, region, regions, sink, cus, "4 multiply-only + 4 load-only waves");
    run<2, 6>(buf, region, regions, sink, cus, "4 multiply-only waves, partners idle");
    run<0, 6>(buf, region, regions, sink, cus, "all waves load + multiply");
    run<5, 6>(buf, region, regions, sink, cus, "same, operand from LDS (ds_read_b128)");
    run<1, 12>(buf, region, regions, sink, cus, "4 multiply-only + 4 load-only waves");
    run<2, 12>(buf, region, regions, sink, cus, "4 multiply-only waves, partners idle");
    run<0, 12>(buf, region, regions, sink, cus, "all waves load + multiply");
    run<5, 12>(buf, region, regions, sink, cus, "same, operand from LDS (ds_read_b128)");
    run<2, 3>(buf, region, regions, sink, cus, "4 multiply-only waves, partners idle");
    return 0;
}
