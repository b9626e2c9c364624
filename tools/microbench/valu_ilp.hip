// Micro-benchmark (round 6): how fast does ONE wave per SIMD run the feature code of the point kernels (Cody-Waite sincos + bf16 hi/lo split)?
// The epilogue / feature phases of dpn_fwd_tiles_kernel run at ~10 cycles per VALU instruction (profiles/round6_pp_*): is that the dependent-issue
// latency of a serial chain (one sincos after the other, as hipcc emits the unrolled loop), and does evaluating W angles in LOCKSTEP (the
// same operation on W independent angles back to back) recover the 4-cycle issue rate?
//   hipcc --offload-arch=gfx950 -O3 -o valu_ilp valu_ilp.hip && ./valu_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define DEV __device__ __forceinline__
DEV u32 pack2(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(u32, __builtin_convertvector(v, bf16x2)); }
DEV float bf_lo(u32 w) { return __uint_as_float(w << 16); }
DEV float bf_hi(u32 w) { return __uint_as_float(w & 0xFFFF0000u); }
DEV void sincos_precise(float th, float& s, float& c) {
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
// W angles in lockstep: every operation on all W angles before the next operation
template <int W>
DEV void sincos_lockstep(const float (&th)[W], float (&s)[W], float (&c)[W]) {
    float k[W], r[W], r2[W], sp[W], cp[W];
#pragma unroll
    for (int i = 0; i < W; ++i) k[i] = rintf(th[i] * 0.63661977236758134f);
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = fmaf(k[i], -1.5707963705062866f, th[i]);
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = fmaf(k[i], 4.371138828673793e-08f, r[i]);
#pragma unroll
    for (int i = 0; i < W; ++i) r2[i] = r[i] * r[i];
#pragma unroll
    for (int i = 0; i < W; ++i) sp[i] = fmaf(r2[i], 2.7183114939898219064e-6f, -1.9839334836096632576e-4f);
#pragma unroll
    for (int i = 0; i < W; ++i) cp[i] = fmaf(r2[i], 2.4433157826443582e-5f, -1.3887316255057415e-3f);
#pragma unroll
    for (int i = 0; i < W; ++i) sp[i] = fmaf(sp[i], r2[i], 8.3333293858894631756e-3f);
#pragma unroll
    for (int i = 0; i < W; ++i) cp[i] = fmaf(cp[i], r2[i], 4.1666645683529456e-2f);
#pragma unroll
    for (int i = 0; i < W; ++i) sp[i] = fmaf(sp[i], r2[i], -1.6666666641626524100e-1f);
#pragma unroll
    for (int i = 0; i < W; ++i) cp[i] = fmaf(cp[i], r2[i], -0.5f);
#pragma unroll
    for (int i = 0; i < W; ++i) sp[i] = fmaf(sp[i] * r2[i], r[i], r[i]);
#pragma unroll
    for (int i = 0; i < W; ++i) cp[i] = fmaf(cp[i], r2[i], 1.0f);
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int q = ((int)k[i]) & 3;
        const float ss = (q & 1) ? cp[i] : sp[i];
        const float cc = (q & 1) ? sp[i] : cp[i];
        s[i] = (q & 2) ? -ss : ss;
        c[i] = ((q + 1) & 2) ? -cc : cc;
    }
}
// MODE 0: serial (one angle after the other, the kernels' form); MODE W > 0: lockstep over W angles; PIN: an empty asm between the chains keeps hipcc
// from interleaving the serial form by itself (what register pressure does to it inside the real kernel)
template <int MODE, bool PIN>
__global__ __launch_bounds__(256) void k(const float* in, u32* out, unsigned long long* cyc, int reps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float x = in[t];
    u32 acc_hi = 0, acc_lo = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float s, c;
                sincos_precise(x * (1.0f + 0.37f * i), s, c);
                const u32 hi = pack2(s, c);
                acc_hi ^= hi;
                acc_lo ^= pack2(s - bf_lo(hi), c - bf_hi(hi));
                if constexpr (PIN) asm volatile("" : "+v"(acc_hi), "+v"(acc_lo));
            }
        } else {
#pragma unroll
            for (int b = 0; b < 8 / MODE; ++b) {
                float th[MODE], s[MODE], c[MODE];
#pragma unroll
                for (int i = 0; i < MODE; ++i) th[i] = x * (1.0f + 0.37f * (b * MODE + i));
                sincos_lockstep<MODE>(th, s, c);
                u32 hi[MODE];
#pragma unroll
                for (int i = 0; i < MODE; ++i) hi[i] = pack2(s[i], c[i]);
#pragma unroll
                for (int i = 0; i < MODE; ++i) { s[i] -= bf_lo(hi[i]); c[i] -= bf_hi(hi[i]); }
#pragma unroll
                for (int i = 0; i < MODE; ++i) { acc_hi ^= hi[i]; acc_lo ^= pack2(s[i], c[i]); }
                if constexpr (PIN) asm volatile("" : "+v"(acc_hi), "+v"(acc_lo));
            }
        }
        x += 0.001f;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[t] = acc_hi + acc_lo;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, bool PIN>
void run(const char* name, const float* in, u32* out, unsigned long long* cyc, int blocks) {
    const int reps = 200;
    k<MODE, PIN><<<blocks, 256>>>(in, out, cyc, reps);
    hipDeviceSynchronize();
    k<MODE, PIN><<<blocks, 256>>>(in, out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= blocks;
    printf("%-44s %7.1f cycles per (sincos + hi/lo split), one wave per SIMD\n", name, m / (reps * 8.0));
}
int main() {
    const int blocks = 256;
    float* in; u32* out; unsigned long long* cyc;
    hipMalloc(&in, blocks * 256 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    std::vector<float> h(blocks * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)(i % 9000);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, true>("serial, chains pinned apart", in, out, cyc, blocks);
    run<0, false>("serial, hipcc free to interleave", in, out, cyc, blocks);
    run<2, true>("lockstep 2", in, out, cyc, blocks);
    run<4, true>("lockstep 4", in, out, cyc, blocks);
    run<8, true>("lockstep 8", in, out, cyc, blocks);
    return 0;
}
