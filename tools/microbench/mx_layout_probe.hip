// Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x e4m3), found by experiment: which (lane half, byte) of A meets which of B,
// and which bytes a lane's E8M0 scale applies to.  hipcc --offload-arch=gfx950 -O2 -o mx_layout_probe tools/microbench/mx_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// a, b: [64 lanes][32 bytes]; sa, sb: [64] scale bytes; out: [64][16]
__global__ void probe(const unsigned char* a, const unsigned char* b, const int* sa, const int* sb, float* out) {
    const int l = threadIdx.x;
    i32x8 va, vb;
    for (int q = 0; q < 8; ++q) { va[q] = reinterpret_cast<const int*>(a + l * 32)[q]; vb[q] = reinterpret_cast<const int*>(b + l * 32)[q]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}

int main() {
    unsigned char *da, *db; int *dsa, *dsb; float* dout;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 4096);
    std::vector<unsigned char> a(2048), b(2048); std::vector<int> sa(64, 127), sb(64, 127); std::vector<float> out(1024);
    auto run = [&]() {
        hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
        hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
    };
    const unsigned char ONE = 0x38;      // 1.0 in e4m3 (bias 7: exponent field 7 << 3)
    // test 1: which (hA, jA) meets which (hB, jB): row 0 of A, column 0 of B
    printf("test 1: for A byte (h, j) the B bytes (h, j) it is multiplied with\n");
    int identity = 0, total = 0;
    for (int hA = 0; hA < 2; ++hA) for (int jA = 0; jA < 32; ++jA) {
        std::fill(a.begin(), a.end(), 0); a[(hA * 32 + 0) * 32 + jA] = ONE;           // lane (i = 0, hA)
        for (int hB = 0; hB < 2; ++hB) for (int jB = 0; jB < 32; ++jB) {
            std::fill(b.begin(), b.end(), 0); b[(hB * 32 + 0) * 32 + jB] = ONE;       // lane (n = 0, hB)
            run();
            if (out[0] != 0.f) { ++total; if (hA == hB && jA == jB) ++identity; else printf("  A(%d,%2d) x B(%d,%2d) = %g\n", hA, jA, hB, jB, out[0]); }
        }
    }
    printf("  %d pairs meet, %d of them the identity pairs\n", total, identity);
    // test 2: whose scale applies to which bytes.  A = B = ones everywhere; raise the A scale of lane (0, h*) by 2^3 and see which A bytes
    // of row 0 got scaled, byte by byte (B one-hot at (hB, jB))
    for (int hs = 0; hs < 2; ++hs) {
        std::fill(a.begin(), a.end(), ONE);
        std::fill(sa.begin(), sa.end(), 127); sa[hs * 32 + 0] = 130;
        printf("test 2: A scale of lane (row 0, h = %d) raised by 2^3: bytes of row 0 that come out 8x:", hs);
        for (int hB = 0; hB < 2; ++hB) for (int jB = 0; jB < 32; ++jB) {
            std::fill(b.begin(), b.end(), 0); b[(hB * 32 + 0) * 32 + jB] = ONE;
            run();
            if (out[0] == 8.f) printf(" (%d,%d)", hB, jB); else if (out[0] != 1.f) printf(" (%d,%d)=%g?", hB, jB, out[0]);
        }
        printf("\n");
    }
    // test 3: opsel / byte position of the scale: put 130 in byte 1 with opsel 0 -> no effect expected
    std::fill(sa.begin(), sa.end(), 127 | (130 << 8)); std::fill(a.begin(), a.end(), ONE); std::fill(b.begin(), b.end(), ONE);
    run(); printf("test 3: all ones, scale byte0 = 127, byte1 = 130, opsel 0: out[0] = %g (64 expected)\n", out[0]);
    std::fill(sa.begin(), sa.end(), 127); sa[0] = 128; run(); printf("        lane (0,0) scale 128: out[0] = %g (96 if it covers 32 k)\n", out[0]);
    return 0;
}
