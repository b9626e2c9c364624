// BASELINE configs[4] experiment: the encoder's forward GEMMs (q / k / v / out projections, the two 1x1 convolutions; reference
// model/attn.py:177-196, model/transformer_net.py:36-42) on the fp8 matrix cores of gfx950 (OCP e4m3, v_mfma_f32_32x32x16_fp8_fp8, fp32
// accumulate), switched on by DPN_ENCODER_FP8=1 and OFF in the product: measured 2-3 % error on the encoder output -- the hyper-network
// turns that output into the point MLPs' weights, and the six PDE losses move by tens of percent (profiles/round2_fp8_encoder_experiment.json).
//
//   C[M][N] = epilogue( sum_k A[m][k] W[n][k] + bias[n] )           A: [M][K] row-major, W: [N][K] row-major (nn.Linear / Conv1d k=1 layout)
//
// Scaling: every row of A and every row of W gets its own power-of-two-free scale amax / 448 (448 = largest e4m3 value), computed in
// the kernel from the fp32 operands (a first pass over the 32 x K tile, which stays in L1 / L2 for the second), so the product of a row
// pair is exact up to the two 3-bit-mantissa roundings: C = s_a[m] s_w[n] sum_k q(a / s_a) q(w / s_w).
// One workgroup = four waves = 32 rows x 128 columns; a wave owns one 32 x 32 tile, walks K in steps of 16 (lane (i, h) supplies the
// eight consecutive k = 16 ks + 8 h .. + 7 of row / column i, as for the bf16 shape).  Not tuned: the 287-row problems of one field are
// launch-latency-bound whatever computes them, the 17 507-row problems of the 61-lead batch are what the timing in the experiment is about.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dpn_hip.h"
#ifdef DPN_EXPERIMENTS
#include "../../include/dpn_hip_experiments.h"
#endif

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define DEV __device__ __forceinline__

namespace {

DEV float gelu_exact8(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

struct Fp8Args {
    const float *A, *W, *bias;
    float *C, *aux_out;
    int M, N, K, lda, ldw, ldc, epi;
};

// amax of the lane's row over K (lanes i and i + 32 share a row: one shuffle joins the halves)
DEV float row_amax(const float* row, int K, int h, bool valid) {
    float m = 0.f;
    if (valid) {
        for (int k0 = 8 * h; k0 < K; k0 += 16) {
            const float4 a = *reinterpret_cast<const float4*>(row + k0), b = *reinterpret_cast<const float4*>(row + k0 + 4);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
        }
    }
    return fmaxf(m, __shfl_xor(m, 32));
}
DEV long quant8(const float* p, float inv_scale, bool valid) {      // eight consecutive values -> eight e4m3 in one 64-bit operand
    if (!valid) return 0;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a.x * inv_scale, a.y * inv_scale, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a.z * inv_scale, a.w * inv_scale, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(b.x * inv_scale, b.y * inv_scale, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(b.z * inv_scale, b.w * inv_scale, hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

#ifdef DPN_EXPERIMENTS
__global__ __launch_bounds__(256) void dpn_gemm_fp8_kernel(Fp8Args a) {
    __shared__ float s_row[32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = (blockIdx.x * 4 + wave) * 32;
    const bool row_ok = m0 + i < a.M, col_ok = n0 + i < a.N;
    const float* arow = a.A + (int64_t)(m0 + (row_ok ? i : 0)) * a.lda;
    const float* wrow = a.W + (int64_t)(n0 + (col_ok ? i : 0)) * a.ldw;
    const float sa = fmaxf(row_amax(arow, a.K, h, row_ok), 1e-30f) * (1.0f / 448.0f);
    const float sw = fmaxf(row_amax(wrow, a.K, h, col_ok), 1e-30f) * (1.0f / 448.0f);
    if (wave == 0 && h == 0) s_row[i] = sa;
    __syncthreads();
    if (n0 >= a.N) return;
    const float ia = 1.0f / sa, iw = 1.0f / sw;
    f32x16 acc = {};
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        const long qa = quant8(arow + k0 + 8 * h, ia, row_ok);
        const long qw = quant8(wrow + k0 + 8 * h, iw, col_ok);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(qa, qw, acc, 0, 0, 0);
    }
    // D: lane (j = i, h) holds column n0 + j, rows (r & 3) + 8 (r >> 2) + 4 h
    const float b = (a.bias && col_ok) ? a.bias[n0 + i] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m0 + m < a.M && col_ok) {
            float v = acc[r] * s_row[m] * sw + b;
            const int64_t idx = (int64_t)(m0 + m) * a.ldc + n0 + i;
            if (a.epi == DPN_EPI_GELU) { if (a.aux_out) a.aux_out[idx] = v; v = gelu_exact8(v); }
            a.C[idx] = v;
        }
    }
}
#endif  // DPN_EXPERIMENTS (the non-scaled form)


// ---- the block-scaled (MX) form: v_mfma_scale_f32_32x32x64_f8f6f4, the only large-K fp8 MFMA of gfx950 (twice the rate of the instruction
// above).  Operand layout, found by experiment (tools/microbench/mx_layout_probe.hip): of the 64 k of an instruction, lane (i, h) holds
// k = 16 h .. 16 h + 15 in its bytes 0-15 and k = 32 + 16 h .. + 15 in its bytes 16-31 (row / column i); the E8M0 scale byte of lane (i, 0)
// applies to the block k = 0..31 (bytes 0-15 of BOTH lane halves), that of lane (i, 1) to k = 32..63 (bytes 16-31 of both).  Scales: one
// power of two per 32 consecutive k of a row (the OCP MX definition), exponent = ceil(log2(amax_block / 448)), so the block's largest value
// lands in the top binade of e4m3.  The hardware applies 2^(ea + ew) to the block's partial sum: no scaling epilogue.
typedef __attribute__((ext_vector_type(8))) int i32x8;
struct Blk { i32x8 q; int e; };
DEV int mx_exponent(float m) {                       // smallest e with m / 2^e <= 448 (an all-zero block takes 2^-127)
    if (!(m > 0.f)) return -127;
    int ex;
    const float fr = frexpf(m * (1.0f / 448.0f), &ex);              // m / 448 = fr * 2^ex, fr in [0.5, 1)
    int e = (fr == 0.5f) ? ex - 1 : ex;
    return e < -127 ? -127 : (e > 127 ? 127 : e);
}
// p = the row's 64 values of this instruction; lane half h
DEV Blk quant64_mx(const float* p, int h, bool valid) {
    Blk b;
    float v[2][16], m[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = valid ? *reinterpret_cast<const float4*>(p + 32 * g + 16 * h + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[g][4 * q] = x.x; v[g][4 * q + 1] = x.y; v[g][4 * q + 2] = x.z; v[g][4 * q + 3] = x.w;
            m[g] = fmaxf(m[g], fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
        }
    int e[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) e[g] = mx_exponent(fmaxf(m[g], __shfl_xor(m[g], 32)));        // a block's 32 values sit in both lane halves
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float inv = ldexpf(1.0f, -e[g]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][4 * q] * inv, v[g][4 * q + 1] * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][4 * q + 2] * inv, v[g][4 * q + 3] * inv, w, true);
            b.q[4 * g + q] = w;
        }
    }
    b.e = (h ? e[1] : e[0]) + 127;
    return b;
}

__global__ __launch_bounds__(256) void dpn_gemm_fp8_mx_kernel(Fp8Args a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = (blockIdx.x * 4 + wave) * 32;
    if (n0 >= a.N) return;
    const bool row_ok = m0 + i < a.M, col_ok = n0 + i < a.N;
    const float* arow = a.A + (int64_t)(m0 + (row_ok ? i : 0)) * a.lda;
    const float* wrow = a.W + (int64_t)(n0 + (col_ok ? i : 0)) * a.ldw;
    f32x16 acc = {};
    for (int k0 = 0; k0 < a.K; k0 += 64) {
        const Blk qa = quant64_mx(arow + k0, h, row_ok);
        const Blk qw = quant64_mx(wrow + k0, h, col_ok);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa.q, qw.q, acc, 0, 0, 0, qa.e, 0, qw.e);      // formats: 0 = e4m3 for both
    }
    const float b = (a.bias && col_ok) ? a.bias[n0 + i] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m0 + m < a.M && col_ok) {
            float v = acc[r] + b;
            const int64_t idx = (int64_t)(m0 + m) * a.ldc + n0 + i;
            if (a.epi == DPN_EPI_GELU) { if (a.aux_out) a.aux_out[idx] = v; v = gelu_exact8(v); }
            a.C[idx] = v;
        }
    }
}

}  // namespace

extern "C" int dpn_gemm_fp8_mx(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int epi,
                               float* aux_out, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 3) || (ldw & 3) || (epi != DPN_EPI_NONE && epi != DPN_EPI_GELU)) return -1;
    Fp8Args a{A, W, bias, C, aux_out, M, N, K, lda, ldw, ldc, epi};
    hipLaunchKernelGGL(dpn_gemm_fp8_mx_kernel, dim3((N + 127) / 128, (M + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

#ifdef DPN_EXPERIMENTS                       // the non-scaled fp8 form (per-row scales): shelved, experiment library only
extern "C" int dpn_gemm_fp8(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int epi,
                            float* aux_out, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || (K & 15) || (lda & 3) || (ldw & 3) || (epi != DPN_EPI_NONE && epi != DPN_EPI_GELU)) return -1;
    Fp8Args a{A, W, bias, C, aux_out, M, N, K, lda, ldw, ldc, epi};
    hipLaunchKernelGGL(dpn_gemm_fp8_kernel, dim3((N + 127) / 128, (M + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
#endif
