/* Entry points of SHELVED EXPERIMENTS -- not part of the product library.
 *
 * libdpn_hip.so exports what include/dpn_hip.h declares: the path bench.py / train.py can reach.  The kernels below were built, measured and not adopted
 * (DESIGN.md sections 1, 4c; tools/experiments/README.md); they are compiled only with -DDPN_EXPERIMENTS, into deepphysinet_amd/libdpn_hip_exp.so
 * (`python -m deepphysinet_amd.build --experiments`; it also holds every product symbol, so one handle serves an experiment run), and the Python side
 * reaches them through deepphysinet_amd._lib.load_experiments() only when the matching frozen switch (deepphysinet_amd/config.py) is on. */
#ifndef DPN_HIP_EXPERIMENTS_H
#define DPN_HIP_EXPERIMENTS_H
#include "dpn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* EXPERIMENT (DPN_CONV16=1): the token convolution (model/embed.py:45-47) on f16 hi+lo MFMA with operands split once per step.
 * dpn_conv16_split: each im2col row (of x [batch*T][C], circular, as dpn_im2col_circ3) and each weight row [conv_n][3C] scaled by a power of two
 * (row maximum into [8, 16): biased exponents in xe / we) and written as f16 hi and lo MFMA-fragment images: per (16-row strip, 32-k block)
 * 2 KB = [hi | lo][lane = (k % 32) / 8 * 16 + row % 16][8 f16]; xs holds ceil(batch*T / 16) strips, ws ceil(conv_n / 16), each of Kp / 32 blocks,
 * Kp = dpn_conv16_kp(3C) (zero-filled behind 3C). */
int dpn_conv16_split(const float* x, int T, int C, int batch, const float* conv_w, int conv_n, void* xs, int32_t* xe, void* ws, int32_t* we, void* stream);
/* dpn_conv16, the GEMM on those images: parts[s][m][n] = sum over the s-th K-slice of x(m, k) w(n, k), fp32,
 * scales undone (M = batch*T rows, N = conv_n, `slices` K-slices of whole 32-k blocks, at most 16 blocks each; dpn_embed_assemble adds the
 * slices in order).  f16 hi+lo,
 * three products, fp32 accumulate (fp32-class); a row's scale depends on that row only, so a field's result does not depend on its batch. */
int64_t dpn_conv16_kp(int K);
int dpn_conv16(const void* xs, const int32_t* xe, const void* ws, const int32_t* we, int M, int N, int Kp, int slices, float* parts, void* stream);


/* The same kernel for any small GEMM, one launch for up to DPN_WGRAD_MAX_PROBLEMS of them:  C[m][n] = sum_k A(m, k) B(n, k) (+ bias[n]),
 * A(m, k) = A[m * a_sm + k * a_sk], B(n, k) = B[n * b_sn + k * b_sk] (element strides: "k runs over rows" is a_sm = 1, a_sk = ld; "k is
 * contiguous" is a_sm = ld, a_sk = 1); asum[m] = sum_k A(m, k) (optional).  slices > 1 cuts K; with reduce = 0 the caller adds the partial
 * results partials[problem][slice][M * N (+ M)] itself (the data embedding's assemble kernel does).  Used for the token convolution
 * (embed.py:45-47: x_unfolded W^T with K = 3 * 2405) instead of the exact-fp32 dpn_sgemm_batch. */
typedef struct DpnGemm16Problem {
    const float* A; const float* B; float* C; float* asum; const float* bias;
    int32_t M, N, K, ldc;
    int64_t a_sm, a_sk, b_sn, b_sk;
} DpnGemm16Problem;
int64_t dpn_gemm16_partial_floats(int n, const DpnGemm16Problem* problems, int slices);
int dpn_gemm16(int n, const DpnGemm16Problem* problems /* host array */, int slices, float* partials, int reduce, void* stream);


/* BASELINE configs[4] experiment (OFF in the product; DPN_ENCODER_FP8=1 routes the encoder layers' forward GEMMs here): C[M][N] =
 * epilogue(A[M][K] . W[N][K]^T + bias[N]) on the fp8 matrix cores (OCP e4m3 operands quantised in the kernel with one scale per row of A
 * and per row of W, fp32 accumulate); K a multiple of 16, lda / ldw multiples of 4; epi = DPN_EPI_NONE or DPN_EPI_GELU (aux_out
 * receives the pre-activation).  Replaces nothing of the reference by default: its measured parity error is why. */
int dpn_gemm_fp8(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int epi,
                 float* aux_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
