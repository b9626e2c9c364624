// Row-local fused nodes of the grid encoder ("MetaNet") for MI355X.
//
// Reference behaviour (paths relative to /root/reference/DeepPhysiNet):
//   model/attn.py:177-196            q/k/v and out projections of an AttentionLayer
//   model/transformer_net.py:28-44   x1 = norm1(x + attn_out);  out = norm2(x1 + conv2(gelu(conv1(x1))))   (1x1 convolutions = per-token linears)
//   model/transformer_net.py:54-72   encoder.norm after the last layer;  :129 the output projection
// Everything in an encoder layer except the attention itself is ROW-LOCAL: a token row never needs another row.  dpn_enc_fwd / dpn_enc_bwd
// therefore run the whole stretch between two attention kernels as ONE launch: a workgroup owns 16 (or 32) token rows and carries them through
//   forward :  out-projection + residual + LayerNorm1 + conv1 + GELU + conv2 + residual + LayerNorm2 + the NEXT layer's q/k/v projections
//              (or encoder.norm + the output projection behind the last layer)
//   backward:  d x = residual cotangent + dq Wq + dk Wk + dv Wv  (or the projection / encoder.norm backward), LayerNorm2 backward, conv2^T,
//              GELU', conv1^T + residual, LayerNorm1 backward, out-projection^T (the attention backward's input)
// with the row block in LDS / registers between the GEMMs; it writes exactly what the backward pass and the weight-gradient GEMMs need.
// (Round 3 ran this stretch as five launches forward and four backward per layer: 31.6 us / layer forward on 287 x 256 tensors.)
//
// Arithmetic: the 256 x 256 GEMMs run on v_mfma_f32_16x16x32_f16 with BOTH operands split in two f16 values,
//     x = hi + 2^-11 lo',   hi = f16(x),  lo' = f16(2^11 (x - hi))        (22 significant bits; round-to-nearest both times)
//     x.w ~ hi.hi + 2^-11 (hi.lo' + lo'.hi)                                (three products, fp32 accumulate, two accumulators)
// fp32-class: tools/precision_encoder_split.py -- encoder output 1e-6 of the fp64 run, like torch's own fp32 run (bf16 hi+lo: 1.2e-5;
// bf16 needs a three-way split = six products).  Activation rows are scaled by a power of two per token (row maximum into [8, 16)) before
// the split, so any magnitude (backward cotangents carry loss factors up to 1e14) stays inside f16's range; weights are split unscaled by
// dpn_enc_pack, which raises a status flag for |w| >= 32768 (a 256-wide linear layer with such weights has no fp32 meaning either).
// LayerNorm, GELU (exact erf), residual adds, biases: fp32 on the row block, the reference's formulas (two-pass variance, eps 1e-5).
//
// Weight stream: dpn_enc_pack writes every matrix as MFMA A-fragments in consumption order (per 16-channel tile and 32-wide k-step: hi
// plane, lo plane, 1 KB each), once for x W^T (forward) and once for g W (backward).  A workgroup streams the images of its six GEMMs
// (1.5 MB) from L2 straight into registers (buffer_load_dwordx4 with a scalar running offset), kPF k-steps ahead and ACROSS the row passes
// and barriers between the GEMMs (LDS-only barriers: vmcnt stays in flight).  That stream at the CU's ~55 B/clk L2 path is the kernel's
// bound (~12 us per launch); the MFMAs (2 304 per workgroup) take a third of it.
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dpn_hip.h"
#ifdef DPN_EXPERIMENTS                       // shelved kernels' entry points (dpn_conv16*, dpn_gemm16): the experiment library only
#include "../../include/dpn_hip_experiments.h"
#else
// (internal in the product build: the strided-GEMM description dpn_wgrad16's problems are expressed in; public only with the experiments header)
typedef struct DpnGemm16Problem {
    const float* A; const float* B; float* C; float* asum; const float* bias;
    int32_t M, N, K, ldc;
    int64_t a_sm, a_sk, b_sn, b_sk;
} DpnGemm16Problem;
#endif

#define DEV __device__ __forceinline__

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 256;
constexpr int kThreads = 512, kWaves = 8;
constexpr int kImgBytes = 256 * 1024;            // one packed 256 x 256 image: [tile 16][k-step 8][hi | lo][lane 64][16 B]
#ifndef DPN_ENC_PF
#define DPN_ENC_PF 8
#endif
// weight fragments in flight: k-steps ahead (4 KB per wave and k-step); 8 = a whole GEMM (16-row workgroups; the 32-row form has the registers for 4)
template <int NTT> constexpr int pf_of() { return NTT == 1 ? DPN_ENC_PF : 4; }
constexpr int kYS = 260;                         // row stride (floats) of the fp32 staging block
constexpr int kXImg = 16384;                     // X image of 16 tokens: [hi | lo][slot 32][position 16][16 B]
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

// Experiment build (-DDPN_ENC_TIMELINE, tools/enc_timeline.py): lane 0 of every wave writes the shader clock at the phase boundaries.
#ifdef DPN_ENC_TIMELINE
unsigned* g_enc_timeline = nullptr;
#define ENC_STAMP(I) do { if (a.tl && lane == 0) a.tl[((size_t)blockIdx.x * kWaves + wave) * 32 + (I)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#define ENC_TL_ARG unsigned* tl;
#define ENC_TL_SET(A) (A).tl = g_enc_timeline
#else
#define ENC_STAMP(I) do { } while (0)
#define ENC_TL_ARG
#define ENC_TL_SET(A) do { } while (0)
#endif

DEV void barrier_lds() {                         // LDS-only workgroup barrier: global loads (weight prefetch) and stores stay in flight
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- all-reduce over the 32 lanes of a half wave (a token row lives in one half: 32 lanes x 8 channels)
template <int CTRL>
DEV float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false)); }
DEV float swz16(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F)); }   // lane ^ 16
DEV float half_sum(float v) {                    // fixed order: the same tree for every row
    v += dpp<0xB1>(v);                           // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);                           // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);                          // row_half_mirror: the other quad of the 8
    v += dpp<0x140>(v);                          // row_mirror: the other 8 of the 16
    v += swz16(v);
    return v;
}
DEV float half_max(float v) {
    v = fmaxf(v, dpp<0xB1>(v));
    v = fmaxf(v, dpp<0x4E>(v));
    v = fmaxf(v, dpp<0x141>(v));
    v = fmaxf(v, dpp<0x140>(v));
    v = fmaxf(v, swz16(v));
    return v;
}

// exact-erf GELU and its derivative (torch's GeluCUDAKernelImpl / GeluBackwardCUDAKernelImpl, approximate='none'; transformer_net.py:26,41)
DEV float gelu_exact(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
DEV float gelu_exact_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
    return cdf + x * pdf;
}

// ---- split: v -> hi + 2^-11 lo' (two f16 planes of eight values = one 16-byte fragment slot each)
DEV void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 a = (_Float16)v[e];
        h[e] = a;
        l[e] = (_Float16)((v[e] - (float)a) * kLoScale);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// X image position of (slot = k / 8, token n): quad index slot * 16 + (n ^ F(slot & 7)), F = {0,1,2,3,12,13,14,15}.  The readers (B fragments:
// lane = (n, g), slot 4 ks + g) touch 16 distinct bank quads per ds_read_b128 lane group with any F that is constant per k-step up to an
// XOR preserving {0-3,12-15} / {4-11}; the writers (8 consecutive slots of ONE token per ds_write_b128 lane group) get 8 distinct quads.
DEV int xpos(int slot, int n) { return slot * 16 + (n ^ ((slot & 3) | ((slot & 4) ? 12 : 0))); }

// Row-pass lane mapping: wave w, lane l -> token 2 w + (l >> 5) of each 16-token tile, channels 8 (l & 31) .. + 7.
struct RowLane {
    int tsel, slot, n16;                         // n16: token inside a 16-token tile
    DEV RowLane(int wave, int lane) : tsel(lane >> 5), slot(lane & 31), n16(2 * wave + (lane >> 5)) {}
};

// scale the row so that its maximum lands in [8, 16), split it and store it as the B-fragment image of its tile; the inverse scale goes to rscale
// (exponent arithmetic on the bits: powers of two, exact)
DEV void row_to_ximg(const float (&v)[8], const float extra_max, const RowLane& rl, char* ximg_tile, float* rscale_tile) {
    float m = extra_max;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    m = half_max(m);
    int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
    eb = eb < 4 ? 4 : eb;                                                   // zero / tiny rows: any scale does
    const float sc = __builtin_bit_cast(float, (257 - eb) << 23);           // 2^(3 - (eb - 127))
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = v[e] * sc;
    u32x4 hi, lo;
    split8(s, hi, lo);
    char* p = ximg_tile + xpos(rl.slot, rl.n16) * 16;
    *reinterpret_cast<u32x4*>(p) = hi;
    *reinterpret_cast<u32x4*>(p + kXImg / 2) = lo;
    if (rl.slot == 0) rscale_tile[rl.n16] = __builtin_bit_cast(float, (eb - 3) << 23);   // 2^((eb - 127) - 3)
}
DEV float row_absmax(const float (&v)[8]) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    return m;
}

DEV void load8(float (&v)[8], const float* p, bool ok) {
    if (ok) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
}
DEV void store8(float* p, const float (&v)[8], bool ok) {
    if (ok) {
        reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}
DEV void lds8(float (&v)[8], const float* p) {
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// LayerNorm over the 256 channels of the lane's row (nn.LayerNorm: biased variance, eps 1e-5, two passes)
DEV void layer_norm8(const float (&v)[8], const float (&gam)[8], const float (&bet)[8], float (&xhat)[8], float (&y)[8], float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    const float mean = half_sum(s) * (1.f / kD);
    float d[8], q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { d[e] = v[e] - mean; q = fmaf(d[e], d[e], q); }
    rstd = 1.0f / sqrtf(half_sum(q) * (1.f / kD) + 1e-5f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { xhat[e] = d[e] * rstd; y[e] = fmaf(xhat[e], gam[e], bet[e]); }
}
// its input gradient: gx = rstd (g gamma - mean(g gamma) - xhat mean(g gamma xhat)); the parameter sums g xhat, g accumulate into pg / pb
DEV void layer_norm_bwd8(const float (&g)[8], const float (&xhat)[8], const float rstd, const float (&gam)[8], float (&gx)[8], float (&pg)[8],
                         float (&pb)[8]) {
    float t[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { t[e] = g[e] * gam[e]; s1 += t[e]; s2 = fmaf(t[e], xhat[e], s2); pg[e] = fmaf(g[e], xhat[e], pg[e]); pb[e] += g[e]; }
    const float m1 = half_sum(s1) * (1.f / kD), m2 = half_sum(s2) * (1.f / kD);
#pragma unroll
    for (int e = 0; e < 8; ++e) gx[e] = rstd * (t[e] - m1 - xhat[e] * m2);
}

// ---- the weight stream of a wave: tiles 2 w, 2 w + 1 of every image, [k-step][hi | lo] 1 KB fragments
struct WStream {
    __amdgpu_buffer_rsrc_t rs;
    int voff, tile_off;
    DEV void init(const void* wpack, int64_t bytes, int wave, int lane) {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wpack), 0, (int)bytes, 0x00020000);
        voff = lane * 16;
        tile_off = wave * 2 * 16384;
    }
    DEV void load(u32x4 (&dst)[2][2], int img_off, int ks) const {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                dst[t][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + p * 1024, img_off + tile_off + t * 16384 + ks * 2048, 0));
    }
};
template <int PF> struct Ring { u32x4 a[PF][2][2]; };

DEV f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int NTT>
struct Acc {
    f32x4 m[2][NTT], c[2][NTT];                  // main (hi.hi) and cross (hi.lo' + lo'.hi) accumulators of the wave's two channel tiles
    DEV void zero() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) { m[t][tt] = (f32x4)0.f; c[t][tt] = (f32x4)0.f; }
    }
};

// acc += W[wave's 32 channels][256] . X^T[256][16 NTT tokens]: eight k-steps; the ring holds this image's k-steps 0 .. kPF-1 on entry and the next
// image's on exit (HAS_NEXT).  xr0 / xr1: the lane's read bases in the X image for even / odd k-steps.
template <int NTT, bool HAS_NEXT>
DEV void gemm(const WStream& ws, Ring<pf_of<NTT>()>& R, const int cur_off, const int next_off, const char* xr0, const char* xr1, Acc<NTT>& acc) {
    u32x4 B[2][NTT][2];
    auto loadB = [&](const int ks, const int buf) __attribute__((always_inline)) {
        const char* base = ((ks & 1) ? xr1 : xr0) + ks * 1024;
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            B[buf][tt][0] = *reinterpret_cast<const u32x4*>(base + tt * kXImg);
            B[buf][tt][1] = *reinterpret_cast<const u32x4*>(base + tt * kXImg + kXImg / 2);
        }
    };
    constexpr int kPF = pf_of<NTT>();
    loadB(0, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) loadB(ks + 1, (ks + 1) & 1);
        const int s = ks % kPF, b = ks & 1;
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
            for (int t = 0; t < 2; ++t) acc.m[t][tt] = mfma(R.a[s][t][0], B[b][tt][0], acc.m[t][tt]);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc.c[t][tt] = mfma(R.a[s][t][0], B[b][tt][1], acc.c[t][tt]);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc.c[t][tt] = mfma(R.a[s][t][1], B[b][tt][0], acc.c[t][tt]);
        }
        if (ks + kPF < 8) ws.load(R.a[s], cur_off, ks + kPF);
        else if (HAS_NEXT) ws.load(R.a[s], next_off, ks + kPF - 8);
    }
}

// accumulator element (tile t, token tile tt, register j) = channel 32 w + 16 t + 4 g + j of token 16 tt + (lane & 15), g = lane >> 4
template <int NTT>
DEV void acc_to_staging(const Acc<NTT>& acc, float* Y, const float* rscale, int wave, int lane) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const float rs = rscale[tt * 16 + n];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(acc.c[t][tt][j], kLoInv, acc.m[t][tt][j]) * rs;
            *reinterpret_cast<f32x4*>(Y + (tt * 16 + n) * kYS + wave * 32 + t * 16 + g * 4) = v;
        }
    }
}
// the same values (+ bias) straight to a [rows][256] tensor in global memory
template <int NTT>
DEV void acc_to_global(const Acc<NTT>& acc, float* out, const float* bias_lds, const float* rscale, int row0, int rows, int wave, int lane) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const float rs = rscale[tt * 16 + n];
        const int row = row0 + tt * 16 + n;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = wave * 32 + t * 16 + g * 4;
            f32x4 b = (f32x4)0.f;
            if (bias_lds) b = *reinterpret_cast<const f32x4*>(bias_lds + ch);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(acc.c[t][tt][j], kLoInv, acc.m[t][tt][j]) * rs + b[j];
            if (row < rows) *reinterpret_cast<f32x4*>(out + (int64_t)row * kD + ch) = v;
        }
    }
}

// ---- L2 warm-up helpers.  One field sample is 18 workgroups on 18 of the 256 CUs, spread over the 8 XCDs (block b runs on XCD b % 8:
// observed placement, used for speed only), and every XCD's L2 has to pull the 1.5 MB of weight images from the memory side before its
// workgroups can stream them: with cold L2s the launch measures 46 k cycles (forward) / 54 k (backward), with the images L2-resident
// 41 k / 44 k (profiles/round4_enc_timeline*.txt).  So the launch carries 64 extra workgroups on otherwise idle CUs -- eight per XCD, each
// touching one eighth of every image's 128-byte lines once -- which finish in the launch's first microseconds: the later GEMMs of the
// 18 working workgroups then hit L2.  Nothing depends on them (a missed line is simply fetched by the stream itself).
constexpr int kHelpers = 64;
DEV void l2_warm_helper(const void* wpack, const int (&img)[6], const int n_img, const int hb) {
    const int part = hb >> 3, t = threadIdx.x;                               // eight parts x (eight blocks: one per XCD)
    float sum = 0.f;
    for (int i = t >> 8; i < n_img; i += 2) {                                 // 256 lines per part and image: threads 0-255 / 256-511 alternate images
        const char* p = static_cast<const char*>(wpack) + img[i] + (part * 256 + (t & 255)) * 128;
        sum += *reinterpret_cast<const float*>(p);
    }
    asm volatile("" ::"v"(sum));
}

// ------------------------------------------------------------------------------------------------ forward
constexpr int kNVecF = 12;
enum { VF_BO = 0, VF_G1, VF_BE1, VF_BC1, VF_BC2, VF_G2, VF_BE2, VF_GF, VF_BEF, VF_BN0, VF_BN1, VF_BN2 };
struct FwdArgs {
    const void* wpack;
    int64_t wbytes;
    int img[6];                                  // byte offsets of the images in consumption order
    int rows, n_main, n_img;                     // n_main: working workgroups (blocks behind them are L2 warm-up helpers)
    const float *o, *x, *xin;
    const float* vec[kNVecF];
    float *x1, *xhat1, *rstd1, *pre, *act, *x2, *xhat2, *rstd2, *xf, *xhatf, *rstdf, *y0, *y1, *y2;
    const float *emb_parts, *emb_bias, *emb_pos, *emb_te, *emb_token;       // !TAIL: assemble the input rows here (DpnEncFwd.emb_*)
    float* emb_out;
    int64_t emb_part_stride;
    int emb_n_parts, emb_n_tok;
    ENC_TL_ARG
};
template <int NTT>
struct Lds {
    static constexpr int kX = 0;                                     // X images (up to three in the backward head)
    static constexpr int kY = 3 * NTT * kXImg;
    static constexpr int kVec = kY + NTT * 16 * kYS * 4;
    static constexpr int kRs = kVec + kNVecF * 256 * 4;
    static constexpr int kRed = kRs + NTT * 16 * 4;                  // [wave][512]: LayerNorm parameter partial sums (backward)
    static constexpr int kBytes = kRed + kWaves * 512 * 4;
};

constexpr int kEmbAhead = 9;                   // slices of the token convolution fetched at once by the launch that assembles x0 (6 / 9 / 18 measured alike: the 18
                                               // workgroups pull 311 KB each through their CU's L2 path, +3.9 us on this launch for the 6.9-us launch it replaces)
template <bool TAIL, int NEXT, int NTT>
__global__ __launch_bounds__(kThreads) void dpn_enc_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ximg = smem + Lds<NTT>::kX;
    float* Y = reinterpret_cast<float*>(smem + Lds<NTT>::kY);
    float* vecs = reinterpret_cast<float*>(smem + Lds<NTT>::kVec);
    float* rscale = reinterpret_cast<float*>(smem + Lds<NTT>::kRs);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= a.n_main) { l2_warm_helper(a.wpack, a.img, a.n_img, blockIdx.x - a.n_main); return; }
    const int row0 = blockIdx.x * 16 * NTT;
    ENC_STAMP(0);
    const RowLane rl(wave, lane);
    const int n_r = lane & 15, g_r = lane >> 4;
    const char* xr0 = ximg + (g_r * 16 + (n_r ^ g_r)) * 16;
    const char* xr1 = ximg + (g_r * 16 + (n_r ^ g_r ^ 12)) * 16;
    // order of the first loads: the input rows (the first GEMM waits for them), the parameter vectors, THEN the weight stream -- vector memory
    // returns in order, a row load issued behind the ring's 32 KB would wait for all of it
    float res[NTT][8], vin[NTT][8];
    bool ok[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int row = row0 + tt * 16 + rl.n16;
        ok[tt] = row < a.rows;
        if constexpr (!TAIL) {
            if (a.emb_parts) {
                // x0 = cat(learnable_token, value_embedding) + positional table + lead-time embedding, assembled HERE from the token convolution's split-K
                // slices (it was a launch of its own, dpn_embed_assemble: 6.9 us for 0.3 MB of output); same order of additions, every load in flight at once
                const int c0 = rl.slot * 8;
                const bool tok = row < a.emb_n_tok;
                const int64_t po = ((int64_t)(tok ? 0 : row - a.emb_n_tok)) * kD + c0;
                float v[8];
                if (tok) load8(v, a.emb_token + (int64_t)row * kD + c0, ok[tt]);
                else load8(v, a.emb_parts + po, ok[tt]);
                if (!tok)
                    for (int p0 = 1; p0 < a.emb_n_parts; p0 += kEmbAhead) {                  // kEmbAhead slices' loads in flight, added in slice order
                        float t6[kEmbAhead][8];
#pragma unroll
                        for (int u = 0; u < kEmbAhead; ++u) load8(t6[u], a.emb_parts + (int64_t)min(p0 + u, a.emb_n_parts - 1) * a.emb_part_stride + po, ok[tt]);
#pragma unroll
                        for (int u = 0; u < kEmbAhead; ++u)
                            if (p0 + u < a.emb_n_parts) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += t6[u][e];
                            }
                    }
                float b8[8], p8[8], e8[8];
                load8(b8, a.emb_bias + c0, !tok && a.emb_bias != nullptr);
                load8(p8, a.emb_pos + (int64_t)row * kD + c0, ok[tt]);
                load8(e8, a.emb_te + c0, true);
#pragma unroll
                for (int e = 0; e < 8; ++e) vin[tt][e] = ok[tt] ? ((tok ? v[e] : v[e] + b8[e]) + p8[e]) + e8[e] : 0.f;
                store8(a.emb_out + (int64_t)row * kD + c0, vin[tt], ok[tt]);
                continue;
            }
        }
        load8(vin[tt], (TAIL ? a.o : a.xin) + (int64_t)row * kD + rl.slot * 8, ok[tt]);
        if constexpr (TAIL) load8(res[tt], a.x + (int64_t)row * kD + rl.slot * 8, ok[tt]);
    }
    constexpr int kVecLoads = (kNVecF * 64 + kThreads - 1) / kThreads;
    float4 pv[kVecLoads];
#pragma unroll
    for (int q = 0; q < kVecLoads; ++q) {
        const int i = tid + q * kThreads;
        const float* src = i < kNVecF * 64 ? a.vec[i >> 6] : nullptr;
        pv[q] = src ? reinterpret_cast<const float4*>(src)[i & 63] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    WStream ws;
    ws.init(a.wpack, a.wbytes, wave, lane);
    constexpr int kPF = pf_of<NTT>();
    Ring<kPF> R;
#pragma unroll
    for (int ks = 0; ks < kPF; ++ks) ws.load(R.a[ks], a.img[0], ks);
#pragma unroll
    for (int q = 0; q < kVecLoads; ++q) {
        const int i = tid + q * kThreads;
        if (i < kNVecF * 64) reinterpret_cast<float4*>(vecs)[i] = pv[q];
    }
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) row_to_ximg(vin[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
    ENC_STAMP(1);
    barrier_lds();
    ENC_STAMP(2);
    Acc<NTT> acc;
    constexpr int kFirstNext = TAIL ? 3 : 0;
    if constexpr (TAIL) {
        // ---- out-projection, residual, LayerNorm1 (attn.py:196, transformer_net.py:33-37)
        acc.zero();
        gemm<NTT, true>(ws, R, a.img[0], a.img[1], xr0, xr1, acc);
        ENC_STAMP(3);
        acc_to_staging<NTT>(acc, Y, rscale, wave, lane);
        barrier_lds();
        ENC_STAMP(4);
        // (every row pass: arithmetic and the X image first, barrier, and only then the saved-state stores -- nobody on the chain waits for them)
        float sa[NTT][8], sb[NTT][8], sr[NTT];
        {
            float gam[8], bet[8], bo[8];
            lds8(bo, vecs + VF_BO * 256 + rl.slot * 8); lds8(gam, vecs + VF_G1 * 256 + rl.slot * 8); lds8(bet, vecs + VF_BE1 * 256 + rl.slot * 8);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                float v[8];
                lds8(v, Y + (tt * 16 + rl.n16) * kYS + rl.slot * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = res[tt][e] + (v[e] + bo[e]);
                layer_norm8(v, gam, bet, sb[tt], res[tt], sr[tt]);           // res <- x1: the residual of the feed-forward block
                row_to_ximg(res[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
            }
        }
        ENC_STAMP(5);
        barrier_lds();
        ENC_STAMP(6);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int64_t off = (int64_t)(row0 + tt * 16 + rl.n16) * kD + rl.slot * 8;
            store8(a.x1 + off, res[tt], ok[tt]);
            store8(a.xhat1 + off, sb[tt], ok[tt]);
            if (ok[tt] && rl.slot == 0) a.rstd1[row0 + tt * 16 + rl.n16] = sr[tt];
        }
        // ---- conv1 + GELU (transformer_net.py:38-41)
        acc.zero();
        gemm<NTT, true>(ws, R, a.img[1], a.img[2], xr0, xr1, acc);
        ENC_STAMP(7);
        acc_to_staging<NTT>(acc, Y, rscale, wave, lane);
        barrier_lds();
        ENC_STAMP(8);
        {
            float b1[8];
            lds8(b1, vecs + VF_BC1 * 256 + rl.slot * 8);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                lds8(sa[tt], Y + (tt * 16 + rl.n16) * kYS + rl.slot * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { sa[tt][e] += b1[e]; sb[tt][e] = gelu_exact(sa[tt][e]); }
                row_to_ximg(sb[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
            }
        }
        ENC_STAMP(9);
        barrier_lds();
        ENC_STAMP(10);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int64_t off = (int64_t)(row0 + tt * 16 + rl.n16) * kD + rl.slot * 8;
            store8(a.pre + off, sa[tt], ok[tt]);
            store8(a.act + off, sb[tt], ok[tt]);
        }
        // ---- conv2, residual, LayerNorm2 (transformer_net.py:42-44) (+ encoder.norm behind the last layer, :68)
        acc.zero();
        gemm<NTT, NEXT != 0>(ws, R, a.img[2], a.img[3], xr0, xr1, acc);
        ENC_STAMP(11);
        acc_to_staging<NTT>(acc, Y, rscale, wave, lane);
        barrier_lds();
        ENC_STAMP(12);
        float sc[NEXT == 2 ? NTT : 1][8], sd[NEXT == 2 ? NTT : 1][8], sq[NEXT == 2 ? NTT : 1];
        {
            float gam[8], bet[8], b2[8];
            lds8(b2, vecs + VF_BC2 * 256 + rl.slot * 8); lds8(gam, vecs + VF_G2 * 256 + rl.slot * 8); lds8(bet, vecs + VF_BE2 * 256 + rl.slot * 8);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                float v[8];
                lds8(v, Y + (tt * 16 + rl.n16) * kYS + rl.slot * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = res[tt][e] + (v[e] + b2[e]);
                layer_norm8(v, gam, bet, sb[tt], sa[tt], sr[tt]);            // sa <- x2, the layer output
                if constexpr (NEXT == 2) {
                    float gf[8], bf[8];
                    lds8(gf, vecs + VF_GF * 256 + rl.slot * 8); lds8(bf, vecs + VF_BEF * 256 + rl.slot * 8);
                    layer_norm8(sa[tt], gf, bf, sd[tt], sc[tt], sq[tt]);
                    row_to_ximg(sc[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
                } else if constexpr (NEXT == 1) {
                    row_to_ximg(sa[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
                }
            }
        }
        ENC_STAMP(13);
        if constexpr (NEXT != 0) barrier_lds();
        ENC_STAMP(14);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int64_t off = (int64_t)(row0 + tt * 16 + rl.n16) * kD + rl.slot * 8;
            store8(a.x2 + off, sa[tt], ok[tt]);
            store8(a.xhat2 + off, sb[tt], ok[tt]);
            if (ok[tt] && rl.slot == 0) a.rstd2[row0 + tt * 16 + rl.n16] = sr[tt];
            if constexpr (NEXT == 2) {
                store8(a.xf + off, sc[tt], ok[tt]);
                store8(a.xhatf + off, sd[tt], ok[tt]);
                if (ok[tt] && rl.slot == 0) a.rstdf[row0 + tt * 16 + rl.n16] = sq[tt];
            }
        }
    }
    if constexpr (NEXT == 1) {
        // ---- the next layer's q / k / v projections (attn.py:183-185): three GEMMs on the same row block, straight to global memory
        acc.zero();
        gemm<NTT, true>(ws, R, a.img[kFirstNext], a.img[kFirstNext + 1], xr0, xr1, acc);
        acc_to_global<NTT>(acc, a.y0, vecs + VF_BN0 * 256, rscale, row0, a.rows, wave, lane);
        ENC_STAMP(15);
        acc.zero();
        gemm<NTT, true>(ws, R, a.img[kFirstNext + 1], a.img[kFirstNext + 2], xr0, xr1, acc);
        acc_to_global<NTT>(acc, a.y1, vecs + VF_BN1 * 256, rscale, row0, a.rows, wave, lane);
        ENC_STAMP(16);
        acc.zero();
        gemm<NTT, false>(ws, R, a.img[kFirstNext + 2], 0, xr0, xr1, acc);
        acc_to_global<NTT>(acc, a.y2, vecs + VF_BN2 * 256, rscale, row0, a.rows, wave, lane);
        ENC_STAMP(17);
    } else if constexpr (NEXT == 2) {
        // ---- the output projection (transformer_net.py:129)
        acc.zero();
        gemm<NTT, false>(ws, R, a.img[kFirstNext], 0, xr0, xr1, acc);
        acc_to_global<NTT>(acc, a.y0, vecs + VF_BN0 * 256, rscale, row0, a.rows, wave, lane);
    }
}

// ------------------------------------------------------------------------------------------------ backward
enum { VB_G2 = 0, VB_G1, VB_GF };
struct BwdArgs {
    const void* wpack;
    int64_t wbytes;
    int img[6];
    int rows, n_main, n_img;
    const float *res, *dq, *dk, *dv;             // HEAD 1
    const float *dmeta, *xhatf, *rstdf;          // HEAD 2
    const float* gin;                            // HEAD 0
    const float *xhat2, *rstd2, *pre, *xhat1, *rstd1;
    const float* vec[3];
    float *gs2, *dpre, *gs1, *dout, *gx;
    float *partial_f, *partial2, *partial1;      // [workgroups][512]: sums over the workgroup's rows of g xhat | g
    float* gx_head;                              // BODY 0: the first gx_head_rows rows of gx once more (the learnable tokens' gradient slot)
    int gx_head_rows;
    ENC_TL_ARG
};

// the wave's LayerNorm parameter partials (lanes 0-31 after joining the two token halves) -> red[wave][512]
DEV void partials_to_lds(float (&pg)[8], float (&pb)[8], float* red, int wave, int lane) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[e] += __shfl_xor(pg[e], 32); pb[e] += __shfl_xor(pb[e], 32); }
    if (lane < 32) {
        float* r = red + wave * 512 + lane * 8;
        reinterpret_cast<f32x4*>(r)[0] = (f32x4){pg[0], pg[1], pg[2], pg[3]};
        reinterpret_cast<f32x4*>(r)[1] = (f32x4){pg[4], pg[5], pg[6], pg[7]};
        reinterpret_cast<f32x4*>(r + 256)[0] = (f32x4){pb[0], pb[1], pb[2], pb[3]};
        reinterpret_cast<f32x4*>(r + 256)[1] = (f32x4){pb[4], pb[5], pb[6], pb[7]};
    }
}
DEV void partials_to_global(const float* red, float* partial, int tid) {        // fixed order over the eight waves
    float s = red[tid];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[w * 512 + tid];
    partial[(int64_t)blockIdx.x * 512 + tid] = s;
}

template <int HEAD, bool BODY, int NTT>
__global__ __launch_bounds__(kThreads) void dpn_enc_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ximg = smem + Lds<NTT>::kX;
    float* Y = reinterpret_cast<float*>(smem + Lds<NTT>::kY);
    float* vecs = reinterpret_cast<float*>(smem + Lds<NTT>::kVec);
    float* rscale = reinterpret_cast<float*>(smem + Lds<NTT>::kRs);
    float* red = reinterpret_cast<float*>(smem + Lds<NTT>::kRed);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= a.n_main) { l2_warm_helper(a.wpack, a.img, a.n_img, blockIdx.x - a.n_main); return; }
    const int row0 = blockIdx.x * 16 * NTT;
    ENC_STAMP(0);
    const RowLane rl(wave, lane);
    const int n_r = lane & 15, g_r = lane >> 4;
    const char* xr0 = ximg + (g_r * 16 + (n_r ^ g_r)) * 16;
    const char* xr1 = ximg + (g_r * 16 + (n_r ^ g_r ^ 12)) * 16;
    bool ok[NTT];
    int64_t off[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int row = row0 + tt * 16 + rl.n16;
        ok[tt] = row < a.rows;
        off[tt] = (int64_t)row * kD + rl.slot * 8;
    }
    // Order of the first loads (vector memory returns in order): the head's input rows, the parameter vectors, the weight stream's first
    // k-steps, and last the saved rows the later row passes need -- in flight from the start, each would otherwise be a cold round trip on
    // the chain, but nothing before the first GEMM waits for them.
    float hq[HEAD == 1 ? NTT : 1][8], hk[HEAD == 1 ? NTT : 1][8], hv[NTT][8], g[NTT][8];
    float xhf[HEAD == 2 ? NTT : 1][8], rsf[HEAD == 2 ? NTT : 1];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        if constexpr (HEAD == 1) {
            load8(hq[tt], a.dq + off[tt], ok[tt]); load8(hk[tt], a.dk + off[tt], ok[tt]); load8(hv[tt], a.dv + off[tt], ok[tt]);
        } else if constexpr (HEAD == 2) {
            load8(hv[tt], a.dmeta + off[tt], ok[tt]);
        } else {
            load8(g[tt], a.gin + off[tt], ok[tt]);
        }
    }
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 3 * 64) {
        const float* src = a.vec[tid >> 6];
        if (src) pv = reinterpret_cast<const float4*>(src)[tid & 63];
    }
    WStream ws;
    ws.init(a.wpack, a.wbytes, wave, lane);
    constexpr int kPF = pf_of<NTT>();
    Ring<kPF> R;
#pragma unroll
    for (int ks = 0; ks < kPF; ++ks) ws.load(R.a[ks], a.img[0], ks);
    float xh2[BODY ? NTT : 1][8], prv[BODY ? NTT : 1][8], xh1[BODY ? NTT : 1][8], rs2[BODY ? NTT : 1], rs1[BODY ? NTT : 1];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        if constexpr (HEAD == 1) load8(g[tt], a.res + off[tt], ok[tt]);                 // g <- the residual-branch cotangent; the GEMM is added to it
        if constexpr (HEAD == 2) {
            load8(xhf[tt], a.xhatf + off[tt], ok[tt]);
            rsf[tt] = ok[tt] ? a.rstdf[row0 + tt * 16 + rl.n16] : 0.f;
        }
        if constexpr (BODY) {
            load8(xh2[tt], a.xhat2 + off[tt], ok[tt]);
            load8(prv[tt], a.pre + off[tt], ok[tt]);
            load8(xh1[tt], a.xhat1 + off[tt], ok[tt]);
            rs2[tt] = ok[tt] ? a.rstd2[row0 + tt * 16 + rl.n16] : 0.f;
            rs1[tt] = ok[tt] ? a.rstd1[row0 + tt * 16 + rl.n16] : 0.f;
        }
    }
    if (tid < 3 * 64) reinterpret_cast<float4*>(vecs)[tid] = pv;
    Acc<NTT> acc;
    constexpr int kHeadImgs = HEAD == 1 ? 3 : HEAD == 2 ? 1 : 0;
    if constexpr (HEAD == 1) {
        // ---- d x = res + dq Wq + dk Wk + dv Wv (attn.py:183-185 backward): one K = 768 reduction, common row scale for the three operands
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const float m = fmaxf(row_absmax(hq[tt]), fmaxf(row_absmax(hk[tt]), row_absmax(hv[tt])));
            row_to_ximg(hq[tt], m, rl, ximg + tt * kXImg, rscale + tt * 16);
            row_to_ximg(hk[tt], m, rl, ximg + (NTT + tt) * kXImg, rscale + tt * 16);
            row_to_ximg(hv[tt], m, rl, ximg + (2 * NTT + tt) * kXImg, rscale + tt * 16);
        }
        ENC_STAMP(1);
        barrier_lds();
        ENC_STAMP(2);
        acc.zero();
        gemm<NTT, true>(ws, R, a.img[0], a.img[1], xr0, xr1, acc);
        gemm<NTT, true>(ws, R, a.img[1], a.img[2], xr0 + NTT * kXImg, xr1 + NTT * kXImg, acc);
        gemm<NTT, BODY>(ws, R, a.img[2], a.img[3], xr0 + 2 * NTT * kXImg, xr1 + 2 * NTT * kXImg, acc);
        ENC_STAMP(3);
        acc_to_staging<NTT>(acc, Y, rscale, wave, lane);
        barrier_lds();
        ENC_STAMP(4);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            float v[8];
            lds8(v, Y + (tt * 16 + rl.n16) * kYS + rl.slot * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[tt][e] += v[e];
        }
    } else if constexpr (HEAD == 2) {
        // ---- output projection and encoder.norm backward (transformer_net.py:129, :68)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) row_to_ximg(hv[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
        ENC_STAMP(1);
        barrier_lds();
        ENC_STAMP(2);
        acc.zero();
        gemm<NTT, BODY>(ws, R, a.img[0], a.img[1], xr0, xr1, acc);
        ENC_STAMP(3);
        acc_to_staging<NTT>(acc, Y, rscale, wave, lane);
        barrier_lds();
        ENC_STAMP(4);
        float gam[8], pg[8], pb[8];
        lds8(gam, vecs + VB_GF * 256 + rl.slot * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { pg[e] = 0.f; pb[e] = 0.f; }
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            float v[8];
            lds8(v, Y + (tt * 16 + rl.n16) * kYS + rl.slot * 8);
            layer_norm_bwd8(v, xhf[tt], rsf[tt], gam, g[tt], pg, pb);
        }
        partials_to_lds(pg, pb, red, wave, lane);
        barrier_lds();
        partials_to_global(red, a.partial_f, tid);
        barrier_lds();                                               // red is reused by LayerNorm2's partials
    } else {
        barrier_lds();                                               // the parameter vectors are in LDS
    }
    if constexpr (!BODY) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            store8(a.gx + off[tt], g[tt], ok[tt]);
            if (row0 + tt * 16 + rl.n16 < a.gx_head_rows) store8(a.gx_head + off[tt], g[tt], ok[tt]);
        }
        return;
    } else {
        // (every row pass: arithmetic and the X image first, barrier, and only then the stores of the weight-gradient operands)
        // ---- LayerNorm2 backward (transformer_net.py:44): gs2 = d(x1 + ffn); it is both the conv2 cotangent and the residual branch
        float res2[NTT][8], sv[NTT][8];
        {
            float gam[8], pg[8], pb[8];
            lds8(gam, vecs + VB_G2 * 256 + rl.slot * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { pg[e] = 0.f; pb[e] = 0.f; }
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                layer_norm_bwd8(g[tt], xh2[tt], rs2[tt], gam, res2[tt], pg, pb);
                row_to_ximg(res2[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
            }
            partials_to_lds(pg, pb, red, wave, lane);
        }
        ENC_STAMP(5);
        barrier_lds();
        ENC_STAMP(6);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) store8(a.gs2 + off[tt], res2[tt], ok[tt]);
        partials_to_global(red, a.partial2, tid);
        // ---- conv2^T and GELU' (transformer_net.py:41-42)
        acc.zero();
        gemm<NTT, true>(ws, R, a.img[kHeadImgs], a.img[kHeadImgs + 1], xr0, xr1, acc);
        ENC_STAMP(7);
        acc_to_staging<NTT>(acc, Y, rscale, wave, lane);
        barrier_lds();
        ENC_STAMP(8);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            lds8(sv[tt], Y + (tt * 16 + rl.n16) * kYS + rl.slot * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[tt][e] *= gelu_exact_grad(prv[tt][e]);
            row_to_ximg(sv[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
        }
        ENC_STAMP(9);
        barrier_lds();
        ENC_STAMP(10);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) store8(a.dpre + off[tt], sv[tt], ok[tt]);
        // ---- conv1^T + residual branch, LayerNorm1 backward (transformer_net.py:37-38)
        acc.zero();
        gemm<NTT, true>(ws, R, a.img[kHeadImgs + 1], a.img[kHeadImgs + 2], xr0, xr1, acc);
        ENC_STAMP(11);
        acc_to_staging<NTT>(acc, Y, rscale, wave, lane);
        barrier_lds();
        ENC_STAMP(12);
        {
            float gam[8], pg[8], pb[8];
            lds8(gam, vecs + VB_G1 * 256 + rl.slot * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { pg[e] = 0.f; pb[e] = 0.f; }
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                float v[8];
                lds8(v, Y + (tt * 16 + rl.n16) * kYS + rl.slot * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += res2[tt][e];
                layer_norm_bwd8(v, xh1[tt], rs1[tt], gam, sv[tt], pg, pb);
                row_to_ximg(sv[tt], 0.f, rl, ximg + tt * kXImg, rscale + tt * 16);
            }
            partials_to_lds(pg, pb, red, wave, lane);                // red: read by partials_to_global(partial2) two barriers ago
        }
        ENC_STAMP(13);
        barrier_lds();
        ENC_STAMP(14);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) store8(a.gs1 + off[tt], sv[tt], ok[tt]);
        partials_to_global(red, a.partial1, tid);
        // ---- out-projection^T (attn.py:196): the attention backward's input
        acc.zero();
        gemm<NTT, false>(ws, R, a.img[kHeadImgs + 2], 0, xr0, xr1, acc);
        acc_to_global<NTT>(acc, a.dout, nullptr, rscale, row0, a.rows, wave, lane);
        ENC_STAMP(15);
    }
}

// ------------------------------------------------------------------------------------------------ weight images
struct PackArgs {
    const float* W[DPN_ENC_MAX_MATS];
    int n;
    char* out;
    int* status;
};
// one thread = one 16-byte fragment slot of both planes: image 0 is W as stored ([out][in]: y = x W^T contracts over `in`), image 1 is W^T
// (g W contracts over `out`).  Fragment (tile mt, k-step ks, lane (m, g)) holds A[16 mt + m][32 ks + 8 g .. + 7].
DEV void pack_body(const PackArgs& a, const int idx) {
    const int lane = idx & 63, ks = (idx >> 6) & 7, mt = (idx >> 9) & 15, im = (idx >> 13) & 1, mat = idx >> 14;
    if (mat >= a.n) return;
    const float* W = a.W[mat];
    const int r = 16 * mt + (lane & 15), c0 = 32 * ks + 8 * (lane >> 4);
    float v[8];
    if (im == 0) {
        const float4 x0 = *reinterpret_cast<const float4*>(W + r * kD + c0), x1 = *reinterpret_cast<const float4*>(W + r * kD + c0 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W[(c0 + e) * kD + r];
    }
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) bad |= !(fabsf(v[e]) < 32768.f);
    if (bad && a.status) atomicOr(a.status, 1);
    u32x4 hi, lo;
    split8(v, hi, lo);
    char* dst = a.out + (int64_t)(mat * 2 + im) * kImgBytes + ((mt * 8 + ks) * 2) * 1024 + lane * 16;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 1024) = lo;
}
__global__ __launch_bounds__(256) void dpn_enc_pack_kernel(PackArgs a) { pack_body(a, blockIdx.x * 256 + threadIdx.x); }

// Everything of the encoder's forward that depends on the step's inputs only -- the weight images, the im2col rows of the circular token
// convolution (embed.py:45-47) and the lead-time positional encodings of the encoder (embed.py:58) and of the VariableNets
// (variable_net.py:46) -- in ONE launch (they were four): block ranges [pack | im2col | lead-time PE].
struct PrepArgs {
    PackArgs pack;
    int pack_blocks, col_blocks;
    const float* x; int T, C; int64_t col_total; float* xu;
    const float* h; int batch; const float* fa; int na; float* oa; const float* fb; int nb; float* ob;
};
struct ConvSplitArgs {                           // dpn_conv16's operand images: one block per im2col row / per weight row
    const float* x; int T, C, batch; const float* conv_w;
    int xs_blocks, ws_blocks, Kp;
    _Float16 *xs, *ws; int *xe, *we;
};
// one row of K values -> power-of-two scale (row maximum into [8, 16)), f16 hi / lo fragment slots of Kp values (zeros behind K), the biased
// exponent.  ONE pass over memory: a thread's values (k = tid + 256 u: coalesced loads, all in flight at once) stay in registers between the
// maximum and the split; the f16 halves pass through LDS so that the stores are whole 16-byte fragment slots (2-byte stores straight from the
// registers: 58 store instructions per thread, each touching eight lines -- the launch took 20 us instead of 9).
// Layout = MFMA-fragment images, as the encoder's weight images: per (16-row strip, 32-k block) 2 KB = [hi | lo][lane = (k % 32) / 8 * 16 +
// row % 16][8 f16], so that a wave's fragment load is ONE contiguous KB.  (Row-major planes made every 16-byte piece of a load its own cache
// line access -- 64 tag look-ups per instruction: the loads alone were 7 of the kernel's 15 us.)  `strip`: the strip's first element, c = row % 16.
constexpr int kSplitIters = 29;                  // rows of up to 7 424 values (3 x 2 405 = 7 215)
template <class Val>
DEV void split_row_planes(const Val& val, const int K, const int Kp, _Float16* strip, const int c, int* e_out) {
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) _Float16 stage[2][kSplitIters * 256];
    const int tid = threadIdx.x;
    float v[kSplitIters];
#pragma unroll
    for (int u = 0; u < kSplitIters; ++u) {
        const int k = tid + 256 * u;
        v[u] = val(k < K ? k : K - 1);           // (clamped index: an unconditional load; the tail is zeroed below)
    }
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < kSplitIters; ++u) {
        v[u] = (tid + 256 * u < K) ? v[u] : 0.f;
        m = fmaxf(m, fabsf(v[u]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
    eb = eb < 4 ? 4 : eb;
    const float sc = __builtin_bit_cast(float, (257 - eb) << 23);
#pragma unroll
    for (int u = 0; u < kSplitIters; ++u) {
        const int k = tid + 256 * u;
        const float x = v[u] * sc;
        const _Float16 h_ = (_Float16)x;
        stage[0][k] = h_;
        stage[1][k] = (_Float16)((x - (float)h_) * kLoScale);
    }
    __syncthreads();
    for (int j = tid; j * 8 < Kp; j += 256) {    // slot j = k / 8: block j / 4, k-group j % 4
        _Float16* slot = strip + (j >> 2) * 1024 + ((j & 3) * 16 + c) * 8;
        *reinterpret_cast<u32x4*>(slot) = *reinterpret_cast<const u32x4*>(&stage[0][j * 8]);
        *reinterpret_cast<u32x4*>(slot + 512) = *reinterpret_cast<const u32x4*>(&stage[1][j * 8]);
    }
    if (tid == 0) *e_out = eb;
}
__global__ __launch_bounds__(256) void dpn_enc_prep_kernel(PrepArgs a) {
    int blk = blockIdx.x;
    if (blk < a.pack_blocks) { pack_body(a.pack, blk * 256 + threadIdx.x); return; }
    blk -= a.pack_blocks;
    if (blk < a.col_blocks) {                                                // out[b*T + t][c*3 + tap] = x[b*T + (t + tap - 1) mod T][c]
        const int64_t i = (int64_t)blk * 256 + threadIdx.x;
        if (i >= a.col_total) return;
        const int64_t tg = i / (3 * a.C);
        const int r = (int)(i - tg * 3 * a.C), c = r / 3, tap = r - 3 * c;
        const int64_t b = tg / a.T;
        int ts = (int)(tg - b * a.T) + tap - 1;
        ts = ts < 0 ? ts + a.T : (ts >= a.T ? ts - a.T : ts);
        a.xu[i] = a.x[(b * a.T + ts) * a.C + c];
        return;
    }
    blk -= a.col_blocks;
    const int i = blk * 256 + threadIdx.x;                                   // SineCosPE of the scalar lead time (position_encoding.py:35-50)
    if (i >= a.batch * (a.na + a.nb)) return;
    const int b = i / (a.na + a.nb), j = i - b * (a.na + a.nb);
    const float hv = a.h[b];
    if (j < a.na) { const float s_ = hv * a.fa[j]; a.oa[(int64_t)b * 2 * a.na + 2 * j] = sinf(s_); a.oa[(int64_t)b * 2 * a.na + 2 * j + 1] = cosf(s_); }
    else { const int jj = j - a.na; const float s_ = hv * a.fb[jj]; a.ob[(int64_t)b * 2 * a.nb + 2 * jj] = sinf(s_); a.ob[(int64_t)b * 2 * a.nb + 2 * jj + 1] = cosf(s_); }
}

#ifdef DPN_EXPERIMENTS
// Experiment (DPN_CONV16=1): the token convolution's operands split once per step for dpn_conv16.  Its own launch: inside dpn_enc_prep the
// 29 KB of LDS and the registers of split_row_planes slowed every other block range of that launch (encoder forward 183 -> 189 us).
__global__ __launch_bounds__(256) void dpn_conv16_split_kernel(ConvSplitArgs a) {
    int blk = blockIdx.x;
    if (blk < a.xs_blocks) {                                                 // im2col row blk = (b, t) as planes: k = c * 3 + tap, as xu below
        const int64_t b = blk / a.T;
        const int t = blk - (int)b * a.T;
        const float* xb = a.x + b * a.T * a.C;
        const int T = a.T, C = a.C;
        auto val = [&](const int k) __attribute__((always_inline)) {
            const int c = k / 3, tap = k - 3 * c;
            int ts = t + tap - 1;
            ts = ts < 0 ? ts + T : (ts >= T ? ts - T : ts);
            return xb[(int64_t)ts * C + c];
        };
        split_row_planes(val, 3 * C, a.Kp, a.xs + (int64_t)(blk >> 4) * 32 * a.Kp, blk & 15, a.xe + blk);
        return;
    }
    blk -= a.xs_blocks;
    if (blk < a.ws_blocks) {                                                 // weight row blk
        const float* wr = a.conv_w + (int64_t)blk * 3 * a.C;
        auto val = [&](const int k) __attribute__((always_inline)) { return wr[k]; };
        split_row_planes(val, 3 * a.C, a.Kp, a.ws + (int64_t)(blk >> 4) * 32 * a.Kp, blk & 15, a.we + blk);
        return;
    }
}

// ------------------------------------------------------------------------------------------------ the token convolution as a GEMM on pre-split planes
// parts[s][m][n] = sum_{k in slice s} x(m, k) w(n, k)  (model/embed.py:45-47: circular Conv1d(C, 256, 3) over the tokens = im2col rows times
// the weight rows, K = 3 C = 7 215).  The operands were split ONCE by dpn_enc_prep (f16 hi / lo planes, one power-of-two scale per row), so the
// loop is loads and MFMAs only.  A workgroup (4 waves) owns a 128 x 64 tile of one K-slice; wave w owns rows 32 w .. 32 w + 31 (two A strips x
// hi | lo: four 16-byte loads per 32-k block, private) against all four column tiles, whose B fragments (4 tiles x hi | lo) are staged through
// LDS, two per wave and block, double-buffered with one LDS-only barrier per block: 24 MFMAs per wave and block for 8 KB of LDS reads (a
// 16 x 128 wave tile does the same MFMAs for 16 KB and measured 2 600 cycles per block: the LDS reads of eight waves, not the MFMAs).
// Straight-line code for the slice's blocks (behind a loop's back edge hipcc waits for every load in flight), loads four blocks ahead.
constexpr int kConvBlocks = 16;                 // 32-k blocks per K-slice at most (blocks behind the slice's end multiply zeros)
struct Conv16Args { const _Float16 *xs, *ws; const int *xe, *we; int M, N, Kp, blocks_per_slice, slices, tn, tm; float* parts; };
__global__ __launch_bounds__(256) void dpn_conv16_kernel(Conv16Args a) {
    __shared__ __attribute__((aligned(16))) char bf[2][8][1024];             // [buffer][column tile * 2 + plane][lane x 16 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (K-slice, tile): all tiles of a slice on ONE XCD (block b runs on XCD b % 8 -- observed placement, used for speed only), so
    // that the slice's 1 MB of planes is fetched into that XCD's L2 once and the 4 x / 3 x re-reads of the tiles hit there (spread over the
    // XCDs the launch moved 74 MB through the fabric: 21 -> 15 us with the tile change alone)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, tiles = a.tn * a.tm;
    const int sl = xcd + 8 * (jj / tiles), tile = jj % tiles;
    if (sl >= a.slices) return;
    const int n0 = (tile % a.tn) * 64, m0 = (tile / a.tn) * 128;
    const int total = a.Kp / 32, kb0 = sl * a.blocks_per_slice;
    const int nb = min(total - kb0, a.blocks_per_slice);
    const int c = lane & 15, g = lane >> 4;
    const int row0 = m0 + wave * 32 + c, col = n0 + wave * 16 + c;           // this lane's A rows (row0, row0 + 16), and the column it stages (tile `wave`)
    // fragment images (dpn_enc_prep): per (16-row strip, 32-k block) 2 KB = [hi | lo][lane][16 B]; a strip holds Kp / 32 blocks
    const int nblk = a.Kp / 32;
    const int64_t xbytes = (int64_t)((a.M + 15) / 16) * nblk * 2048, wbytes = (int64_t)((a.N + 15) / 16) * nblk * 2048;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.xs), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.ws), 0, (int)wbytes, 0x00020000);
    // (strips outside the problem: an offset behind the descriptor reads as zero; rows of a partial last strip hold whatever the buffer held:
    // they only reach output rows that are not stored)
    int xvo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int strip = (m0 >> 4) + wave * 2 + r;
        xvo[r] = strip * 16 < a.M ? (int)((int64_t)strip * nblk * 2048) + lane * 16 : 0x7ffff000;
    }
    const int wstrip = (n0 >> 4) + wave;
    const int wvo = wstrip * 16 < a.N ? (int)((int64_t)wstrip * nblk * 2048) + lane * 16 : 0x7ffff000;
    constexpr int D = 6;                         // blocks in flight: the planes were written a launch ago by other XCDs, the first touch is a memory round trip
    u32x4 ah[D][2], al[D][2], bh[D], bl[D];
    // (every block issues exactly six loads, blocks behind the slice's end from an offset behind the descriptor = zeros: with loads under
    // conditions hipcc cannot count them and waits for ALL of them at every block -- one memory round trip per block)
    auto fetch = [&](const int i, const int s_) __attribute__((always_inline)) {
        const int so = i < nb ? (kb0 + i) * 2048 : 0x7ffff000;               // one block of a strip: 2 KB [hi | lo]
#ifdef CONV_ABL_NOLOAD                           // (ablation builds, wrong results on purpose: tools/variant_build.py --unit=5)
        if (i >= 0) return;
#endif
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            ah[s_][r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[r], so, 0));
            al[s_][r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[r] + 1024, so, 0));
        }
        bh[s_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0));
        bl[s_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo + 1024, so, 0));
    };
    f32x4 am[2][4], ac[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) { am[r][t] = (f32x4)0.f; ac[r][t] = (f32x4)0.f; }
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, d);
#pragma unroll
    for (int i = 0; i < kConvBlocks; ++i) {
        const int d = i % D, buf = i & 1;
        u32x4 bq[8];                                                         // all eight B fragments first, into registers of their own
#ifdef CONV_ABL_NOLDS
#pragma unroll
        for (int f = 0; f < 8; ++f) bq[f] = (f & 1) ? bl[d] : bh[d];
#else
        *reinterpret_cast<u32x4*>(&bf[buf][2 * wave][lane * 16]) = bh[d];
        *reinterpret_cast<u32x4*>(&bf[buf][2 * wave + 1][lane * 16]) = bl[d];
        barrier_lds();                                                       // (LDS only: the blocks in flight stay in flight)
#pragma unroll
        for (int f = 0; f < 8; ++f) bq[f] = *reinterpret_cast<const u32x4*>(&bf[buf][f][lane * 16]);
#endif
#ifdef CONV_ABL_NOMFMA
#pragma unroll
        for (int f = 0; f < 8; ++f) asm volatile("" ::"v"(bq[f]));
        asm volatile("" ::"v"(ah[d][0]), "v"(ah[d][1]), "v"(al[d][0]), "v"(al[d][1]));
        if (i + D < kConvBlocks) fetch(i + D, d);
        continue;
#endif
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) am[r][t] = mfma(ah[d][r], bq[2 * t], am[r][t]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) ac[r][t] = mfma(ah[d][r], bq[2 * t + 1], ac[r][t]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) ac[r][t] = mfma(al[d][r], bq[2 * t], ac[r][t]);
        if (i + D < kConvBlocks) fetch(i + D, d);                            // refill the slot this block just used
        // (double buffer: block i + 1 writes the other buffer; its barrier orders block i + 2's writes behind this block's reads)
    }
    float* out = a.parts + (int64_t)sl * a.M * a.N;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int er[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int rr = m0 + wave * 32 + 16 * r + g * 4 + j; er[j] = rr < a.M ? a.xe[rr] : 130; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cc = n0 + t * 16 + c;
            if (cc >= a.N) continue;
            const int ec = a.we[cc];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = m0 + wave * 32 + 16 * r + g * 4 + j;
                if (rr < a.M) out[(int64_t)rr * a.N + cc] = __builtin_ldexpf(fmaf(ac[r][t][j], kLoInv, am[r][t][j]), er[j] + ec - 260);
            }
        }
    }
}
#endif  // DPN_EXPERIMENTS (dpn_conv16_split_kernel, dpn_conv16_kernel)

// ------------------------------------------------------------------------------------------------ weight gradients
// d W[M][N] = G^T X over the token rows (G: the cotangent [rows][M] of a linear's output, X: its input [rows][N]); d b = column sums of G.
// (attn.py:183-196, transformer_net.py:38-42, :129 backward; embed.py:45-47: the token convolution's weight gradient with X = the im2col rows.)
// One launch for all linears of the encoder stack: a workgroup owns a 64 x 64 tile of one d W and walks the rows 32 at a time.  The operand
// layouts need no transposition: a lane that loads eight consecutive rows of ONE column holds exactly an MFMA fragment slot (k = row).  The
// G columns of a wave are private (its own 16 x 64 strip of the tile), the X fragments are shared through LDS (double-buffered, one barrier
// per 32 rows).  Same f16 hi+lo split as the GEMMs above; the range is handled like a running softmax maximum: every 32-row block is scaled
// by a power of two taken from the largest magnitude seen SO FAR in that operand strip, and when a block raises it the accumulators are
// rescaled (exact: powers of two) -- cotangent rows that differ by many decades (a field whose loss is 100 x the median's) cost nothing.
// Long reductions (batches of fields) are cut into row slices whose partial tiles dpn_wgrad16_reduce adds in a fixed order.
// The same kernel serves every small GEMM with a short output and any operand orientation (dpn_gemm16): C[m][n] = sum_k A(m, k) B(n, k) with
// element strides per operand -- A(m, k) = A[m * a_sm + k * a_sk] -- so "rows are k" (the weight gradients: a_sm = 1, a_sk = ld) and "k is
// contiguous" (x W^T: a_sm = ld, a_sk = 1) are the same code; the eight k-values of a fragment slot are eight dword loads either way.
struct WgProblem {
    const float *G, *X;                          // A (its 16-column strip is private to a wave), B (shared through LDS)
    float *dW, *db;                              // C, and the sums over k of A (the bias gradient when A is a cotangent) or NULL
    const float* bias;                           // [N] added to every row of C, or NULL
    int M, N, rows, ldw;                         // rows = K
    int64_t g_sm, g_sk, x_sn, x_sk;
    int64_t part_off;                            // floats: this problem's partials [slices][M * N (+ M with db)]
};
constexpr int kWgMaxProblems = DPN_WGRAD_MAX_PROBLEMS, kWgMaxJobs = DPN_GEMM_MAX_JOBS;
struct WgJob { const float* partial; float* out_a; float* out_b; int nblocks, pad; };
struct WgArgs {
    WgProblem p[kWgMaxProblems];
    WgJob job[kWgMaxJobs];
    int n, slices, rows_per_slice, n_tiles;
    float* partials;
    int start[kWgMaxProblems + 1];               // first workgroup of each problem: the grid is the LIST of its tiles (x slices), then one per job
    float inv_tx[kWgMaxProblems];
    int per_slice[kWgMaxProblems];
    int tx[kWgMaxProblems];                      // tiles along N  (int: with a 2-byte table hipcc folded 2 * pi into the BASE of the scalar load of
                                                 //  start[pi] -- scalar loads drop the low two address bits of the base: every odd problem read start[pi - 1])
};
static_assert(sizeof(WgArgs) <= 4096, "kernel arguments");

DEV float wave_absmax_uniform(float m) {         // maximum over the 64 lanes as a wave-uniform value
    m = half_max(m);
    return fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 0)),
                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 32)));
}
DEV int exp_bits(float m) { const int e = (__builtin_bit_cast(int, m) >> 23) & 0xff; return e < 13 ? 13 : e; }

// Tile: kWgWaves waves, each a 16-column strip of G against ALL kWgWaves 16-column tiles of X: a lane splits 16 values per block for
// 3 kWgWaves MFMAs.  Four waves (64 x 64): ~150 VALU per 12 MFMAs.  Eight waves (128 x 128, -DDPN_WG_WAVES=8) halve the split work per MFMA
// and were measured SLOWER -- 27.5 against 23.1 us for the 25 problems of 287 rows (tools/wgrad16_bench.py) -- each wave then reads 16 KB of
// LDS per block for 24 MFMAs (the bound dpn_conv16 met with the same wave tile).
#ifndef DPN_WG_WAVES
#define DPN_WG_WAVES 4
#endif
constexpr int kWgWaves = DPN_WG_WAVES, kWgTile = 16 * kWgWaves;
__global__ __launch_bounds__(64 * kWgWaves) void dpn_wgrad16_kernel(WgArgs a) {
    __shared__ __attribute__((aligned(16))) char xf[2][kWgWaves][2][1024];   // [buffer][n-tile][hi | lo][lane x 16 B]
    __shared__ int xe[2][kWgWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // One workgroup per tile that exists: a (widest N, tallest M, problems) box around 25 problems of 4 x 4 .. 8 x 4 tiles and ONE of 113 x 4
    // (the token convolution) was 31 640 workgroups for ~950 tiles.
    // (consecutive tiles go to different XCDs; giving every XCD a contiguous range of the list measured 26.7 -> 31.5 us: profiles/round6_xcd_contiguous.txt)
    const int bid = blockIdx.x;
    if (bid >= a.n_tiles) {                      // ride-along job: LayerNorm parameter sums (fixed order over the row blocks)
        const WgJob& j = a.job[bid - a.n_tiles];
        float s1 = 0.f, s2 = 0.f;
        if (tid >= 256) return;
#pragma unroll 8
        for (int b = 0; b < j.nblocks; ++b) { s1 += j.partial[(int64_t)b * 512 + tid]; s2 += j.partial[(int64_t)b * 512 + 256 + tid]; }
        j.out_a[tid] = s1;
        j.out_b[tid] = s2;
        return;
    }
    // (decoded on the SCALAR unit: the waves of this kernel are VALU-bound, and a search that stops early is a chain of dependent scalar loads)
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kWgMaxProblems; ++i) pi += (int)((unsigned)(a.start[i] - 1 - bid) >> 31);     // entries behind the last problem hold n_tiles
    pi = __builtin_amdgcn_readfirstlane(pi);
    const WgProblem& p = a.p[pi];
    const int tx = a.tx[pi], per_slice = a.per_slice[pi];
    int rel = bid - a.start[pi], sl = 0;
    while (rel >= per_slice) { rel -= per_slice; ++sl; }                    // (slices: a handful)
    const int by = __builtin_amdgcn_readfirstlane((int)(((float)rel + 0.5f) * a.inv_tx[pi])), bx = rel - by * tx;
    const int m0 = by * kWgTile, n0 = bx * kWgTile;
    const int r_begin = sl * a.rows_per_slice, r_end = min(p.rows, r_begin + a.rows_per_slice);
    const int nk = r_end > r_begin ? (r_end - r_begin + 31) / 32 : 0;
    const int c = lane & 15, g = lane >> 4;
    const int gm = m0 + wave * 16 + c, xn = n0 + wave * 16 + c;              // this lane's column of G (its A strip) and of X (the tile it stages)
    const bool gok = gm < p.M, xok = xn < p.N;
    // (everything the loop needs from the problem record in registers: read through the reference, each use is a scalar load from the kernel
    // arguments with its own wait -- measured 1.5 us per 32-row block; loads are unconditional from a clamped address, zeroed by a select)
    const int64_t g_sk = p.g_sk, x_sk = p.x_sk;
    const float* gp = p.G + (gok ? gm * p.g_sm : 0);
    const float* xp = p.X + (xok ? xn * p.x_sn : 0);
    constexpr int kWgDepth = 4;                  // rows in flight: blocks of 32 (a dependent global round trip per block would be a chain of latencies)
    float gv[kWgDepth][8], xv[kWgDepth][8];
    // Fast paths, chosen per operand: buffer loads -- one instruction per load, the block's base a scalar, and whatever lies outside the
    // descriptor (lanes outside the problem: offset 2^31 - 16) reads as zero without a mask.
    //   mode 1 "rows are k" (the weight gradients: element stride 1 along m / n): eight dword loads per block, the lane's eight row offsets
    //          computed once; the descriptor ends behind the slice's last row, so rows beyond it are zero too.
    //   mode 2 "k is contiguous" (x W^T: the token convolution, the hyper-network heads): the lane's eight k-values are 32 contiguous bytes
    //          = two 16-byte loads (dword-aligned rows: K = 7 215 does not make them 16-byte aligned, buffer loads do not ask for it); the
    //          k-tail of the last block is zeroed when the block is consumed.
    //   mode 0 generic strides (~6 VALU per load on 64-bit addresses and 2 per element on masks: ~500 VALU per 32-row block and wave
    //          against 12 MFMAs).
    const int g_mode = (p.g_sm == 1 && (int64_t)p.rows * g_sk * 4 < (1ll << 31)) ? 1 : (g_sk == 1 && (int64_t)p.M * p.g_sm * 4 < (1ll << 31)) ? 2 : 0;
    const int x_mode = (p.x_sn == 1 && (int64_t)p.rows * x_sk * 4 < (1ll << 31)) ? 1 : (x_sk == 1 && (int64_t)p.N * p.x_sn * 4 < (1ll << 31)) ? 2 : 0;
    __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0,
        g_mode == 1 ? (int)((int64_t)r_end * g_sk * 4) : g_mode == 2 ? (int)((int64_t)p.M * p.g_sm * 4) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0,
        x_mode == 1 ? (int)((int64_t)r_end * x_sk * 4) : x_mode == 2 ? (int)((int64_t)p.N * p.x_sn * 4) : 0, 0x00020000);
    int gvo[8], xvo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gvo[e] = !gok ? 0x7ffffff0 : g_mode == 2 ? (int)((gm * p.g_sm + g * 8) * 4) + 16 * (e & 1) : (int)(((g * 8 + e) * g_sk + gm) * 4);
        xvo[e] = !xok ? 0x7ffffff0 : x_mode == 2 ? (int)((xn * p.x_sn + g * 8) * 4) + 16 * (e & 1) : (int)(((g * 8 + e) * x_sk + xn) * 4);
    }
    auto fetch1 = [&](const int mode, const __amdgpu_buffer_rsrc_t rs, const int (&vo)[8], const int64_t sk, const float* base, const int ks,
                      float (&d)[8]) __attribute__((always_inline)) {
        if (mode == 1) {
            const int so = (int)((r_begin + ks * 32) * sk * 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[e], so, 0));
        } else if (mode == 2) {
            const int so = (r_begin + ks * 32) * 4;
            const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo[0], so, 0));
            const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo[1], so, 0));
            d[0] = lo[0]; d[1] = lo[1]; d[2] = lo[2]; d[3] = lo[3]; d[4] = hi[0]; d[5] = hi[1]; d[6] = hi[2]; d[7] = hi[3];
        } else {
            // generic strides: unconditional loads from a clamped row, landing in the ring untouched; rows / columns outside the problem are
            // zeroed by a multiplication when the block is CONSUMED.  (A select makes the compiler sink every load into a branch of its own,
            // and a mask applied at fetch time puts an s_waitcnt vmcnt(0) in front of every barrier: both measured at ~1.5 us per block.)
            const int r0 = r_begin + ks * 32 + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t r = r0 + e < r_end ? r0 + e : r_begin;
                d[e] = base[r * sk];
            }
        }
    };
    auto fetch = [&](int ks, float (&gd)[8], float (&xd)[8]) __attribute__((always_inline)) {
        fetch1(g_mode, grs, gvo, g_sk, gp, ks, gd);
        fetch1(x_mode, xrs, xvo, x_sk, xp, ks, xd);
    };
    auto mask1 = [&](const int mode, const bool ok, const int ks, float (&d)[8]) __attribute__((always_inline)) {
        if (mode == 1) return;
        const int r0 = r_begin + ks * 32 + g * 8;
        if (mode == 2 && r0 + 8 <= r_end) return;                            // (only the k-tail of the last block; wave-divergent at most there)
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] *= (r0 + e < r_end && ok) ? 1.f : 0.f;
    };
    auto mask_rows = [&](int ks, float (&gd)[8], float (&xd)[8]) __attribute__((always_inline)) {
        mask1(g_mode, gok, ks, gd);
        mask1(x_mode, xok, ks, xd);
    };
    f32x4 am[kWgWaves], ac[kWgWaves];
    int eg_cur = 13, ex_mine = 13, ex_seen[kWgWaves];
#pragma unroll
    for (int t = 0; t < kWgWaves; ++t) { am[t] = (f32x4)0.f; ac[t] = (f32x4)0.f; ex_seen[t] = 13; }
    float bsum = 0.f;
    const bool want_b = bx == 0 && p.db != nullptr;
#pragma unroll
    for (int d = 0; d < kWgDepth; ++d)
        if (d < nk) fetch(d, gv[d], xv[d]);
    for (int ks0 = 0; ks0 < nk; ks0 += kWgDepth) {
#pragma unroll
        for (int d = 0; d < kWgDepth; ++d) {
            const int ks = ks0 + d;
            if (ks >= nk) break;
            const int buf = d & 1;                                           // kWgDepth is even: consecutive blocks alternate buffers
            // ---- this wave's G strip: running scale, split (registers only)
            mask_rows(ks, gv[d], xv[d]);
            float mg = 0.f, mx = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { mg = fmaxf(mg, fabsf(gv[d][e])); mx = fmaxf(mx, fabsf(xv[d][e])); bsum += gv[d][e]; }
            const int eg = exp_bits(wave_absmax_uniform(mg)), ex = exp_bits(wave_absmax_uniform(mx));
            if (eg > eg_cur) {
                const int dd = max(eg_cur - eg, -200);
#pragma unroll
                for (int t = 0; t < kWgWaves; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { am[t][j] = __builtin_ldexpf(am[t][j], dd); ac[t][j] = __builtin_ldexpf(ac[t][j], dd); }
                eg_cur = eg;
            }
            if (ex > ex_mine) ex_mine = ex;
            float gs[8], xs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gs[e] = __builtin_ldexpf(gv[d][e], 140 - eg_cur); xs[e] = __builtin_ldexpf(xv[d][e], 140 - ex_mine); }      // maximum into [2^13, 2^14)
            u32x4 ahi, alo, bhi, blo;
            split8(gs, ahi, alo);
            split8(xs, bhi, blo);
            *reinterpret_cast<u32x4*>(&xf[buf][wave][0][lane * 16]) = bhi;
            *reinterpret_cast<u32x4*>(&xf[buf][wave][1][lane * 16]) = blo;
            if (lane == 0) xe[buf][wave] = ex_mine;
            if (ks + kWgDepth < nk) fetch(ks + kWgDepth, gv[d], xv[d]);      // refill the slot: kWgDepth blocks ahead
            barrier_lds();                                                   // (LDS only: a __syncthreads() would drain the rows in flight)
#pragma unroll
            for (int t = 0; t < kWgWaves; ++t) {
                const int et = xe[buf][t];
                if (et > ex_seen[t]) {
                    const int dd = max(ex_seen[t] - et, -200);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { am[t][j] = __builtin_ldexpf(am[t][j], dd); ac[t][j] = __builtin_ldexpf(ac[t][j], dd); }
                    ex_seen[t] = et;
                }
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(&xf[buf][t][0][lane * 16]);
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(&xf[buf][t][1][lane * 16]);
                am[t] = mfma(ahi, b0, am[t]);
                ac[t] = mfma(ahi, b1, ac[t]);
                ac[t] = mfma(alo, b0, ac[t]);
            }
            // (double-buffered X fragments: the next block's stores go to the other buffer; the barrier of that block orders them against
            // this block's reads of THIS buffer two blocks later)
        }
    }
    // ---- epilogue: undo the scales, write the tile (or the slice's partial tile)
    const bool direct = a.slices == 1;
    float* out = direct ? p.dW : a.partials + p.part_off + (int64_t)sl * ((int64_t)p.M * p.N + (p.db ? p.M : 0));
    const int ldo = direct ? p.ldw : p.N;
#pragma unroll
    for (int t = 0; t < kWgWaves; ++t) {
        const int sh = eg_cur + ex_seen[t] - 280;                            // (eg - 140) + (ex - 140)
        const int col = n0 + t * 16 + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = m0 + wave * 16 + g * 4 + j;
            if (row < p.M && col < p.N)
                out[(int64_t)row * ldo + col] = __builtin_ldexpf(fmaf(ac[t][j], kLoInv, am[t][j]), max(sh, -250)) + ((direct && p.bias) ? p.bias[col] : 0.f);
        }
    }
    if (want_b) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (lane < 16 && gok) (direct ? p.db : out + (int64_t)p.M * p.N)[gm] = bsum;
    }
}

// partial tiles of the row slices -> d W, d b (fixed order)
__global__ __launch_bounds__(256) void dpn_wgrad16_reduce_kernel(WgArgs a) {
    const WgProblem& p = a.p[blockIdx.y];
    const int64_t mn = (int64_t)p.M * p.N, per = mn + (p.db ? p.M : 0);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const float* src = a.partials + p.part_off + i;
    float v = src[0];
    for (int s = 1; s < a.slices; ++s) v += src[(int64_t)s * per];
    if (i < mn) p.dW[(i / p.N) * p.ldw + (i % p.N)] = v + (p.bias ? p.bias[i % p.N] : 0.f);
    else p.db[i - mn] = v;
}


// ------------------------------------------------------------------------------------------------ attention forward, 16-row query tiles
// FullAttention (attn.py:50-68): o = softmax(q k^T / sqrt(32)) v per head, 8 heads x 32, L <= 288 tokens per field.  Round 3's kernel
// (dpn_attn_fwd_kernel, exact-fp32 MFMA, 32-row tiles, K and V of the head staged through 151 KB of LDS) measured 11 us per layer on a
// workload of a few MFLOP: it is launch-to-first-data plus a chain of LDS stages.  Here q, k and v never pass through LDS: with the f16
// hi+lo split of the GEMMs above a lane's eight consecutive head channels of ONE token row ARE an MFMA fragment slot (scores: k = head
// channel, exactly one k-step), and eight consecutive token rows of one channel are a slot of the P V product (k = key).  Workgroup =
// (16 query rows, head): four waves, wave w takes key tiles w, w + 4, ... of the scores and key blocks w, w + 4, w + 8 of P V; only the
// 16 x 288 score / probability block and the waves' partial outputs go through LDS (27 KB).  Rows of q and k are scaled by a power of
// two each (the scales factor out of the head-channel contraction), v by one power of two per wave (its rows are contracted over).
constexpr int kALmax = 288, kASS = 292;
struct Attn16Args { const float *q, *k, *v; float *out, *P; int L; float scale; };

DEV void load8g(float (&v)[8], const float* p) {                           // eight consecutive floats (32-byte aligned rows of a [..][256] tensor)
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// scale the row that the four lanes (i, g = 0..3) hold (8 values each) so that its maximum lands in [8, 16); returns the biased exponent used
DEV int scale_row32(float (&v)[8]) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    int eb = (__builtin_bit_cast(int, m) >> 23) & 0xff;
    eb = eb < 4 ? 4 : eb;
    const float sc = __builtin_bit_cast(float, (257 - eb) << 23);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sc;
    return eb;
}

__global__ __launch_bounds__(256) void dpn_attn16_fwd_kernel(Attn16Args a) {
    __shared__ __attribute__((aligned(16))) float Ss[16 * kASS];
    __shared__ __attribute__((aligned(16))) float part[4][16 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y, q0 = blockIdx.x * 16, L = a.L;
    const int64_t ro = (int64_t)blockIdx.z * L * kD;
    const float* q = a.q + ro;
    const float* k = a.k + ro;
    const float* v = a.v + ro;
    const int c = lane & 15, g = lane >> 4;
    // ---- every load of the kernel in flight at once: the query fragment, this wave's key tiles (scores) and value blocks (P V)
    // (unconditional loads from clamped rows, zeroed by a multiplication where they are consumed: a select would put every load into a
    // branch of its own with its own wait)
    float qv[8], kv[5][8], vv[3][2][8];
    load8g(qv, q + (int64_t)min(q0 + c, L - 1) * kD + head * 32 + g * 8);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int jt = min(wave + 4 * u, 17), j = jt * 16 + c;                   // 18 key tiles of 16
        load8g(kv[u], k + (int64_t)min(j, L - 1) * kD + head * 32 + g * 8);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int ks = min(wave + 4 * u, 8);                                 // 9 key blocks of 32
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[u][nt][e] = v[(int64_t)min(ks * 32 + g * 8 + e, L - 1) * kD + head * 32 + nt * 16 + c];
    }
    {
        const float mq = q0 + c < L ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] *= mq;
    }
    // ---- scores: S[i][j] = scale q_i . k_j
    const int eq = scale_row32(qv);
    u32x4 qhi, qlo;
    split8(qv, qhi, qlo);
    // the accumulator rows of this lane are queries 4 g + jj: their scales sit in lanes (4 g + jj, any g)
    float qs[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) qs[jj] = __builtin_bit_cast(float, (__shfl(eq, 4 * g + jj) - 3) << 23) * a.scale;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int jt = wave + 4 * u;
        if (jt >= 18) break;
        {
            const float mk = jt * 16 + c < L ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) kv[u][e] *= mk;
        }
        const int ek = scale_row32(kv[u]);
        u32x4 khi, klo;
        split8(kv[u], khi, klo);
        f32x4 am = (f32x4)0.f, ac = (f32x4)0.f;
        am = mfma(qhi, khi, am);
        ac = mfma(qhi, klo, ac);
        ac = mfma(qlo, khi, ac);
        const float ks_ = __builtin_bit_cast(float, (ek - 3) << 23);
        const int col = jt * 16 + c;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            Ss[(4 * g + jj) * kASS + col] = col < L ? fmaf(ac[jj], kLoInv, am[jj]) * (qs[jj] * ks_) : -INFINITY;
    }
    __syncthreads();
    // ---- softmax: wave w takes rows 4 w .. 4 w + 3, the 64 lanes stride the 288 columns; probabilities to LDS and to P (saved for the backward)
    {
        float e[4][5], mx[4], sm[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            mx[rr] = -INFINITY;
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int col = lane + 64 * u;
                e[rr][u] = col < kALmax ? Ss[(4 * wave + rr) * kASS + col] : -INFINITY;
                mx[rr] = fmaxf(mx[rr], e[rr][u]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) mx[rr] = fmaxf(mx[rr], __shfl_xor(mx[rr], o));
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            sm[rr] = 0.f;
#pragma unroll
            for (int u = 0; u < 5; ++u) { e[rr][u] = __expf(e[rr][u] - mx[rr]); sm[rr] += e[rr][u]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) sm[rr] += __shfl_xor(sm[rr], o);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * wave + rr;
            const bool rok = q0 + row < L;
            const float inv = 1.f / sm[rr];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int col = lane + 64 * u;
                if (col < kALmax) {
                    const float p = rok ? e[rr][u] * inv : 0.f;
                    Ss[row * kASS + col] = p;
                    if (rok) a.P[(((int64_t)blockIdx.z * 8 + head) * kALmax + q0 + row) * kALmax + col] = p;
                }
            }
        }
    }
    __syncthreads();
    // ---- O = P V: this wave's key blocks; one power-of-two scale for all its value fragments
    float vm = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                vv[u][nt][e] *= (wave + 4 * u < 9 && (wave + 4 * u) * 32 + g * 8 + e < L) ? 1.f : 0.f;
                vm = fmaxf(vm, fabsf(vv[u][nt][e]));
            }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vm = fmaxf(vm, __shfl_xor(vm, o));
    int ev = (__builtin_bit_cast(int, vm) >> 23) & 0xff;
    ev = ev < 4 ? 4 : ev;
    const float vsc = __builtin_bit_cast(float, (257 - ev) << 23), vinv = __builtin_bit_cast(float, (ev - 3) << 23);
    f32x4 om[2] = {(f32x4)0.f, (f32x4)0.f}, oc[2] = {(f32x4)0.f, (f32x4)0.f};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int ks = wave + 4 * u;
        if (ks >= 9) break;
        float pv[8];
        lds8(pv, Ss + c * kASS + ks * 32 + g * 8);
        u32x4 phi, plo;
        split8(pv, phi, plo);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float sv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[e] = vv[u][nt][e] * vsc;
            u32x4 vhi, vlo;
            split8(sv, vhi, vlo);
            om[nt] = mfma(phi, vhi, om[nt]);
            oc[nt] = mfma(phi, vlo, oc[nt]);
            oc[nt] = mfma(plo, vhi, oc[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) part[wave][(4 * g + jj) * 32 + nt * 16 + c] = fmaf(oc[nt][jj], kLoInv, om[nt][jj]) * vinv;
    __syncthreads();
    for (int i = tid; i < 16 * 32; i += 256) {
        const int row = i >> 5, col = i & 31;
        if (q0 + row < L) a.out[ro + (int64_t)(q0 + row) * kD + head * 32 + col] = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
    }
}

template <class K>
int set_lds(K kernel, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
int img_off(int mat, int nn) { return (mat * 2 + nn) * kImgBytes; }
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: the "already set" mark is a bit per device (atomic: two host threads may launch
// first at the same time; setting the attribute twice is harmless, skipping it on a second GPU is a launch failure)
bool first_use_on_device(std::atomic<unsigned long long>& done, unsigned long long& bit) {
    int d = 0;
    (void)hipGetDevice(&d);
    bit = 1ull << (d & 63);
    return (done.load(std::memory_order_acquire) & bit) == 0;
}

}  // namespace

extern "C" {

#ifdef DPN_ENC_TIMELINE
void dpn_enc_debug_set_timeline(void* buf) { g_enc_timeline = static_cast<unsigned*>(buf); }
#endif

int64_t dpn_enc_pack_bytes(int n_mats) { return (int64_t)n_mats * 2 * kImgBytes; }

int dpn_enc_pack(int n_mats, const float* const* weights, void* packed, int* status_dev, void* stream) {
    if (n_mats <= 0 || n_mats > DPN_ENC_MAX_MATS || !weights || !packed) return -1;
    PackArgs a{};
    for (int i = 0; i < n_mats; ++i) {
        if (!weights[i]) return -1;
        a.W[i] = weights[i];
    }
    a.n = n_mats; a.out = static_cast<char*>(packed); a.status = status_dev;
    hipLaunchKernelGGL(dpn_enc_pack_kernel, dim3(n_mats * 2 * 16 * 8 * 64 / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

static int gemm16_launch(int n, const DpnGemm16Problem* problems, int n_jobs, const DpnColsumJob* jobs, int slices, float* partials, int reduce,
                         void* stream) {
    if (n < 0 || n > kWgMaxProblems || n_jobs < 0 || n_jobs > kWgMaxJobs || (n && !problems) || (n_jobs && !jobs) || n + n_jobs == 0) return -1;
    if (slices < 1 || (slices > 1 && !partials)) return -1;
    WgArgs a{};
    a.n = n; a.slices = slices; a.partials = partials;
    int max_k = 0;
    int64_t off = 0, per_max = 0, tiles = 0;
    for (int i = 0; i < n; ++i) {
        const DpnGemm16Problem& q = problems[i];
        if (!q.A || !q.B || !q.C || q.M <= 0 || q.N <= 0 || q.K <= 0 || q.ldc < q.N) return -1;
        a.p[i] = WgProblem{q.A, q.B, q.C, q.asum, q.bias, q.M, q.N, q.K, q.ldc, q.a_sm, q.a_sk, q.b_sn, q.b_sk, off};
        const int64_t per = (int64_t)q.M * q.N + (q.asum ? q.M : 0);
        off += (int64_t)slices * per;
        per_max = per_max > per ? per_max : per;
        const int tx = (q.N + kWgTile - 1) / kWgTile, ty = (q.M + kWgTile - 1) / kWgTile;
        if ((int64_t)tx * ty > (1 << 20) || tiles + (int64_t)tx * ty * slices > (1 << 30)) return -1;   // (1 << 20: the kernel's float division of a tile index by tx is exact below 4e6)
        a.start[i] = (int)tiles;
        a.tx[i] = tx; a.per_slice[i] = tx * ty; a.inv_tx[i] = 1.0f / (float)tx;
        tiles += (int64_t)tx * ty * slices;
        max_k = max_k > q.K ? max_k : q.K;
    }
    for (int i = n; i <= kWgMaxProblems; ++i) a.start[i] = (int)tiles;
    a.n_tiles = (int)tiles;
    a.rows_per_slice = ((max_k + slices - 1) / slices + 31) / 32 * 32;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].partial || !jobs[i].out_a || !jobs[i].out_b || jobs[i].n_blocks <= 0) return -1;
        a.job[i] = WgJob{jobs[i].partial, jobs[i].out_a, jobs[i].out_b, jobs[i].n_blocks, 0};
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dpn_wgrad16_kernel, dim3((unsigned)tiles + n_jobs), dim3(64 * kWgWaves), 0, s, a);
    if (slices > 1 && n > 0 && reduce)
        hipLaunchKernelGGL(dpn_wgrad16_reduce_kernel, dim3((unsigned)((per_max + 255) / 256), n), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

static int64_t gemm16_partial_floats(int n, const DpnGemm16Problem* problems, int slices) {
    if (n <= 0 || !problems || slices <= 1) return 0;
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) tot += (int64_t)slices * ((int64_t)problems[i].M * problems[i].N + (problems[i].asum ? problems[i].M : 0));
    return tot;
}
#ifdef DPN_EXPERIMENTS
int64_t dpn_gemm16_partial_floats(int n, const DpnGemm16Problem* problems, int slices) { return gemm16_partial_floats(n, problems, slices); }
int dpn_gemm16(int n, const DpnGemm16Problem* problems, int slices, float* partials, int reduce, void* stream) {
    return gemm16_launch(n, problems, 0, nullptr, slices, partials, reduce, stream);
}
#endif

static int wgrad_as_gemm16(int n, const DpnWgradProblem* problems, DpnGemm16Problem* out) {
    for (int i = 0; i < n; ++i) {
        const DpnWgradProblem& q = problems[i];
        if (!q.G || !q.X || !q.dW || q.M <= 0 || q.N <= 0 || q.rows <= 0 || q.ldg < q.M || q.ldx < q.N || q.ldw < q.N) return -1;
        out[i] = DpnGemm16Problem{q.G, q.X, q.dW, q.db, nullptr, q.M, q.N, q.rows, q.ldw, 1, q.ldg, 1, q.ldx};
    }
    return 0;
}

int64_t dpn_wgrad16_partial_floats(int n, const DpnWgradProblem* problems, int slices) {
    if (n <= 0 || n > kWgMaxProblems || !problems || slices <= 1) return 0;
    DpnGemm16Problem g[kWgMaxProblems];
    if (wgrad_as_gemm16(n, problems, g)) return 0;
    return gemm16_partial_floats(n, g, slices);
}

int dpn_wgrad16(int n, const DpnWgradProblem* problems, int n_jobs, const DpnColsumJob* jobs, int slices, float* partials, void* stream) {
    if (n < 0 || n > kWgMaxProblems || (n && !problems)) return -1;
    DpnGemm16Problem g[kWgMaxProblems];
    if (wgrad_as_gemm16(n, problems, g)) return -1;
    return gemm16_launch(n, g, n_jobs, jobs, slices, partials, 1, stream);
}

int dpn_attn16_fwd(const float* q, const float* k, const float* v, int L, int batch, float* out, float* P, void* stream) {
    if (!q || !k || !v || !out || !P || L <= 0 || L > kALmax || batch <= 0 || batch > 32767) return -1;
    Attn16Args a{q, k, v, out, P, L, 1.0f / sqrtf(32.0f)};
    hipLaunchKernelGGL(dpn_attn16_fwd_kernel, dim3((L + 15) / 16, 8, batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

int dpn_enc_prep(const DpnEncPrep* p, void* stream) {
    if (!p || p->n_mats < 0 || p->n_mats > DPN_ENC_MAX_MATS || (p->n_mats && (!p->weights || !p->packed))) return -1;
    if ((p->x != nullptr) != (p->xu != nullptr) || (p->x && (p->T <= 0 || p->C <= 0 || p->batch <= 0))) return -1;
    if (p->h && (p->batch <= 0 || !p->freqs_a || !p->out_a || p->n_a <= 0 || p->n_b < 0 || (p->n_b > 0 && (!p->freqs_b || !p->out_b)))) return -1;
    PrepArgs a{};
    for (int i = 0; i < p->n_mats; ++i) {
        if (!p->weights[i]) return -1;
        a.pack.W[i] = p->weights[i];
    }
    a.pack.n = p->n_mats; a.pack.out = static_cast<char*>(p->packed); a.pack.status = p->status_dev;
    a.pack_blocks = p->n_mats * 64;
    a.x = p->x; a.T = p->T; a.C = p->C; a.xu = p->xu;
    a.col_total = p->x ? (int64_t)p->batch * p->T * p->C * 3 : 0;
    a.col_blocks = (int)((a.col_total + 255) / 256);
    a.h = p->h; a.batch = p->batch; a.fa = p->freqs_a; a.na = p->n_a; a.oa = p->out_a; a.fb = p->freqs_b; a.nb = p->h ? p->n_b : 0; a.ob = p->out_b;
    const int pe_blocks = p->h ? (p->batch * (p->n_a + a.nb) + 255) / 256 : 0;
    const int blocks = a.pack_blocks + a.col_blocks + pe_blocks;
    if (blocks <= 0) return -1;
    hipLaunchKernelGGL(dpn_enc_prep_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

#ifdef DPN_EXPERIMENTS
int64_t dpn_conv16_kp(int K) { return K <= 0 ? 0 : ((int64_t)K + 31) / 32 * 32; }

int dpn_conv16_split(const float* x, int T, int C, int batch, const float* conv_w, int conv_n, void* xs, int32_t* xe, void* ws, int32_t* we, void* stream) {
    if (!x || !conv_w || !xs || !xe || !ws || !we || T <= 0 || C <= 0 || batch <= 0 || conv_n <= 0) return -1;
    ConvSplitArgs a{};
    a.x = x; a.T = T; a.C = C; a.batch = batch; a.conv_w = conv_w;
    a.Kp = (int)dpn_conv16_kp(3 * C);
    if (a.Kp > kSplitIters * 256) return -1;
    if (((int64_t)batch * T + 15) / 16 * 16 * a.Kp * 4 >= 0x7ffff000ll) return -1;      // (the images are addressed through one 2 GB buffer descriptor)
    a.xs_blocks = batch * T; a.ws_blocks = conv_n;
    a.xs = static_cast<_Float16*>(xs); a.ws = static_cast<_Float16*>(ws); a.xe = xe; a.we = we;
    hipLaunchKernelGGL(dpn_conv16_split_kernel, dim3(a.xs_blocks + a.ws_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

int dpn_conv16(const void* xs, const int32_t* xe, const void* ws, const int32_t* we, int M, int N, int Kp, int slices, float* parts, void* stream) {
    if (!xs || !xe || !ws || !we || !parts || M <= 0 || N <= 0 || Kp <= 0 || (Kp & 31) || slices <= 0) return -1;
    if ((int64_t)(M + 15) / 16 * 16 * Kp * 4 >= 0x7ffff000ll || (int64_t)(N + 15) / 16 * 16 * Kp * 4 >= 0x7ffff000ll) return -1;
    const int total = Kp / 32, per = (total + slices - 1) / slices;
    if ((int64_t)(slices - 1) * per >= total || per > kConvBlocks) return -1;    // every slice owns at least one block, at most kConvBlocks
    const int tn = (N + 63) / 64, tm = (M + 127) / 128;
    Conv16Args a{static_cast<const _Float16*>(xs), static_cast<const _Float16*>(ws), xe, we, M, N, Kp, per, slices, tn, tm, parts};
    const int64_t grid = 8ll * ((slices + 7) / 8) * tn * tm;                      // 8 XCD lanes x (slices per XCD) x tiles
    if (grid > 0x7fffffff) return -1;
    hipLaunchKernelGGL(dpn_conv16_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
#endif  // DPN_EXPERIMENTS

int dpn_enc_fwd(const DpnEncFwd* p, void* stream) {
    if (!p || !p->wpack || p->rows <= 0 || p->n_mats <= 0 || p->n_mats > DPN_ENC_MAX_MATS || p->next < 0 || p->next > 2) return -1;
    if (!p->tail && p->next != 1) return -1;
    const bool wide = p->row_tiles == 2;
    if (p->row_tiles != 1 && p->row_tiles != 2) return -1;
    FwdArgs a{};
    a.wpack = p->wpack; a.wbytes = dpn_enc_pack_bytes(p->n_mats); a.rows = p->rows;
    auto mat_ok = [&](int m) { return m >= 0 && m < p->n_mats; };
    int ni = 0;
    if (p->tail) {
        if (!mat_ok(p->m_o) || !mat_ok(p->m_c1) || !mat_ok(p->m_c2)) return -1;
        if (!p->o || !p->x || !p->bo || !p->g1 || !p->be1 || !p->bc1 || !p->bc2 || !p->g2 || !p->be2) return -1;
        if (!p->x1 || !p->xhat1 || !p->rstd1 || !p->pre || !p->act || !p->x2 || !p->xhat2 || !p->rstd2) return -1;
        a.img[ni++] = img_off(p->m_o, 0); a.img[ni++] = img_off(p->m_c1, 0); a.img[ni++] = img_off(p->m_c2, 0);
    } else if (p->emb_parts) {
        if (!p->emb_pos || !p->emb_te || !p->emb_out || p->emb_n_parts < 1 || p->emb_n_tok < 0 || (p->emb_n_tok > 0 && !p->emb_token) || p->emb_part_stride < 0) return -1;
        a.emb_parts = p->emb_parts; a.emb_bias = p->emb_bias; a.emb_pos = p->emb_pos; a.emb_te = p->emb_te; a.emb_token = p->emb_token; a.emb_out = p->emb_out;
        a.emb_part_stride = p->emb_part_stride; a.emb_n_parts = p->emb_n_parts; a.emb_n_tok = p->emb_n_tok;
    } else if (!p->xin) return -1;
    if (p->next == 1) {
        if (!mat_ok(p->m_n0) || !mat_ok(p->m_n1) || !mat_ok(p->m_n2) || !p->y0 || !p->y1 || !p->y2) return -1;
        a.img[ni++] = img_off(p->m_n0, 0); a.img[ni++] = img_off(p->m_n1, 0); a.img[ni++] = img_off(p->m_n2, 0);
    } else if (p->next == 2) {
        if (!mat_ok(p->m_n0) || !p->y0 || !p->gf || !p->bef || !p->xf || !p->xhatf || !p->rstdf) return -1;
        a.img[ni++] = img_off(p->m_n0, 0);
    }
    a.o = p->o; a.x = p->x; a.xin = p->xin;
    const float* vec[kNVecF] = {p->bo, p->g1, p->be1, p->bc1, p->bc2, p->g2, p->be2, p->gf, p->bef, p->bn0, p->bn1, p->bn2};
    for (int i = 0; i < kNVecF; ++i) a.vec[i] = vec[i];
    a.x1 = p->x1; a.xhat1 = p->xhat1; a.rstd1 = p->rstd1; a.pre = p->pre; a.act = p->act; a.x2 = p->x2; a.xhat2 = p->xhat2; a.rstd2 = p->rstd2;
    a.xf = p->xf; a.xhatf = p->xhatf; a.rstdf = p->rstdf; a.y0 = p->y0; a.y1 = p->y1; a.y2 = p->y2;
    ENC_TL_SET(a);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rpw = wide ? 32 : 16;
    a.n_main = (p->rows + rpw - 1) / rpw; a.n_img = ni;
    // the warm-up helpers pay while the working workgroups are few (one or a few field samples); a batch of fields warms the L2s by itself
    const dim3 grid(a.n_main + (a.n_main <= 64 && !getenv("DPN_ENC_NO_HELPERS") ? kHelpers : 0)), block(kThreads);
    static std::atomic<unsigned long long> done{0};
    unsigned long long dev_bit;
#define DPN_ENC_FWD_CASES(X) X(true, 0) X(true, 1) X(true, 2) X(false, 1)
    if (first_use_on_device(done, dev_bit)) {
#define X(T, N) if (set_lds(dpn_enc_fwd_kernel<T, N, 1>, Lds<1>::kBytes) || set_lds(dpn_enc_fwd_kernel<T, N, 2>, Lds<2>::kBytes)) return -2;
        DPN_ENC_FWD_CASES(X)
#undef X
        done.fetch_or(dev_bit, std::memory_order_release);
    }
#define X(T, N) if ((p->tail != 0) == T && p->next == N) { \
        if (wide) hipLaunchKernelGGL((dpn_enc_fwd_kernel<T, N, 2>), grid, block, Lds<2>::kBytes, s, a); \
        else hipLaunchKernelGGL((dpn_enc_fwd_kernel<T, N, 1>), grid, block, Lds<1>::kBytes, s, a); }
    DPN_ENC_FWD_CASES(X)
#undef X
    return (int)hipGetLastError();
}

int dpn_enc_bwd(const DpnEncBwd* p, void* stream) {
    if (!p || p->rows <= 0 || p->head < 0 || p->head > 2 || (p->row_tiles != 1 && p->row_tiles != 2)) return -1;
    if (p->head == 0 && !p->body) return -1;
    if (!p->wpack || p->n_mats <= 0 || p->n_mats > DPN_ENC_MAX_MATS) return -1;
    const bool wide = p->row_tiles == 2;
    BwdArgs a{};
    a.wpack = p->wpack; a.wbytes = dpn_enc_pack_bytes(p->n_mats); a.rows = p->rows;
    auto mat_ok = [&](int m) { return m >= 0 && m < p->n_mats; };
    int ni = 0;
    if (p->head == 1) {
        if (!mat_ok(p->m_h0) || !mat_ok(p->m_h1) || !mat_ok(p->m_h2) || !p->res || !p->dq || !p->dk || !p->dv) return -1;
        a.img[ni++] = img_off(p->m_h0, 1); a.img[ni++] = img_off(p->m_h1, 1); a.img[ni++] = img_off(p->m_h2, 1);
    } else if (p->head == 2) {
        if (!mat_ok(p->m_h0) || !p->dmeta || !p->xhatf || !p->rstdf || !p->gf || !p->partial_f) return -1;
        a.img[ni++] = img_off(p->m_h0, 1);
    } else if (!p->gin) return -1;
    if (p->body) {
        if (!mat_ok(p->m_c2) || !mat_ok(p->m_c1) || !mat_ok(p->m_o)) return -1;
        if (!p->xhat2 || !p->rstd2 || !p->pre || !p->xhat1 || !p->rstd1 || !p->g2 || !p->g1) return -1;
        if (!p->gs2 || !p->dpre || !p->gs1 || !p->dout || !p->partial2 || !p->partial1) return -1;
        a.img[ni++] = img_off(p->m_c2, 1); a.img[ni++] = img_off(p->m_c1, 1); a.img[ni++] = img_off(p->m_o, 1);
    } else if (!p->gx) return -1;
    a.res = p->res; a.dq = p->dq; a.dk = p->dk; a.dv = p->dv; a.dmeta = p->dmeta; a.xhatf = p->xhatf; a.rstdf = p->rstdf; a.gin = p->gin;
    a.xhat2 = p->xhat2; a.rstd2 = p->rstd2; a.pre = p->pre; a.xhat1 = p->xhat1; a.rstd1 = p->rstd1;
    a.vec[VB_G2] = p->g2; a.vec[VB_G1] = p->g1; a.vec[VB_GF] = p->gf;
    a.gs2 = p->gs2; a.dpre = p->dpre; a.gs1 = p->gs1; a.dout = p->dout; a.gx = p->gx;
    a.gx_head = p->gx_head; a.gx_head_rows = (p->gx_head && !p->body) ? p->gx_head_rows : 0;
    a.partial_f = p->partial_f; a.partial2 = p->partial2; a.partial1 = p->partial1;
    ENC_TL_SET(a);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rpw = wide ? 32 : 16;
    a.n_main = (p->rows + rpw - 1) / rpw; a.n_img = ni;
    const dim3 grid(a.n_main + (a.n_main <= 64 && ni > 0 && !getenv("DPN_ENC_NO_HELPERS") ? kHelpers : 0)), block(kThreads);
    static std::atomic<unsigned long long> done{0};
    unsigned long long dev_bit;
#define DPN_ENC_BWD_CASES(X) X(0, true) X(1, true) X(2, true) X(1, false)
    if (first_use_on_device(done, dev_bit)) {
#define X(H, B) if (set_lds(dpn_enc_bwd_kernel<H, B, 1>, Lds<1>::kBytes) || set_lds(dpn_enc_bwd_kernel<H, B, 2>, Lds<2>::kBytes)) return -2;
        DPN_ENC_BWD_CASES(X)
#undef X
        done.fetch_or(dev_bit, std::memory_order_release);
    }
#define X(H, B) if (p->head == H && (p->body != 0) == B) { \
        if (wide) hipLaunchKernelGGL((dpn_enc_bwd_kernel<H, B, 2>), grid, block, Lds<2>::kBytes, s, a); \
        else hipLaunchKernelGGL((dpn_enc_bwd_kernel<H, B, 1>), grid, block, Lds<1>::kBytes, s, a); }
    DPN_ENC_BWD_CASES(X)
#undef X
    return (int)hipGetLastError();
}

}  // extern "C"
