// MI355X (gfx950 / CDNA4) point path of the DeepPhysiNet physics-informed training step.
//
// Reference behaviour restated here (paths relative to /root/reference/DeepPhysiNet):
//   model/variable_net.py:49-87      VariableNet.forward (hyper-network coordinate MLP)
//   model/physics_net.py:49-54       six VariableNets share coord / coord_data
//   utils/position_encoding.py:35-50 SineCosPE
//   interface/interface_physics.py:90-95    gradient()  (autograd.grad, create_graph)
//   interface/interface_physics.py:97-185   six residual losses
//   interface/interface_physics.py:232-262  inverse_norm (+clip)
//   interface/interface_physics.py:322-332  encoding_coord
// DESIGN.md section 3 derives the restructured algorithm (reverse-sweep Jacobian, rank-1 fc.2,
// single-GEMM weight gradients) implemented below; oracle/kernel_model.py states it in torch.
//
// Kernel inventory
//   dpn_pack_*          fp32 weights -> MFMA-fragment-ordered bf16 (hi/lo) + permuted vectors
//   dpn_fwd_kernel      fused PE + MLP chain + reverse sweep + Jacobian contraction (activations never leave registers)
//   dpn_residual_kernel de-norm, clip, six residuals, wave-shuffle loss reduction, analytic cotangents
//   dpn_bwd_kernel      per-point cotangent streams -> operands of the weight-gradient reductions
//   dpn_wgrad_kernel    points-reduction GEMMs (split over point ranges)
//   dpn_finish_*        split reduction, un-permutation, rank-1 fc.2 gradients
//   dpn_sgemm_batch / dpn_sgemm_ln / dpn_sgemm     exact-fp32 MFMA GEMMs of the per-field tensors (encoder, heads), with the fused
//                       epilogues / LayerNorm prologue / ride-along reductions the encoder nodes need
//   dpn_gradnorm / dpn_adam                        global-norm clip + Adam over flat moment buffers
// No kernel in this file uses atomics: every reduction is fixed-order, the whole step is bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/dpn_hip.h"
#include "dpn_layout.h"

using namespace dpn;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short u16;

#define DEV __device__ __forceinline__

// Translation units.  The library is built from this file TWICE (deepphysinet_amd/build.py): DPN_TU=1 holds the point forward and
// backward kernels and is compiled with -mllvm -amdgpu-mfma-vgpr-form (accumulators in VGPRs: hipcc otherwise parks them in AGPRs and
// pays a v_accvgpr_read for every element an epilogue touches -- 1775 of the forward kernel's 6457 VALU instructions); DPN_TU=2 holds
// everything else (the weight-gradient kernel measures slower in that form).  Without DPN_TU the file is one unit.
#if !defined(DPN_TU) || DPN_TU == 1
#define DPN_HAS_POINT 1
#else
#define DPN_HAS_POINT 0
#endif
#if !defined(DPN_TU) || DPN_TU == 2
#define DPN_HAS_REST 1
#else
#define DPN_HAS_REST 0
#endif

// ------------------------------------------------------------------------------------------------ small helpers
typedef unsigned int u32;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) u32 u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

DEV u16 f2bf(float x) {                       // round-to-nearest-even, finite inputs (pack kernel)
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
DEV float bf2f(u16 b) { return __uint_as_float(((unsigned)b) << 16); }
DEV u32 pack2(float a, float b) {             // one v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(u32, __builtin_convertvector(v, bf16x2));
}
DEV float bf_lo(u32 w) { return __uint_as_float(w << 16); }
DEV float bf_hi(u32 w) { return __uint_as_float(w & 0xFFFF0000u); }

template <int NS>
struct Frag {                                 // one k-step operand fragment: 8 bf16 per lane as 4 packed words, hi [+ lo]
    u32x4 w[NS];
};
DEV bf16x8 as_bf(u32x4 w) { return __builtin_bit_cast(bf16x8, w); }
template <int NS>
DEV void frag_set2(Frag<NS>& f, const int p, float a, float b) {      // elements 2p, 2p+1
    const u32 hi = pack2(a, b);
    f.w[0][p] = hi;
    if constexpr (NS == 2) f.w[1][p] = pack2(a - bf_lo(hi), b - bf_hi(hi));
}

// (max(x, 0) used to be ONE inline-asm v_max_f32 here -- fmaxf() makes hipcc canonicalise its operand first, an extra v_max x,x per
//  element.  Inline asm is invisible to the compiler's hazard recogniser: when the scheduler made that v_max the FIRST reader of an MFMA
//  result, no wait states were inserted and it read the accumulator before the matrix core had written it (3 % errors in the hi+lo mode,
//  depending on unrelated code around it).  The ReLU is now a select on the compare that builds the mask bit; tools/mfma_hazard_check.py
//  scans the generated assembly for any inline-asm reader of a fresh MFMA result.)

DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// Cody-Waite reduction by pi/2 + minimax polynomials, |err| ~1e-7 for |theta| up to a few hundred.
DEV void sincos_precise(float th, float& s, float& c) {
    const float k = rintf(th * 0.63661977236758134f);
    float r = fmaf(k, -1.5707963705062866f, th);               // float(pi/2); the fma keeps k*hi exact
    r = fmaf(k, 4.371138828673793e-08f, r);                     // pi/2 - float(pi/2)
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
// NS == 1 (plain bf16 operands): the features are rounded to 8 mantissa bits anyway -> hardware v_sin/v_cos
template <int NS>
DEV void sincos_t(float th, float& s, float& c) {
    if constexpr (NS == 1) { s = __sinf(th); c = __cosf(th); }
    else sincos_precise(th, s, c);
}

// ------------------------------------------------------------------------------------------------ weight packing
struct PackArgs {
    DpnNetPtrs net[kNets];
    char* packed;
    int ns;
    int form;              // 0: the seven-GEMM stream (ring kernels), 1: the fused five-GEMM stream (dpn_fwd_tiles_kernel; dpn_layout.h)
    // a batch of fields in ONE launch (grid.y = kNets * n_fields): field f reads the hyper-network outputs of field 0 moved by f * heads_stride /
    // f * evec_stride floats (w1b1, w2b2 | evec; the static tensors are shared) and writes its packed block at packed + f * packed_stride bytes
    int n_fields;
    long heads_stride, evec_stride, packed_stride;
};
// the pointer table of (net, field): wave-uniform scalar arithmetic on a copy of the kernel argument
DEV DpnNetPtrs pack_net(const PackArgs& a, const int net, const int field) {
    DpnNetPtrs P = a.net[net];
    P.w1b1 += field * a.heads_stride;
    P.w2b2 += field * a.heads_stride;
    P.evec += field * a.evec_stride;
    return P;
}

DEV float pack_src(const DpnNetPtrs& P, int kb, int lane, int e) {
    const int i = lane & 31, h = lane >> 5;
    if (kb < kS1) {                                   // S0: w1, rows o, K = PE3 slots
        const int T = kb / 12, ks = kb % 12;
        return P.w1b1[(32 * T + i) * P.ld_w1b1 + pe3_ch(ks, h, e)];
    } else if (kb >= kS5) {                           // S5: w1^T rows rho (PE slots), K over o
        const int rel = kb - kS5, T = rel / 16, ks = rel % 16;
        return P.w1b1[chain_ch(ks, h, e) * P.ld_w1b1 + gpe_row_to_pe3_ch(32 * T + i)];
    }
    if (kb < kS2) {                                   // S1: w2 (8 tiles x 16 k-steps), then Wd (8 tiles x 12 k-steps)
        const int rel = kb - kS1;
        if (rel < 128) return P.w2b2[(32 * (rel / 16) + i) * P.ld_w2b2 + chain_ch(rel % 16, h, e)];
        const int r2 = rel - 128;
        return P.Wd[(32 * (r2 / 12) + i) * kPe + pe6_ch(r2 % 12, h, e)];
    } else if (kb < kS3) {                            // S2: W1 rows o
        const int rel = kb - kS2, T = rel / 16, ks = rel % 16;
        return P.W1[(32 * T + i) * kHidden + chain_ch(ks, h, e)];
    } else if (kb < kS4) {                            // S3: W1^T rows i, K over o
        const int rel = kb - kS3, T = rel / 16, ks = rel % 16;
        return P.W1[chain_ch(ks, h, e) * kHidden + (32 * T + i)];
    } else {                                          // S4: w2^T rows i, K over o
        const int rel = kb - kS4, T = rel / 16, ks = rel % 16;
        return P.w2b2[chain_ch(ks, h, e) * P.ld_w2b2 + (32 * T + i)];
    }
}

// vectors in [h][T][r] order (channel 32T + drow32(r,h)); u = W2^T wo; const0 = wo.bf2 + bo
// form 1 (dpn_layout.h): C2 = W1 cvec + bf1, A2 = w2^T wo, Bv = Wd^T wo (PE6 slot order), const0 += 2 wo.cvec
// EIGHT blocks per net (part = 0..7): block `part` owns the 32 vector entries idx = 32 part .. 32 part + 31; its 256 threads are 8 groups of 32, group og
// sums the reduction index o over [32 og, 32 og + 32) and the eight partial sums are joined in a fixed order through LDS.  (One block per net walking
// 256-long chains of dependent loads -- three of them in the fused form -- was the long pole of the launch: 31-40 us.)
constexpr int kVecParts = 8;
DEV void pack_vectors(const PackArgs& a, const DpnNetPtrs& P, char* packed_net, const int part) {      // packed_net: this (field, net)'s packed block
    float* vec = reinterpret_cast<float*>(packed_net + (long)kPackKB * 1024 * a.ns);
    const int tid = threadIdx.x, og = tid >> 5;
    const int idx = 32 * part + (tid & 31);
    const int h = idx >> 7, T = (idx >> 4) & 7, r = idx & 15;
    const int ch = 32 * T + drow32(r, h);
    const int c6 = idx < kPe ? pe6_ch(idx >> 4, (idx >> 3) & 1, idx & 7) : 0;                // Bv: idx = PE6 slot 16 ks + 8 h + e (idx < 192)
    __shared__ float red[3][8][32];
    float up = 0.f, a2 = 0.f, bv = 0.f;
    if (a.form == 1) {
#pragma unroll 8
        for (int o = 32 * og; o < 32 * og + 32; ++o) {
            const float w = P.wo[o];
            up = fmaf(w, P.W2[o * kHidden + ch], up);
            a2 = fmaf(w, P.w2b2[o * P.ld_w2b2 + ch], a2);                                    // (w2^T wo)[ch]
            bv = fmaf(w, P.Wd[o * kPe + c6], bv);                                            // (Wd^T wo)[pe6 channel of this slot]
        }
    } else {
#pragma unroll 8
        for (int o = 32 * og; o < 32 * og + 32; ++o) up = fmaf(P.wo[o], P.W2[o * kHidden + ch], up);
    }
    red[0][og][tid & 31] = up; red[1][og][tid & 31] = a2; red[2][og][tid & 31] = bv;
    __syncthreads();
    if (og == 0) {
        const int c = tid;
        auto sum8 = [&](const int q) { return ((red[q][0][c] + red[q][1][c]) + (red[q][2][c] + red[q][3][c])) + ((red[q][4][c] + red[q][5][c]) + (red[q][6][c] + red[q][7][c])); };
        vec[kVecB1 * 256 + idx] = P.w1b1[ch * P.ld_w1b1 + kPe];
        vec[kVecU * 256 + idx] = sum8(0);
        vec[kVecWo * 256 + idx] = P.wo[ch];
        if (a.form == 1) {
            // (C2 = W1 cvec + bf1 is written by the fused kernel's own tile role: it needs a pass over W1, i.e. the matrix cores)
            vec[kVecA2 * 256 + idx] = sum8(1);
            vec[kVecBv * 256 + idx] = idx < kPe ? sum8(2) : 0.f;
        } else {
            vec[kVecCvec * 256 + idx] = P.w2b2[ch * P.ld_w2b2 + kHidden] + P.bd[ch] + P.evec[ch];
            vec[kVecBf1 * 256 + idx] = P.bf1[ch];
            vec[kVecB2BdE_unused * 256 + idx] = 0.f;
        }
    }
    if (part != 0) return;
    // const0 = wo . (bf2 [+ 2 cvec]) + bo: block 0 of the net, all 256 threads
    __shared__ float red1[256];
    const float cv_nat = P.w2b2[tid * P.ld_w2b2 + kHidden] + P.bd[tid] + P.evec[tid];        // cvec[tid], natural order
    red1[tid] = P.wo[tid] * (P.bf2[tid] + (a.form == 1 ? 2.0f * cv_nat : 0.f));
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red1[tid] += red1[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        vec[kNumVecs * 256 + 0] = red1[0] + P.bo[0];
        vec[kNumVecs * 256 + 1] = (float)a.form; vec[kNumVecs * 256 + 2] = 0.f; vec[kNumVecs * 256 + 3] = 0.f;
    }
}

#if DPN_HAS_REST
__global__ __launch_bounds__(256) void dpn_pack_matrices_kernel(PackArgs a) {
    const int mcols = gridDim.x - kVecParts;                                          // block columns of matrix fragments, then kVecParts of vector blocks
    const int field = blockIdx.y / kNets, net = blockIdx.y - field * kNets;
    const int ns = a.ns;
    const DpnNetPtrs P = pack_net(a, net, field);
    char* packed_net = a.packed + field * a.packed_stride + (long)net * pack_bytes_per_net(ns);
    if ((int)blockIdx.x >= mcols) { pack_vectors(a, P, packed_net, blockIdx.x - mcols); return; }
    uint4* dst = reinterpret_cast<uint4*>(packed_net);
    const int total = kPackKB * 64;                   // (kb, lane) pairs (form 0; the fused form has its own kernel, dpn_pack_fused_kernel)
    for (int u = blockIdx.x * 256 + threadIdx.x; u < total; u += mcols * 256) {
        const int kb = u >> 6;
        const int lane = u & 63;
        u16 hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = pack_src(P, kb, lane, e);
            hi[e] = f2bf(x);
            lo[e] = f2bf(x - bf2f(hi[e]));
        }
        uint4 w;
        w.x = hi[0] | (hi[1] << 16); w.y = hi[2] | (hi[3] << 16); w.z = hi[4] | (hi[5] << 16); w.w = hi[6] | (hi[7] << 16);
        dst[(kb * ns) * 64 + lane] = w;
        if (ns == 2) {
            w.x = lo[0] | (lo[1] << 16); w.y = lo[2] | (lo[3] << 16); w.z = lo[4] | (lo[5] << 16); w.w = lo[6] | (lo[7] << 16);
            dst[(kb * ns + 1) * 64 + lane] = w;
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused form: products + packing in ONE launch
// A = W1 w2, B = W1 Wd (and C2 = W1 cvec + bf1) on the exact-fp32 matrix instruction, each 32 x 32 result tile split hi / lo and written straight into
// the fragment stream (A: rows o AND, transposed, rows j; B: rows o over PE6 slots) -- no fp32 scratch, no second launch (rounds before: a 24-problem
// dpn_sgemm_batch launch + dpn_pack_matrices_kernel, 17 + 10 us on the chain between the hyper-network heads and the forward kernel).
// Block roles per net (blockIdx.x): [0, 64) tiles of A | [64, 112) tiles of B (columns in PE6 SLOT order) | [112, 120) C2 | [120, 132) the w1 / w1^T
// fragments | [132, 140) the vector blocks (pack_vectors).
constexpr int kFusedBlocks = 64 + 48 + 8 + 12 + kVecParts;
DEV f32x16 pk_mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__global__ __launch_bounds__(256) void dpn_pack_fused_kernel(PackArgs a) {
    // Workgroups go to the eight XCDs round-robin in dispatch order: every XCD gets a CONTIGUOUS range of the (net, block) list, so that a net's W1 / w2 / Wd
    // are fetched into one or two L2s instead of all eight: 20.0 -> 16.2 us per launch (tools/pack_probe.py, profiles/round6_xcd_contiguous.txt).
#ifdef PACK_NO_XCD_REMAP
    const int field = blockIdx.y / kNets, net = blockIdx.y - field * kNets, bx = blockIdx.x, ns = a.ns;
#else
    // (a batch of fields: grid.y = kNets * n_fields, the list is (field, net, block): an XCD then works on whole fields)
    const int lin = blockIdx.x + kFusedBlocks * blockIdx.y, virt = (lin & 7) * (kFusedBlocks * kNets / 8 * a.n_fields) + (lin >> 3);
    static_assert(kFusedBlocks * kNets % 8 == 0, "remap");
    const int fnet = virt / kFusedBlocks, bx = virt - fnet * kFusedBlocks, ns = a.ns, field = fnet / kNets, net = fnet - field * kNets;
#endif
#ifdef PACK_ABL_MASK        // ablation builds (wrong results on purpose, timing only: tools/variant_build.py --unit=1 -DPACK_ABL_MASK=m): only the roles in bit mask m run.
    // Round 6, tools/pack_probe.py (us per launch): all 19.9-20.2; role 0 alone 7.7, 1: 6.5, 2: 8.0, 3: 5.7, 4: 5.1; {0,1} 12.1, {0,1,2} 17.7, {3,4} 6.9, {0,1,3,4} 16.8:
    // the three MFMA-tile roles do not hide behind each other.  Staging the block's W1 rows through LDS with coalesced loads (each lane fetches 16-byte pieces
    // of its own row today) was built and changed nothing (21.1-21.9 us): it is not the request pattern.  profiles/round6_pack_fused_roles.txt
    if (!((PACK_ABL_MASK >> (bx < 64 ? 0 : bx < 112 ? 1 : bx < 120 ? 2 : bx < 132 ? 3 : 4)) & 1)) return;
#endif
    const DpnNetPtrs P = pack_net(a, net, field);
    char* packed_net = a.packed + field * a.packed_stride + (long)net * pack_bytes_per_net(ns);
    if (bx >= 132) { pack_vectors(a, P, packed_net, bx - 132); return; }
    uint4* dst = reinterpret_cast<uint4*>(packed_net);
    auto put = [&](const int kb, const int lane, const float (&x)[8]) __attribute__((always_inline)) {
        u16 hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { hi[e] = f2bf(x[e]); lo[e] = f2bf(x[e] - bf2f(hi[e])); }
        uint4 w;
        w.x = hi[0] | (hi[1] << 16); w.y = hi[2] | (hi[3] << 16); w.z = hi[4] | (hi[5] << 16); w.w = hi[6] | (hi[7] << 16);
        dst[(kb * ns) * 64 + lane] = w;
        if (ns == 2) {
            w.x = lo[0] | (lo[1] << 16); w.y = lo[2] | (lo[3] << 16); w.z = lo[4] | (lo[5] << 16); w.w = lo[6] | (lo[7] << 16);
            dst[(kb * ns + 1) * 64 + lane] = w;
        }
    };
    if (bx >= 120) {                                          // w1 (kS0 .. kS1) and w1^T (kS5 .. kPackKB): 192 (kb) x 64 lanes over 12 blocks
        for (int u = (bx - 120) * 256 + threadIdx.x; u < 192 * 64; u += 12 * 256) {
            int kb = u >> 6;
            const int lane = u & 63;
            if (kb >= kS1) kb += kS5 - kS1;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = pack_src(P, kb, lane, e);
            put(kb, lane, x);
        }
        return;
    }
    // ---- a 32 x 32 tile of W1 . R, R = w2 (role 0), Wd with its columns in PE6 slot order (role 1), cvec as a single column (role 2)
    const int role = bx < 64 ? 0 : bx < 112 ? 1 : 2;
    const int rb = role == 0 ? bx : role == 1 ? bx - 64 : bx - 112;
    const int To = rb & 7, Tc = role == 2 ? 0 : rb >> 3;      // row tile (o), column tile (j / slot)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 31, kh = lane >> 5;
    __shared__ float red[4][16][64];
    __shared__ float tile[32][33];
    const float* W1r = P.W1 + (32 * To + col) * kHidden;
    int ccol = 0;                                             // this lane's column of R
    if (role == 0) ccol = 32 * Tc + col;
    else if (role == 1) { const int sl = 32 * Tc + col; ccol = pe6_ch(sl >> 4, (sl >> 3) & 1, sl & 7); }
    f32x16 acc = (f32x16)0.f;
    float av[32], bv[32];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k0 = 64 * wv + 8 * m + 4 * kh;
        const f32x4 q = *reinterpret_cast<const f32x4*>(W1r + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            av[4 * m + e] = q[e];
            if (role == 0) bv[4 * m + e] = P.w2b2[(long)k * P.ld_w2b2 + ccol];
            else if (role == 1) bv[4 * m + e] = P.Wd[k * kPe + ccol];
            else bv[4 * m + e] = col == 0 ? (P.w2b2[(long)k * P.ld_w2b2 + kHidden] + P.bd[k] + P.evec[k]) : 0.f;
        }
    }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) acc = pk_mfma_f32(av[kk], bv[kk], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wv + 4 * q;                                         // element (r, lane) of the tile: row drow32(r, kh), column col
        tile[drow32(r, kh)][col] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
    }
    __syncthreads();
    if (role == 2) {                                                      // C2[o] = (W1 cvec)[o] + bf1[o], in the vectors' [h][T][r] order
        if (threadIdx.x < 32) {
            const int o = 32 * To + threadIdx.x, w_ = o & 31;
            float* vec = reinterpret_cast<float*>(packed_net + (long)kPackKB * 1024 * ns);
            vec[kVecC2 * 256 + ((w_ >> 2) & 1) * 128 + (o >> 5) * 16 + (w_ & 3) + 4 * (w_ >> 3)] = tile[threadIdx.x][0] + P.bf1[o];
        }
        return;
    }
    // ---- the tile as fragments: thread = (form, k-step of the tile's pair, lane)
    const int form = threadIdx.x >> 7, ksl = (threadIdx.x >> 6) & 1, i = lane & 31, h = lane >> 5;
    float x[8];
    if (role == 1) {
        if (form == 1) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = tile[i][16 * ksl + 8 * h + e];                       // PE6 slot (h, e) of k-step 2 Tc + ksl
        put(kFB + To * 12 + 2 * Tc + ksl, lane, x);
    } else if (form == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = tile[i][16 * ksl + 8 * (e >> 2) + 4 * h + (e & 3)];  // A rows o, K = chain(h1): column j = chain_ch(ks, h, e)
        put(kFA + To * 16 + 2 * Tc + ksl, lane, x);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = tile[16 * ksl + 8 * (e >> 2) + 4 * h + (e & 3)][i];  // A^T rows j, K = chain(t2): row o = chain_ch(ks, h, e)
        put(kFAT + Tc * 16 + 2 * To + ksl, lane, x);
    }
}

#endif  // DPN_HAS_REST

// ------------------------------------------------------------------------------------------------ weight stream
// All four waves of a workgroup walk the same packed weight block chunk by chunk (one chunk = the A fragments of one 32-row
// output tile for 12 or 16 k-steps).  Chunks travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per
// wave-instruction, no staging registers, no ds_write pass) into a ring of 4 slots, three chunks ahead of the MFMAs.
// Per chunk: counted s_waitcnt vmcnt (my pieces of chunk c have landed; the DMAs of c+1, c+2 stay in flight), one raw
// s_barrier (everybody's pieces have landed, everybody is done with the slot refilled next), issue chunk c+3, multiply chunk c.
// vmcnt retires in order on gfx9-class hardware (the compiler's own counted waits rely on it); other VMEM traffic of the wave
// (saved-state stores) only makes the counted wait stricter, never weaker.
DEV void dma16(const char* gsrc_lane, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc_lane),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
}
template <int OFF> DEV void dma16_at(const char* gsrc_lane, char* lds_wave_base) {     // OFF: the instruction's immediate, added to both addresses
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc_lane),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, OFF, 0);
}
DEV void dma16_nt(const char* gsrc_lane, char* lds_wave_base) {       // read-once streams (weight-gradient operands): non-temporal hint
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc_lane),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 2);
}
DEV void dma4(const char* gsrc_lane, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc_lane),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 4, 0, 0);
}
template <int N> DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

DEV void wait_vmcnt_n(const int n) {        // n is a compile-time constant after unrolling: the switch folds to one s_waitcnt
    switch (n) {
        case 0: wait_vmcnt<0>(); break; case 1: wait_vmcnt<1>(); break; case 2: wait_vmcnt<2>(); break; case 3: wait_vmcnt<3>(); break; case 4: wait_vmcnt<4>(); break; case 5: wait_vmcnt<5>(); break; case 6: wait_vmcnt<6>(); break; case 7: wait_vmcnt<7>(); break; case 8: wait_vmcnt<8>(); break; case 9: wait_vmcnt<9>(); break; case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break; case 12: wait_vmcnt<12>(); break; case 13: wait_vmcnt<13>(); break; case 14: wait_vmcnt<14>(); break; case 15: wait_vmcnt<15>(); break; case 16: wait_vmcnt<16>(); break; case 17: wait_vmcnt<17>(); break; case 18: wait_vmcnt<18>(); break; case 19: wait_vmcnt<19>(); break; case 20: wait_vmcnt<20>(); break; case 21: wait_vmcnt<21>(); break; case 22: wait_vmcnt<22>(); break; case 23: wait_vmcnt<23>(); break; case 24: wait_vmcnt<24>(); break; case 25: wait_vmcnt<25>(); break; case 26: wait_vmcnt<26>(); break; case 27: wait_vmcnt<27>(); break; case 28: wait_vmcnt<28>(); break; case 29: wait_vmcnt<29>(); break; case 30: wait_vmcnt<30>(); break; case 31: wait_vmcnt<31>(); break; case 32: wait_vmcnt<32>(); break; case 33: wait_vmcnt<33>(); break; case 34: wait_vmcnt<34>(); break; case 35: wait_vmcnt<35>(); break; case 36: wait_vmcnt<36>(); break; case 37: wait_vmcnt<37>(); break; case 38: wait_vmcnt<38>(); break; case 39: wait_vmcnt<39>(); break; case 40: wait_vmcnt<40>(); break; case 41: wait_vmcnt<41>(); break; case 42: wait_vmcnt<42>(); break; case 43: wait_vmcnt<43>(); break; case 44: wait_vmcnt<44>(); break; case 45: wait_vmcnt<45>(); break; case 46: wait_vmcnt<46>(); break; case 47: wait_vmcnt<47>(); break; case 48: wait_vmcnt<48>(); break;
        default: wait_vmcnt<0>(); break;
    }
}
// k-steps of chunk c in stream order (dpn_layout.h): w1 8x12 | w2 8x16 | Wd 8x12 | W1 8x16 | W1^T 8x16 | w2^T 8x16 | w1^T 6x16
DPN_HD __attribute__((always_inline)) int stream_nk(int c, int end) { return (c < 0 || c >= end) ? 0 : (c < 8 ? 12 : c < 16 ? 16 : c < 24 ? 12 : 16); }

template <int NS>
struct Pipe {
    static constexpr int kRing = 4;
    static constexpr int kSlotBytes = 16 * 1024 * NS;
#ifdef DPN_FWD_PHASES
    u32 ph[6] = {0, 0, 0, 0, 0, 0}, pc0 = 0;      // experiment build: cycles in vmcnt wait / barrier / DMA issue / reads + MFMAs / last block / epilogue
#define DPN_PH_CLOCK(V) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); V = (u32)t_; } while (0)
#endif
    const char* g;       // global address of the next chunk to issue (wave-uniform)
    char* lds;
    int end;             // number of chunks this kernel may touch (54 forward, 24 backward)
    int wave, lane;

    DEV void init(const void* gsrc, char* lds_base, int end_chunks) {
        g = reinterpret_cast<const char*>(gsrc); lds = lds_base; end = end_chunks;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); lane = threadIdx.x & 63;
    }
#ifdef DPN_ABL_HALFDMA                        // ablation (wrong results): half of each chunk is fetched -- is the step time the DMA's?
    DEV static int dmas(int nk) { return nk * NS / 8; }
#elif defined(DPN_ABL_QUARTERDMA)
    DEV static int dmas(int nk) { return nk * NS / 16; }
#else
    DEV static int dmas(int nk) { return nk * NS / 4; }                 // DMA instructions per wave for a chunk of nk k-steps
#endif
    DEV void issue(const int c) {                                       // chunk c -> slot c % 4
        const int n = dmas(stream_nk(c, end));
        char* slot = lds + (c & (kRing - 1)) * kSlotBytes;
        // The slot is a byte image of the chunk; wave w copies the CONTIGUOUS quarter [w n KiB, (w+1) n KiB) of it, so that its pieces
        // differ only in the instruction's immediate offset (which moves the global and the LDS address alike, < 4 KiB): one
        // address pair + one M0 per four pieces instead of five address instructions per piece (a sixth of the kernel's issue slots).
        const char* src = g + wave * (n * 1024) + lane * 16;
        char* dst = slot + wave * (n * 1024);
        if (n > 0) dma16_at<0>(src, dst);
        if (n > 1) dma16_at<1024>(src, dst);
        if (n > 2) dma16_at<2048>(src, dst);
        if (n > 3) dma16_at<3072>(src, dst);
        if constexpr (NS == 2) {
            if (n > 4) dma16_at<0>(src + 4096, dst + 4096);
            if (n > 5) dma16_at<1024>(src + 4096, dst + 4096);
            if (n > 6) dma16_at<2048>(src + 4096, dst + 4096);
            if (n > 7) dma16_at<3072>(src + 4096, dst + 4096);
        }
        g += n * 4096;
    }
    DEV void prime() { issue(0); issue(1); issue(2); }
    // (Counting the saved-state stores of the last three epilogues into the allowed vmcnt -- they retire in order with the DMAs, so
    //  leaving them out makes the wait stricter than needed -- was measured with the timeline probe: no change, 3356 vs 3097 cycles per
    //  fc1 chunk in the hi+lo mode.  Not kept.)
    DEV void acquire(const int c) {                                     // after this, every wave may read chunk c from LDS
        __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from stretching live ranges across pipeline steps
#ifdef DPN_FWD_PHASES
        u32 c0, c1, c2, c3;
        DPN_PH_CLOCK(c0);
        if (pc0) ph[5] += c0 - pc0;             // since the end of the previous chunk's multiply: its epilogue
#endif
        wait_vmcnt_n(dmas(stream_nk(c + 1, end)) + dmas(stream_nk(c + 2, end)));
#ifdef DPN_FWD_PHASES
        DPN_PH_CLOCK(c1);
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // compiler-level ordering only: no s_waitcnt is emitted
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef DPN_FWD_PHASES
        DPN_PH_CLOCK(c2);
#endif
        issue(c + 3);
#ifdef DPN_FWD_PHASES
        DPN_PH_CLOCK(c3);
        ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; pc0 = c3;
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
    DEV unsigned buf(const int c) const {                               // LDS byte address of slot c % 4
        return (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (c & (kRing - 1)) * kSlotBytes;
    }
    DEV void drain() { wait_vmcnt<0>(); }                               // no DMA may outlive the workgroup's LDS allocation
};

// SWAP = false: Out[channel][point] (+)= W[channel][k] * Act[k][point]   (weights as A, chained layout)
// SWAP = true : Out[point][channel] (+)= Act[point][k] * W[channel][k]   (same packed weights as B: the result lands
//               channel-per-lane / points-in-registers, which is the K-operand layout of the weight-gradient GEMMs)
// The A fragments are read with inline-asm ds_read_b128: hipcc orders every LDS read it can see behind ALL outstanding
// LDS-DMA (s_waitcnt vmcnt(0)), which would drain the three chunks in flight at every step.  The reads of a chunk are
// issued in blocks of four k-steps, one block ahead of the MFMAs that consume them, with counted lgkmcnt waits
// (LDS operations retire in order; anything else on the counter only makes the wait stricter).
template <int NS>
struct WBlock { u32x4 w[4][NS]; };

template <int NS, int KS0>
DEV void lds_load_block(WBlock<NS>& b, unsigned addr) {
    if constexpr (NS == 1) {
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=&v"(b.w[0][0]), "=&v"(b.w[1][0]), "=&v"(b.w[2][0]), "=&v"(b.w[3][0])
                     : "v"(addr), "n"((KS0 + 0) * 1024), "n"((KS0 + 1) * 1024), "n"((KS0 + 2) * 1024), "n"((KS0 + 3) * 1024)
                     : "memory");
    } else {
        asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%10\n\tds_read_b128 %2, %8 offset:%11\n\tds_read_b128 %3, %8 offset:%12\n\t"
                     "ds_read_b128 %4, %8 offset:%13\n\tds_read_b128 %5, %8 offset:%14\n\tds_read_b128 %6, %8 offset:%15\n\tds_read_b128 %7, %8 offset:%16"
                     : "=&v"(b.w[0][0]), "=&v"(b.w[0][1]), "=&v"(b.w[1][0]), "=&v"(b.w[1][1]), "=&v"(b.w[2][0]), "=&v"(b.w[2][1]), "=&v"(b.w[3][0]), "=&v"(b.w[3][1])
                     : "v"(addr), "n"((KS0 * 2 + 0) * 1024), "n"((KS0 * 2 + 1) * 1024), "n"((KS0 * 2 + 2) * 1024), "n"((KS0 * 2 + 3) * 1024),
                       "n"((KS0 * 2 + 4) * 1024), "n"((KS0 * 2 + 5) * 1024), "n"((KS0 * 2 + 6) * 1024), "n"((KS0 * 2 + 7) * 1024)
                     : "memory");
    }
}
// wait until at most N LDS operations issued after this block are outstanding; the "+v" ties keep every consumer below the wait
template <int NS, int N>
DEV void lds_wait_block(WBlock<NS>& b) {
    if constexpr (NS == 1)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b.w[0][0]), "+v"(b.w[1][0]), "+v"(b.w[2][0]), "+v"(b.w[3][0]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(b.w[0][0]), "+v"(b.w[0][1]), "+v"(b.w[1][0]), "+v"(b.w[1][1]), "+v"(b.w[2][0]), "+v"(b.w[2][1]),
                     "+v"(b.w[3][0]), "+v"(b.w[3][1]) : "n"(N) : "memory");
    // hipcc would otherwise hoist register-only MFMAs above the asm wait.  VALU / SALU / VMEM / transcendental instructions
    // (the previous tile's epilogue) MAY cross: they are what fills the issue slots in the shadow of the MFMAs.
    __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x20 | 0x40 | 0x400);
}

template <int NS, bool SWAP, int KS0>
DEV void mma_block(const WBlock<NS>& b, const Frag<NS>* act, f32x16& acc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bf16x8 whi = as_bf(b.w[k][0]);
        if constexpr (NS == 2) {
            const bf16x8 wlo = as_bf(b.w[k][1]);
            if constexpr (SWAP) { acc = mfma(as_bf(act[KS0 + k].w[1]), whi, acc); acc = mfma(as_bf(act[KS0 + k].w[0]), wlo, acc); }
            else { acc = mfma(whi, as_bf(act[KS0 + k].w[1]), acc); acc = mfma(wlo, as_bf(act[KS0 + k].w[0]), acc); }
        }
        if constexpr (SWAP) acc = mfma(as_bf(act[KS0 + k].w[0]), whi, acc);
        else acc = mfma(whi, as_bf(act[KS0 + k].w[0]), acc);
    }
}

template <int NS, int NK, bool SWAP>
DEV void mma_chunk(unsigned slot_addr, const Frag<NS>* act, f32x16& acc) {
    static_assert(NK == 12 || NK == 16, "chunks are 12 or 16 k-steps");
    const unsigned addr = slot_addr + (threadIdx.x & 63) * 16;
    WBlock<NS> b0, b1;
    lds_load_block<NS, 0>(b0, addr);
    lds_load_block<NS, 4>(b1, addr);
    lds_wait_block<NS, 4 * NS>(b0);
    mma_block<NS, SWAP, 0>(b0, act, acc);
    lds_load_block<NS, 8>(b0, addr);
    lds_wait_block<NS, 4 * NS>(b1);
    mma_block<NS, SWAP, 4>(b1, act, acc);
    if constexpr (NK == 16) {
        lds_load_block<NS, 12>(b1, addr);
        lds_wait_block<NS, 4 * NS>(b0);
        mma_block<NS, SWAP, 8>(b0, act, acc);
        lds_wait_block<NS, 0>(b1);
        mma_block<NS, SWAP, 12>(b1, act, acc);
    } else {
        lds_wait_block<NS, 0>(b0);
        mma_block<NS, SWAP, 8>(b0, act, acc);
    }
}

// ------------------------------------------------------------------------------------------------ per-lane context
struct Lane {
    int lane, j, h;
    int64_t pt;        // global point index of this lane's column
    bool valid;
    float xi[3];       // normalised coordinates
    float fr32[16];    // freq32[8*(m>>2) + 4h + (m&3)]
    float fr16[8];     // freq16[8*(m>>2) + 4h + (m&3)]
    u32x4 idA, idB;       // identity B-operand fragments for the MFMA transposes (columns 0..15 / 16..31)
};

DEV void lane_init(Lane& L, const float* x, const float* y, const float* t, int64_t n, const float* freqs, const DpnGeometry& geo,
                   int64_t tile32) {
    L.lane = threadIdx.x & 63;
    L.j = L.lane & 31;
    L.h = L.lane >> 5;
    L.pt = tile32 * 32 + L.j;
    L.valid = L.pt < n;
    const int64_t pc = L.valid ? L.pt : (n - 1);
    L.xi[0] = L.xi[1] = L.xi[2] = 0.f;
    if (x) {
        L.xi[0] = x[pc] / geo.dx / geo.lon_m1;   // interface_physics.py:324-326 (two fp32 divisions, like the reference)
        L.xi[1] = y[pc] / geo.dy / geo.lat_m1;
        L.xi[2] = t[pc] / geo.pred_t_span;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) L.fr32[m] = freqs[8 * (m >> 2) + 4 * L.h + (m & 3)];
#pragma unroll
    for (int m = 0; m < 8; ++m) L.fr16[m] = freqs[32 + 8 * (m >> 2) + 4 * L.h + (m & 3)];
    // identity: column jj of a 32-column tile <- k-slot (h = (jj>>3)&1, e = jj&7) of k-step (jj>>4)
    const int mine = (((L.j >> 3) & 1) == L.h) ? (L.j & 7) : -1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const u32 v = ((mine == 2 * p) ? 0x3F80u : 0u) | ((mine == 2 * p + 1) ? 0x3F800000u : 0u);
        L.idA[p] = (L.j < 16) ? v : 0u;
        L.idB[p] = (L.j >= 16) ? v : 0u;
    }
}

// coordinate PE fragments; BWD builds Z0 = g*pe + sum_c gj[c] * dpe/dxi_c instead (backward stream)
template <int NS, bool BWD>
DEV void build_pe3(const Lane& L, Frag<NS>* act, float g, const float* gj) {
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
        const int c = ks >> 2;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float fr = L.fr32[4 * (ks & 3) + p];
            float s, co;
            sincos_t<NS>(L.xi[c] * fr, s, co);
            if constexpr (BWD) {
                const float gf = gj[c] * fr;
                frag_set2<NS>(act[ks], p, fmaf(g, s, gf * co), fmaf(g, co, -gf * s));
            } else {
                frag_set2<NS>(act[ks], p, s, co);
            }
        }
    }
}

// coordinate features supplied by the caller in the reference's channel order (f*6 + fn*3 + c), scaled by g
template <int NS>
DEV void load_pe3(const float* row, int h, Frag<NS>* act, float g) {
#pragma unroll
    for (int ks = 0; ks < 12; ++ks)
#pragma unroll
        for (int p = 0; p < 4; ++p) frag_set2<NS>(act[ks], p, g * row[pe3_ch(ks, h, 2 * p)], g * row[pe3_ch(ks, h, 2 * p + 1)]);
}

// data PE fragments (SineCosPE(6,16) of coord_data, variable_net.py:73), scaled by g
template <int NS>
DEV void build_pe6_ks(const Lane& L, const float* cd6, Frag<NS>* act, float g, const int ks) {
    const float v = cd6[ks >> 1];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float s, co;
        sincos_t<NS>(v * L.fr16[4 * (ks & 1) + p], s, co);
        frag_set2<NS>(act[ks], p, g * s, g * co);
    }
}
template <int NS>
DEV void build_pe6(const Lane& L, const float* cd6, Frag<NS>* act, float g) {
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) build_pe6_ks<NS>(L, cd6, act, g, ks);
}

// The permuted bias vectors live in LDS (filled once, before any DMA is in flight) and are read with inline-asm ds_read_b128
// for the same reason as the weight fragments: a read hipcc can see is ordered behind every outstanding LDS-DMA, and a
// global load it can see is waited for with vmcnt(0), which drains the DMA ring as well.
struct Vec16 { f32x4 q[4]; };
DEV void lds_read_vec16(Vec16& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v.q[0]), "=&v"(v.q[1]), "=&v"(v.q[2]), "=&v"(v.q[3]) : "v"(addr) : "memory");
}
DEV unsigned vec_addr(unsigned vec_base, int which, int h, int T) { return vec_base + (which * 256 + h * 128 + T * 16) * 4; }
DEV void acc_init_vec(f32x16& acc, unsigned vec_base, int which, int h, int T, float scale) {
    Vec16 v;
    lds_read_vec16(v, vec_addr(vec_base, which, h, T));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[4 * q] = scale * v.q[q][0]; acc[4 * q + 1] = scale * v.q[q][1]; acc[4 * q + 2] = scale * v.q[q][2]; acc[4 * q + 3] = scale * v.q[q][3];
    }
}
DEV float lds_read_f32(unsigned addr) {
    float r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    return r;
}

// ------------------------------------------------------------------------------------------------ K-layout operand matrices
// Every matrix the points-reduction GEMMs consume is stored "channel-per-lane": for each 32-point tile and each 32-column
// tile, lane (col = lane&31, h = lane>>5) owns 16 bf16 = the values of its column at points drow32(r,h), r = 0..15, i.e.
// exactly the A/B fragments of v_mfma_f32_32x32x16_bf16 with K = points (k-step kk = registers 8kk..8kk+7).  Columns are in
// SLOT order (column 32*ct + jj <-> k-step 2ct + (jj>>4), slot (h=(jj>>3)&1, e=jj&7)).
// A matrix held point-per-lane (as chained fragments) is brought into this layout by multiplying with an identity B
// operand: one extra MFMA per 16 channels instead of an LDS round trip.
struct KMat {
    char* base;
    int64_t tiles32;
    int ct_per_tile;      // column tiles: 8 (256 columns) or 6 (192)
};
// one 32-point tile of a matrix = [2 k-steps][ct_per_tile column tiles][64 lanes][16 B]: a linear image of what the
// weight-gradient kernel wants in LDS (its global -> LDS transfer is a plain 1-KB-per-wave-instruction DMA)
DEV char* kmat_ptr(const KMat& m, int net, int ns, int s, int64_t tile32, int ct, int lane, int kk) {
    return m.base + (((int64_t)net * ns + s) * m.tiles32 + tile32) * (m.ct_per_tile * 2048) + ((kk * m.ct_per_tile + ct) * 64 + lane) * 16;
}
DEV void store_d_as_k(const KMat& m, int net, int ns, int s, int64_t tile32, int ct, int lane, const f32x16& d) {
    uint4 a, b;
    a.x = pack2(d[0], d[1]); a.y = pack2(d[2], d[3]); a.z = pack2(d[4], d[5]); a.w = pack2(d[6], d[7]);
    b.x = pack2(d[8], d[9]); b.y = pack2(d[10], d[11]); b.z = pack2(d[12], d[13]); b.w = pack2(d[14], d[15]);
    // streaming stores: 0.4 GB of operands per launch pass through once and must not evict the L2-resident weight stream
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#ifdef TS_ABL_NOSTORE            // ablation build (timing only, wrong results on purpose): the packed values stay alive, nothing is written
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
    return;
#endif
    __builtin_nontemporal_store(u32x4_t{a.x, a.y, a.z, a.w}, reinterpret_cast<u32x4_t*>(kmat_ptr(m, net, ns, s, tile32, ct, lane, 0)));
    __builtin_nontemporal_store(u32x4_t{b.x, b.y, b.z, b.w}, reinterpret_cast<u32x4_t*>(kmat_ptr(m, net, ns, s, tile32, ct, lane, 1)));
}
// transpose-store the two fragments (k-steps 2ct, 2ct+1) that make up column tile ct; zero rows of invalid points
template <int NS, int NSTORE>
DEV void store_tile_k(const KMat& m, int net, int64_t tile32, int ct, const Lane& L, const Frag<NS>& f0, const Frag<NS>& f1, bool partial) {
#pragma unroll
    for (int s = 0; s < NSTORE; ++s) {
        u32x4 a0 = f0.w[s], a1 = f1.w[s];
        if (partial && !L.valid) { a0 = (u32x4)0u; a1 = (u32x4)0u; }
        f32x16 d = (f32x16)0.f;
        d = mfma(as_bf(a0), as_bf(L.idA), d);
        d = mfma(as_bf(a1), as_bf(L.idB), d);
        store_d_as_k(m, net, NSTORE, s, tile32, ct, L.lane, d);
    }
}

// saved-state / operand addressing ---------------------------------------------------------------------------
struct SavedView {       // written by dpn_fwd
    KMat T1;             // [6][NS] x 256 columns: t1 = m1 (.) (w2^T v), the cotangent in front of the first ReLU
    KMat M2;             // [6][1]  x 256 columns, relu-2 mask as bf16 0/1
    uint4* m1;           // [6][tiles32][64] lane-format bits of relu mask 1
};
// v = d out / d c is NOT saved: it is affine in the second mask, v = W1^T (m2 (.) u) + 2 wo (W1 = cat_fc1.fc.0.weight, u = fc.2.weight^T wo),
// so the two weight-gradient products it entered factor through the 0/1 matrix that is saved anyway,
//   sum_pt v (x) z1 = W1^T diag(u) (M2^T Z1) + 2 wo (x) colsum(Z1)      (likewise with G6),
// -- one 512-byte mask row per point and net instead of a 1-KB hi+lo row written once and read twice, two MFMAs per fragment pair instead
// of three, and the 256 x 256 factor applied once per net in fp32 (dpn_finish_vside_fc2_kernel) instead of once per point in split bf16.
DEV SavedView saved_view(void* base, int64_t n_pad, int ns) {
    SavedView s;
    char* b = reinterpret_cast<char*>(base);
    const int64_t mat = (int64_t)kNets * ns * n_pad * 512;
    const int64_t tiles32 = n_pad / 32;
    s.T1 = KMat{b, tiles32, 8};
    s.M2 = KMat{b + mat, tiles32, 8};
    s.m1 = reinterpret_cast<uint4*>(b + mat + (int64_t)kNets * n_pad * 512);
    return s;
}
DPN_HD int64_t saved_state_bytes(int64_t n_pad, int ns) { return (int64_t)kNets * ns * n_pad * 512 + (int64_t)kNets * n_pad * 512 + (int64_t)kNets * n_pad * 32; }
static int64_t saved_bytes(int64_t n_pad, int ns) { return saved_state_bytes(n_pad, ns); }

struct OperandView {     // written by dpn_bwd_points
    KMat Z1;             // [6][NS] x 256   (round 5: Z is not an operand any more, dpn_finish_gside_kernel)
    KMat Z0;             // [6][NS] x 192
    KMat PE6;            // [1][NS] x 192   per-POINT table of the data features (sin / cos of coord_data), written once by the net-0 workgroups
    float* gnet;         // [6][n_pad]      per-net cotangent of the normalised field, zero for padding points
};
// Round 5: G6 = g pe6 (the Y operand of S2 = M2^T G6) is no longer written per point AND NET: it is a per-point table times a per-net scalar, so
// dpn_wgrad_kernel forms it in registers from the table fragment it has just read (seven VALU instructions per element beside the MFMAs) and the table
// -- 768 B per point in the hi+lo mode, shared by the six nets -- stays in the memory-side cache.  Stage 1 writes 2 560 -> 1 792 B per point and net.
// (The same was built for Z0 = g pe3 + gJ_c d pe3 / d xi_c -- the partner column of one pe3 table through a DPP move -- and measured: stage 1 91 us
// instead of 130, but product 3's tile loop no longer fits 256 registers beside its 128 accumulators and two X planes, each reload of a spilled value
// waits for the LDS-DMA ring as well, and dpn_wgrad_kernel went 198 -> 337 us; profiles/round5_operand_tables.txt.  Z0 stays a per-net operand.)
DEV OperandView operand_view(void* base, int64_t n_pad, int ns) {
    OperandView o;
    char* b = reinterpret_cast<char*>(base);
    const int64_t m256 = (int64_t)kNets * ns * n_pad * 512, m192 = (int64_t)kNets * ns * n_pad * 384, t192 = (int64_t)ns * n_pad * 384;
    const int64_t tiles32 = n_pad / 32;
    o.Z1 = KMat{b, tiles32, 8};
    o.Z0 = KMat{b + m256, tiles32, 6};
    o.PE6 = KMat{b + m256 + m192, tiles32, 6};
    o.gnet = reinterpret_cast<float*>(b + m256 + m192 + t192);
    return o;
}
static int64_t operand_bytes(int64_t n_pad, int ns) { return (int64_t)kNets * ns * n_pad * 512 + (int64_t)(kNets + 1) * ns * n_pad * 384 + (int64_t)kNets * n_pad * 4 + 1024; }

// ------------------------------------------------------------------------------------------------ forward + Jacobian
struct FwdArgs {
    const float *x, *y, *t, *coord_data, *freqs, *pe_in;
    int64_t n, n_pad;
    DpnGeometry geo;
    const char* packed;
    float* out_n;
    float* jac_n;
    void* saved;
    const float* ref;        // [N][6] added to the output in place of coord_data (VariableNet.forward's own ref_data argument), else null
#ifdef DPN_TIMELINE
    unsigned* timeline;      // [blocks][6 nets][8 wave slots][64]: s_memtime (low word) at the start of every pipeline step (experiment build only)
#endif
};

// Experiment build (-DDPN_TIMELINE, tools/timeline_build.py): every wave keeps the shader clock at the start of each pipeline step in one
// VGPR (lane i <- stamp i, v_writelane: no memory traffic, no counters touched besides the s_memtime's own lgkmcnt, which is empty at a
// step boundary) and writes the register out at the end.  Stamp 0 = kernel entry, 1 = ring primed / prologue done, 2 + C = step C, 62 = exit.
#ifdef DPN_TIMELINE
#define DPN_STAMP(I)                                                                                          \
    do {                                                                                                      \
        unsigned long long t_;                                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(tl) : "s"((u32)t_), "n"(I));                         \
    } while (0)
#else
#define DPN_STAMP(I) do { } while (0)
#endif

#ifdef DPN_FWD_PHASES
#define DPN_PH_AFTER_MMA do { u32 c4_; DPN_PH_CLOCK(c4_); pipe.ph[3] += c4_ - pipe.pc0; pipe.pc0 = c4_; } while (0)
#else
#define DPN_PH_AFTER_MMA do { } while (0)
#endif
// One pipeline step on chunk C: make it readable (and put chunk C+3 in flight), multiply it, and run the epilogue of the
// PREVIOUS tile in the shadow of these MFMAs (it only touches that tile's accumulator).
#define DPN_STEP(C, NK, SWAP, ACT, ACC, EPI_PREV)                                    \
    do {                                                                             \
        DPN_STAMP(2 + (C));                                                          \
        pipe.acquire(C);                                                             \
        mma_chunk<NS, (NK), (SWAP)>(pipe.buf(C), (ACT), (ACC));                      \
        DPN_PH_AFTER_MMA;                                                            \
        EPI_PREV;                                                                    \
    } while (0)

template <int NS>
__global__ __launch_bounds__(256, 1) void dpn_fwd_kernel(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds_w[Pipe<NS>::kRing * Pipe<NS>::kSlotBytes];

    const int net = blockIdx.y;
    const int wave = threadIdx.x >> 6;
    const int64_t tile32 = (int64_t)blockIdx.x * 4 + wave;
    const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
    __shared__ __attribute__((aligned(16))) float lds_vec_store[kNumVecs * 256 + 4];
#ifdef DPN_TIMELINE
    u32 tl = 0;
    DPN_STAMP(0);
#endif
    {   // permuted fp32 vectors of this net -> LDS (published by the barrier below, before the first DMA is issued)
        const float* gv = reinterpret_cast<const float*>(pk + (long)kPackKB * 1024 * NS);
        for (int i = threadIdx.x; i < kNumVecs * 256 + 4; i += 256) lds_vec_store[i] = gv[i];
    }
    const unsigned lds_vec = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds_vec_store;
    Lane L;
    lane_init(L, a.x, a.y, a.t, a.n, a.freqs, a.geo, tile32);
    const int h = L.h;
    const bool partial = (tile32 * 32 + 32 > a.n);
    const int64_t pc = L.valid ? L.pt : (a.n - 1);
    float cd6[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) cd6[c] = a.coord_data[pc * 6 + c];
    const float ref_data = (a.ref ? a.ref : a.coord_data)[pc * 6 + net];
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    const bool save = a.saved != nullptr;

    Pipe<NS> pipe;
    __syncthreads();
    pipe.init(pk, lds_w, 54);
    pipe.prime();
    DPN_STAMP(1);

    f32x16 acc[8];
    u32 m1w[4] = {0u, 0u, 0u, 0u};
    Frag<NS> actA[16], actB[16];

    // ---------------- L1: pre1 = w1 . pe + b1 ; h1 = relu -> actA ; relu mask -> m1w
    auto epi1 = [&](const int T) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = acc[T][r], p1 = acc[T][r + 1];
            const bool on0 = p0 > 0.f, on1 = p1 > 0.f;
            m1w[T >> 1] |= (on0 ? (1u << (16 * (T & 1) + r)) : 0u) | (on1 ? (2u << (16 * (T & 1) + r)) : 0u);
            frag_set2<NS>(actA[2 * T + (r >> 3)], (r & 7) >> 1, on0 ? p0 : 0.f, on1 ? p1 : 0.f);
        }
        // pin the mask word HERE: left alone, the scheduler postpones the compares to the first use of m1w (after fc1) and keeps the
        // tile's 16 pre-activations alive until then -- in AGPRs for bf16, in SCRATCH for the hi+lo mode, whose reloads drain the DMA ring
        asm volatile("" : "+v"(m1w[T >> 1]));
    };
    {
        Frag<NS> pe[12];
        if (a.pe_in) load_pe3<NS>(a.pe_in + pc * kPe, h, pe, 1.0f);     // caller-encoded coordinates (PhysicsNet.forward surface)
        else build_pe3<NS, false>(L, pe, 0.f, nullptr);
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            acc_init_vec(acc[T], lds_vec, kVecB1, h, T, 1.0f);
            if (T == 0) DPN_STEP(T, 12, false, pe, acc[T], (void)0);
            else DPN_STEP(T, 12, false, pe, acc[T], epi1(T - 1));
        }
        epi1(7);
    }
    // ---------------- L2 + data: c = w2 . h1 + Wd . pe6 + (b2 + bd + e) -> actB ; cdot = wo . c
    float cdot = 0.f;
    auto epi2 = [&](const int T) __attribute__((always_inline)) {
        Vec16 wv;
        lds_read_vec16(wv, vec_addr(lds_vec, kVecWo, h, T));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cdot = fmaf(wv.q[q][0], acc[T][4 * q], cdot); cdot = fmaf(wv.q[q][1], acc[T][4 * q + 1], cdot);
            cdot = fmaf(wv.q[q][2], acc[T][4 * q + 2], cdot); cdot = fmaf(wv.q[q][3], acc[T][4 * q + 3], cdot);
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) frag_set2<NS>(actB[2 * T + (r >> 3)], (r & 7) >> 1, acc[T][r], acc[T][r + 1]);
    };
    {
        // pass A: all eight tiles of w2 . h1 (h1 = actA dies here); pass B: + Wd . pe6, epilogue one tile late
        // the data PE (96 sin/cos + packing per lane) is built two k-steps per tile UNDER the MFMAs of pass A, whose tiles have no
        // epilogue of their own, and pinned there (left to the scheduler it lands in one block in front of pass B)
        Frag<NS> pe6[12];
        auto pe6_part = [&](const int T) __attribute__((always_inline)) {
            if (NS == 1 && T < 6) {                              // hi+lo fragments: twice the registers, the early build spills
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    build_pe6_ks<NS>(L, cd6, pe6, 1.0f, 2 * T + j);
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) asm volatile("" : "+v"(pe6[2 * T + j].w[s2]));
                }
            }
        };
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            acc_init_vec(acc[T], lds_vec, kVecCvec, h, T, 1.0f);
            DPN_STEP(8 + T, 16, false, actA, acc[T], pe6_part(T));
        }
        if constexpr (NS == 2) build_pe6<NS>(L, cd6, pe6, 1.0f);
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            if (T == 0) DPN_STEP(16 + T, 12, false, pe6, acc[T], (void)0);
            else DPN_STEP(16 + T, 12, false, pe6, acc[T], epi2(T - 1));
        }
        epi2(7);
    }
    // ---------------- fc1: pre2 = W1 . c + bf1 ; a = relu ; out = u.a + 2 wo.c + const ; t2 = m2 (.) u -> actA ; M2 -> saved
    const float const0 = lds_read_f32(lds_vec + kNumVecs * 256 * 4);
    // the packed stream carries its form behind const0 (dpn_pack_vectors): this kernel multiplies the seven-GEMM stream.  A buffer packed in the fused
    // form (or a DPN_FWD_KERNEL switch flipped between pack and launch) would give silently wrong fields: trap instead.
    if (lds_read_f32(lds_vec + (kNumVecs * 256 + 1) * 4) != 0.f) __builtin_trap();
    float adot = 0.f;
    auto epi3 = [&](const int T) __attribute__((always_inline)) {
        Frag<1> mk0, mk1;
        Vec16 uv;
        lds_read_vec16(uv, vec_addr(lds_vec, kVecU, h, T));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float uu[4] = {uv.q[q][0], uv.q[q][1], uv.q[q][2], uv.q[q][3]};
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const int r = 4 * q + i;
                const float p0 = acc[T][r], p1 = acc[T][r + 1];
                const bool on0 = p0 > 0.f, on1 = p1 > 0.f;
                const float t0 = on0 ? uu[i] : 0.f, t1 = on1 ? uu[i + 1] : 0.f;      // t2 = m2 (.) u
                adot = fmaf(p0, t0, adot);                                           // relu(p) * u == p * (m2 * u): no separate max
                adot = fmaf(p1, t1, adot);
                frag_set2<NS>(actA[2 * T + (r >> 3)], (r & 7) >> 1, t0, t1);
                const u32 mw = (on0 ? 0x3F80u : 0u) | (on1 ? 0x3F800000u : 0u);
                if (r < 8) mk0.w[0][(r & 7) >> 1] = mw; else mk1.w[0][(r & 7) >> 1] = mw;
            }
        }
        // pin t2 and the mask words in VGPRs HERE: left alone, the scheduler keeps the 128 compare results as lane masks in SGPRs
        // (spilling them through v_writelane / v_readlane) and materialises every select in one 1000-instruction block after the GEMM
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) asm volatile("" : "+v"(actA[2 * T].w[s2]), "+v"(actA[2 * T + 1].w[s2]));
        asm volatile("" : "+v"(mk0.w[0]), "+v"(mk1.w[0]));
        // (Saving the mask as bit words instead -- the compares' lane masks ARE the transposed words, 1 KB per tile and net instead of
        //  16 KB, v_writelane into one VGPR -- and expanding them to 0 / 1 fragments in dpn_wgrad_kernel was built and measured: this
        //  kernel unchanged, dpn_wgrad_kernel +4.5 % (the expansion sits in front of the MFMAs of its product's workgroups), step +1.1 %
        //  hi+lo and +3.2 % single bf16 on the same box.  Not kept: tools/experiments/m2_bit_masks.patch.)
        if (save) store_tile_k<1, 1>(sv.M2, net, tile32, T, L, mk0, mk1, partial);
    };
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        acc_init_vec(acc[T], lds_vec, kVecBf1, h, T, 1.0f);
        if (T == 0) DPN_STEP(24 + T, 16, false, actB, acc[T], (void)0);
        else DPN_STEP(24 + T, 16, false, actB, acc[T], epi3(T - 1));
    }
    epi3(7);
    {
        float o = adot + 2.0f * cdot;
        o += __shfl_xor(o, 32);
        if (L.valid && h == 0) a.out_n[L.pt * 6 + net] = o + const0 + ref_data;   // + ref_data (variable_net.py:86)
    }
    if (save) sv.m1[((int64_t)net * (a.n_pad / 32) + tile32) * 64 + L.lane] = make_uint4(m1w[0], m1w[1], m1w[2], m1w[3]);
    if (!save && !a.jac_n) { pipe.drain(); return; }
    // ---------------- reverse sweep: v = W1^T t2 + 2 wo -> actB (not saved: SavedView)
    auto epiv = [&](const int T) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) frag_set2<NS>(actB[2 * T + (r >> 3)], (r & 7) >> 1, acc[T][r], acc[T][r + 1]);
    };
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        acc_init_vec(acc[T], lds_vec, kVecWo, h, T, 2.0f);
        if (T == 0) DPN_STEP(32 + T, 16, false, actA, acc[T], (void)0);
        else DPN_STEP(32 + T, 16, false, actA, acc[T], epiv(T - 1));
    }
    epiv(7);
    // ---------------- y = w2^T v ; t1 = m1 (.) y -> actA (+ saved T1)
    auto epiy = [&](const int T) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const u32 bits = m1w[T >> 1] >> (16 * (T & 1) + r);
            frag_set2<NS>(actA[2 * T + (r >> 3)], (r & 7) >> 1, (bits & 1u) ? acc[T][r] : 0.f, (bits & 2u) ? acc[T][r + 1] : 0.f);
        }
        if (save) store_tile_k<NS, NS>(sv.T1, net, tile32, T, L, actA[2 * T], actA[2 * T + 1], partial);
    };
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        acc[T] = (f32x16)0.f;
        if (T == 0) DPN_STEP(40 + T, 16, false, actB, acc[T], (void)0);
        else DPN_STEP(40 + T, 16, false, actB, acc[T], epiy(T - 1));           // the w1^T chunks always exist: prefetching them is harmless
    }
    epiy(7);
    if (!a.jac_n) { pipe.drain(); return; }
    // ---------------- gpe = w1^T t1 (6 tiles), contracted with d(pe)/d(xi) in registers
    float jc[3] = {0.f, 0.f, 0.f};
    auto epij = [&](const int T) __attribute__((always_inline)) {
        if (a.pe_in) {
            // caller-encoded coordinates (PhysicsNet.forward surface): hand back d out / d pe_in itself, [N][6][192] in the
            // reference's channel order, and let the caller's autograd chain it through its own encoding (generic path, scattered stores)
            if (L.valid) {
                float* o = a.jac_n + (L.pt * 6 + net) * kPe;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[pe3_ch(2 * T + (r >> 3), h, r & 7)] = acc[T][r];
            }
            return;
        }
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {                // register pair (sin, cos) of one angle
            const int r = 2 * rp;
            const int ks = 2 * T + (r >> 3), p = (r & 7) >> 1, c = ks >> 2;
            const float fr = L.fr32[4 * (ks & 3) + p];
            float s, co;
            sincos_t<NS>(L.xi[c] * fr, s, co);
            jc[c] = fmaf(acc[T][r], fr * co, jc[c]);
            jc[c] = fmaf(acc[T][r + 1], -fr * s, jc[c]);
        }
    };
#pragma unroll
    for (int T = 0; T < 6; ++T) {
        acc[T] = (f32x16)0.f;
        if (T == 0) DPN_STEP(48 + T, 16, false, actA, acc[T], (void)0);
        else DPN_STEP(48 + T, 16, false, actA, acc[T], epij(T - 1));
    }
    epij(5);
    pipe.drain();
#ifdef DPN_TIMELINE
    DPN_STAMP(62);
#ifdef DPN_FWD_PHASES
    if (L.lane >= 56 && L.lane < 62) tl = pipe.ph[L.lane - 56];         // slots 56..61: the phase sums of this wave
#endif
    if (a.timeline) a.timeline[(((int64_t)blockIdx.x * kNets + net) * 8 + wave) * 64 + L.lane] = tl;
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) jc[c] += __shfl_xor(jc[c], 32);
    if (L.valid && h == 0 && !a.pe_in) {
        float* o = a.jac_n + (L.pt * 6 + net) * 3;
        o[0] = jc[0] / a.geo.lon_m1 / a.geo.dx;          // chain rule through x/dx/(lon-1), in the reference's backward order
        o[1] = jc[1] / a.geo.lat_m1 / a.geo.dy;
        o[2] = jc[2] / a.geo.pred_t_span;
    }
}

#if DPN_HAS_POINT && defined(DPN_EXPERIMENT_FWD2)
#include "../../tools/experiments/dpn_fwd2_eight_waves.h"       // shelved eight-wave variant (measured slower; see its header and DESIGN.md)
#endif

#if DPN_HAS_REST
// g_pe[n][c] = sum_k g_out[n][k] * gpe[n][k][c]: the cotangent of caller-encoded coordinates (PhysicsNet.forward backward w.r.t. coord_x)
__global__ __launch_bounds__(192) void dpn_contract_gpe_kernel(const float* g_out, const float* gpe, int64_t n, float* g_pe) {
    const int64_t pt = blockIdx.x;
    const int c = threadIdx.x;
    float s_ = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) s_ = fmaf(g_out[pt * 6 + k], gpe[(pt * 6 + k) * kPe + c], s_);
    g_pe[pt * kPe + c] = s_;
}

// ------------------------------------------------------------------------------------------------ residuals
struct ResArgs {
    const float *out_n, *jac_n, *f;
    int64_t n;
    DpnGeometry geo;
    DpnPhysics ph;
    const float *gl, *gtot;
    double* loss_sums;
    float *g_out, *g_jxi;
};

// the criterion's per-element value rho(r) and slope rho'(r) (DpnPhysics.criterion): every criterion the reference's builder offers is a function of
// input - target alone, so `loss(lhs, 0)` (:104) and the gas law's `loss(p, rho R T)` (:179) are both mean(rho(r))
DEV float crit_value(const float r, const int kind, const float beta) {
    const float ar = fabsf(r);
    if (kind == DPN_CRIT_L1) return ar;
    return ar < beta ? 0.5f * r * r / beta : ar - 0.5f * beta;             // nn.SmoothL1Loss
}
DEV float crit_slope(const float r, const int kind, const float beta) {
    if (kind == DPN_CRIT_MSE) return 2.0f * r;
    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
    if (kind == DPN_CRIT_L1) return sg;
    return fabsf(r) < beta ? r / beta : sg;
}

DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void dpn_residual_kernel(ResArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < a.n;
    const int64_t ic = valid ? i : a.n - 1;
    constexpr float C_P = 1005.f, L_V = 2.5e6f, R_V = 461.5f, R_D = 287.f, EPS = 1e-6f;
    float val[6], msk[6], J[6][3];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float v = a.out_n[ic * 6 + k] * a.ph.std[k] + a.ph.mean[k];       // inverse_norm (interface_physics.py:250)
        float dv = a.ph.std[k];                                          // d val / d out
        if (a.ph.sq_on[k]) { dv = 2.f * v * a.ph.std[k]; v = v * v + a.ph.sq_add[k]; }   // three-factor min_max: squared, shifted (:244-247)
        float m = 1.f;
        if (a.ph.clip_on[k]) {                                           // torch.clip: gradient passes where lo <= v <= hi
            m = (v >= a.ph.clip_lo[k] && v <= a.ph.clip_hi[k]) ? 1.f : 0.f;
            v = fminf(fmaxf(v, a.ph.clip_lo[k]), a.ph.clip_hi[k]);
        }
        val[k] = v; msk[k] = m * dv;
#pragma unroll
        for (int c = 0; c < 3; ++c) J[k][c] = a.jac_n[(ic * 6 + k) * 3 + c] * msk[k];
    }
    const float u = val[0], v = val[1], p = val[2], T = val[3], q = val[4], rho = val[5];
    const float fc = a.f[ic];
    const float omega = J[2][2] + u * J[2][0] + v * J[2][1];
    const float A = J[3][2] + u * J[3][0] + v * J[3][1];
    const float B = J[4][2] + u * J[4][0] + v * J[4][1];
    const float tc = T - 273.15f;
    const float e_s = 6.112f * expf(17.67f * tc / (tc + 243.5f)) * 100.f;                 // get_qs :181-185
    const float qs_raw = 0.622f * e_s / (p - 0.378f * e_s);
    const float q_s = (qs_raw != qs_raw) ? qs_raw : fmaxf(qs_raw, 1e-6f);          // torch.maximum propagates NaN (:166)
    const float delta = (omega < 0.f && q >= q_s) ? 1.f : 0.f;
    const float R = (1.f + 0.608f * q) * R_D;
    const float Fv = (L_V * R - C_P * R_V * T) / (C_P * R_V + T * T + L_V * L_V * q_s) * q_s * T;   // precedence as written :161-163
    const float K = delta * Fv / (p + EPS);
    float r[6];
    r[0] = J[0][2] + u * J[0][0] + v * J[0][1] + J[2][0] / rho - fc * v;                   // :97-104
    r[1] = J[1][2] + u * J[1][0] + v * J[1][1] + J[2][1] / rho + fc * u;                   // :106-114
    r[2] = J[5][2] + u * J[5][0] + v * J[5][1] + rho * J[0][0] + rho * J[1][1];            // :116-124
    r[3] = C_P * A - omega / (rho + EPS) + L_V * B;                                        // :126-144
    r[4] = -omega * K + B;                                                                 // :146-175
    r[5] = p - rho * (1.f + 0.608f * q) * R_D * T;                                         // :177-179
    if (a.loss_sums) {
        // fp64 partial sums (residual^2 spans 1e-20..1e+20 across equations): wave shuffle tree, then the four waves of the block
        // in a fixed order -> one [6] row per block.  No atomics: dpn_residual_finish adds the rows in a fixed order, so the
        // losses are run-to-run deterministic (and 3.5k serialised fp64 atomics are gone from the step).
        __shared__ double wsum[4][6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            double s = 0.0;
            if (valid) s = a.ph.criterion == DPN_CRIT_MSE ? (double)r[e] * (double)r[e] : (double)crit_value(r[e], a.ph.criterion, a.ph.beta);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][e] = s;
        }
        __syncthreads();
        if (threadIdx.x < 6)
            a.loss_sums[(int64_t)blockIdx.x * 6 + threadIdx.x] =
                ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
    }
    if (!a.g_out || !valid) return;
    float g[6];
    const float inv_n = a.ph.reduce_sum ? 1.0f : 1.0f / (float)a.n;       // reduction "sum": the criterion does not divide by the number of points
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        // upstream weight of loss e: cotangent of losses[e] plus cotangent of the in-kernel total (1 when neither is given)
        const float w = (a.gl || a.gtot) ? ((a.gl ? a.gl[e] : 0.f) + (a.gtot ? a.gtot[0] : 0.f)) : 1.f;
        g[e] = a.ph.factor[e] * w * crit_slope(r[e], a.ph.criterion, a.ph.beta) * inv_n;     // d(factor*mean(rho(r)))/dr ; MSE: 2 r
    }
    const float ir = 1.f / rho, ire = 1.f / (rho + EPS);
    float gv[6], gJ[6][3];
    gv[0] = g[0] * J[0][0] + g[1] * (J[1][0] + fc) + g[2] * J[5][0] + g[3] * (C_P * J[3][0] - J[2][0] * ire + L_V * J[4][0]) + g[4] * (-J[2][0] * K + J[4][0]);
    gv[1] = g[0] * (J[0][1] - fc) + g[1] * J[1][1] + g[2] * J[5][1] + g[3] * (C_P * J[3][1] - J[2][1] * ire + L_V * J[4][1]) + g[4] * (-J[2][1] * K + J[4][1]);
    gv[2] = g[4] * omega * delta * Fv / ((p + EPS) * (p + EPS)) + g[5];
    gv[3] = -g[5] * rho * (1.f + 0.608f * q) * R_D;
    gv[4] = -g[5] * rho * 0.608f * R_D * T;
    gv[5] = -g[0] * J[2][0] * ir * ir - g[1] * J[2][1] * ir * ir + g[2] * (J[0][0] + J[1][1]) + g[3] * omega * ire * ire - g[5] * (1.f + 0.608f * q) * R_D * T;
    gJ[0][0] = g[0] * u + g[2] * rho; gJ[0][1] = g[0] * v;             gJ[0][2] = g[0];
    gJ[1][0] = g[1] * u;              gJ[1][1] = g[1] * v + g[2] * rho; gJ[1][2] = g[1];
    gJ[2][0] = g[0] * ir - g[3] * u * ire - g[4] * u * K;
    gJ[2][1] = g[1] * ir - g[3] * v * ire - g[4] * v * K;
    gJ[2][2] = -g[3] * ire - g[4] * K;
    gJ[3][0] = g[3] * C_P * u; gJ[3][1] = g[3] * C_P * v; gJ[3][2] = g[3] * C_P;
    const float gq = g[3] * L_V + g[4];
    gJ[4][0] = gq * u; gJ[4][1] = gq * v; gJ[4][2] = gq;
    gJ[5][0] = g[2] * u; gJ[5][1] = g[2] * v; gJ[5][2] = g[2];
    const float sc[3] = {1.f / a.geo.lon_m1 / a.geo.dx, 1.f / a.geo.lat_m1 / a.geo.dy, 1.f / a.geo.pred_t_span};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float go = gv[k] * msk[k];
        if (a.ph.sq_on[k] && msk[k] != 0.f) {
            // the squared form is not affine: J = jac * d val / d out depends on `out` as well -- d J / d out = jac * 2 std^2 (inside the clip bounds)
            float t = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) t = fmaf(gJ[k][c], a.jac_n[(i * 6 + k) * 3 + c], t);
            go = fmaf(t, 2.f * a.ph.std[k] * a.ph.std[k], go);
        }
        a.g_out[i * 6 + k] = go;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.g_jxi[(i * 6 + k) * 3 + c] = gJ[k][c] * msk[k] * sc[c];
    }
}

__global__ __launch_bounds__(384) void dpn_residual_finish_kernel(const double* partials, int64_t n, DpnPhysics ph, float* losses) {
    // (a batch of fields: one workgroup per field, its block rows and its seven outputs side by side)
    partials += (int64_t)blockIdx.x * ((n + 255) / 256) * 6;
    losses += (int64_t)blockIdx.x * 7;
    // partials: [ceil(n/256)][6] block rows of dpn_residual.  Wave e adds equation e (lane l takes rows l, l+64, ... in order, then a
    // fixed shuffle tree).  losses[0..5]: the six scaled terms; losses[6]: their sum in the reference's order of additions (:301)
    __shared__ float l[6];
    const int e = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nblk = (n + 255) / 256;
    double s = 0.0;
    for (int64_t b = lane; b < nblk; b += 64) s += partials[b * 6 + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { l[e] = (float)((double)(float)(ph.reduce_sum ? s : s / (double)n) * (double)ph.factor[e]); losses[e] = l[e]; }   // .float() * factor (:104)
    __syncthreads();
    if (threadIdx.x == 0) losses[6] = ((((l[0] + l[1]) + l[3]) + l[2]) + l[4]) + l[5];   // montion_u + montion_v + energy + continous + vapor + gas
}

__global__ __launch_bounds__(256) void dpn_smooth_l1_kernel(const float* out_n, const float* labels, int64_t n, float beta, float scale,
                                                            double* loss_sum, float* g_out, int accumulate, const float* scale_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // one element of [N][6]
    const bool valid = i < n * 6;
    float l = 0.f;
    if (valid) {
        const float d = out_n[i] - labels[i];
        const float ad = fabsf(d);
        l = (ad < beta) ? 0.5f * d * d / beta : ad - 0.5f * beta;    // nn.SmoothL1Loss(beta), weights_loss.py:15-19
        if (g_out) {
            const float gv = scale * (scale_dev ? scale_dev[0] : 1.f) * ((ad < beta) ? d / beta : (d > 0.f ? 1.f : -1.f));
            g_out[i] = accumulate ? g_out[i] + gv : gv;             // accumulate: joins the PDE cotangent of the same points
        }
    }
    if (!loss_sum) return;
    double s = (double)l;                          // one fp64 partial per block, fixed order, no atomics: the caller adds the blocks up
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss_sum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

#endif  // DPN_HAS_REST

// ------------------------------------------------------------------------------------------------ backward, stage 1
struct BwdArgs {
    const float *x, *y, *t, *coord_data, *freqs, *pe_in;
    int64_t n, n_pad;
    DpnGeometry geo;
    const char* packed;
    const float *g_out, *g_jxi;
    const float* g_scale;    // device scalar multiplied into both cotangent streams as they are read (an upstream cotangent on unit-cotangent streams), or null
    void* saved;
    void* operands;
    int reverse;             // tile-split kernel: walk nets / tiles in the opposite order (DPN_BWD_ORDER=reverse: a cache-residency probe)
#ifdef DPN_TIMELINE
    unsigned* timeline;      // experiment build: [net][workgroup][4 waves][48] shader clocks at the phase boundaries of dpn_bwd_tiles_kernel
#endif
};

template <int NS>
__global__ __launch_bounds__(256, 1) void dpn_bwd_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds_w[Pipe<NS>::kRing * Pipe<NS>::kSlotBytes];
    const int net = blockIdx.y;
    const int wave = threadIdx.x >> 6;
    const int64_t tile32 = (int64_t)blockIdx.x * 4 + wave;
    const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
    __shared__ __attribute__((aligned(16))) float lds_vec_store[kNumVecs * 256 + 4];
    {
        const float* gv = reinterpret_cast<const float*>(pk + (long)kPackKB * 1024 * NS);
        for (int i = threadIdx.x; i < kNumVecs * 256 + 4; i += 256) lds_vec_store[i] = gv[i];
    }
    const unsigned lds_vec = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds_vec_store;
    Lane L;
    lane_init(L, a.x, a.y, a.t, a.n, a.freqs, a.geo, tile32);
    const int h = L.h;
    const int64_t pc = L.valid ? L.pt : (a.n - 1);
    float cd6[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) cd6[c] = a.coord_data[pc * 6 + c];
    const float gsc = a.g_scale ? a.g_scale[0] : 1.0f;
    const float g = L.valid ? gsc * a.g_out[pc * 6 + net] : 0.f;   // padding points carry a zero cotangent: every operand row is zero
    float gj[3] = {0.f, 0.f, 0.f};
    if (a.g_jxi && L.valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gj[c] = gsc * a.g_jxi[(pc * 6 + net) * 3 + c];
    }
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    OperandView ov = operand_view(a.operands, a.n_pad, NS);
    const int64_t tiles32 = a.n_pad / 32;
    const uint4 m1v = sv.m1[((int64_t)net * tiles32 + tile32) * 64 + L.lane];
    const u32 m1w[4] = {m1v.x, m1v.y, m1v.z, m1v.w};
    if (h == 0) ov.gnet[(int64_t)net * a.n_pad + tile32 * 32 + L.j] = g;

    Pipe<NS> pipe;
    __syncthreads();
    pipe.init(pk, lds_w, 8);                      // the w1 chunks only (round 5: the w2 / Wd products of Z are gone)
    pipe.prime();

    f32x16 acc[8];
    Frag<NS> actA[16];
#ifdef DPN_TIMELINE
    u32 tl = 0;          // the step macro stamps; the backward kernel does not report (experiment build only)
#endif
    // ---------------- Z0 = g * pe + sum_c gJ_c * dpe/dxi_c ; Z1 = m1 (.) (w1 Z0 + g b1)
    auto epi1 = [&](const int T) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const u32 bits = m1w[T >> 1] >> (16 * (T & 1) + r);
            frag_set2<NS>(actA[2 * T + (r >> 3)], (r & 7) >> 1, (bits & 1u) ? acc[T][r] : 0.f, (bits & 2u) ? acc[T][r + 1] : 0.f);
        }
        store_tile_k<NS, NS>(ov.Z1, net, tile32, T, L, actA[2 * T], actA[2 * T + 1], false);
    };
    {
        Frag<NS> z0[12];
        if (net == 0) {                           // the per-point table of the data features (OperandView): written once for the six nets
            build_pe6<NS>(L, cd6, z0, 1.0f);
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) store_tile_k<NS, NS>(ov.PE6, 0, tile32, ct, L, z0[2 * ct], z0[2 * ct + 1], false);
        }
        if (a.pe_in) load_pe3<NS>(a.pe_in + pc * kPe, h, z0, g);
        else build_pe3<NS, true>(L, z0, g, gj);
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) store_tile_k<NS, NS>(ov.Z0, net, tile32, ct, L, z0[2 * ct], z0[2 * ct + 1], false);
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            acc_init_vec(acc[T], lds_vec, kVecB1, h, T, g);
            if (T == 0) DPN_STEP(T, 12, false, z0, acc[T], (void)0);
            else DPN_STEP(T, 12, false, z0, acc[T], epi1(T - 1));
        }
        epi1(7);
        pipe.drain();
    }
}

#if DPN_HAS_POINT
#include "dpn_fwd_tiles.h"                                       // tile-split forward / backward kernels (the hi+lo mode's default)
#include "dpn_fwd_pp.h"                                          // round 6: the ping-pong form of the tile-split forward (one persistent 8-wave workgroup per CU)
#include "dpn_fwd_tiles_persist.h"                               // round 6: persistent workgroups, the next item's features built by the wave that idles through the last GEMM
#endif

#if DPN_HAS_REST
// ------------------------------------------------------------------------------------------------ backward, stage 2
// Points-reduction GEMMs  D[so][si] = sum_pt X[pt][so] * Y[pt][si]  for the four products of a net:
//   P0: G    = M2^T Z    (256x256)  + mvec = M2^T g, q  = Z^T 1
//   P1: S1   = M2^T Z1   (256x256)  + mv1  = M2^T g, q1 = Z1^T 1, sum g      dw2 = W1^T diag(u) S1 + 2 wo (x) q1   (dpn_finish_vside_fc2_kernel;
//   P2: S2   = M2^T G6   (256x192)  +                q6 = G6^T 1             dWd = W1^T diag(u) S2 + 2 wo (x) q6    v is affine in m2: SavedView)
//   P3: dw1  = T1^T Z0   (256x192)  + db1 = T1^T g
// grid = (sum of the four products' point-range counts, 6 nets): each workgroup owns the whole output of its product for its range of
// 32-point tiles (SplitPlan below says how many ranges each product is cut into) and writes one partial sum per range;
// dpn_finish_* add the ranges in a fixed order.  Operands are already MFMA fragments in global memory (K-layout, written by
// dpn_fwd / dpn_bwd_points), so a tile travels global -> LDS as a plain byte image.
constexpr int kPartFloats = 65536 + 49152 * 2 + 7 * 256;           // per (split, net)
DPN_HD int part_off(int prod) { return prod == 1 ? 0 : prod == 2 ? 65536 : 114688; }
constexpr int kPartVec = 163840;                                    // -, -, mv1, db1, [sum g], q1, q6 (slots 0, 1 were product 0's mvec, q)

struct WgradArgs {
    int64_t n, n_pad;
    int splits[4];          // point ranges per product (SplitPlan)
    void* saved;
    void* operands;
    float* partials;
    int reverse;            // every workgroup walks its point range from the end (DPN_WGRAD_ORDER=reverse: a cache-residency probe; the sums' order changes)
#ifdef DPN_WGRAD_PHASES
    unsigned* phases;       // experiment build: [net][workgroup][8 waves][8]: cycles in wait / barrier / issue / compute, tiles
#endif
};
#ifdef DPN_WGRAD_PHASES
#define DPN_WG_CLOCK(V) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); V = (u32)t_; } while (0)
#else
#define DPN_WG_CLOCK(V) do { } while (0)
#endif


// one workgroup = 8 waves (2 x 4): wave (wm, wn) owns rows 128wm.. and columns 64wn.. of the product.  The X and Y fragments
// of each 32-point tile are moved global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction, no staging
// registers) into a ring of RING slots, RING-1 tiles ahead; every operand byte is fetched from HBM exactly once per product.
// Synchronisation per tile: counted s_waitcnt vmcnt (this wave's pieces of the tile have landed) + one raw s_barrier (all
// pieces have landed AND everybody is done with the slot that is refilled next).
//
// The body is compiled once per product (WgradShape): the slot holds exactly that product's planes -- X: one plane for the 0/1 mask,
// NS for T1; Y: NS planes of 8 or 6 column tiles -- so every wave issues the same number of 1-KB pieces per tile without dummy loads
// (48 / 48 / 40 / 56 pieces in the hi+lo mode: 6 / 6 / 5 / 7 per wave), and the ring is as deep as 160 KB of LDS allow for THAT slot:
// three slots for the mask products in the hi+lo mode where the common 66-KB slot allowed two.  The kernel's time per tile is the
// latency of a tile's loads under load, not its bytes (measured: halving a product's bytes with the ring depth unchanged changed
// nothing), so the number of tiles in flight is what counts.
template <int NS, int PROD>
struct WgradShape {
    static constexpr int nct = PROD < 2 ? 8 : 6;                        // column tiles of Y (Z, Z1: 256 columns; G6, Z0: 192)
    static constexpr int nsx = PROD == 3 ? NS : 1;                      // planes of X (the mask has no lo part)
    static constexpr int kX = nsx * 16384, kYPlane = nct * 2048, kY = NS * kYPlane;
    static constexpr int kPieces = (kX + kY) / 1024;
    static constexpr int kIssue = (kPieces + 7) / 8;                    // 1-KB pieces per wave per tile; if they do not divide (single bf16, 192 columns:
    static constexpr int kPad = kIssue * 8 - kPieces;                   // 28 pieces), the last waves re-read one fixed kilobyte into a dummy area
    static constexpr int kGOff = kX + kY;                               // 8 per-wave copies of g[64]
    static constexpr int kPadOff = kGOff + 8 * 256;
    static constexpr int kSlot = kPadOff + (kPad ? 1024 : 0);
    static constexpr int PER_TILE = kIssue + 1;                         // DMA instructions per wave per tile
    static constexpr int RING = (160 * 1024) / kSlot < 5 ? (160 * 1024) / kSlot : 5;
};
template <int NS>
constexpr int wgrad_lds_bytes() {
    int m = 0;
    const int v[3] = {WgradShape<NS, 1>::RING * WgradShape<NS, 1>::kSlot, WgradShape<NS, 2>::RING * WgradShape<NS, 2>::kSlot,
                      WgradShape<NS, 3>::RING * WgradShape<NS, 3>::kSlot};
    for (int k = 0; k < 3; ++k) m = v[k] > m ? v[k] : m;
    return m;
}

template <int NS, int PROD>
DEV void wgrad_body(const WgradArgs& a, char* lds, const int split, const int net) {
    using S = WgradShape<NS, PROD>;
    constexpr int nct = S::nct, nsx = S::nsx, kSlot = S::kSlot, RING = S::RING, PER_TILE = S::PER_TILE, ncol = nct * 32;
    const int64_t tiles = a.n_pad / 32;
    const int64_t per = (tiles + a.splits[PROD] - 1) / a.splits[PROD];
    const int64_t t0 = (int64_t)split * per;
    int64_t t1 = t0 + per < tiles ? t0 + per : tiles;
    if (t1 < t0) t1 = t0;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wave tile: 256-column products (P1) 2 x 4 waves of 128 rows x 64 columns (4 x 2 MFMA tiles); 192-column products (P2, P3) 4 x 2 waves of 64 rows x
    // 96 columns (2 x 3 tiles).  Until the end of round 5 the 192-column products kept the 2 x 4 arrangement with the wn = 3 waves idle: six waves with
    // eight tiles each on THREE SIMDs (wave w runs on SIMD w % 4) -- tools/wgrad_phase_probe.py showed the multiply phase of the wm = 1 waves at 4 200
    // cycles per tile against 2 550 for their SIMD partners and SIMD 3 idle.  Eight waves with six tiles each: 12 tile-products per SIMD instead of 16.
#ifdef DPN_WGRAD_2X4_ONLY                                                   // A/B build (tools/variant_build.py wg2x4 -DDPN_WGRAD_2X4_ONLY): the former arrangement
    constexpr bool kWide = true;
#else
    constexpr bool kWide = nct == 8;
#endif
    constexpr int MT = kWide ? 4 : 2, NT = kWide ? 2 : 3;
    const int wm = kWide ? wave >> 2 : wave >> 1, wn = kWide ? wave & 3 : wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const bool active = wn * NT < nct;                                  // (always true but in the A/B build, whose wn = 3 waves have no columns at 192)
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    OperandView ov = operand_view(a.operands, a.n_pad, NS);
    const char* xb = (PROD == 3) ? sv.T1.base : sv.M2.base;                                      // 8 column tiles
    static_assert(PROD >= 1 && PROD <= 3, "products: 1 = M2^T Z1, 2 = M2^T G6, 3 = T1^T Z0");
    const char* yb = (PROD == 1) ? ov.Z1.base : (PROD == 2) ? ov.PE6.base : ov.Z0.base;   // nct column tiles; product 2: the per-point TABLE (no net index)
    const int64_t ynet = (PROD == 2) ? 0 : net;
    const float* gnet = ov.gnet + (int64_t)net * a.n_pad;

    // every wave issues PER_TILE DMA instructions per tile: piece q = wave + 8*j of the (X planes, Y planes) image, + its own g copy.
    // (Letting only the four waves with wm == tile & 1 issue a tile -- twice the pieces each, in the shadow of their SIMD partners'
    //  MFMAs -- was measured: the issue phase shrinks from 1 900 to 1 100 cycles per tile and the barrier wait grows by as much,
    //  360 us against 364 us in the hi+lo mode.  Not kept.)
    // piece q = wave + 8 j of the slot image: where it comes from (address of tile 0, bytes per tile) and where it goes -- worked out once,
    // so that issuing a tile is straight-line code
    const char* pbase[S::kIssue];
    int pstride[S::kIssue], pdst[S::kIssue];
#pragma unroll
    for (int j = 0; j < S::kIssue; ++j) {
        const int q = wave + 8 * j;
        if (S::kPad && q >= S::kPieces) {                               // a cache hit after the first time
            pbase[j] = xb + ((int64_t)net * nsx * tiles + (t0 < tiles ? t0 : 0)) * 16384; pstride[j] = 0; pdst[j] = S::kPadOff;
        } else if (q < nsx * 16) {
            pbase[j] = xb + ((int64_t)net * nsx + q / 16) * tiles * 16384 + (q % 16) * 1024; pstride[j] = 16384; pdst[j] = q * 1024;
        } else {
            const int qy = q - nsx * 16;
            pbase[j] = yb + (ynet * NS + qy / (2 * nct)) * tiles * S::kYPlane + (qy % (2 * nct)) * 1024; pstride[j] = S::kYPlane; pdst[j] = q * 1024;
        }
    }
    auto issue = [&](int64_t tile, int slot) __attribute__((always_inline)) {
        char* sl = lds + slot * kSlot;
        if (a.reverse && t1 > t0) tile = t0 + (t1 - 1 - tile);
#pragma unroll
        for (int j = 0; j < S::kIssue; ++j) {
            // read-once operand streams carry the non-temporal hint; the per-point pe6 table of product 2 is read by six nets' workgroups and should
            // stay in the memory-side cache (PMC, round 5: with the hint on it the table came from HBM six times: 1 052 MB against 890 algorithmic)
            if (PROD == 2 && wave + 8 * j >= nsx * 16) dma16(pbase[j] + tile * pstride[j] + lane * 16, sl + pdst[j]);
            else dma16_nt(pbase[j] + tile * pstride[j] + lane * 16, sl + pdst[j]);
        }
        dma4(reinterpret_cast<const char*>(gnet + tile * 32) + lane * 4, sl + S::kGOff + wave * 256);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n2 = 0; n2 < NT; ++n2) acc[m][n2] = (f32x16)0.f;
    float vecA[MT], vecB[NT], gsum = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) vecA[m] = 0.f;
#pragma unroll
    for (int n2 = 0; n2 < NT; ++n2) vecB[n2] = 0.f;
    // which wave of the waves that hold a row tile's (column tile's) fragments adds up its row-side (column-side) vector
    auto owns_row = [&](const int m) __attribute__((always_inline)) { return (kWide && nct == 6) ? wn == m % 3 : wn == m; };   // 4 wn for 4 m | 2 wn for 2 m
    auto owns_col = [&](const int n2) __attribute__((always_inline)) { return wm == n2; };                              // 2 wm for 2 n | 4 wm, three used

    // LDS reads by inline asm with a counted wait: a read hipcc can see is ordered behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)
    // in front of the first ds_read of every tile), which serialised fetch and compute -- DMA alone 156 us, compute alone 131 us,
    // together 246 us before this, measured with stage-exit builds
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    auto rd128 = [&](u32x4& v, const unsigned addr) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    };
    auto compute = [&](const int slot_) __attribute__((always_inline)) {
        const unsigned buf = lds_base + slot_ * kSlot;
        const unsigned gl = buf + S::kGOff + wave * 256;
        constexpr int KB = (NS == 1) ? 2 : 1;                       // single-bf16 fragments: both k-steps of the tile are read up front
        u32x4 gqa[KB][2], faa[KB][nsx][MT], fba[KB][NS][NT];
        // one base register per stream, the fragment index as the instruction's immediate offset (a full address per read costs a VGPR each for its
        // slot-independent part -- ~30 of them, hoisted out of the tile loop -- and spilled once the in-register operand forming of products 2 / 3 came in)
        const unsigned baseG = gl + h * 16;
        const unsigned baseA = buf + wm * (MT * 1024) + lane * 16, baseB = buf + S::kX + wn * (NT * 1024) + lane * 16;
        auto rd128o = [&](u32x4& v, const unsigned base, const int off) __attribute__((always_inline)) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "i"(off) : "memory");
        };
        auto issue_reads = [&](const int kk, const int b) __attribute__((always_inline)) {
            rd128o(gqa[b][0], baseG, 64 * kk);
            rd128o(gqa[b][1], baseG, 64 * kk + 32);
#pragma unroll
            for (int s2 = 0; s2 < nsx; ++s2)
#pragma unroll
                for (int m = 0; m < MT; ++m) rd128o(faa[b][s2][m], baseA, s2 * 16384 + (kk * 8 + m) * 1024);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                for (int n2 = 0; n2 < NT; ++n2) rd128o(fba[b][s2][n2], baseB, s2 * S::kYPlane + (kk * nct + n2) * 1024);
        };
        constexpr int kReads = 2 + MT * nsx + NT * NS;              // LDS reads per k-step
        if constexpr (NS == 1) { issue_reads(0, 0); issue_reads(1, 1); }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int b = (NS == 1) ? kk : 0;
            if constexpr (NS == 2) issue_reads(kk, 0);
            u32x4 (&fa)[nsx][MT] = faa[b];
            u32x4 (&fb)[NS][NT] = fba[b];
            u32x4& gq0 = gqa[b][0];
            u32x4& gq1 = gqa[b][1];
            // every value passes through the wait, so no use can be scheduled in front of it (LDS reads retire in order: with the
            // second k-step's reads still behind, the first k-step is complete at lgkmcnt(kReads))
            if (NS == 1 && kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(kReads) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(gq0), "+v"(gq1), "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fb[0][0]), "+v"(fb[0][1]));
            if constexpr (MT == 4) asm volatile("" : "+v"(fa[0][2]), "+v"(fa[0][3]));
            if constexpr (NT == 3) asm volatile("" : "+v"(fb[0][2]));
            if constexpr (NS == 2) asm volatile("" : "+v"(fb[1][0]), "+v"(fb[1][1]));
            if constexpr (NS == 2 && NT == 3) asm volatile("" : "+v"(fb[1][2]));
            if constexpr (nsx == 2) asm volatile("" : "+v"(fa[1][0]), "+v"(fa[1][1]));
            if constexpr (nsx == 2 && MT == 4) asm volatile("" : "+v"(fa[1][2]), "+v"(fa[1][3]));
            const float gp[8] = {__uint_as_float(gq0[0]), __uint_as_float(gq0[1]), __uint_as_float(gq0[2]), __uint_as_float(gq0[3]),
                                 __uint_as_float(gq1[0]), __uint_as_float(gq1[1]), __uint_as_float(gq1[2]), __uint_as_float(gq1[3])};
            if constexpr (PROD == 2) {
                // the Y fragments just read are the per-point TABLE pe6 (hi [+ lo]); this net's operand G6 = g pe6 is formed here: element e of a register
                // pair <-> point e of the lane's eight (gp[e])
#pragma unroll
                for (int n2 = 0; n2 < NT; ++n2)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        float v0 = bf_lo(fb[0][n2][p]), v1 = bf_hi(fb[0][n2][p]);
                        if constexpr (NS == 2) { v0 += bf_lo(fb[1][n2][p]); v1 += bf_hi(fb[1][n2][p]); }
                        const float z0 = gp[2 * p] * v0, z1 = gp[2 * p + 1] * v1;
                        const u32 hi = pack2(z0, z1);
                        fb[0][n2][p] = hi;
                        if constexpr (NS == 2) fb[1][n2][p] = pack2(z0 - bf_lo(hi), z1 - bf_hi(hi));
                    }
            }
            if (PROD == 1 && wave == 0 && i == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) gsum += gp[e];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (owns_row(m) && PROD != 2) {         // row-side vector: sum_pt X[pt][row] * g[pt]; the four waves that
                                                                         // hold this row tile's fragments take one tile each (all on
                                                                         // the wn = 0 waves it made them the workgroup's critical path:
                                                                         // +700 cycles per tile, everybody else waiting at the barrier);
                                                                         // 192-column products: the wn = 3 waves sit idle, three share
                    float d = 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < nsx; ++s2) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) { d = fmaf(bf_lo(fa[s2][m][p]), gp[2 * p], d); d = fmaf(bf_hi(fa[s2][m][p]), gp[2 * p + 1], d); }
                    }
                    vecA[m] += d;
                }
#pragma unroll
                for (int n2 = 0; n2 < NT; ++n2) {
                    if constexpr (NS == 2) {
                        acc[m][n2] = mfma(as_bf(fa[0][m]), as_bf(fb[1][n2]), acc[m][n2]);
                        if constexpr (nsx == 2) acc[m][n2] = mfma(as_bf(fa[1][m]), as_bf(fb[0][n2]), acc[m][n2]);
                    }
                    acc[m][n2] = mfma(as_bf(fa[0][m]), as_bf(fb[0][n2]), acc[m][n2]);
                }
            }
            if (PROD != 3) {                                             // column-side vector: q = sum_pt Y[pt][col] (Z, Z1, G6), one column tile per wm
#pragma unroll
                for (int n2 = 0; n2 < NT; ++n2) {
                    if (!owns_col(n2)) continue;
                    float d = 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                        for (int p = 0; p < 4; ++p) d += bf_lo(fb[s2][n2][p]) + bf_hi(fb[s2][n2][p]);
                    vecB[n2] += d;
                }
            }
        }
    };

    // prologue: RING-1 tiles in flight (out-of-range tiles re-read the last valid one; their data is never used)
    const int64_t tl = t1 > t0 ? t1 - 1 : t0;
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) issue(t0 + r < t1 ? t0 + r : tl, r);
    int slot = 0;
#ifdef DPN_WGRAD_PHASES
    u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, ph_wait = 0, ph_bar = 0, ph_issue = 0, ph_comp = 0;
#endif
    for (int64_t tile = t0; tile < t1; ++tile) {
        DPN_WG_CLOCK(c0);
        wait_vmcnt<(RING - 2) * PER_TILE>();                             // this wave's pieces of `tile` have landed
        DPN_WG_CLOCK(c1);
        __builtin_amdgcn_s_barrier();                                    // ... and everybody else's; and compute(tile-1) is finished everywhere
        DPN_WG_CLOCK(c2);
        {
            const int64_t nt = tile + RING - 1;
            issue(nt < t1 ? nt : tl, (slot + RING - 1) % RING);          // refill the slot that compute(tile-1) just released
        }
        DPN_WG_CLOCK(c3);
        if (active) compute(slot);
        DPN_WG_CLOCK(c4);
#ifdef DPN_WGRAD_PHASES
        ph_wait += c1 - c0; ph_bar += c2 - c1; ph_issue += c3 - c2; ph_comp += c4 - c3;
#endif
        slot = (slot + 1) % RING;
    }
#ifdef DPN_WGRAD_PHASES
    if (a.phases && lane == 0) {
        unsigned* o = a.phases + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
        o[0] = ph_wait; o[1] = ph_bar; o[2] = ph_issue; o[3] = ph_comp; o[4] = (unsigned)(t1 - t0); o[5] = PROD;
    }
#endif
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // ---- write this split's partial sums: natural [row slot][col] order
    float* part = a.partials + ((int64_t)split * kNets + net) * kPartFloats;
    float* out = part + part_off(PROD);
    if (active) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n2 = 0; n2 < NT; ++n2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = wm * (MT * 32) + 32 * m + drow32(r, h);
                    const int cc = wn * (NT * 32) + 32 * n2 + i;
                    out[rr * ncol + cc] = acc[m][n2][r];
                }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float v = vecA[m] + __shfl_xor(vecA[m], 32);
        if (owns_row(m) && h == 0) {
            const int rr = wm * (MT * 32) + 32 * m + i;
            if (PROD == 1) part[kPartVec + 2 * 256 + rr] = v;          // mv1 (= mvec again: the hyper-network's half of the reduction does not wait for P0)
            if (PROD == 3) part[kPartVec + 3 * 256 + rr] = v;          // db1
        }
    }
    if (PROD == 1 && wave == 0) {
        const float gs = gsum + __shfl_xor(gsum, 32);                  // lanes 0 and 32 hold the two halves
        if (lane == 0) part[kPartVec + 4 * 256] = gs;
    }
    if (PROD != 3 && active) {
        const int qo = kPartVec + (PROD == 1 ? 5 : 6) * 256;                             // q1 = colsum(Z1), q6 = colsum(G6)
#pragma unroll
        for (int n2 = 0; n2 < NT; ++n2) {
            const float v = vecB[n2] + __shfl_xor(vecB[n2], 32);
            if (owns_col(n2) && h == 0) part[qo + wn * (NT * 32) + 32 * n2 + i] = v;
        }
    }
}

template <int NS>
__global__ __launch_bounds__(512, 2) void dpn_wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[wgrad_lds_bytes<NS>()];
    // workgroup -> (product, point range): uniform scalar walk.  (Round 5 tried an XCD-aware placement -- the six nets' product-2 workgroups of one range on
    // ONE XCD, so that its L2 serves the per-point pe6 table to five of them: 186.7 / 186.1 us against 184.7 / 188.6 us for this linear walk on one box.
    // No difference: the table's re-reads are served by the memory-side cache either way.)
    int prod = 1, split = blockIdx.x;
    const int net = blockIdx.y;
    while (prod < 3 && split >= a.splits[prod]) { split -= a.splits[prod]; ++prod; }
    if (prod == 1) wgrad_body<NS, 1>(a, lds, split, net);
    else if (prod == 2) wgrad_body<NS, 2>(a, lds, split, net);
    else wgrad_body<NS, 3>(a, lds, split, net);
}

// ------------------------------------------------------------------------------------------------ backward, stage 3
// Round 5: two halves.  (1) dpn_finish_rows_kernel -> dpn_finish_vside_kernel: what the hyper-network's backward waits for (d w1b1, d w2b2,
// d evec; dWd, d bd ride in the same launch).  (2) dpn_finish_gside_kernel -> dpn_finish_fc2_kernel: gradients of static tensors only
// (cat_fc1.fc.0 / fc.2, out_fc) -- the host may run them on a side branch beside the encoder's backward chain (dpn_wgrad_finish_parts).
struct FinishArgs {
    DpnNetPtrs net[kNets];
    DpnNetGradPtrs grad[kNets];
    const char* packed;
    const float* partials;
    float* scratch_s1;      // [6][256][256] S1 = M2^T Z1, natural order             (dpn_finish_rows_kernel -> vside, gside)
    float* scratch_s2;      // [6][256][192] S2 = M2^T G6
    float* scratch_mv;      // [6][256]      mvec = M2^T g
    float* scratch_u;       // [6][256]      u = W2^T wo, natural order
    float* scratch_q1;      // [6][256]      colsum(Z1)
    float* scratch_q6;      // [6][256]      colsum(G6) (192 used)
    float* scratch_sg;      // [8]           sum g per net
    float* scratch_rp;      // [6][8][256]   per column tile: sum_i W1[o][i] G[o][i]      (gside -> fc2)
    int splits[4], ns;      // point ranges per product, as dpn_wgrad_kernel cut them (splits[0] = 0: the product M2^T Z is gone)
    int64_t n;
};

DEV int slot_of_ch(int ch) { return (ch & ~15) + 8 * ((ch >> 2) & 1) + 4 * ((ch >> 3) & 1) + (ch & 3); }
// original PE3 / PE6 channel -> slot index 16*ks + 8*h + e
DEV int slot_of_pe3(int orig) {
    const int f = orig / 6, fn = (orig % 6) / 3, c = orig % 3;
    const int a = 32 * c + f;
    const int ks = a >> 3, h = (a >> 2) & 1, p = a & 3;
    return 16 * ks + 8 * h + 2 * p + fn;
}
DEV int slot_of_pe6(int orig) {
    const int f = orig / 12, fn = (orig % 12) / 6, c6 = orig % 6;
    const int a = 16 * c6 + f;
    const int ks = a >> 3, h = (a >> 2) & 1, p = a & 3;
    return 16 * ks + 8 * h + 2 * p + fn;
}

// NQ sums over the partial buffers of the point ranges (ks[q] of them for sum q: the count of the product that wrote it), ALL their
// loads in flight at once (a loop of load -> wait -> add, which is what hipcc makes of the obvious code, costs one HBM round trip per
// range and per sum: 40 in a row per thread).  Ranges beyond ks[q] re-read the last one and are not added; the additions keep the
// range order, so the result does not depend on how the loads are grouped.
constexpr int kMaxSplits = 20;                  // choose_plan() never returns more for one product
template <int NQ, int MAXS>
DEV void part_sums_n(const float* partials, const int (&ks)[NQ], int net, const int (&off)[NQ], float (&out)[NQ]) {
    float v[MAXS][NQ];
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            // (plain loads: with the non-temporal hint -- "read once" -- the partials dpn_wgrad has just written are fetched past the memory-side cache: the finish
            //  stage's first half 28.8 against 25.4 us alone, dpn_finish_rows 18.5 against 15.7 us in the step; profiles/round6_nontemporal_hints.txt)
            v[k][q] = partials[((int64_t)min(k, ks[q] - 1) * kNets + net) * kPartFloats + off[q]];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) t += (k < ks[q]) ? v[k][q] : 0.f;
        out[q] = t;
    }
}
template <int NQ>
DEV void part_sums(const float* partials, const int (&ks)[NQ], int net, const int (&off)[NQ], float (&out)[NQ]) {
    int most = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) most = ks[q] > most ? ks[q] : most;
    if (most <= 12) part_sums_n<NQ, 12>(partials, ks, net, off, out);
    else if (most <= 16) part_sums_n<NQ, 16>(partials, ks, net, off, out);
    else part_sums_n<NQ, kMaxSplits>(partials, ks, net, off, out);
}

// one block per (output row o, net): reduces the point ranges, un-permutes: rows o of S1, S2 and of d(w1b1); the row's vector entries
__global__ __launch_bounds__(256) void dpn_finish_rows_kernel(FinishArgs a) {
    const int o = blockIdx.x, net = blockIdx.y, i = threadIdx.x;
    const DpnNetGradPtrs& Gd = a.grad[net];
    const int so = slot_of_ch(o), si = slot_of_ch(i);
    // thread 0..4 also own the row's vector entries: fetched with everything else, not after the reduction
    float rowv[1] = {0.f};
    const bool has_vec = i < 4 || (i == 4 && o == 0);
    if (has_vec) {
        // mv1 = M2^T g and db1 = T1^T g (row slot so), q1 = colsum(Z1) (column slot of channel o), q6 = colsum(G6) (PE6 channel o), sum g
        const int offv[1] = {i == 0 ? kPartVec + 2 * 256 + so : i == 1 ? kPartVec + 3 * 256 + so : i == 2 ? kPartVec + 5 * 256 + so
                             : i == 3 ? kPartVec + 6 * 256 + slot_of_pe6(o < kPe ? o : 0) : kPartVec + 4 * 256};
        const int ksv[1] = {i == 1 ? a.splits[3] : i == 3 ? a.splits[2] : a.splits[1]};
        part_sums<1>(a.partials, ksv, net, offv, rowv);
    }
    if (i < kPe) {
        const int off3[3] = {part_off(1) + so * 256 + si, part_off(2) + so * 192 + slot_of_pe6(i), part_off(3) + so * 192 + slot_of_pe3(i)};
        const int ks3[3] = {a.splits[1], a.splits[2], a.splits[3]};
        float g3[3];
        part_sums<3>(a.partials, ks3, net, off3, g3);
        a.scratch_s1[((int64_t)net * 256 + o) * 256 + i] = g3[0];
        a.scratch_s2[((int64_t)net * 256 + o) * kPe + i] = g3[1];
        Gd.w1b1[o * Gd.ld_w1b1 + i] = g3[2];
    } else {
        const int off1[1] = {part_off(1) + so * 256 + si};
        const int ks1[1] = {a.splits[1]};
        float g1[1];
        part_sums<1>(a.partials, ks1, net, off1, g1);
        a.scratch_s1[((int64_t)net * 256 + o) * 256 + i] = g1[0];
    }
    if (i == 0) {
        // u[o] = (W2^T wo)[o] from the packed vectors ([h][T][r] order)
        const float* vec = reinterpret_cast<const float*>(a.packed + (long)net * pack_bytes_per_net(a.ns) + (long)kPackKB * 1024 * a.ns);
        const int T = o >> 5, w = o & 31, hh = (w >> 2) & 1, r = (w & 3) + 4 * (w >> 3);
        a.scratch_u[net * 256 + o] = vec[kVecU * 256 + hh * 128 + T * 16 + r];
        a.scratch_mv[net * 256 + o] = rowv[0];
    }
    if (i == 1) Gd.w1b1[o * Gd.ld_w1b1 + 192] = rowv[0];
    if (i == 2) a.scratch_q1[net * 256 + o] = rowv[0];
    if (i == 3 && o < kPe) a.scratch_q6[net * 256 + o] = rowv[0];
    if (i == 4 && o == 0) a.scratch_sg[net] = rowv[0];
}

// The factor that turns the mask-side sums into the gradients that used to need v per point (SavedView):
//   d(w2b2)[o][i] = sum_j W1[j][o] u[j] S1[j][i] + 2 wo[o] q1[i]        S1 = M2^T Z1   (i < 256),  column 256: S1 -> M2^T g, q1 -> sum g
//   dWd[o][i]     = sum_j W1[j][o] u[j] S2[j][i] + 2 wo[o] q6[i]        S2 = M2^T G6
// and d evec = d bd = column 256.  One workgroup per 32 x 32 output tile: grid (8 row tiles x 15 column tiles [8 of d w2, the vector, 6 of
// dWd], 6 nets); the four waves take 64 of the 256 j each on the exact-fp32 matrix instruction (operands straight from global memory: both
// are contiguous along the lane index) and their partial tiles are added in a fixed order through LDS.
DEV f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
constexpr int kVsideBlocks = 8 * 15;
DEV void finish_vside_body(const FinishArgs& a, const int bx, const int net, float (&red)[4][16][64], float (&qs)[32]) {
    const int rt = bx & 7, ctile = bx >> 3;
    const int kind = ctile < 8 ? 0 : ctile == 8 ? 1 : 2;                  // d w2 | vector column | dWd
    const DpnNetPtrs& P = a.net[net];
    const DpnNetGradPtrs& Gd = a.grad[net];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 31, kh = lane >> 5;
    const int o0 = 32 * rt, n0 = kind == 0 ? 32 * ctile : kind == 2 ? 32 * (ctile - 9) : 0, ncol = kind == 0 ? 256 : kind == 2 ? kPe : 1;
    const float* B = kind == 0 ? a.scratch_s1 + (int64_t)net * 65536 : kind == 2 ? a.scratch_s2 + (int64_t)net * 256 * kPe : a.scratch_mv + net * 256;
    const float* U = a.scratch_u + net * 256;
    const int ldb = ncol;
    const bool colok = n0 + col < ncol;
    // the rank-one term's column factor: q1 / q6 / sum g
    float qv = 0.f;
    if (wv == 0 && colok) qv = kind == 0 ? a.scratch_q1[net * 256 + n0 + col] : kind == 2 ? a.scratch_q6[net * 256 + n0 + col] : a.scratch_sg[net];
    f32x16 acc = (f32x16)0.f;
    float av[32], bv[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
        const int j = 64 * wv + 2 * kk + kh;
        av[kk] = P.W1[j * 256 + o0 + col] * U[j];
        bv[kk] = colok ? B[(int64_t)j * ldb + n0 + col] : 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) acc = mfma_f32_32x32x2(av[kk], bv[kk], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][r][lane] = acc[r];
    if (wv == 0 && kh == 0) qs[col] = qv;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wv + 4 * q;                                         // element (r, lane) of the tile: row drow32(r, kh), column col
        const float sum = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
        const int o = o0 + drow32(r, kh), i = n0 + col;
        if (!colok) continue;
        const float v = sum + 2.0f * P.wo[o] * qs[col];
        if (kind == 0) Gd.w2b2[o * Gd.ld_w2b2 + i] = v;
        else if (kind == 2) Gd.Wd[o * kPe + i] = v;
        else { Gd.w2b2[o * Gd.ld_w2b2 + 256] = v; Gd.evec[o] = v; Gd.bd[o] = v; }
    }
}

// G = M2^T Z without Z (round 5).  Z = Z1 w2^T + G6 Wd^T + g cvec^T is linear in the three per-point operands, so
//   G[o][i] = sum_j S1[o][j] w2[i][j] + sum_k S2[o][k] Wd[i][k] + mvec[o] cvec[i]          cvec = b2 + bd + e
// -- a 256 x 256 x 448 exact-fp32 GEMM per net on the sums dpn_wgrad_kernel produces anyway.  From it d cat_fc1.fc.0.weight = diag(u) G and the
// per-tile parts of r[o] = sum_i W1[o][i] G[o][i] (+ bf1 mvec, added by dpn_finish_fc2_kernel).  One workgroup per 32 x 32 tile of G; wave wv
// takes j in [64 wv, 64 wv + 64) and k in [48 wv, 48 wv + 48); lane (col, kh) holds four consecutive reduction indices per load of its row.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
DEV void finish_gside_body(const FinishArgs& a, const int bx, const int net, float (&red)[4][16][64]) {
    const int rt = bx & 7, ct = bx >> 3;
    const DpnNetPtrs& P = a.net[net];
    const DpnNetGradPtrs& Gd = a.grad[net];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 31, kh = lane >> 5;
    const int o0 = 32 * rt, i0 = 32 * ct;
    const float* S1 = a.scratch_s1 + ((int64_t)net * 256 + o0 + col) * 256;
    const float* S2 = a.scratch_s2 + ((int64_t)net * 256 + o0 + col) * kPe;
    const float* w2 = P.w2b2 + (int64_t)(i0 + col) * P.ld_w2b2;
    const float* Wd = P.Wd + (i0 + col) * kPe;
    f32x4u a1[8], b1[8], a2[6], b2[6];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k = 64 * wv + 8 * m + 4 * kh;
        a1[m] = *reinterpret_cast<const f32x4u*>(S1 + k);
        b1[m] = *reinterpret_cast<const f32x4u*>(w2 + k);
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int k = 48 * wv + 8 * m + 4 * kh;
        a2[m] = *reinterpret_cast<const f32x4u*>(S2 + k);
        b2[m] = *reinterpret_cast<const f32x4u*>(Wd + k);
    }
    f32x16 acc = (f32x16)0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma_f32_32x32x2(a1[m][e], b1[m][e], acc);
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma_f32_32x32x2(a2[m][e], b2[m][e], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
    const int i = i0 + col;
    const float cv = P.w2b2[(int64_t)i * P.ld_w2b2 + 256] + P.bd[i] + P.evec[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wv + 4 * q;                                         // element (r, lane) of the tile: row drow32(r, kh), column col
        const int o = o0 + drow32(r, kh);
        const float G = ((red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane])) + a.scratch_mv[net * 256 + o] * cv;
        Gd.W1[o * 256 + i] = a.scratch_u[net * 256 + o] * G;
        float rp = P.W1[o * 256 + i] * G;                                 // this tile's part of r[o]: the 32 columns sit on the 32 lanes of a half-wave
#pragma unroll
        for (int sft = 16; sft > 0; sft >>= 1) rp += __shfl_xor(rp, sft);
        if (col == 0) a.scratch_rp[((int64_t)net * 8 + ct) * 256 + o] = rp;
    }
}

// ONE launch for the two independent consumers of dpn_finish_rows_kernel's sums: blocks [0, n_v) = the W1^T diag(u) factor (what the hyper-network's
// backward waits for), the rest = G = S1 w2^T + S2 Wd^T + ... (static tensors only) -- side by side instead of one behind the other.  n_v = 0 or
// kVsideBlocks, the grid decides which halves run (dpn_wgrad_finish_parts).
constexpr int kGsideBlocks = 64;
__global__ __launch_bounds__(256) void dpn_finish_sides_kernel(FinishArgs a, int n_v) {
    __shared__ float red[4][16][64];
    __shared__ float qs[32];
    // As dispatched, XCD = blockIdx.x mod 8 = the ROW tile (the grid's x extent is a multiple of 8): the 15 x 6 (or 8 x 6) tiles that read the same 32 columns of
    // W1 (rows of S1: each lane its own row) sat on one L2.  Every XCD takes a contiguous range of the (net, block) list instead, as in dpn_pack_fused_kernel.
#ifdef FINISH_NO_XCD_REMAP
    const int net = blockIdx.y, bx = blockIdx.x;
#else
    const int gx = gridDim.x, total = gx * kNets, lin = blockIdx.x + gx * blockIdx.y, xcd = lin & 7;
    const int virt = xcd * (total >> 3) + min(xcd, total & 7) + (lin >> 3), net = virt / gx, bx = virt - net * gx;
#endif
    if (bx < n_v) finish_vside_body(a, bx, net, red, qs);
    else finish_gside_body(a, bx - n_v, net, red);
}

// one block per (row o', net): r, then dW2 = wo (x) r, dbf2, dwo (with colsum(Z) = w2 q1 + Wd q6 + sum g cvec), dbo, dbf1
__global__ __launch_bounds__(256) void dpn_finish_fc2_kernel(FinishArgs a) {
    __shared__ float red[2][256];
    const int net = blockIdx.y, op = blockIdx.x, o = threadIdx.x;
    const DpnNetPtrs& P = a.net[net];
    const DpnNetGradPtrs& Gd = a.grad[net];
    const float* rp = a.scratch_rp + (int64_t)net * 8 * 256 + o;
    const float mv = a.scratch_mv[net * 256 + o];
    const float r = (((rp[0] + rp[256]) + (rp[512] + rp[768])) + ((rp[1024] + rp[1280]) + (rp[1536] + rp[1792]))) + P.bf1[o] * mv;
    const float wop = P.wo[op];
    Gd.W2[op * 256 + o] = wop * r;
    red[0][o] = P.W2[op * 256 + o] * r;
    red[1][o] = a.scratch_q1[net * 256 + o] * P.w2b2[(int64_t)op * P.ld_w2b2 + o] + (o < kPe ? a.scratch_q6[net * 256 + o] * P.Wd[op * kPe + o] : 0.f);
    if (op == 0) Gd.bf1[o] = a.scratch_u[net * 256 + o] * mv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (o < s) { red[0][o] += red[0][o + s]; red[1][o] += red[1][o + s]; }
        __syncthreads();
    }
    if (o == 0) {
        const float sg = a.scratch_sg[net];
        const float cv = P.w2b2[(int64_t)op * P.ld_w2b2 + 256] + P.bd[op] + P.evec[op];
        const float zsum = red[1][0] + sg * cv;                           // colsum(Z)[op]
        Gd.bf2[op] = wop * sg;
        Gd.wo[op] = red[0][0] + P.bf2[op] * sg + 2.f * zsum;
        if (op == 0) Gd.bo[0] = sg;
    }
}

// ------------------------------------------------------------------------------------------------ small fp32 GEMM
// The encoder (L = 287 tokens, d = 256) and the hyper-network heads are a few dozen tiny fp32 GEMMs per step; library
// GEMMs cost 19-75 us each at these shapes (latency-bound).  One LDS-tiled fp32 kernel serves all of them:
//   C[M][N] = op(A)[M][K] * op(B)[K][N] (+ bias[N]) (+ C)      op = identity or transpose, row-major with leading dims
//   optionally colsum[N] = sum_m op(A)^T ... is NOT needed; instead rowsum of op(A)^T is offered through `asum`:
//   asum[M] = sum_k op(A)[m][k]  (used as the bias gradient when op(A) = grad_out^T).
struct SgemmArgs {
    const float *A, *B, *bias;
    float *C, *asum, *ws;
    int M, N, K, lda, ldb, ldc, ta, tb, accumulate, k_per_split;
};

__global__ __launch_bounds__(256) void dpn_sgemm_kernel(SgemmArgs a) {
    constexpr int BM = 32, BN = 32, BK = 32;
    __shared__ float As[BK][BM + 1];
    __shared__ float Bs[BK][BN + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, 2 x 2 outputs each
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * a.k_per_split;
    const int kend = (kbeg + a.k_per_split < a.K) ? kbeg + a.k_per_split : a.K;
    const bool split = gridDim.z > 1;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float rs = 0.f;
    const bool do_asum = a.asum != nullptr && blockIdx.x == 0;
    float ra[4], rb[4];
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = threadIdx.x + 256 * i;                  // 1024 elements per tile
            {
                const int kk = a.ta ? (e >> 5) : (e & 31), mm = a.ta ? (e & 31) : (e >> 5);
                const int gm = m0 + mm, gk = k0 + kk;
                ra[i] = (gm < a.M && gk < kend) ? (a.ta ? a.A[(int64_t)gk * a.lda + gm] : a.A[(int64_t)gm * a.lda + gk]) : 0.f;
            }
            {
                const int kk = a.tb ? (e & 31) : (e >> 5), nn = a.tb ? (e >> 5) : (e & 31);
                const int gk = k0 + kk, gn = n0 + nn;
                rb[i] = (gk < kend && gn < a.N) ? (a.tb ? a.B[(int64_t)gn * a.ldb + gk] : a.B[(int64_t)gk * a.ldb + gn]) : 0.f;
            }
        }
    };
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = threadIdx.x + 256 * i;
            As[a.ta ? (e >> 5) : (e & 31)][a.ta ? (e & 31) : (e >> 5)] = ra[i];
            Bs[a.tb ? (e & 31) : (e >> 5)][a.tb ? (e >> 5) : (e & 31)] = rb[i];
        }
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);                       // next tile in flight under the FMAs
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            const float a0 = As[kk][ty], a1 = As[kk][ty + 16], b0 = Bs[kk][tx], b1 = Bs[kk][tx + 16];
            acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        if (do_asum && threadIdx.x < BM) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) rs += As[kk][threadIdx.x];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gm = m0 + ty + 16 * i, gn = n0 + tx + 16 * j;
            if (gm < a.M && gn < a.N) {
                if (split) {                                      // deterministic split-K: partial tile -> workspace[z][M][N]
                    a.ws[((int64_t)blockIdx.z * a.M + gm) * a.N + gn] = acc[i][j];
                } else {
                    const float v = acc[i][j] + (a.bias ? a.bias[gn] : 0.f);
                    float* c = a.C + (int64_t)gm * a.ldc + gn;
                    *c = a.accumulate ? (*c + v) : v;
                }
            }
        }
    if (do_asum && threadIdx.x < BM && m0 + threadIdx.x < a.M) {
        if (split) a.ws[(int64_t)gridDim.z * a.M * a.N + (int64_t)blockIdx.z * a.M + m0 + threadIdx.x] = rs;
        else a.asum[m0 + threadIdx.x] = rs;
    }
}

// second pass of the split-K path: fixed-order sum over the splits (+ bias, + C)
__global__ __launch_bounds__(256) void dpn_sgemm_reduce_kernel(SgemmArgs a, int splits) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t mn = (int64_t)a.M * a.N;
    if (idx < mn) {
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += a.ws[z * mn + idx];
        const int gm = (int)(idx / a.N), gn = (int)(idx % a.N);
        v += a.bias ? a.bias[gn] : 0.f;
        float* c = a.C + (int64_t)gm * a.ldc + gn;
        *c = a.accumulate ? (*c + v) : v;
    }
    if (a.asum && idx < a.M) {
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += a.ws[splits * mn + (int64_t)z * a.M + idx];
        a.asum[idx] = v;
    }
}

// Several independent small GEMMs in ONE launch (blockIdx.z = problem), each optionally a sum of up to 3 products
// (C = sum_t op(A_t) op(B_t)): the three q/k/v projections of an attention layer, or the input- and weight-gradient
// GEMMs of a linear layer, cost one launch instead of 2-6.  Two k-tiles are kept in flight in registers.
constexpr int kBatchMaxProblems = DPN_GEMM_MAX_PROBLEMS, kBatchMaxTerms = DPN_GEMM_MAX_TERMS, kBatchTermPool = 32, kBatchMaxJobs = DPN_GEMM_MAX_JOBS;
struct SgemmTerm {
    const float* A;
    const float* B;
    int lda, ldb, K, pad;
};
struct SgemmProblem {
    const float* bias;
    float *C, *asum;
    const float* aux;          // epilogue operand [M][ldc] (epi 2, 3)
    float* aux_out;            // pre-activation output [M][ldc] (epi 1, optional)
    int M, N, ldc, ta, tb, nterms, term0, epi;
};
// exact-erf GELU and its derivative, the formulas of torch's GeluCUDAKernelImpl / GeluBackwardCUDAKernelImpl (approximate='none')
DEV float gelu_exact(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
DEV float gelu_exact_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;      // M_2_SQRTPI * M_SQRT1_2 * 0.5
    return cdf + x * pdf;
}
struct SgemmColsum { const float* partial; float* out_a; float* out_b; int nblocks, pad; };
struct SgemmBatch {
    SgemmProblem p[kBatchMaxProblems];
    SgemmTerm t[kBatchTermPool];          // the accumulated A.B terms of all problems (problem i owns t[term0 .. term0+nterms))
    SgemmColsum job[kBatchMaxJobs];       // ride-along column sums (LayerNorm parameter gradients): blockIdx.z = n, n + 1, ...
    int n;
};

// exact-fp32 matrix cores: v_mfma_f32_32x32x2_f32 (A: lane l holds A[l&31][l>>5], B: lane l holds B[l>>5][l&31]); bitwise an fmaf chain.
typedef float f32x1;
DEV f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int BK, int NBUF, int NT = 256, int TM = 1, int TN = 1>
__global__ __launch_bounds__(NT) void dpn_sgemm_batch_kernel(SgemmBatch batch) {
    // one workgroup = one 32x32 output tile.  K is walked in 32-wide tiles that are loaded COALESCED (the fast index follows each
    // operand's contiguous dimension) into double-buffered LDS, two tiles ahead in registers; inside a tile the four waves take
    // BK/4 k-values each (v_mfma_f32_32x32x2_f32, exact fp32) and their partial tiles are summed through LDS in a fixed order.
    // <64, 2>: double-buffered LDS, two k-tiles in flight (long / multi-term reductions); 64-deep tiles = 34 KB of LDS = four
    // workgroups per CU, so the 500-800 tiles of a backward launch are one round (128-deep: two per CU, measured +7 us per step;
    // 96-deep: slower, the non-power-of-two index arithmetic).  <256, 1>: the whole K of a 256-wide encoder
    // GEMM is ONE tile -- one LDS stage, one barrier pair, 16 loads per operand in flight (used when every problem is a single tile);
    // it runs with EIGHT waves (NT = 512): half the MFMA chain and half the loads per thread of a latency-bound tile, -15 us per step.
    // Round 6: TM x TN 32 x 32 sub-tiles per workgroup (<64, 2, 512, 2, 2>: a 64 x 64 output tile, each of its four quadrants on two waves that split the
    // k-values of a k-tile): half the L2 -> LDS traffic per MAC of the 32 x 32 form (a 32 x 32 tile loads 2 x 32 x K values for 32 x 32 x K MACs: the token
    // convolution moved 74 MB through L2 for 12 MB of operands) and four times the MFMA work per barrier.
    constexpr int BM = 32 * TM, BN = 32 * TN, NLA = BK * BM / NT, NLB = BK * BN / NT;    // loads per operand per thread per k-tile
    constexpr int NW = NT / 64, NQ = TM * TN, KW = NW / NQ;   // waves; sub-tiles; waves per sub-tile, each taking BK / KW k-values of a k-tile
    static_assert(NW % NQ == 0 && BK % (2 * KW) == 0 && (BK * BM) % NT == 0 && (BK * BN) % NT == 0, "tile shape");
    // (Round 6: XCD-contiguous renumbering of the workgroups -- every XCD a contiguous range of the (problem, tile) list, as in dpn_pack_fused_kernel -- is SLOWER
    //  here: the heads' forward 16.2 -> 22.4 us, their backward 27.1 -> 48 us, the token convolution unchanged: a problem's tiles on ONE L2 queue on the same
    //  lines; spread over eight L2s each serves an eighth of them.  A skew of the column tile by (row tile + 3 problem), so that tiles sharing a B panel
    //  leave the XCD they share when gridDim.x is a multiple of 8, changes nothing.  profiles/round6_xcd_contiguous.txt)
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (bz >= batch.n) {
        // ride-along job: out_a[c] = sum_b partial[b][c], out_b[c] = sum_b partial[b][256 + c] (fixed order) -- the reduction of
        // dpn_add_ln_bwd's per-block partial sums, finished in the shadow of the GEMM tiles instead of in a launch of its own
        if (bx || by) return;
        const SgemmColsum& j = batch.job[bz - batch.n];
        const int c = threadIdx.x;
        if (c >= 256) return;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int b = 0; b < j.nblocks; ++b) { s1 += j.partial[(int64_t)b * 512 + c]; s2 += j.partial[(int64_t)b * 512 + 256 + c]; }
        j.out_a[c] = s1;
        j.out_b[c] = s2;
        return;
    }
    const SgemmProblem& a = batch.p[bz];
    const int m0 = by * BM, n0 = bx * BN;
    if (m0 >= a.M || n0 >= a.N) return;
    // 34 KB (<64,2>) or 68 KB (<256,1>) of LDS: four or two workgroups per CU.  The partial tiles of the waves (eight: 33 KB) reuse the
    // staging buffers after the last k-tile.  Both forms run with eight waves: -15 us (<256,1>) and -53 us (<64,2>) per step against four.
    struct Stage { float A[NBUF][BK][BM + 1]; float B[NBUF][BK][BN + 1]; };
    __shared__ Stage stage;
    float (&As)[NBUF][BK][BM + 1] = stage.A;
    float (&Bs)[NBUF][BK][BN + 1] = stage.B;
    float (*part)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(&stage);           // the partial tiles reuse the staging buffers
    const int wave_q = (threadIdx.x >> 6) % NQ, wave_k = (threadIdx.x >> 6) / NQ, qm = wave_q / TN, qn = wave_q % TN;
    static_assert(sizeof(Stage) >= NW * 32 * 33 * sizeof(float), "partial tiles must fit in the staging buffers");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc = (f32x16)0.f;
    float rs = 0.f;
    const bool do_asum = a.asum != nullptr && bx == 0;
    int total = 0;
    for (int t = 0; t < a.nterms; ++t) total += (batch.t[a.term0 + t].K + BK - 1) / BK;
    const int ktiles0 = (batch.t[a.term0].K + BK - 1) / BK;
    float ra[NLA], rb[NLB];
    int lt = 0, lk0 = 0;                     // load cursor: term, k offset inside the term (terms may have different K)
    auto gload = [&]() __attribute__((always_inline)) {
        const SgemmTerm& T = batch.t[a.term0 + lt];
        const float* A = T.A;
        const float* B = T.B;
        const int lda = T.lda, ldb = T.ldb, K = T.K, k0 = lk0;
#pragma unroll
        for (int q = 0; q < NLA; ++q) {
            const int e = threadIdx.x + NT * q;
            const int kk = a.ta ? (e / BM) : (e % BK), mm = a.ta ? (e % BM) : (e / BK);
            const int gm = m0 + mm, gk = k0 + kk;
            ra[q] = (gm < a.M && gk < K) ? (a.ta ? A[(int64_t)gk * lda + gm] : A[(int64_t)gm * lda + gk]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NLB; ++q) {
            const int e = threadIdx.x + NT * q;
            const int kk = a.tb ? (e % BK) : (e / BN), nn = a.tb ? (e / BK) : (e % BN);
            const int gk = k0 + kk, gn = n0 + nn;
            rb[q] = (gk < K && gn < a.N) ? (a.tb ? B[(int64_t)gn * ldb + gk] : B[(int64_t)gk * ldb + gn]) : 0.f;
        }
        lk0 += BK;
        if (lk0 >= K) { lk0 = 0; ++lt; }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NLA; ++q) {
            const int e = threadIdx.x + NT * q;
            As[buf][a.ta ? (e / BM) : (e % BK)][a.ta ? (e % BM) : (e / BK)] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < NLB; ++q) {
            const int e = threadIdx.x + NT * q;
            Bs[buf][a.tb ? (e % BK) : (e / BN)][a.tb ? (e / BK) : (e % BN)] = rb[q];
        }
    };
    if constexpr (NBUF == 2) {
        gload();
        lstore(0);
        if (total > 1) gload();
        __syncthreads();
        for (int it = 0; it < total; ++it) {
            const int buf = it & 1;
            if (it + 1 < total) lstore(buf ^ 1);                  // tile it+1 (loaded during the previous iteration) -> other LDS buffer
            if (it + 2 < total) gload();                          // tile it+2 in flight under the MFMAs
#pragma unroll
            for (int u = 0; u < BK / (2 * KW); ++u) {
                const int kk = wave_k * (BK / KW) + 2 * u + h;
                acc = mfma_f32(As[buf][kk][32 * qm + i], Bs[buf][kk][32 * qn + i], acc);
            }
            if (do_asum && it < ktiles0 && threadIdx.x < BM) {
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) rs += As[buf][kk][threadIdx.x];
            }
            __syncthreads();
        }
    } else {
        gload();
        for (int it = 0; it < total; ++it) {
            lstore(0);
            __syncthreads();
            if (it + 1 < total) gload();                          // next tile in flight under the MFMAs
#pragma unroll 8
            for (int u = 0; u < BK / (2 * KW); ++u) {
                const int kk = wave_k * (BK / KW) + 2 * u + h;
                acc = mfma_f32(As[0][kk][32 * qm + i], Bs[0][kk][32 * qn + i], acc);
            }
            if (do_asum && it < ktiles0 && threadIdx.x < BM) {
#pragma unroll 8
                for (int kk = 0; kk < BK; ++kk) rs += As[0][kk][threadIdx.x];
            }
            __syncthreads();
        }
    }
    // ---- fixed-order reduction over the waves of a sub-tile (wave = wave_k * NQ + wave_q)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][drow32(r, h) * 33 + i] = acc[r];
    __syncthreads();
#pragma unroll
    for (int e = threadIdx.x; e < 1024 * NQ; e += NT) {
        const int sq = e >> 10, r = (e >> 5) & 31, c = e & 31, o = r * 33 + c;
        const int gr = m0 + 32 * (sq / TN) + r, gc = n0 + 32 * (sq % TN) + c;
        if (gr < a.M && gc < a.N) {
            float v = part[sq][o];
#pragma unroll
            for (int w_ = 1; w_ < KW; ++w_) v += part[w_ * NQ + sq][o];     // fixed order: k-part 0, 1, ... (left fold)
            v += a.bias ? a.bias[gc] : 0.f;
            const int64_t idx = (int64_t)gr * a.ldc + gc;
            if (a.epi == DPN_EPI_GELU) { if (a.aux_out) a.aux_out[idx] = v; v = gelu_exact(v); }
            else if (a.epi == DPN_EPI_MUL_GELU_GRAD) v *= gelu_exact_grad(a.aux[idx]);
            else if (a.epi == DPN_EPI_ADD) v += a.aux[idx];
            a.C[idx] = v;
        }
    }
    if (do_asum && threadIdx.x < BM && m0 + threadIdx.x < a.M) a.asum[m0 + threadIdx.x] = rs;
}

// ------------------------------------------------------------------------------------------------ LayerNorm folded into a GEMM's A operand
// A dependent kernel costs >= 4.5 us on this machine whatever it does, and LayerNorm (forward or backward) on 287 x 256 does almost
// nothing.  Its consumer is always a GEMM over the FULL 256-wide rows (K = d_model = 256 = one LDS tile), so the consumer can apply it
// to its own A tile: every workgroup of a row block recomputes the 32 row statistics (cheap), the n0 == 0 workgroups write the
// normalised rows / statistics (forward) or the row gradients and the per-block parameter partial sums (backward) for later use.
//   mode 1:  A = LN(x + r) * gamma + beta      (writes y, xhat, rstd)                       -- transformer_net.py:37,44
//   mode 2:  A = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))               (writes gs and partial[block][2][256])
struct LnGemmArgs {
    int mode, M, N, tb, ldb, ldc, epi;
    const float *x, *r, *gamma, *beta, *rstd_in;
    float *y_out, *xhat_out, *rstd_out, *partial;
    const float *B, *bias;
    float* C;
    const float* aux;
    float* aux_out;
};
template <int MODE, int NT = 512>
__global__ __launch_bounds__(NT) void dpn_sgemm_ln_kernel(LnGemmArgs a) {
    constexpr int BK = 256;
    constexpr int NW = NT / 64;                 // waves: BK / NW k-values each (eight: half the MFMA chain and loads of a latency-bound tile)
    constexpr int TPR = NT / 32, CW = 256 / TPR;  // threads per row of the A tile, columns per thread
    __shared__ float As[BK][33];
    __shared__ float Bs[BK][33];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const bool keep = blockIdx.x == 0;
    // ---- B tile -> registers (stored to LDS after the LayerNorm phase, which may borrow Bs)
    constexpr int NLB = BK * 32 / NT;
    float rb[NLB];
#pragma unroll
    for (int q = 0; q < NLB; ++q) {
        const int e = t + NT * q;
        const int kk = a.tb ? (e & 255) : (e >> 5), nn = a.tb ? (e >> 8) : (e & 31);
        const int gn = n0 + nn;
        rb[q] = (gn < a.N) ? (a.tb ? a.B[(int64_t)gn * a.ldb + kk] : a.B[(int64_t)kk * a.ldb + gn]) : 0.f;
    }
    // ---- A rows in registers: thread t holds columns [CW c, CW c + CW) of row m (c = t % TPR, m = t / TPR): a row is TPR adjacent lanes,
    // so the row statistics are log2(TPR) xor-shuffles -- no LDS pass, no barrier
    const int m = t / TPR, c0 = (t % TPR) * CW, gm = m0 + m;
    const bool ok = gm < a.M;
    float v[CW], w[MODE == 2 ? CW : 1], gam[CW];
#pragma unroll
    for (int j = 0; j < CW / 4; ++j) {
        const float4 x4 = ok ? *reinterpret_cast<const float4*>(a.x + (int64_t)gm * 256 + c0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g4 = *reinterpret_cast<const float4*>(a.gamma + c0 + 4 * j);
        v[4 * j] = x4.x; v[4 * j + 1] = x4.y; v[4 * j + 2] = x4.z; v[4 * j + 3] = x4.w;
        gam[4 * j] = g4.x; gam[4 * j + 1] = g4.y; gam[4 * j + 2] = g4.z; gam[4 * j + 3] = g4.w;
        if (MODE == 1) {
            if (ok && a.r) {
                const float4 r4 = *reinterpret_cast<const float4*>(a.r + (int64_t)gm * 256 + c0 + 4 * j);
                v[4 * j] += r4.x; v[4 * j + 1] += r4.y; v[4 * j + 2] += r4.z; v[4 * j + 3] += r4.w;
            }
        } else {
            const float4 r4 = ok ? *reinterpret_cast<const float4*>(a.r + (int64_t)gm * 256 + c0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            w[4 * j] = r4.x; w[4 * j + 1] = r4.y; w[4 * j + 2] = r4.z; w[4 * j + 3] = r4.w;
        }
    }
    auto row_sum = [&](float s_) __attribute__((always_inline)) {            // over the 8 lanes of the row, fixed order
        s_ += __shfl_xor(s_, 1); s_ += __shfl_xor(s_, 2); s_ += __shfl_xor(s_, 4);
        if constexpr (TPR == 16) s_ += __shfl_xor(s_, 8);
        return s_;
    };
    if (MODE == 2 && keep) {
        // parameter partial sums of this row block need g and xhat column-wise: park them in As / Bs (both still unused), one pass
#pragma unroll
        for (int j = 0; j < CW; ++j) { As[c0 + j][m] = v[j]; Bs[c0 + j][m] = w[j]; }
        __syncthreads();
        if (t < 256) {
            float dg = 0.f, db = 0.f;
#pragma unroll 8
            for (int q = 0; q < 32; ++q) { dg = fmaf(As[t][q], Bs[t][q], dg); db += As[t][q]; }
            a.partial[(int64_t)blockIdx.y * 512 + t] = dg;
            a.partial[(int64_t)blockIdx.y * 512 + 256 + t] = db;
        }
        __syncthreads();
    }
    if (MODE == 1) {
        float s_ = 0.f;
#pragma unroll
        for (int j = 0; j < CW; ++j) s_ += v[j];
        const float mean = row_sum(s_) * (1.f / 256.f);
        float q_ = 0.f;
#pragma unroll
        for (int j = 0; j < CW; ++j) { const float d = v[j] - mean; q_ = fmaf(d, d, q_); }
        const float rstd = 1.0f / sqrtf(row_sum(q_) * (1.f / 256.f) + 1e-5f);
        if (keep && ok && (t % TPR) == 0) a.rstd_out[gm] = rstd;
#pragma unroll
        for (int j = 0; j < CW / 4; ++j) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.beta + c0 + 4 * j);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            float xh[4], yy[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { xh[e] = (v[4 * j + e] - mean) * rstd; yy[e] = fmaf(xh[e], gam[4 * j + e], bb[e]); v[4 * j + e] = yy[e]; }
            if (keep && ok) {
                *reinterpret_cast<float4*>(a.xhat_out + (int64_t)gm * 256 + c0 + 4 * j) = make_float4(xh[0], xh[1], xh[2], xh[3]);
                *reinterpret_cast<float4*>(a.y_out + (int64_t)gm * 256 + c0 + 4 * j) = make_float4(yy[0], yy[1], yy[2], yy[3]);
            }
        }
    } else {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CW; ++j) { const float tk = v[j] * gam[j]; s1 += tk; s2 = fmaf(tk, w[j], s2); }
        const float m1 = row_sum(s1) * (1.f / 256.f), m2 = row_sum(s2) * (1.f / 256.f);
        const float rs_ = ok ? a.rstd_in[gm] : 0.f;
#pragma unroll
        for (int j = 0; j < CW / 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * j + e] = rs_ * (v[4 * j + e] * gam[4 * j + e] - m1 - w[4 * j + e] * m2);
            if (keep && ok) *reinterpret_cast<float4*>(a.y_out + (int64_t)gm * 256 + c0 + 4 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        }
    }
    // ---- transformed A tile and B tile -> LDS
#pragma unroll
    for (int j = 0; j < CW; ++j) As[c0 + j][m] = v[j];
#pragma unroll
    for (int q = 0; q < NLB; ++q) {
        const int e = t + NT * q;
        Bs[a.tb ? (e & 255) : (e >> 5)][a.tb ? (e >> 8) : (e & 31)] = rb[q];
    }
    __syncthreads();
    // ---- the GEMM proper: NW waves x BK / NW k each (exact fp32 MFMA), partial tiles joined in a fixed order
    f32x16 acc = (f32x16)0.f;
#pragma unroll 8
    for (int u = 0; u < BK / (2 * NW); ++u) {
        const int kk = wave * (BK / NW) + 2 * u + h;
        acc = mfma_f32(As[kk][i], Bs[kk][i], acc);
    }
    __syncthreads();
    float (*partt)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(&As[0][0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) partt[wave][drow32(r, h) * 33 + i] = acc[r];
    __syncthreads();
#pragma unroll
    for (int e = t; e < 1024; e += NT) {
        const int r = e >> 5, c = e & 31, o = r * 33 + c;
        if (m0 + r < a.M && n0 + c < a.N) {
            float vv = partt[0][o];
#pragma unroll
            for (int w_ = 1; w_ < NW; ++w_) vv += partt[w_][o];            // fixed order: wave 0, 1, ... (left fold)
            vv += a.bias ? a.bias[n0 + c] : 0.f;
            const int64_t idx = (int64_t)(m0 + r) * a.ldc + n0 + c;
            if (a.epi == DPN_EPI_GELU) { if (a.aux_out) a.aux_out[idx] = vv; vv = gelu_exact(vv); }
            else if (a.epi == DPN_EPI_MUL_GELU_GRAD) vv *= gelu_exact_grad(a.aux[idx]);
            else if (a.epi == DPN_EPI_ADD) vv += a.aux[idx];
            a.C[idx] = vv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused clip + Adam
// clip_grad_norm_(max_norm) followed by torch.optim.Adam(lr, betas, eps, weight_decay) (L2-in-gradient, not AdamW), as in
// interface_physics.py:514-515 / cfg:151-155, for a LIST of tensors per launch (pointer table in the kernel arguments).
constexpr int kAdamMaxTensors = 72;
constexpr int kAdamChunk = 2048;                 // elements per block
struct AdamTable {
    float* p[kAdamMaxTensors];
    const float* g[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    int chunk_start[kAdamMaxTensors + 1];        // prefix sum of ceil(numel / kAdamChunk)
    int numel[kAdamMaxTensors];
    int n;
};
// Optimiser state kept by the caller as ONE flat buffer per moment, tensor i at offset chunk_start[i] * kAdamChunk (each tensor padded to
// whole chunks): no per-tensor state pointers, so 160 tensors fit in the kernel arguments and a PhysicsNet is one launch per pass.
constexpr int kAdamFlatMaxTensors = 160;
struct AdamTableFlat {
    float* p[kAdamFlatMaxTensors];
    const float* g[kAdamFlatMaxTensors];
    int chunk_start[kAdamFlatMaxTensors + 1];
    int numel[kAdamFlatMaxTensors];
    float* m_flat;
    float* v_flat;
    int n;
};
static_assert(sizeof(AdamTableFlat) + 64 <= 4096, "kernel arguments are limited to 4 KB");
DEV float* table_m(const AdamTable& t, int ti) { return t.m[ti]; }
DEV float* table_v(const AdamTable& t, int ti) { return t.v[ti]; }
DEV float* table_m(const AdamTableFlat& t, int ti) { return t.m_flat + (int64_t)t.chunk_start[ti] * kAdamChunk; }
DEV float* table_v(const AdamTableFlat& t, int ti) { return t.v_flat + (int64_t)t.chunk_start[ti] * kAdamChunk; }
template <class Table>
DEV int adam_find(const Table& t, int blk) {
    int lo = 0, hi = t.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.chunk_start[mid] <= blk) lo = mid; else hi = mid - 1; }
    return lo;
}
template <class Table>
__global__ __launch_bounds__(256) void dpn_gradnorm_kernel(Table t, double* partial, int* step, int bump_step) {
    // one fp64 partial per block (no atomics: 2.7k serialised fp64 atomics on one address cost more than reading the gradients);
    // dpn_gradnorm_reduce_kernel adds them in a fixed order -> the clip coefficient is run-to-run deterministic
    if (bump_step && blockIdx.x == 0 && threadIdx.x == 0) *step += 1;       // device-side step counter: graph replays advance it
    const int ti = adam_find(t, blockIdx.x);
    const int base = (blockIdx.x - t.chunk_start[ti]) * kAdamChunk;
    const float* g = t.g[ti];
    const int end = min(base + kAdamChunk, t.numel[ti]);
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {                         // 16-byte loads over the aligned body, scalars for the tail
        const int end4 = base + ((end - base) & ~3);
        for (int i = base + 4 * threadIdx.x; i < end4; i += 1024) {
            const float4 q = *reinterpret_cast<const float4*>(g + i);
            s = fmaf(q.x, q.x, s); s = fmaf(q.y, q.y, s); s = fmaf(q.z, q.z, s); s = fmaf(q.w, q.w, s);
        }
        for (int i = end4 + threadIdx.x; i < end; i += 256) s = fmaf(g[i], g[i], s);
    } else {
        for (int i = base + threadIdx.x; i < end; i += 256) s = fmaf(g[i], g[i], s);
    }
    double d = (double)s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void dpn_gradnorm_reduce_kernel(const double* partial, int n, double* sumsq) {
    double d = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) d += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) *sumsq = (red[0] + red[1]) + (red[2] + red[3]);
}
template <class Table>
__global__ __launch_bounds__(256) void dpn_adam_kernel(Table t, const double* sumsq, const int* step, float lr, float b1, float b2, float eps,
                                                       float wd, float max_norm, float* out_norm, const float* hyper) {
    // hyper (optional, device): [lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale] read at run time, so that a step captured in
    // a hipGraph follows a learning-rate schedule (a by-value lr is frozen into the graph); grad_scale multiplies every gradient
    // before the norm and the update (1 / world_size after a SUM all-reduce)
    float gscale = 1.f;
    if (hyper) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4]; max_norm = hyper[5]; gscale = hyper[6]; }
    const float total = (float)sqrt(*sumsq) * gscale;
    if (out_norm && blockIdx.x == 0 && threadIdx.x == 0) *out_norm = total;
    const float coef = fminf(max_norm / (total + 1e-6f), 1.0f) * gscale;    // clip_grad_norm_'s clamp(max_norm / (norm + 1e-6), max=1)
    const float st = (float)(*step);
    const float bc1 = 1.f - powf(b1, st), bc2s = sqrtf(1.f - powf(b2, st));
    const float step_size = lr / bc1;
    const int ti = adam_find(t, blockIdx.x);
    const int base = (blockIdx.x - t.chunk_start[ti]) * kAdamChunk;
    float* p = t.p[ti]; const float* g = t.g[ti]; float* m = table_m(t, ti); float* v = table_v(t, ti);
    const int end = min(base + kAdamChunk, t.numel[ti]);
    auto upd = [&](float& pi, const float graw, float& mi, float& vi) __attribute__((always_inline)) {
        const float gi = fmaf(wd, pi, graw * coef);
        mi = fmaf(b1, mi, (1.f - b1) * gi);                                  // lerp(m, g, 1-b1)
        vi = fmaf(b2, vi, (1.f - b2) * gi * gi);
        pi = pi - step_size * mi / (sqrtf(vi) / bc2s + eps);
    };
    int scalar_from = base;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        const int end4 = base + ((end - base) & ~3);
        for (int i = base + 4 * threadIdx.x; i < end4; i += 1024) {
            // the moments and the gradient are touched by nobody else: streamed past the caches (the parameters stay cacheable, the
            // next step reads them first)
            typedef __attribute__((ext_vector_type(4))) float f32x4_t;
            float4 P = *reinterpret_cast<float4*>(p + i);
            // (round 6 once more, rocprofv3 in the step: the gradient loaded without the hint 27.2, all three loads without 27.6, stores too 28.3 against 25.2 us)
            const f32x4_t Mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(m + i));
            const f32x4_t Vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(v + i));
            const f32x4_t Gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(g + i));
            float4 M = make_float4(Mv[0], Mv[1], Mv[2], Mv[3]), V = make_float4(Vv[0], Vv[1], Vv[2], Vv[3]);
            const float4 G = make_float4(Gv[0], Gv[1], Gv[2], Gv[3]);
            upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
            *reinterpret_cast<float4*>(p + i) = P;
            __builtin_nontemporal_store(f32x4_t{M.x, M.y, M.z, M.w}, reinterpret_cast<f32x4_t*>(m + i));
            __builtin_nontemporal_store(f32x4_t{V.x, V.y, V.z, V.w}, reinterpret_cast<f32x4_t*>(v + i));
        }
        scalar_from = end4;
    }
    for (int i = scalar_from + threadIdx.x; i < end; i += 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        upd(pi, g[i], mi, vi);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// ------------------------------------------------------------------------------------------------ MFMA layout self-test
__global__ void dpn_selftest_kernel(float* out) {
    // A = I (32x32 over two k-steps of 16) against B1[k][j] = k and B2[k][j] = j: D1[i][j] = i, D2[i][j] = j.
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    f32x16 acc1 = (f32x16)0.f, acc2 = (f32x16)0.f;
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 A, B1, B2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * h + e;        // this (ks,h,e) k-slot <-> index k (any bijection works as long as A and B agree)
            A[e] = (__bf16)((k == i) ? 1.f : 0.f);
            B1[e] = (__bf16)(float)k;
            B2[e] = (__bf16)(float)i;
        }
        acc1 = mfma(A, B1, acc1);
        acc2 = mfma(A, B2, acc2);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { out[lane * 16 + r] = acc1[r]; out[1024 + lane * 16 + r] = acc2[r]; }
}

#endif  // DPN_HAS_REST

// ------------------------------------------------------------------------------------------------ C ABI
static inline int64_t pad_points(int64_t n) { return ((n + 127) / 128) * 128; }
// hi+lo mode: the tile-split kernels (dpn_fwd_tiles.h; 64 points per workgroup, two workgroups per CU).  Caller-encoded coordinates and the
// single-bf16 mode stay on the ring kernels.  DPN_FWD_KERNEL / DPN_BWD_KERNEL = ring | tiles override (read per call: the tests compare the two
// decompositions inside one process).
static inline int order_reversed(const char* knob) {          // read on every launch: a measurement switch (tools set it between runs of one process)
    const char* e = getenv(knob);
    return e && e[0] == 'r' ? 1 : 0;                     // "reverse"
}
#ifndef DPN_FWD_PP_DEFAULT
#define DPN_FWD_PP_DEFAULT 0
#endif
static inline bool use_pp() {
    const char* e = getenv("DPN_FWD_PP");
    return e ? (e[0] == '1') : (DPN_FWD_PP_DEFAULT != 0);
}
#ifndef DPN_FWD_PERSIST_DEFAULT
#define DPN_FWD_PERSIST_DEFAULT 0
#endif
static inline bool use_persist() {                      // DPN_FWD_PERSIST=0|1, read per call (A/B and bitwise comparison inside one process)
    const char* e = getenv("DPN_FWD_PERSIST");
    return e ? (e[0] == '1') : (DPN_FWD_PERSIST_DEFAULT != 0);
}
static inline int cu_count() {                          // compute units of the current device (one persistent workgroup each)
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
static inline bool use_tiles(const char* knob, int prec, bool has_pe_in) {
    const char* force = getenv(knob);
    return (force ? (force[0] == 't') : (prec == 2)) && !has_pe_in;
}
#if DPN_HAS_REST
// Point ranges per product.  One 8-wave workgroup per CU (the LDS ring fills it) and a kernel time that falls as 1 / workgroups up to
// one round (measured, hi+lo mode, 37 265 points: 120 workgroups 671 us, 192 452 us, 240 396 us, 288 562 us -- the tail round), so
// the plan fills one round of the 256 CUs: 42 workgroups per net.  Hi+lo mode: dw1 = T1^T Z0 is the one product whose slot (two X planes)
// leaves room for a ring of two only, so its tiles take longest and it gets the most ranges; measured (tools/wgrad_overlap_probe.py,
// profiles/round3_wgrad_plans.txt): 10,11,10,11 327 us, 11,11,9,11 303 us, 10,10,9,13 277 us, 10,10,8,14 277 us, 9,9,8,16 280 us --
// a plateau at 5.0 TB/s.  Single bf16 (rings of four and five): 10,11,10,11 156 us, 10,10,9,13 160-163 us.
struct SplitPlan { int s[4]; int most; };
// Round 5: three products (s[0] = 0: M2^T Z is gone, dpn_finish_gside_kernel); product 2 forms its Y operand G6 = g pe6 in registers from the
// per-point table (OperandView), which makes ITS tiles the slowest: it gets the most ranges per byte.  Sweeps of the 42 ranges per net
// (tools/wgrad_overlap_probe.py, profiles/round5_wgrad_plans.txt; eager launches back to back): hi+lo 14,12,16 277 us, 15,12,15 244, 14,13,15 234,
// 13,13,16 228, 15,13,14 233, 13,14,15 224; single bf16 14,13,15 163 us, 15,13,14 136, 14,12,16 137, 13,13,16 129, 15,12,15 133, 16,13,13 132.
static inline SplitPlan choose_plan(int64_t n_pad, int ns) {
    int64_t c = n_pad / 32 / 16;
    if (c < 1) c = 1;
    SplitPlan p;
    if (c >= 10) p = (ns == 2) ? SplitPlan{{0, 13, 14, 15}, 15} : SplitPlan{{0, 13, 13, 16}, 16};
    else p = SplitPlan{{0, (int)c, (int)c, (int)c}, (int)c};
#ifdef DPN_EXPERIMENT_SPLITS                     // timing experiments only: DPN_WGRAD_PLAN="9,12,10,11"
    if (const char* e = getenv("DPN_WGRAD_PLAN")) {
        if (sscanf(e, "%d,%d,%d,%d", &p.s[0], &p.s[1], &p.s[2], &p.s[3]) == 4) {
            p.most = 1;
            p.s[0] = 0;
            for (int k = 1; k < 4; ++k) { if (p.s[k] < 1) p.s[k] = 1; if (p.s[k] > kMaxSplits) p.s[k] = kMaxSplits; if (p.s[k] > p.most) p.most = p.s[k]; }
        }
    }
#endif
    return p;
}
#endif  // DPN_HAS_REST
#if DPN_HAS_REST
// A device-clock stamp as a graph node: HIP event records inside a stream capture are not timing events (and torch refuses external events on ROCm),
// so a measurement INSIDE a replayed hipGraph puts this one-thread kernel in front of and behind the launch it brackets.  wall_clock64() is the
// constant-rate counter HIP events read (hipDeviceAttributeWallClockRate, 100 MHz on gfx950).
__global__ void dpn_clock_stamp_kernel(unsigned long long* ring, unsigned int* cursor, unsigned int cap) {
    const unsigned long long t = wall_clock64();
    ring[atomicAdd(cursor, 1u) % cap] = t;
}
#endif
constexpr int64_t kFinishScratchFloats = (int64_t)kNets * 65536 + (int64_t)kNets * 256 * 192 + 4 * kNets * 256 + 8 + (int64_t)kNets * 8 * 256;   // S1 | S2 | mv | u | q1 | q6 | sum g | r parts (FinishArgs)
static inline int ck(hipError_t e) { return (int)e; }

extern "C" {

#if DPN_HAS_REST
int dpn_version(void) { return 2; }

int dpn_clock_stamp(unsigned long long* ring, unsigned int* cursor, unsigned int cap, void* stream) {
    if (!ring || !cursor || cap == 0) return -1;
    hipLaunchKernelGGL(dpn_clock_stamp_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), ring, cursor, cap);
    return ck(hipGetLastError());
}

int dpn_clock_rate_khz(int* khz) {
    if (!khz) return -1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return ck(e);
    return ck(hipDeviceGetAttribute(khz, hipDeviceAttributeWallClockRate, dev));
}

// which packed form (dpn_pack_weights_form) the forward launch of this precision mode expects: 1 = fused (tile-split kernel), 0 = ring stream
int dpn_fwd_form(int prec, int has_pe_in) { return use_tiles("DPN_FWD_KERNEL", prec, has_pe_in != 0) ? 1 : 0; }

int dpn_sizes(int64_t n, int prec, DpnSizes* out) {
    if (!out || n <= 0 || (prec != 1 && prec != 2)) return -1;
    const int64_t n_pad = pad_points(n);
    out->n_pad = n_pad;
    out->packed = (((int64_t)kNets * pack_bytes_per_net(prec) + 255) / 256) * 256;
    out->saved = saved_bytes(n_pad, prec);
    out->operands = operand_bytes(n_pad, prec);
    out->k_splits = choose_plan(n_pad, prec).most;
    out->partials = ((int64_t)out->k_splits * kNets * kPartFloats + kFinishScratchFloats) * 4;
    return 0;
}

static int sgemm_batch_launch(int n_problems, const DpnGemmProblem* problems, int n_jobs, const DpnColsumJob* jobs, void* stream);

// form 0: the seven-GEMM stream of the ring kernels; form 1: the fused five-GEMM stream of dpn_fwd_tiles_kernel (dpn_layout.h) -- ONE launch forms
// A = W1 w2, B = W1 Wd and C2 = W1 cvec + bf1 on the exact-fp32 matrix instruction and writes them as fragments (dpn_pack_fused_kernel)
int dpn_pack_weights_batch(const DpnNetPtrs nets[DPN_NETS], int n_fields, int64_t heads_stride, int64_t evec_stride, int prec, int form, void* packed,
                           int64_t packed_stride, void* stream) {
    if (!nets || !packed || (prec != 1 && prec != 2) || (form != 0 && form != 1) || n_fields < 1 || n_fields > 65535 / kNets) return -1;
    if (n_fields > 1 && (heads_stride <= 0 || evec_stride <= 0 || packed_stride < (int64_t)kNets * pack_bytes_per_net(prec) || (packed_stride & 15))) return -1;
    PackArgs a;
    for (int k = 0; k < kNets; ++k) a.net[k] = nets[k];
    a.packed = reinterpret_cast<char*>(packed);
    a.ns = prec;
    a.form = form;
    a.n_fields = n_fields; a.heads_stride = heads_stride; a.evec_stride = evec_stride; a.packed_stride = packed_stride;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (form == 1) {                                                      // products + packing in one launch (no fp32 scratch)
        hipLaunchKernelGGL(dpn_pack_fused_kernel, dim3(kFusedBlocks, kNets * n_fields), dim3(256), 0, s, a);
        return ck(hipGetLastError());
    }
    hipLaunchKernelGGL(dpn_pack_matrices_kernel, dim3(40 + kVecParts, kNets * n_fields), dim3(256), 0, s, a);
    return ck(hipGetLastError());
}

int dpn_pack_weights_form(const DpnNetPtrs nets[DPN_NETS], int prec, int form, void* packed, void* stream) {
    return dpn_pack_weights_batch(nets, 1, 0, 0, prec, form, packed, 0, stream);
}

int dpn_pack_weights(const DpnNetPtrs nets[DPN_NETS], int prec, void* packed, void* stream) {
    return dpn_pack_weights_form(nets, prec, dpn_fwd_form(prec, 0), packed, stream);
}

#endif  // DPN_HAS_REST
#if DPN_HAS_POINT
#ifdef DPN_TIMELINE
static unsigned* g_timeline = nullptr;
int dpn_debug_set_timeline(void* buf) { g_timeline = reinterpret_cast<unsigned*>(buf); return 0; }    // experiment build only, not in dpn_hip.h
#endif
static int fwd_launch(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, const float* ref_data, int64_t n,
                      const float* freqs, const DpnGeometry* geo, const void* packed, int prec, float* out_n, float* jac_n, void* saved, void* stream,
                      int n_nets);
int dpn_fwd_ref(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, const float* ref_data, int64_t n,
                const float* freqs, const DpnGeometry* geo, const void* packed, int prec, float* out_n, float* jac_n, void* saved, void* stream) {
    return fwd_launch(x, y, t, pe_in, coord_data, ref_data, n, freqs, geo, packed, prec, out_n, jac_n, saved, stream, kNets);
}
// the first n_nets VariableNets only (inference: nothing is saved); the other columns of out_n / jac_n are left untouched
int dpn_fwd_ref_nets(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, const float* ref_data, int64_t n,
                     const float* freqs, const DpnGeometry* geo, const void* packed, int prec, int n_nets, float* out_n, float* jac_n, void* stream) {
    if (n_nets < 1 || n_nets > kNets) return -1;
    return fwd_launch(x, y, t, pe_in, coord_data, ref_data, n, freqs, geo, packed, prec, out_n, jac_n, nullptr, stream, n_nets);
}
static int fwd_launch(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, const float* ref_data, int64_t n,
                      const float* freqs, const DpnGeometry* geo, const void* packed, int prec, float* out_n, float* jac_n, void* saved, void* stream,
                      int n_nets) {
    if (!coord_data || !freqs || !geo || !packed || !out_n || n <= 0 || (prec != 1 && prec != 2)) return -1;
    if (!pe_in && (!x || !y || !t)) return -1;
#ifdef DPN_TIMELINE
    FwdArgs a{x, y, t, coord_data, freqs, pe_in, n, pad_points(n), *geo, reinterpret_cast<const char*>(packed), out_n, jac_n, saved, ref_data, g_timeline};
#else
    FwdArgs a{x, y, t, coord_data, freqs, pe_in, n, pad_points(n), *geo, reinterpret_cast<const char*>(packed), out_n, jac_n, saved, ref_data};
#endif
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(a.n_pad / 128), n_nets);
#ifdef DPN_EXPERIMENT_FWD2
    if (!pe_in && getenv("DPN_FWD2") != nullptr) {          // experiment build only: the shelved eight-wave kernel
        if (prec == 1) hipLaunchKernelGGL(dpn_fwd2_kernel<1>, grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL(dpn_fwd2_kernel<2>, grid, dim3(512), 0, s, a);
        return ck(hipGetLastError());
    }
#endif
    // hi+lo mode: the tile-split kernel (dpn_fwd_tiles.h; 64 points per workgroup, two workgroups per CU).  Caller-encoded coordinates
    // and the single-bf16 mode stay on the ring kernel (one bf16 product per fragment pair cannot pay for the doubled weight stream).
    // DPN_FWD_KERNEL=ring|tiles overrides (A/B measurements, bitwise comparison of the two kernels in the tests).
    if (use_tiles("DPN_FWD_KERNEL", prec, pe_in != nullptr)) {    // (expects the FUSED packed form: dpn_fwd_form)
        // Training shape (saved state AND Jacobian wanted) in the hi+lo mode: the ping-pong form (dpn_fwd_pp.h), bit-identical results.  DPN_FWD_PP=0|1
        // overrides (read per call, like DPN_FWD_KERNEL: the tests compare the two forms inside one process).
        if (prec == 2 && saved && jac_n && use_persist() && a.n_pad / 64 * n_nets < (1ll << 31)) {
            const int64_t items = (a.n_pad / 64) * n_nets;
            const int wgs = 2 * cu_count();                      // two workgroups per CU (72 KB of LDS, <= 256 registers each)
            const dim3 gridp((unsigned)(items < wgs ? items : wgs));
            hipLaunchKernelGGL(dpn_fwd_tiles_persist_kernel<2>, gridp, dim3(256), 0, s, a, n_nets);
            return ck(hipGetLastError());
        }
        if (prec == 2 && saved && jac_n && use_pp()) {
            const int64_t items = (a.n_pad / 128) * n_nets;
            const int cus = cu_count();
            const dim3 gridp((unsigned)(items < cus ? items : cus));
            hipLaunchKernelGGL(dpn_fwd_pp_kernel<2>, gridp, dim3(512), 0, s, a, n_nets);
            return ck(hipGetLastError());
        }
        const dim3 grid64((unsigned)(a.n_pad / 64), n_nets);
        if (prec == 1) hipLaunchKernelGGL(dpn_fwd_tiles_kernel<1>, grid64, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(dpn_fwd_tiles_kernel<2>, grid64, dim3(256), 0, s, a);
        return ck(hipGetLastError());
    }
    if (prec == 1) hipLaunchKernelGGL(dpn_fwd_kernel<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(dpn_fwd_kernel<2>, grid, dim3(256), 0, s, a);
    return ck(hipGetLastError());
}

int dpn_fwd(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n, const float* freqs,
            const DpnGeometry* geo, const void* packed, int prec, float* out_n, float* jac_n, void* saved, void* stream) {
    return dpn_fwd_ref(x, y, t, pe_in, coord_data, nullptr, n, freqs, geo, packed, prec, out_n, jac_n, saved, stream);
}

#endif  // DPN_HAS_POINT
#if DPN_HAS_REST
int dpn_contract_gpe(const float* g_out, const float* gpe, int64_t n, float* g_pe, void* stream) {
    if (!g_out || !gpe || !g_pe || n <= 0) return -1;
    hipLaunchKernelGGL(dpn_contract_gpe_kernel, dim3((unsigned)n), dim3(192), 0, reinterpret_cast<hipStream_t>(stream), g_out, gpe, n, g_pe);
    return ck(hipGetLastError());
}

int dpn_residual(const float* out_n, const float* jac_n, const float* f, int64_t n, const DpnGeometry* geo, const DpnPhysics* phys,
                 const float* gl, const float* gtot, double* loss_sums, float* g_out, float* g_jxi, void* stream) {
    if (!out_n || !jac_n || !f || !geo || !phys || n <= 0 || (g_out && !g_jxi)) return -1;
    if (phys->criterion < DPN_CRIT_MSE || phys->criterion > DPN_CRIT_SMOOTH_L1 || (phys->criterion == DPN_CRIT_SMOOTH_L1 && !(phys->beta > 0.f))) return -1;
    ResArgs a{out_n, jac_n, f, n, *geo, *phys, gl, gtot, loss_sums, g_out, g_jxi};
    hipLaunchKernelGGL(dpn_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return ck(hipGetLastError());
}

int dpn_residual_finish_batch(const double* loss_sums, int64_t n, int n_fields, const DpnPhysics* phys, float* losses, void* stream) {
    if (!loss_sums || !phys || !losses || n <= 0 || n_fields < 1) return -1;
    hipLaunchKernelGGL(dpn_residual_finish_kernel, dim3(n_fields), dim3(384), 0, reinterpret_cast<hipStream_t>(stream), loss_sums, n, *phys, losses);
    return ck(hipGetLastError());
}
int dpn_residual_finish(const double* loss_sums, int64_t n, const DpnPhysics* phys, float* losses, void* stream) {
    return dpn_residual_finish_batch(loss_sums, n, 1, phys, losses, stream);
}

int dpn_smooth_l1(const float* out_n, const float* labels, int64_t n, float beta, float scale, double* loss_sum, float* g_out, int accumulate,
                  const float* scale_dev, void* stream) {
    if (!out_n || !labels || n <= 0) return -1;
    hipLaunchKernelGGL(dpn_smooth_l1_kernel, dim3((unsigned)((n * 6 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       out_n, labels, n, beta, scale, loss_sum, g_out, accumulate, scale_dev);
    return ck(hipGetLastError());
}

#endif  // DPN_HAS_REST
#if DPN_HAS_POINT
static int bwd_points_launch(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n, const float* freqs,
                             const DpnGeometry* geo, const void* packed, int prec, const float* g_out, const float* g_jxi, const float* g_scale,
                             const void* saved, void* operands, void* stream);
int dpn_bwd_points(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n, const float* freqs,
                   const DpnGeometry* geo, const void* packed, int prec, const float* g_out, const float* g_jxi, const void* saved,
                   void* operands, void* stream) {
    return bwd_points_launch(x, y, t, pe_in, coord_data, n, freqs, geo, packed, prec, g_out, g_jxi, nullptr, saved, operands, stream);
}
int dpn_bwd_points_scaled(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n, const float* freqs,
                          const DpnGeometry* geo, const void* packed, int prec, const float* g_out, const float* g_jxi, const float* g_scale,
                          const void* saved, void* operands, void* stream) {
    return bwd_points_launch(x, y, t, pe_in, coord_data, n, freqs, geo, packed, prec, g_out, g_jxi, g_scale, saved, operands, stream);
}
static int bwd_points_launch(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n, const float* freqs,
                             const DpnGeometry* geo, const void* packed, int prec, const float* g_out, const float* g_jxi, const float* g_scale,
                             const void* saved, void* operands, void* stream) {
    if (!coord_data || !freqs || !geo || !packed || !g_out || !saved || !operands || n <= 0 || (prec != 1 && prec != 2)) return -1;
    if (pe_in ? (g_jxi != nullptr) : (!x || !y || !t)) return -1;
#ifdef DPN_TIMELINE
    BwdArgs a{x, y, t, coord_data, freqs, pe_in, n, pad_points(n), *geo, reinterpret_cast<const char*>(packed), g_out, g_jxi, g_scale,
              const_cast<void*>(saved), operands, order_reversed("DPN_BWD_ORDER"), g_timeline};
#else
    BwdArgs a{x, y, t, coord_data, freqs, pe_in, n, pad_points(n), *geo, reinterpret_cast<const char*>(packed), g_out, g_jxi, g_scale,
              const_cast<void*>(saved), operands, order_reversed("DPN_BWD_ORDER")};
#endif
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(a.n_pad / 128), kNets);
    if (use_tiles("DPN_BWD_KERNEL", prec, pe_in != nullptr)) {    // ring | tiles: bit-identical operands (tests); both read w1 at S0 of either packed form
        const dim3 grid64((unsigned)(a.n_pad / 64), kNets);
        if (prec == 1) hipLaunchKernelGGL(dpn_bwd_tiles_kernel<1>, grid64, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(dpn_bwd_tiles_kernel<2>, grid64, dim3(256), 0, s, a);
        return ck(hipGetLastError());
    }
    if (prec == 1) hipLaunchKernelGGL(dpn_bwd_kernel<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(dpn_bwd_kernel<2>, grid, dim3(256), 0, s, a);
    return ck(hipGetLastError());
}

#endif  // DPN_HAS_POINT
#if DPN_HAS_REST
#ifdef DPN_WGRAD_PHASES
static unsigned* g_wgrad_phases = nullptr;
int dpn_debug_set_wgrad_phases(void* buf) { g_wgrad_phases = reinterpret_cast<unsigned*>(buf); return 0; }   // experiment build only, not in dpn_hip.h
#endif
int dpn_wgrad(int64_t n, int prec, const float* g_out, const void* saved, const void* operands, void* partials, void* stream) {
    if (!g_out || !saved || !operands || !partials || n <= 0 || (prec != 1 && prec != 2)) return -1;
    const SplitPlan plan = choose_plan(pad_points(n), prec);
    WgradArgs a{n, pad_points(n), {plan.s[0], plan.s[1], plan.s[2], plan.s[3]}, const_cast<void*>(saved), const_cast<void*>(operands),
                reinterpret_cast<float*>(partials), order_reversed("DPN_WGRAD_ORDER")};
#ifdef DPN_WGRAD_PHASES
    a.phases = g_wgrad_phases;
#endif
    (void)g_out;   // the per-net cotangents were staged into `operands` by dpn_bwd_points
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(plan.s[1] + plan.s[2] + plan.s[3], kNets);
    if (prec == 1) hipLaunchKernelGGL(dpn_wgrad_kernel<1>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(dpn_wgrad_kernel<2>, grid, dim3(512), 0, s, a);
    return ck(hipGetLastError());
}

// parts: bit 0 = the hyper-network's half (dpn_finish_rows_kernel, dpn_finish_vside_kernel: d w1b1, d w2b2, d evec; dWd, d bd), bit 1 = the half
// that ends in static tensors only (dpn_finish_gside_kernel, dpn_finish_fc2_kernel: d cat_fc1.fc.0 / fc.2, d out_fc) and reads what half 0 left in
// the scratch tail of `partials`: same stream, or another one ordered behind half 0 (a side branch beside the encoder's backward)
int dpn_wgrad_finish_parts(const DpnNetPtrs nets[DPN_NETS], const void* packed, int64_t n, int prec, const void* partials,
                           const DpnNetGradPtrs grads[DPN_NETS], int parts, void* stream) {
    if (!nets || !packed || !partials || !grads || n <= 0 || (prec != 1 && prec != 2) || !(parts & 3)) return -1;
    FinishArgs a;
    for (int k = 0; k < kNets; ++k) { a.net[k] = nets[k]; a.grad[k] = grads[k]; }
    a.packed = reinterpret_cast<const char*>(packed);
    a.partials = reinterpret_cast<const float*>(partials);
    const SplitPlan plan = choose_plan(pad_points(n), prec);
    for (int k = 0; k < 4; ++k) a.splits[k] = plan.s[k];
    a.ns = prec;
    a.n = n;
    // scratch in the tail of the partials buffer (dpn_sizes)
    a.scratch_s1 = const_cast<float*>(a.partials) + (int64_t)plan.most * kNets * kPartFloats;
    a.scratch_s2 = a.scratch_s1 + (int64_t)kNets * 65536;
    a.scratch_mv = a.scratch_s2 + (int64_t)kNets * 256 * 192;
    a.scratch_u = a.scratch_mv + kNets * 256;
    a.scratch_q1 = a.scratch_u + kNets * 256;
    a.scratch_q6 = a.scratch_q1 + kNets * 256;
    a.scratch_sg = a.scratch_q6 + kNets * 256;
    a.scratch_rp = a.scratch_sg + 8;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (parts & 1) hipLaunchKernelGGL(dpn_finish_rows_kernel, dim3(256, kNets), dim3(256), 0, s, a);
    const int n_v = (parts & 1) ? kVsideBlocks : 0, n_g = (parts & 2) ? kGsideBlocks : 0;
    hipLaunchKernelGGL(dpn_finish_sides_kernel, dim3(n_v + n_g, kNets), dim3(256), 0, s, a, n_v);
    if (parts & 2) hipLaunchKernelGGL(dpn_finish_fc2_kernel, dim3(256, kNets), dim3(256), 0, s, a);
    return ck(hipGetLastError());
}

int dpn_wgrad_finish(const DpnNetPtrs nets[DPN_NETS], const void* packed, int64_t n, int prec, const void* partials,
                     const DpnNetGradPtrs grads[DPN_NETS], void* stream) {
    return dpn_wgrad_finish_parts(nets, packed, n, prec, partials, grads, 3, stream);
}

int dpn_sgemm(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
              const float* bias, float* asum, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return -1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tiles = ((N + 31) / 32) * ((M + 31) / 32);
    int splits = 1;
    if (workspace && tiles < 256 && K >= 1024) {                 // long reductions with few output tiles: split K over the grid
        splits = (512 + tiles - 1) / tiles;
        const int maxs = K / 256;
        if (splits > maxs) splits = maxs;
        if (splits > 32) splits = 32;
        while (splits > 1 && (int64_t)splits * ((int64_t)M * N + M) * 4 > workspace_bytes) --splits;
        if (splits < 1) splits = 1;
    }
    int kps = (K + splits - 1) / splits;
    kps = ((kps + 31) / 32) * 32;
    splits = (K + kps - 1) / kps;
    SgemmArgs a{A, B, bias, C, asum, reinterpret_cast<float*>(workspace), M, N, K, lda, ldb, ldc, ta, tb, accumulate, kps};
    hipLaunchKernelGGL(dpn_sgemm_kernel, dim3((N + 31) / 32, (M + 31) / 32, splits), dim3(256), 0, s, a);
    if (splits > 1) {
        const int64_t work = (int64_t)M * N > M ? (int64_t)M * N : M;
        hipLaunchKernelGGL(dpn_sgemm_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a, splits);
    }
    return ck(hipGetLastError());
}

// measured on the encoder backward (A2 phase): threshold 1 -> 778 us, 2 -> 804 us, 3 -> 798 us: multi-tile problems want the
// double-buffered <64,2> pipeline
constexpr int kSingleStageMaxTiles = 1;
constexpr long kSingleStageMaxOutTiles = 512;      // one round of <256,1>; thresholds 256 ... 700 measure the same, none or 1024 worse
static int sgemm_batch_launch(int n_problems, const DpnGemmProblem* problems, int n_jobs, const DpnColsumJob* jobs, void* stream) {
    if (n_problems <= 0 || n_problems > kBatchMaxProblems || !problems || n_jobs < 0 || n_jobs > kBatchMaxJobs || (n_jobs && !jobs)) return -1;
    static_assert(sizeof(SgemmBatch) <= 4096, "kernel arguments");
    SgemmBatch b;
    b.n = n_problems;
    int gx = 0, gy = 0, pool = 0;
    for (int i = 0; i < n_problems; ++i) {
        const DpnGemmProblem& q = problems[i];
        if (q.nterms < 1 || q.nterms > kBatchMaxTerms || pool + q.nterms > kBatchTermPool || !q.C || q.M <= 0 || q.N <= 0) return -1;
        if (q.epi < 0 || q.epi > DPN_EPI_ADD || ((q.epi == DPN_EPI_MUL_GELU_GRAD || q.epi == DPN_EPI_ADD) && !q.aux)) return -1;
        if (q.asum && q.nterms != 1) return -1;                       // row sums of A are defined for a single term
        SgemmProblem& p = b.p[i];
        p.term0 = pool;
        for (int t = 0; t < q.nterms; ++t) {
            const int kt = q.k_term[t] > 0 ? q.k_term[t] : q.K;
            if (!q.A[t] || !q.B[t] || kt <= 0) return -1;
            b.t[pool++] = SgemmTerm{q.A[t], q.B[t], q.lda[t], q.ldb[t], kt, 0};
        }
        p.bias = q.bias; p.C = q.C; p.asum = q.asum; p.aux = q.aux; p.aux_out = q.aux_out; p.epi = q.epi;
        p.M = q.M; p.N = q.N; p.ldc = q.ldc; p.ta = q.ta; p.tb = q.tb; p.nterms = q.nterms;
        gx = gx > (q.N + 31) / 32 ? gx : (q.N + 31) / 32;
        gy = gy > (q.M + 31) / 32 ? gy : (q.M + 31) / 32;
    }
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].partial || !jobs[i].out_a || !jobs[i].out_b || jobs[i].n_blocks <= 0) return -1;
        b.job[i] = SgemmColsum{jobs[i].partial, jobs[i].out_a, jobs[i].out_b, jobs[i].n_blocks, 0};
    }
    const int gz = n_problems + n_jobs;
    // <256,1> (one LDS stage of 256 k) when every problem is a single such tile AND the launch is one round of it (68 KB of LDS: two
    // workgroups per CU, 512 at a time); launches with more output tiles than that (batches of fields: 548 row tiles per GEMM) run
    // better four to a CU in the pipelined <64,2> form (single field: 1.309 -> 1.286 ms per step, the heads' 728-tile launch among
    // them; configs[2]: 51.1 -> 50.4 ms)
    bool single_tile = true;
    long out_tiles = 0;
    for (int i = 0; i < n_problems; ++i) {
        int tiles = 0;
        for (int t = 0; t < b.p[i].nterms; ++t) tiles += (b.t[b.p[i].term0 + t].K + 255) / 256;
        single_tile = single_tile && tiles <= kSingleStageMaxTiles;
        out_tiles += (long)((b.p[i].M + 31) / 32) * ((b.p[i].N + 31) / 32);
    }
    single_tile = single_tile && out_tiles <= kSingleStageMaxOutTiles;
    if (single_tile) {
        hipLaunchKernelGGL((dpn_sgemm_batch_kernel<256, 1, 512>), dim3(gx, gy, gz), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), b);
        return ck(hipGetLastError());
    }
    // Round 6: 64 x 64 output tiles (four 32 x 32 sub-tiles on eight waves: half the L2 -> LDS bytes per MAC) are in the kernel and OFF:
    // measured slower on every launch of the step -- token convolution + assemble 27.3 against 25.0 us (captured and replayed, tools/embed_parts_bench.py), the
    // heads' forward 17.7 against 16.3, their backward 34.6 against 26.9 us (rocprofv3): these GEMMs wait for latency, not for L2 bytes, and a quarter of the
    // workgroups with four times the serial work each is the wrong trade (profiles/round6_sgemm_tile_ab.txt).  DPN_SGEMM_TILE=64 selects it.
    const char* force = getenv("DPN_SGEMM_TILE");
    const bool big = force && force[0] == '6';
    if (big) {
        const int gx64 = (gx + 1) / 2, gy64 = (gy + 1) / 2;
        hipLaunchKernelGGL((dpn_sgemm_batch_kernel<64, 2, 512, 2, 2>), dim3(gx64, gy64, gz), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), b);
    } else {
        hipLaunchKernelGGL((dpn_sgemm_batch_kernel<64, 2, 512>), dim3(gx, gy, gz), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), b);
    }
    return ck(hipGetLastError());
}

int dpn_sgemm_batch(int n_problems, const DpnGemmProblem* problems, void* stream) {
    return sgemm_batch_launch(n_problems, problems, 0, nullptr, stream);
}

int dpn_sgemm_batch_jobs(int n_problems, const DpnGemmProblem* problems, int n_jobs, const DpnColsumJob* jobs, void* stream) {
    return sgemm_batch_launch(n_problems, problems, n_jobs, jobs, stream);
}

int dpn_sgemm_ln(const DpnLnGemm* q, void* stream) {
    if (!q || (q->mode != 1 && q->mode != 2) || q->M <= 0 || q->N <= 0 || !q->x || !q->gamma || !q->B || !q->C || !q->y_out) return -1;
    if (q->mode == 1 && (!q->beta || !q->xhat_out || !q->rstd_out)) return -1;
    if (q->mode == 2 && (!q->r || !q->rstd_in || !q->partial)) return -1;
    if (q->epi < 0 || q->epi > DPN_EPI_ADD || ((q->epi == DPN_EPI_MUL_GELU_GRAD || q->epi == DPN_EPI_ADD) && !q->aux)) return -1;
    LnGemmArgs a{q->mode, q->M, q->N, q->tb, q->ldb, q->ldc, q->epi, q->x, q->r, q->gamma, q->beta, q->rstd_in, q->y_out, q->xhat_out,
                 q->rstd_out, q->partial, q->B, q->bias, q->C, q->aux, q->aux_out};
    const dim3 grid((q->N + 31) / 32, (q->M + 31) / 32);
    if (q->mode == 1) hipLaunchKernelGGL((dpn_sgemm_ln_kernel<1, 512>), grid, dim3(512), 0, reinterpret_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL((dpn_sgemm_ln_kernel<2, 512>), grid, dim3(512), 0, reinterpret_cast<hipStream_t>(stream), a);
    return ck(hipGetLastError());
}

int64_t dpn_clip_adam_scratch_doubles(int n_tensors, const int64_t* numel) {
    if (n_tensors <= 0 || !numel) return -1;
    int64_t chunks = 0;
    for (int i = 0; i < n_tensors; ++i) chunks += (numel[i] + kAdamChunk - 1) / kAdamChunk;
    return 1 + chunks;
}

int dpn_clip_adam(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const int64_t* numel, double* scratch_dev, int* step_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float max_norm, float* out_norm_dev, void* stream) {
    if (n_tensors <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !scratch_dev || !step_dev) return -1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* sumsq = scratch_dev;                 // [0]: sum of squares of all gradients; [1 ..]: one partial per 2048-element chunk
    double* partial = scratch_dev + 1;
    for (int pass = 0; pass < 2; ++pass) {
        int base_chunk = 0;
        for (int t0 = 0; t0 < n_tensors; t0 += kAdamMaxTensors) {
            AdamTable t;
            t.n = (n_tensors - t0 < kAdamMaxTensors) ? n_tensors - t0 : kAdamMaxTensors;
            int chunks = 0;
            for (int i = 0; i < t.n; ++i) {
                if (numel[t0 + i] <= 0 || numel[t0 + i] > 0x7fffffff) return -1;
                t.p[i] = params[t0 + i]; t.g[i] = grads[t0 + i]; t.m[i] = exp_avg[t0 + i]; t.v[i] = exp_avg_sq[t0 + i];
                t.numel[i] = (int)numel[t0 + i];
                t.chunk_start[i] = chunks;
                chunks += (t.numel[i] + kAdamChunk - 1) / kAdamChunk;
            }
            t.chunk_start[t.n] = chunks;
            if (pass == 0) hipLaunchKernelGGL(dpn_gradnorm_kernel<AdamTable>, dim3(chunks), dim3(256), 0, s, t, partial + base_chunk, step_dev, t0 == 0 ? 1 : 0);
            else hipLaunchKernelGGL(dpn_adam_kernel<AdamTable>, dim3(chunks), dim3(256), 0, s, t, (const double*)sumsq, (const int*)step_dev, lr, beta1,
                                    beta2, eps, weight_decay, max_norm, out_norm_dev, (const float*)nullptr);
            base_chunk += chunks;
        }
        if (pass == 0) hipLaunchKernelGGL(dpn_gradnorm_reduce_kernel, dim3(1), dim3(256), 0, s, (const double*)partial, base_chunk, sumsq);
    }
    return ck(hipGetLastError());
}

int64_t dpn_clip_adam_flat_floats(int n_tensors, const int64_t* numel) {
    if (n_tensors <= 0 || !numel) return -1;
    int64_t chunks = 0;
    for (int i = 0; i < n_tensors; ++i) chunks += (numel[i] + kAdamChunk - 1) / kAdamChunk;
    return chunks * kAdamChunk;
}

static int clip_adam_flat_impl(int n_tensors, float* const* params, const float* const* grads, const int64_t* numel, float* exp_avg_flat,
                               float* exp_avg_sq_flat, double* scratch_dev, int* step_dev, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float max_norm, float* out_norm_dev, const float* hyper_dev, void* stream) {
    if (n_tensors <= 0 || !params || !grads || !numel || !exp_avg_flat || !exp_avg_sq_flat || !scratch_dev || !step_dev) return -1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* sumsq = scratch_dev;
    double* partial = scratch_dev + 1;
    for (int pass = 0; pass < 2; ++pass) {
        int base_chunk = 0;
        for (int t0 = 0; t0 < n_tensors; t0 += kAdamFlatMaxTensors) {
            AdamTableFlat t;
            t.n = (n_tensors - t0 < kAdamFlatMaxTensors) ? n_tensors - t0 : kAdamFlatMaxTensors;
            t.m_flat = exp_avg_flat + (int64_t)base_chunk * kAdamChunk;
            t.v_flat = exp_avg_sq_flat + (int64_t)base_chunk * kAdamChunk;
            int chunks = 0;
            for (int i = 0; i < t.n; ++i) {
                if (numel[t0 + i] <= 0 || numel[t0 + i] > 0x7fffffff) return -1;
                t.p[i] = params[t0 + i]; t.g[i] = grads[t0 + i];
                t.numel[i] = (int)numel[t0 + i];
                t.chunk_start[i] = chunks;
                chunks += (t.numel[i] + kAdamChunk - 1) / kAdamChunk;
            }
            t.chunk_start[t.n] = chunks;
            if (pass == 0) hipLaunchKernelGGL(dpn_gradnorm_kernel<AdamTableFlat>, dim3(chunks), dim3(256), 0, s, t, partial + base_chunk, step_dev, t0 == 0 ? 1 : 0);
            else hipLaunchKernelGGL(dpn_adam_kernel<AdamTableFlat>, dim3(chunks), dim3(256), 0, s, t, (const double*)sumsq, (const int*)step_dev, lr,
                                    beta1, beta2, eps, weight_decay, max_norm, out_norm_dev, hyper_dev);
            base_chunk += chunks;
        }
        if (pass == 0) hipLaunchKernelGGL(dpn_gradnorm_reduce_kernel, dim3(1), dim3(256), 0, s, (const double*)partial, base_chunk, sumsq);
    }
    return ck(hipGetLastError());
}

int dpn_clip_adam_flat(int n_tensors, float* const* params, const float* const* grads, const int64_t* numel, float* exp_avg_flat,
                       float* exp_avg_sq_flat, double* scratch_dev, int* step_dev, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float max_norm, float* out_norm_dev, void* stream) {
    return clip_adam_flat_impl(n_tensors, params, grads, numel, exp_avg_flat, exp_avg_sq_flat, scratch_dev, step_dev, lr, beta1, beta2, eps,
                               weight_decay, max_norm, out_norm_dev, nullptr, stream);
}

int dpn_clip_adam_flat_dev(int n_tensors, float* const* params, const float* const* grads, const int64_t* numel, float* exp_avg_flat,
                           float* exp_avg_sq_flat, double* scratch_dev, int* step_dev, const float* hyper_dev, float* out_norm_dev,
                           void* stream) {
    if (!hyper_dev) return -1;
    return clip_adam_flat_impl(n_tensors, params, grads, numel, exp_avg_flat, exp_avg_sq_flat, scratch_dev, step_dev, 0.f, 0.f, 0.f, 0.f,
                               0.f, 0.f, out_norm_dev, hyper_dev, stream);
}

int dpn_selftest(void* scratch_dev, void* stream) {
    if (!scratch_dev) return -1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* out = reinterpret_cast<float*>(scratch_dev);
    hipLaunchKernelGGL(dpn_selftest_kernel, dim3(1), dim3(64), 0, s, out);
    float host[2048];
    if (hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess) return -2;
    if (hipStreamSynchronize(s) != hipSuccess) return -3;
    // D layout claimed in dpn_layout.h: lane (j = lane&31, h = lane>>5), register r  ->  row drow32(r,h), column j
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
            const int j = lane & 31, h = lane >> 5, i = drow32(r, h);
            if (host[lane * 16 + r] != (float)i) return 100 + r;
            if (host[1024 + lane * 16 + r] != (float)j) return 200 + r;
        }
    return 0;
}

#endif  // DPN_HAS_REST

}  // extern "C"
