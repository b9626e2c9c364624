// Forward + Jacobian kernel of the parity-grade (hi+lo) mode, TILE-SPLIT form.  Included by dpn_kernels.hip (point unit).
//
// Same arithmetic as dpn_fwd_kernel (reference model/variable_net.py:49-87 restated as in DESIGN.md section 3; same packed weight
// stream, same fragment algebra of dpn_layout.h, same accumulation order per output tile: saved state and Jacobian are bit-identical),
// different decomposition:
//
//   dpn_fwd_kernel        one 512-register wave per SIMD owns 32 points and ALL eight output tiles of a layer; activations stay in its
//                         registers, the weight fragments are shared through an LDS-DMA ring (one barrier + 8 DMA issues per 48 MFMAs,
//                         paid with an idle matrix pipe: MFMA busy 45 % in the hi+lo mode).
//   dpn_fwd_tiles_kernel  a workgroup = 4 waves x 256 registers owns 64 points; wave w owns output tiles 2w, 2w+1 of every layer for
//                         both 32-point column tiles (2 x 2 accumulators).  The ACTIVATIONS are what is shared: each layer's epilogue
//                         writes its output to LDS already as next layer's B fragments (the accumulator-is-next-B-operand layout makes
//                         that a linear 16-byte-per-lane store), every wave reads all of them back with conflict-free ds_read_b128.
//                         The WEIGHTS are private to a wave (its two tiles), so they go L2 -> VGPR directly: plain 1-KB-per-instruction
//                         global loads two k-steps ahead, no LDS-DMA, no ring, no counted-wait choreography.
//                         72 KB of LDS and 256 registers => TWO workgroups per CU, i.e. two waves per SIMD that belong to different
//                         workgroups: they never meet at a barrier, so one multiplies while the other packs / stores / waits / builds
//                         features.  Two barriers per LAYER (192 MFMAs per wave) instead of one per 48 MFMAs.
//   cost                  a workgroup streams the net's 1.6 MB of fragments per 64 points instead of per 128 (L2 -> CU traffic x2:
//                         ~30 B/clk/CU at the MFMA rate reached, under the 64 B/clk of the vector memory path; weights are L2 hits).
//
// Per k-step and wave: 4 global loads (A: 2 tiles x hi, lo), 4 ds_read_b128 (B: 2 column tiles x hi, lo), 12 MFMAs.
#pragma once

namespace ts {
constexpr int kVecFloats = kNumVecs * 256 + 4;
template <int NS>
struct Cfg {
    static constexpr int kXBytes = 16 * 2 * NS * 1024;            // [k-step 16][column tile 2][hi | lo][64 lanes][16 B]
    static constexpr int kVecOff = kXBytes;
    static constexpr int kRedOff = kVecOff + kVecFloats * 4;      // 6160 B of vectors: the offset stays 16-byte aligned
    static constexpr int kLdsBytes = kRedOff + 4 * 64 * 4;        // [wave][column tile * 32 + j] partial field sums
};

// LDS-only workgroup barrier: this wave's LDS reads / writes have completed; global loads (weight prefetch) and stores in flight STAY
// in flight (a plain __syncthreads() would drain vmcnt as well)
DEV void barrier_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NS>
DEV void mma3(const u32x4 (&a)[NS], const u32x4 (&b)[NS], f32x16& acc) {        // same product order as mma_block (not SWAP)
    if constexpr (NS == 2) {
        acc = mfma(as_bf(a[0]), as_bf(b[1]), acc);
        acc = mfma(as_bf(a[1]), as_bf(b[0]), acc);
    }
    acc = mfma(as_bf(a[0]), as_bf(b[0]), acc);
}

// acc[t][p] += W[tile t][k] * X[k][column tile p] over NK k-steps.  wg: this wave's first tile chunk in the packed stream (wave-uniform;
// the second tile's chunk follows it), xl: LDS X + lane * 16.  A fragments two k-steps ahead, B fragments one.
template <int NS, int NK, int NT>
DEV void gemm(const char* wg, const char* xl, const int lane, f32x16 (&acc)[2][2]) {
    const u32x4* ag = reinterpret_cast<const u32x4*>(wg) + lane;
    u32x4 A[3][NT][NS];
    u32x4 B[2][2][NS];
    auto loadA = [&](const int ks, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s) A[slot][t][s] = ag[((t * NK + ks) * NS + s) * 64];
    };
    auto loadB = [&](const int ks, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < NS; ++s) B[slot][p][s] = *reinterpret_cast<const u32x4*>(xl + ((ks * 2 + p) * NS + s) * 1024);
    };
    loadA(0, 0);
    loadA(1, 1);
    loadB(0, 0);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + 2 < NK) loadA(ks + 2, (ks + 2) % 3);
        if (ks + 1 < NK) loadB(ks + 1, (ks + 1) & 1);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) mma3<NS>(A[ks % 3][t], B[ks & 1][p], acc[t][p]);
    }
}

template <int NS>
DEV void x_store(char* xl, const int ks, const int p, const Frag<NS>& f) {
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x4*>(xl + ((ks * 2 + p) * NS + s) * 1024) = f.w[s];
}

DEV void acc_init(f32x16& acc, const float* vec, const int which, const int h, const int T, const float scale) {
    const f32x4* v = reinterpret_cast<const f32x4*>(vec + which * 256 + h * 128 + T * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 x = v[q];
        acc[4 * q] = scale * x[0]; acc[4 * q + 1] = scale * x[1]; acc[4 * q + 2] = scale * x[2]; acc[4 * q + 3] = scale * x[3];
    }
}

struct Ident { u32x4 a, b; };
DEV Ident make_ident(const int j, const int h) {          // the identity B fragments of lane_init (MFMA transposes of the saved state)
    Ident I;
    const int mine = (((j >> 3) & 1) == h) ? (j & 7) : -1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const u32 v = ((mine == 2 * p) ? 0x3F80u : 0u) | ((mine == 2 * p + 1) ? 0x3F800000u : 0u);
        I.a[p] = (j < 16) ? v : 0u;
        I.b[p] = (j >= 16) ? v : 0u;
    }
    return I;
}
// store_tile_k of the ring kernel: fragments (k-steps 2ct, 2ct+1) of one column tile -> K-layout rows of the 32-point tile
template <int NS, int NSTORE>
DEV void save_tile_k(const KMat& m, const int net, const int64_t tile32, const int ct, const int lane, const Ident& I, const bool zero,
                     const Frag<NS>& f0, const Frag<NS>& f1) {
#pragma unroll
    for (int s = 0; s < NSTORE; ++s) {
        u32x4 a0 = f0.w[s], a1 = f1.w[s];
        if (zero) { a0 = (u32x4)0u; a1 = (u32x4)0u; }
        f32x16 d = (f32x16)0.f;
        d = mfma(as_bf(a0), as_bf(I.a), d);
        d = mfma(as_bf(a1), as_bf(I.b), d);
        store_d_as_k(m, net, NSTORE, s, tile32, ct, lane, d);
    }
}
}  // namespace ts

template <int NS>
__global__ __launch_bounds__(256, 2) void dpn_fwd_tiles_kernel(FwdArgs a) {
    using C = ts::Cfg<NS>;
    __shared__ __attribute__((aligned(16))) char lds[C::kLdsBytes];
    const int net = blockIdx.y;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
    float* vec = reinterpret_cast<float*>(lds + C::kVecOff);
    float* red = reinterpret_cast<float*>(lds + C::kRedOff);
    char* xl = lds + lane * 16;
    {   // permuted fp32 vectors of this net -> LDS (published by the first barrier)
        const float* gv = reinterpret_cast<const float*>(pk + (long)kPackKB * 1024 * NS);
        for (int i = threadIdx.x; i < ts::kVecFloats; i += 256) vec[i] = gv[i];
    }
    const int64_t tile0 = (int64_t)blockIdx.x * 2;                  // first of this workgroup's two 32-point column tiles
    int64_t pc[2];
    bool valid[2], zero_rows[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int64_t pt = (tile0 + p) * 32 + j;
        valid[p] = pt < a.n;
        pc[p] = valid[p] ? pt : (a.n - 1);
        zero_rows[p] = ((tile0 + p) * 32 + 32 > a.n) && !valid[p];    // saved rows of padding points are zero
    }
    const ts::Ident I = ts::make_ident(j, h);
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    const bool save = a.saved != nullptr;
    const int64_t tiles32 = a.n_pad / 32;
    auto chunk = [&](const int kb) __attribute__((always_inline)) { return pk + (long)kb * 1024 * NS; };

    f32x16 acc[2][2];
    Frag<NS> F[2][2][2];                     // [tile t][column tile p][k-step of the tile's pair]: the epilogue's output fragments
    auto x_store_all = [&]() __attribute__((always_inline)) {      // this wave's tiles 2w, 2w+1 are k-steps 4w .. 4w+3 of the next layer
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) ts::x_store<NS>(xl, 4 * w + 2 * t + kk, p, F[t][p][kk]);
    };
    auto init_all = [&](const int which, const float scale) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ts::acc_init(acc[t][0], vec, which, h, 2 * w + t, scale);
            acc[t][1] = acc[t][0];
        }
    };

    // ---------------- coordinate features pe3 -> X (k-steps 0..11): this thread builds k-steps 3w .. 3w+2 of both column tiles
    {
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int ks = 3 * w + kk, c = ks >> 2;                                   // wave-uniform
            const float* src = (c == 0) ? a.x : (c == 1) ? a.y : a.t;
            const float d1 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : a.geo.pred_t_span;
            const float d2 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : 1.0f;   // x / dx / (lon-1): two fp32 divisions (interface_physics.py:324-326); t: one
            float fr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) fr[q] = a.freqs[8 * (ks & 3) + 4 * h + q];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float xi = src[pc[p]] / d1 / d2;
                Frag<NS> f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float s, co;
                    sincos_t<NS>(xi * fr[q], s, co);
                    frag_set2<NS>(f, q, s, co);
                }
                ts::x_store<NS>(xl, ks, p, f);
            }
        }
    }
    ts::barrier_lds();
    // ---------------- L1: pre1 = w1 . pe + b1 ; h1 = relu -> X ; relu mask bits -> m1w
    u32 m1w[2] = {0u, 0u};
    init_all(kVecB1, 1.0f);
    ts::gemm<NS, 12, 2>(chunk(kS0 + 2 * w * 12), xl, lane, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = acc[t][p][r], p1 = acc[t][p][r + 1];
                const bool on0 = p0 > 0.f, on1 = p1 > 0.f;
                m1w[p] |= (on0 ? (1u << (16 * t + r)) : 0u) | (on1 ? (2u << (16 * t + r)) : 0u);
                frag_set2<NS>(F[t][p][r >> 3], (r & 7) >> 1, on0 ? p0 : 0.f, on1 ? p1 : 0.f);
            }
    if (save) {
#pragma unroll
        for (int p = 0; p < 2; ++p)          // word w of the lane's uint4 = tiles 2w (low half), 2w+1 (high half): the ring kernel's m1w[T >> 1]
            reinterpret_cast<u32*>(sv.m1 + ((int64_t)net * tiles32 + tile0 + p) * 64 + lane)[w] = m1w[p];
    }
    ts::barrier_lds();                       // everybody is done reading pe3
    x_store_all();
    ts::barrier_lds();
    // ---------------- L2: c = w2 . h1 + Wd . pe6 + (b2 + bd + e) -> X ; cdot = wo . c
    init_all(kVecCvec, 1.0f);
    ts::gemm<NS, 16, 2>(chunk(kS1 + 2 * w * 16), xl, lane, acc);
    {   // data features pe6 (SineCosPE(6,16) of coord_data): k-steps 3w .. 3w+2 of both column tiles, built while the accumulators wait
        Frag<NS> f6[3][2];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int ks = 3 * w + kk;
            float fr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) fr[q] = a.freqs[32 + 8 * (ks & 1) + 4 * h + q];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float v = a.coord_data[pc[p] * 6 + (ks >> 1)];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float s, co;
                    sincos_t<NS>(v * fr[q], s, co);
                    frag_set2<NS>(f6[kk][p], q, 1.0f * s, 1.0f * co);
                }
            }
        }
        ts::barrier_lds();                   // everybody is done reading h1
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p) ts::x_store<NS>(xl, 3 * w + kk, p, f6[kk][p]);
        ts::barrier_lds();
    }
    ts::gemm<NS, 12, 2>(chunk(kS1 + 128 + 2 * w * 12), xl, lane, acc);
    float cdot[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4* wv = reinterpret_cast<const f32x4*>(vec + kVecWo * 256 + h * 128 + (2 * w + t) * 16);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 x = wv[q];
                cdot[p] = fmaf(x[0], acc[t][p][4 * q], cdot[p]); cdot[p] = fmaf(x[1], acc[t][p][4 * q + 1], cdot[p]);
                cdot[p] = fmaf(x[2], acc[t][p][4 * q + 2], cdot[p]); cdot[p] = fmaf(x[3], acc[t][p][4 * q + 3], cdot[p]);
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) frag_set2<NS>(F[t][p][r >> 3], (r & 7) >> 1, acc[t][p][r], acc[t][p][r + 1]);
        }
    }
    ts::barrier_lds();
    x_store_all();
    ts::barrier_lds();
    // ---------------- fc1: pre2 = W1 . c + bf1 ; out = u . relu(pre2) + 2 wo . c + const ; t2 = m2 (.) u -> X ; M2 -> saved
    init_all(kVecBf1, 1.0f);
    ts::gemm<NS, 16, 2>(chunk(kS2 + 2 * w * 16), xl, lane, acc);
    float adot[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4* uvp = reinterpret_cast<const f32x4*>(vec + kVecU * 256 + h * 128 + (2 * w + t) * 16);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            Frag<1> mk0, mk1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 uq = uvp[q];
                const float uu[4] = {uq[0], uq[1], uq[2], uq[3]};
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const int r = 4 * q + i;
                    const float p0 = acc[t][p][r], p1 = acc[t][p][r + 1];
                    const bool on0 = p0 > 0.f, on1 = p1 > 0.f;
                    const float t0 = on0 ? uu[i] : 0.f, t1 = on1 ? uu[i + 1] : 0.f;          // t2 = m2 (.) u
                    adot[p] = fmaf(p0, t0, adot[p]);                                       // relu(p) * u == p * (m2 * u)
                    adot[p] = fmaf(p1, t1, adot[p]);
                    frag_set2<NS>(F[t][p][r >> 3], (r & 7) >> 1, t0, t1);
                    const u32 mw = (on0 ? 0x3F80u : 0u) | (on1 ? 0x3F800000u : 0u);
                    if (r < 8) mk0.w[0][(r & 7) >> 1] = mw; else mk1.w[0][(r & 7) >> 1] = mw;
                }
            }
            if (save) ts::save_tile_k<1, 1>(sv.M2, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], mk0, mk1);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {            // this wave's share of the field: its 64 channels, both halves of the wave
        float o = adot[p] + 2.0f * cdot[p];
        o += __shfl_xor(o, 32);
        if (h == 0) red[w * 64 + p * 32 + j] = o;
    }
    ts::barrier_lds();
    x_store_all();
    ts::barrier_lds();
    if (w == 0) {                            // lane (j, h) finishes point j of column tile h: the four waves' shares in a fixed order
        const int64_t pt = (tile0 + h) * 32 + j;
        if (pt < a.n) {
            const float const0 = vec[kNumVecs * 256];
            const float o = (red[0 * 64 + h * 32 + j] + red[1 * 64 + h * 32 + j]) + (red[2 * 64 + h * 32 + j] + red[3 * 64 + h * 32 + j]);
            a.out_n[pt * 6 + net] = o + const0 + a.coord_data[pt * 6 + net];           // + ref_data (variable_net.py:86)
        }
    }
    if (!save && !a.jac_n) return;
    // ---------------- reverse sweep: v = W1^T t2 + 2 wo -> X (+ saved V)
    init_all(kVecWo, 2.0f);
    ts::gemm<NS, 16, 2>(chunk(kS3 + 2 * w * 16), xl, lane, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) frag_set2<NS>(F[t][p][r >> 3], (r & 7) >> 1, acc[t][p][r], acc[t][p][r + 1]);
            if (save) ts::save_tile_k<NS, NS>(sv.V, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], F[t][p][0], F[t][p][1]);
        }
    ts::barrier_lds();
    x_store_all();
    ts::barrier_lds();
    // ---------------- y = w2^T v ; t1 = m1 (.) y -> X (+ saved T1)
#pragma unroll
    for (int t = 0; t < 2; ++t) { acc[t][0] = (f32x16)0.f; acc[t][1] = (f32x16)0.f; }
    ts::gemm<NS, 16, 2>(chunk(kS4 + 2 * w * 16), xl, lane, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const u32 bits = m1w[p] >> (16 * t + r);
                frag_set2<NS>(F[t][p][r >> 3], (r & 7) >> 1, (bits & 1u) ? acc[t][p][r] : 0.f, (bits & 2u) ? acc[t][p][r + 1] : 0.f);
            }
            if (save) ts::save_tile_k<NS, NS>(sv.T1, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], F[t][p][0], F[t][p][1]);
        }
    if (!a.jac_n) return;
    ts::barrier_lds();
    x_store_all();
    ts::barrier_lds();
    // ---------------- gpe = w1^T t1 (6 tiles: waves 0..2; both tiles of wave w belong to coordinate c = w), contracted with d(pe)/d(xi)
    if (w >= 3) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) { acc[t][0] = (f32x16)0.f; acc[t][1] = (f32x16)0.f; }
    ts::gemm<NS, 16, 2>(chunk(kS5 + 2 * w * 16), xl, lane, acc);
    {
        const int c = w;
        const float* src = (c == 0) ? a.x : (c == 1) ? a.y : a.t;
        const float d1 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : a.geo.pred_t_span;
        const float d2 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : 1.0f;
        float jc[2] = {0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float xi = src[pc[p]] / d1 / d2;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {            // register pair (sin, cos) of one angle: k-step 2T + (r >> 3) of the coordinate PE
                    const int r = 2 * rp;
                    const float fr = a.freqs[8 * ((2 * t + (r >> 3)) & 3) + 4 * h + ((r & 7) >> 1)];
                    float s, co;
                    sincos_t<NS>(xi * fr, s, co);
                    jc[p] = fmaf(acc[t][p][r], fr * co, jc[p]);
                    jc[p] = fmaf(acc[t][p][r + 1], -fr * s, jc[p]);
                }
            jc[p] += __shfl_xor(jc[p], 32);
        }
        // lane (j, h) stores point j of column tile h; chain rule through x / dx / (lon - 1), in the reference's backward order
        const float mine = h ? jc[1] : jc[0];
        const int64_t pt = (tile0 + h) * 32 + j;
        if (pt < a.n) {
            const float g1 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : a.geo.pred_t_span;
            const float g2 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : 1.0f;
            a.jac_n[(pt * 6 + net) * 3 + c] = mine / g1 / g2;
        }
    }
}
