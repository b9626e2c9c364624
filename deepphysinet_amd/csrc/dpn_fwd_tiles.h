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

struct NoSide { DEV void operator()(int) const {} };

// Ablation builds (tools/variant_build.py; wrong results on purpose, timing only): TS_ABL_NOMFMA keeps the operands alive and drops the
// products, TS_ABL_NOSINCOS replaces the feature evaluation by one multiply, TS_ABL_NOALOAD multiplies with whatever is in the registers.
#ifdef TS_ABL_NOSINCOS
template <int NS> DEV void ts_sincos(float th, float& s, float& c) { s = th; c = th * 0.5f; }
#else
template <int NS> DEV void ts_sincos(float th, float& s, float& c) { sincos_t<NS>(th, s, c); }
#endif

// SWAP = false: a = weight fragment (A operand), b = activation fragment (B operand): Out[channel][point], the chained layout.
// SWAP = true : the activation fragment is the A operand, the weight fragment the B operand: Out[point][channel] -- channel per lane, the
//               K-operand layout of the weight-gradient GEMMs (dpn_bwd_tiles_kernel's Z).  Product order as mma_block in both cases.
template <int NS, bool SWAP = false>
DEV void mma3(const u32x4 (&a)[NS], const u32x4 (&b)[NS], f32x16& acc) {
#ifdef TS_ABL_NOMFMA
    if constexpr (NS == 2) asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
    else asm volatile("" ::"v"(a[0]), "v"(b[0]));
    return;
#endif
    if constexpr (SWAP) {
        if constexpr (NS == 2) {
            acc = mfma(as_bf(b[1]), as_bf(a[0]), acc);
            acc = mfma(as_bf(b[0]), as_bf(a[1]), acc);
        }
        acc = mfma(as_bf(b[0]), as_bf(a[0]), acc);
    } else {
        if constexpr (NS == 2) {
            acc = mfma(as_bf(a[0]), as_bf(b[1]), acc);
            acc = mfma(as_bf(a[1]), as_bf(b[0]), acc);
        }
        acc = mfma(as_bf(a[0]), as_bf(b[0]), acc);
    }
}

// acc[t][p] += W[tile t][k] * X[k][column tile p] over NK k-steps.  wg: this wave's first tile chunk in the packed stream (wave-uniform;
// the second tile's chunk follows it), xl: LDS X + lane * 16.  A fragments kPF - 1 k-steps ahead, B fragments one.
// The first kPF - 1 k-steps' A fragments are loaded by gemm_head, which the kernel calls BEFORE the previous layer's epilogue: the
// weight stream of a layer then starts under that epilogue (packing, saved-state stores, the two barriers) instead of cold behind it --
// and in front of its stores: vmcnt retires in order, a load issued behind the saved-state stores would wait for them as well.
#ifndef TS_PF
#define TS_PF 3      // two k-steps ahead: with the deferred saved-state hand-over (TS_DEFER_SAVES) a fourth slot spills 44 registers (measured slower)
#endif
// Issue priority between the two waves of a SIMD (they belong to different workgroups): 0 = none, 1 = s_setprio 1 around every k-step's
// MFMAs, 2 = s_setprio 1 everywhere EXCEPT the multiply loops (the feature / epilogue / store phases are a workgroup's serial chain; a
// multiplying wave needs one issue slot in eight)
#ifndef TS_PRIO
#define TS_PRIO 2
#endif
// Saved-state hand-over (V, T1, M2 as K-layout rows: two transposing MFMAs, eight packs and two streaming stores per plane and tile) issued
// k-step by k-step inside the NEXT layer's multiply loop instead of in the serial epilogue between two barriers (nobody in the kernel waits
// for it).  0 = in the epilogue.
#ifndef TS_DEFER_SAVES
#define TS_DEFER_SAVES 1
#endif
#ifndef TS_SCHED_GROUPS
#define TS_SCHED_GROUPS 0
#endif
constexpr int kPF = TS_PF;
template <int NS, int NT> struct Head { u32x4 a[kPF - 1][NT][NS]; };

// Weight fragments come through a buffer descriptor (SGPR base, the lane's 16-byte slot as the only VGPR offset, fragment index as scalar /
// immediate offset): no 64-bit address arithmetic between the MFMAs.  tools/microbench/l2_stream2.hip, this loop shape at 3 MFMAs per
// loaded KB: global_load with VGPR addresses 48 % of the bf16 peak, buffer_load 65 %, s_setprio 1 around the MFMAs 62 %.
template <int NS, int NK, int NT>
struct WSrc {
    __amdgpu_buffer_rsrc_t rs;
    int voff;
    int so[NT];                                  // running scalar offsets of the next k-step, one per tile (SALU adds; kept opaque so that
                                                 // the unrolled loop does not turn them into a hundred constants in as many SGPRs)
    DEV void init(const char* wg, const int lane, const int ks0) {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wg), 0, NT * NK * NS * 1024, 0x00020000);
        voff = lane * 16;
#pragma unroll
        for (int t = 0; t < NT; ++t) { so[t] = (t * NK + ks0) * NS * 1024; asm volatile("" : "+s"(so[t])); }
    }
    DEV void next(u32x4 (&dst)[NT][NS]) {        // fragments of the next k-step
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#ifdef TS_ABL_NOALOAD
                asm volatile("" : "=v"(dst[t][s]));
#else
                dst[t][s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + s * 1024, so[t], 0));
#endif
            }
            so[t] += NS * 1024;
            asm volatile("" : "+s"(so[t]));
        }
    }
};
template <int NS, int NK, int NT>
DEV void gemm_head(const char* wg, const int lane, Head<NS, NT>& H) {
    WSrc<NS, NK, NT> src;
    src.init(wg, lane, 0);
#pragma unroll
    for (int k = 0; k < kPF - 1; ++k) src.next(H.a[k]);
}
// side(ks): work of the PREVIOUS layer that nobody waits for (its saved-state transposes and stores), issued k-step by k-step in the shadow
// of this layer's MFMAs instead of in the serial epilogue between two barriers
template <int NS, int NK, int NT, bool SWAP = false, class Side = NoSide, bool PIN = true, bool SG = PIN>
DEV void gemm(const char* wg, const char* xl, const int lane, const Head<NS, NT>& H, f32x16 (&acc)[2][2], const Side& side = Side()) {
    static_assert(NK >= kPF, "k-steps per chunk");
    WSrc<NS, NK, NT> src;
    src.init(wg, lane, kPF - 1);
    u32x4 A[kPF][NT][NS];
    u32x4 B[2][2][NS];
    auto loadB = [&](const int ks, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < NS; ++s) B[slot][p][s] = *reinterpret_cast<const u32x4*>(xl + ((ks * 2 + p) * NS + s) * 1024);
    };
#pragma unroll
    for (int k = 0; k < kPF - 1; ++k)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s) A[k][t][s] = H.a[k][t][s];
    loadB(0, 0);
#if TS_PRIO == 2
    __builtin_amdgcn_s_setprio(0);            // multiply phases at low priority, everything else (the serial chain of a workgroup) at high
#endif
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + kPF - 1 < NK) src.next(A[(ks + kPF - 1) % kPF]);
        if (ks + 1 < NK) loadB(ks + 1, (ks + 1) & 1);
#if TS_SCHED_GROUPS
        // (SG: the forward kernels; the backward stage-1 kernel lives on 168 registers and spills with the longer live ranges)
        // the next k-step's B fragments are READ (4 ds_read_b128) in front of this k-step's MFMAs, not behind them where hipcc's scheduler sinks them to
        // shorten their live ranges (the read latency is then exposed in front of every k-step when no second multiplying wave covers it)
        if (SG && ks + 1 < NK) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NS, 0);        // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * (NS == 2 ? 3 : 1), 0);   // MFMA
        }
#endif
#if TS_PRIO == 1
        __builtin_amdgcn_s_setprio(1);        // the multiplying wave wins issue arbitration against its SIMD partner's loads / packing
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) mma3<NS, SWAP>(A[ks % kPF][t], B[ks & 1][p], acc[t][p]);
#if TS_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        side(ks);
        if constexpr (PIN) {
        // k-steps stay k-steps: the MFMAs are pure values without a position of their own, and instruction selection is free to emit all of a
        // layer's 144-192 of them AFTER the fragment loads of all its k-steps (seen after an unrelated change four layers later: 204 spilled
        // registers in the first layer's loop, every load followed by s_waitcnt vmcnt(0) + a scratch store, 628 us instead of 478).  The empty
        // volatile statement gives the accumulators a place in the order of the loads' running offsets (WSrc::next), which are volatile too.
        // (PIN = false in the backward kernel: its loops carry side work in every layer, which holds the order by itself, and the pin costs it
        // 19 us -- 241 -> 260 us measured on one box.)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        }
    }
#if TS_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif
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
// one plane (hi or lo) of one column tile: two transposing MFMAs, eight packs, two 16-byte streaming stores
DEV void save_plane_k(const KMat& m, const int net, const int nstore, const int s, const int64_t tile32, const int ct, const int lane, const Ident& I,
                      const bool zero, u32x4 a0, u32x4 a1) {
    if (zero) { a0 = (u32x4)0u; a1 = (u32x4)0u; }
    f32x16 d = (f32x16)0.f;
    d = mfma(as_bf(a0), as_bf(I.a), d);
    d = mfma(as_bf(a1), as_bf(I.b), d);
    store_d_as_k(m, net, nstore, s, tile32, ct, lane, d);
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

// the coordinate features of k-step ks (0..11) for the lane's point: one B fragment (hi [+ lo])
template <int NS, class Args>
DEV void pe3_frag(Frag<NS>& f, const Args& a, const int ks, const int h, const int64_t pc) {
    const int c = ks >> 2;
    const float* src = (c == 0) ? a.x : (c == 1) ? a.y : a.t;
    const float d1 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : a.geo.pred_t_span;
    const float d2 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : 1.0f;   // x / dx / (lon-1): two fp32 divisions (interface_physics.py:324-326); t: one
    const float xi = src[pc] / d1 / d2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s, co;
        ts_sincos<NS>(xi * a.freqs[8 * (ks & 3) + 4 * h + q], s, co);
        frag_set2<NS>(f, q, s, co);
    }
}
// the data features (SineCosPE(6,16) of coord_data, variable_net.py:73) of k-step ks (0..11)
template <int NS, class Args>
DEV void pe6_frag(Frag<NS>& f, const Args& a, const int ks, const int h, const int64_t pc, const float g = 1.0f) {
    const float v = a.coord_data[pc * 6 + (ks >> 1)];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s, co;
        ts_sincos<NS>(v * a.freqs[32 + 8 * (ks & 1) + 4 * h + q], s, co);
        frag_set2<NS>(f, q, g * s, g * co);
    }
}
// the same, and dot += sum over the fragment's eight features of feature * bv[slot] (bv: a 192-vector in PE6 slot order 16 ks + 8 h + e, in LDS)
template <int NS, class Args>
DEV void pe6_frag_dot(Frag<NS>& f, const Args& a, const int ks, const int h, const int64_t pc, const float* bv, float& dot) {
    const float v = a.coord_data[pc * 6 + (ks >> 1)];
    const f32x4* b4 = reinterpret_cast<const f32x4*>(bv + 16 * ks + 8 * h);
    const f32x4 b0 = b4[0], b1 = b4[1];
    const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s, co;
        ts_sincos<NS>(v * a.freqs[32 + 8 * (ks & 1) + 4 * h + q], s, co);
        frag_set2<NS>(f, q, s, co);
        dot = fmaf(s, bb[2 * q], dot);
        dot = fmaf(co, bb[2 * q + 1], dot);
    }
}
// backward: Z0 = g * pe + gJ_c * d pe / d xi_c for k-step ks (coordinate c = ks >> 2), build_pe3<BWD> of the ring kernels
template <int NS, class Args>
DEV void z0_frag(Frag<NS>& f, const Args& a, const int ks, const int h, const int64_t pc, const float g, const float gjc) {
    const int c = ks >> 2;
    const float* src = (c == 0) ? a.x : (c == 1) ? a.y : a.t;
    const float d1 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : a.geo.pred_t_span;
    const float d2 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : 1.0f;
    const float xi = src[pc] / d1 / d2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float fr = a.freqs[8 * (ks & 3) + 4 * h + q];
        float s, co;
        ts_sincos<NS>(xi * fr, s, co);
        const float gf = gjc * fr;
        frag_set2<NS>(f, q, fmaf(g, s, gf * co), fmaf(g, co, -gf * s));
    }
}
// d pe3 / d xi_c for the 16 accumulator registers of gpe tile 2c + t (register pair rp = one angle: k-step 2T + (rp >> 2) of the coordinate PE)
template <int NS, class Args>
DEV void dpe_tile(float (&d)[16], const Args& a, const int c, const int t, const int h, const int64_t pc) {
    const float* src = (c == 0) ? a.x : (c == 1) ? a.y : a.t;
    const float d1 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : a.geo.pred_t_span;
    const float d2 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : 1.0f;
    const float xi = src[pc] / d1 / d2;
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
        const int r = 2 * rp;
        const float fr = a.freqs[8 * ((2 * t + (r >> 3)) & 3) + 4 * h + ((r & 7) >> 1)];
        float s, co;
        ts_sincos<NS>(xi * fr, s, co);
        d[r] = fr * co;
        d[r + 1] = -fr * s;
    }
}
// ---- Round 6: the same features with their memory operands FRONT-LOADED (used by dpn_fwd_pp_kernel).  pe3_frag / pe6_frag_dot / dpe_tile above fetch the lane's
// coordinate (through a pointer selected by the wave's k-step) and one frequency per angle INSIDE every fragment; the *_x forms take the coordinate and the
// four frequencies of a fragment as values, so that the caller issues ALL loads of a phase first, then the arithmetic -- identical operations in identical order
// (bit-identical results: tools/fwd_dump.py).  Measured: in the ping-pong kernel, where a service interval is ONE wave per SIMD with nothing to hide a latency
// behind, a fragment of four angles went from 1 350 to ~900 cycles (the arithmetic alone is 4 x 146: tools/microbench/valu_ilp.hip); in dpn_fwd_tiles_kernel and
// dpn_bwd_tiles_kernel the same change measured neutral to -1 % (the partner workgroup's waves cover the latencies there) and they keep the forms above.
template <class Args>
DEV float load_xi(const Args& a, const int c, const int64_t pc) {          // x / dx / (lon-1): two fp32 divisions (interface_physics.py:324-326); t: one
    // all three coordinates are fetched and the VALUE is selected: a pointer selected by the (wave-uniform, run-time) c makes hipcc index the kernel-argument
    // block, which then lives in scratch memory (136 bytes of private segment and a scratch load per use -- seen, round 6)
    const float xr = a.x[pc], yr = a.y[pc], tr = a.t[pc];
    const float raw = (c == 0) ? xr : (c == 1) ? yr : tr;
    const float d1 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : a.geo.pred_t_span;
    const float d2 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : 1.0f;
    return raw / d1 / d2;
}
DEV f32x4 load_fr4(const float* freqs, const int idx) { return *reinterpret_cast<const f32x4*>(freqs + idx); }     // idx % 4 == 0: 16-byte aligned
template <int NS>
DEV void pe3_frag_x(Frag<NS>& f, const float xi, const f32x4 fr) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s, co;
        ts_sincos<NS>(xi * fr[q], s, co);
        frag_set2<NS>(f, q, s, co);
    }
}
template <int NS>
DEV void pe6_frag_x(Frag<NS>& f, const float v, const f32x4 fr, const float g = 1.0f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s, co;
        ts_sincos<NS>(v * fr[q], s, co);
        frag_set2<NS>(f, q, g * s, g * co);
    }
}
template <int NS>
DEV void pe6_frag_dot_x(Frag<NS>& f, const float v, const f32x4 fr, const float* bv8, float& dot) {      // bv8: the fragment's eight entries of the 192-vector (LDS)
    const f32x4* b4 = reinterpret_cast<const f32x4*>(bv8);
    const f32x4 b0 = b4[0], b1 = b4[1];
    const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s, co;
        ts_sincos<NS>(v * fr[q], s, co);
        frag_set2<NS>(f, q, s, co);
        dot = fmaf(s, bb[2 * q], dot);
        dot = fmaf(co, bb[2 * q + 1], dot);
    }
}
template <int NS>
DEV void z0_frag_x(Frag<NS>& f, const float xi, const f32x4 fr4, const float g, const float gjc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float fr = fr4[q];
        float s, co;
        ts_sincos<NS>(xi * fr, s, co);
        const float gf = gjc * fr;
        frag_set2<NS>(f, q, fmaf(g, s, gf * co), fmaf(g, co, -gf * s));
    }
}
// Jacobian contraction of one gpe tile pair (t = 0, 1) of column tile p with d pe3 / d xi_c: jc += sum_r acc[t][r] * d[r], r ascending, t outer -- the order
// of dpe_tile + the caller's loop.  fr[kq] = the four frequencies 8 kq + 4 h .. + 3 of the coordinate's k-step kq
template <int NS>
DEV void jac_contract_x(float& jc, const f32x16 (&acc_t)[2], const float xi, const f32x4 (&fr)[4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float d[16];
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const float f = fr[(2 * t + (rp >> 2)) & 3][rp & 3];
            float s, co;
            ts_sincos<NS>(xi * f, s, co);
            d[2 * rp] = f * co;
            d[2 * rp + 1] = -f * s;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) jc = fmaf(acc_t[t][r], d[r], jc);
    }
}
}  // namespace ts

// Experiment build (-DDPN_TIMELINE -DTS_TIMELINE, tools/tiles_timeline.py): lane 0 of every wave writes the shader clock at the phase
// boundaries below to a.timeline[workgroup][wave][stamp]
#if defined(TS_TIMELINE) && defined(DPN_TIMELINE)
#define TS_STAMP(I) do { if (a.timeline && lane == 0) a.timeline[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 48 + (I)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define TS_STAMP(I) do { } while (0)
#endif

// Round 5: FIVE GEMMs per point and net instead of seven.  W1 (cat_fc1.fc.0.weight) only ever multiplies c = w2 h1 + Wd pe6 + cvec, and its
// transpose only ever meets w2^T on the way back, so the two static-times-hyper products are formed ONCE per net and step,
//     A = W1 w2 [256, 256],   B = W1 Wd [256, 192]        (exact fp32, dpn_pack_weights: csrc FusedForm)
// and   pre2 = A h1 + B pe6 + (W1 cvec + bf1),    wo . c = (w2^T wo) . h1 + (Wd^T wo) . pe6 + wo . cvec,    y = A^T (m2 (.) u) + 2 w2^T wo:
// neither c nor v = d out / d c is formed per point.  409 600 -> 278 528 executed MACs per point and net, and the weight stream a workgroup
// pulls out of L2 per 64 points shrinks from 800 to 544 KB (x NS).  What the backward pass needs (m1, M2, T1 = m1 (.) y) is unchanged; its
// formulas are in the original parameters (dpn_finish_*).  Identity and operand rounding: tools/precision_fused_algebra.py.
template <int NS>
__global__ __launch_bounds__(256, 2) void dpn_fwd_tiles_kernel(FwdArgs a) {
    using C = ts::Cfg<NS>;
    __shared__ __attribute__((aligned(16))) char lds[C::kLdsBytes];
    const int net = blockIdx.y;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
#if TS_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif
    TS_STAMP(0);
    float* vec = reinterpret_cast<float*>(lds + C::kVecOff);
    float* red = reinterpret_cast<float*>(lds + C::kRedOff);
    char* xl = lds + lane * 16;
    {   // permuted fp32 vectors of this net -> LDS (published by the first barrier): 385 x 16 bytes, both loads of a thread in flight
        const u32x4* gv = reinterpret_cast<const u32x4*>(pk + (long)kPackKB * 1024 * NS);
        static_assert(ts::kVecFloats % 4 == 0 && ts::kVecFloats / 4 <= 512, "vector block");
        const int i0 = threadIdx.x, i1 = threadIdx.x + 256;
        const u32x4 v0 = gv[i0];
        const u32x4 v1 = gv[i1 < ts::kVecFloats / 4 ? i1 : i0];
        reinterpret_cast<u32x4*>(vec)[i0] = v0;
        if (i1 < ts::kVecFloats / 4) reinterpret_cast<u32x4*>(vec)[i1] = v1;
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

    ts::Head<NS, 2> H;
    ts::gemm_head<NS, 12, 2>(chunk(kF0 + 2 * w * 12), lane, H);
    // ---------------- coordinate features pe3 -> X (k-steps 0..11): this thread builds k-steps 3w .. 3w+2 of both column tiles
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            Frag<NS> f;
            ts::pe3_frag<NS>(f, a, 3 * w + kk, h, pc[p]);
            ts::x_store<NS>(xl, 3 * w + kk, p, f);
        }
    TS_STAMP(1);
    ts::barrier_lds();
    // ---------------- L1: pre1 = w1 . pe + b1 ; h1 = relu -> X ; relu mask bits -> m1w ; hdot = (w2^T wo) . h1 (this wave's 64 channels)
    u32 m1w[2] = {0u, 0u};
    init_all(kVecB1, 1.0f);
    TS_STAMP(2);
    ts::gemm<NS, 12, 2>(chunk(kF0 + 2 * w * 12), xl, lane, H, acc);
    TS_STAMP(3);
    float hdot[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {            // hdot first, on max(pre1, 0) by v_med3 (no compare result shared with the mask loop below)
        const f32x4* av = reinterpret_cast<const f32x4*>(vec + kVecA2 * 256 + h * 128 + (2 * w + t) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 aq = av[q];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                hdot[p] = fmaf(aq[0], __builtin_amdgcn_fmed3f(acc[t][p][4 * q], 0.f, __builtin_inff()), hdot[p]);
                hdot[p] = fmaf(aq[1], __builtin_amdgcn_fmed3f(acc[t][p][4 * q + 1], 0.f, __builtin_inff()), hdot[p]);
                hdot[p] = fmaf(aq[2], __builtin_amdgcn_fmed3f(acc[t][p][4 * q + 2], 0.f, __builtin_inff()), hdot[p]);
                hdot[p] = fmaf(aq[3], __builtin_amdgcn_fmed3f(acc[t][p][4 * q + 3], 0.f, __builtin_inff()), hdot[p]);
            }
        }
    }
    asm volatile("" : "+v"(hdot[0]), "+v"(hdot[1]));      // the dot products are finished BEFORE the next layer's first weight fragments are requested (register pressure)
    ts::gemm_head<NS, 16, 2>(chunk(kFA + 2 * w * 16), lane, H);
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
    // make the mask words opaque HERE: left alone, the compiler proves (m1w >> k) & 1 == the k-th compare and keeps all 64 compare results
    // alive (as lane masks in SGPRs, spilled through v_writelane, and in scratch) until the y layer's epilogue instead of the two words
    asm volatile("" : "+v"(m1w[0]), "+v"(m1w[1]));
    if (save) {
#pragma unroll
        for (int p = 0; p < 2; ++p)          // word w of the lane's uint4 = tiles 2w (low half), 2w+1 (high half): the ring kernel's m1w[T >> 1]
            reinterpret_cast<u32*>(sv.m1 + ((int64_t)net * tiles32 + tile0 + p) * 64 + lane)[w] = m1w[p];
    }
    TS_STAMP(4);
    ts::barrier_lds();                       // everybody is done reading pe3
    x_store_all();
    TS_STAMP(5);
    ts::barrier_lds();
    // ---------------- pre2 = A h1 + B pe6 + (W1 cvec + bf1)
    init_all(kVecC2, 1.0f);
    TS_STAMP(6);
    ts::gemm<NS, 16, 2>(chunk(kFA + 2 * w * 16), xl, lane, H, acc);
    TS_STAMP(7);
    ts::gemm_head<NS, 12, 2>(chunk(kFB + 2 * w * 12), lane, H);
    float ddot[2] = {0.f, 0.f};              // (Wd^T wo) . pe6 over this wave's k-steps
    {   // data features pe6 (SineCosPE(6,16) of coord_data): k-steps 3w .. 3w+2 of both column tiles, built while the accumulators wait
        Frag<NS> f6[3][2];
        const float* bv = vec + kVecBv * 256;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p) ts::pe6_frag_dot<NS>(f6[kk][p], a, 3 * w + kk, h, pc[p], bv, ddot[p]);
        TS_STAMP(8);
        ts::barrier_lds();                   // everybody is done reading h1
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p) ts::x_store<NS>(xl, 3 * w + kk, p, f6[kk][p]);
        TS_STAMP(9);
        ts::barrier_lds();
    }
    TS_STAMP(10);
    ts::gemm<NS, 12, 2>(chunk(kFB + 2 * w * 12), xl, lane, H, acc);
    TS_STAMP(11);
    if (save || a.jac_n) ts::gemm_head<NS, 16, 2>(chunk(kFAT + 2 * w * 16), lane, H);
    // ---------------- out = u . relu(pre2) + 2 wo . c + const ; t2 = m2 (.) u -> X ; M2 -> saved
    float adot[2] = {0.f, 0.f};
    Frag<1> MK[2][2][2];                     // relu-2 mask as bf16 0 / 1 fragments (one plane)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4* uvp = reinterpret_cast<const f32x4*>(vec + kVecU * 256 + h * 128 + (2 * w + t) * 16);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            Frag<1>& mk0 = MK[t][p][0];
            Frag<1>& mk1 = MK[t][p][1];
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
#if !TS_DEFER_SAVES
            if (save) ts::save_tile_k<1, 1>(sv.M2, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], mk0, mk1);
#endif
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {            // this wave's share of the field: its 64 channels and its 3 k-steps of pe6, both halves of the wave
        float o = adot[p] + 2.0f * (hdot[p] + ddot[p]);
        o += __shfl_xor(o, 32);
        if (h == 0) red[w * 64 + p * 32 + j] = o;
    }
    TS_STAMP(12);
    ts::barrier_lds();
    x_store_all();
    TS_STAMP(13);
    ts::barrier_lds();
    if (w == 0) {                            // lane (j, h) finishes point j of column tile h: the four waves' shares in a fixed order
        const int64_t pt = (tile0 + h) * 32 + j;
        if (pt < a.n) {
            const float const0 = vec[kNumVecs * 256];          // wo . bf2 + bo + 2 wo . cvec
            if (vec[kNumVecs * 256 + 1] != 1.0f) __builtin_trap();     // the packed stream is not in the fused five-GEMM form (its tag sits behind const0): wrong fields otherwise
            const float o = (red[0 * 64 + h * 32 + j] + red[1 * 64 + h * 32 + j]) + (red[2 * 64 + h * 32 + j] + red[3 * 64 + h * 32 + j]);
            a.out_n[pt * 6 + net] = o + const0 + (a.ref ? a.ref : a.coord_data)[pt * 6 + net];           // + ref_data (variable_net.py:86)
        }
    }
    if (!save && !a.jac_n) return;
    // ---------------- reverse sweep: y = A^T t2 + 2 w2^T wo ; t1 = m1 (.) y -> X (+ saved T1)
    init_all(kVecA2, 2.0f);
    TS_STAMP(14);
#if TS_DEFER_SAVES
    {
        auto side = [&](const int ks) __attribute__((always_inline)) {            // M2: four (tile, column tile) units over the 16 k-steps
            if (!save) return;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ks == 4 * u + 1) ts::save_plane_k(sv.M2, net, 1, 0, tile0 + (u & 1), 2 * w + (u >> 1), lane, I, zero_rows[u & 1], MK[u >> 1][u & 1][0].w[0], MK[u >> 1][u & 1][1].w[0]);
        };
        ts::gemm<NS, 16, 2, false>(chunk(kFAT + 2 * w * 16), xl, lane, H, acc, side);
    }
#else
    ts::gemm<NS, 16, 2>(chunk(kFAT + 2 * w * 16), xl, lane, H, acc);
#endif
    TS_STAMP(15);
    if (a.jac_n && w < 3) ts::gemm_head<NS, 16, 2>(chunk(kF5 + 2 * w * 16), lane, H);
    // F (the t2 fragments) is rewritten by this epilogue: every wave has finished reading X(t2) only after the barrier below
    auto side_planes = [&](const KMat& m, const int ks) __attribute__((always_inline)) {       // 4 x NS (tile, column tile, plane) units over 16 k-steps
        if (!save) return;
#pragma unroll
        for (int u = 0; u < 4 * NS; ++u) {
            const int tp = u / NS, s_ = u % NS;
            if (ks == (16 / (4 * NS)) * u + 1)
                ts::save_plane_k(m, net, NS, s_, tile0 + (tp & 1), 2 * w + (tp >> 1), lane, I, zero_rows[tp & 1], F[tp >> 1][tp & 1][0].w[s_], F[tp >> 1][tp & 1][1].w[s_]);
        }
    };
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const u32 bits = m1w[p] >> (16 * t + r);
                frag_set2<NS>(F[t][p][r >> 3], (r & 7) >> 1, (bits & 1u) ? acc[t][p][r] : 0.f, (bits & 2u) ? acc[t][p][r + 1] : 0.f);
            }
#if TS_DEFER_SAVES
            if (save && (!a.jac_n || w >= 3)) ts::save_tile_k<NS, NS>(sv.T1, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], F[t][p][0], F[t][p][1]);
#else
            if (save) ts::save_tile_k<NS, NS>(sv.T1, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], F[t][p][0], F[t][p][1]);
#endif
        }
    if (!a.jac_n) return;
    TS_STAMP(16);
    ts::barrier_lds();
    x_store_all();
    TS_STAMP(17);
    ts::barrier_lds();
    // ---------------- gpe = w1^T t1 (6 tiles: waves 0..2; both tiles of wave w belong to coordinate c = w), contracted with d(pe)/d(xi)
    if (w >= 3) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) { acc[t][0] = (f32x16)0.f; acc[t][1] = (f32x16)0.f; }
    TS_STAMP(18);
#if TS_DEFER_SAVES
    {
        auto side = [&](const int ks) __attribute__((always_inline)) { side_planes(sv.T1, ks); };
        ts::gemm<NS, 16, 2, false>(chunk(kF5 + 2 * w * 16), xl, lane, H, acc, side);
    }
#else
    ts::gemm<NS, 16, 2>(chunk(kF5 + 2 * w * 16), xl, lane, H, acc);
#endif
    TS_STAMP(19);
    {
        const int c = w;
        float jc[2] = {0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float d[16];
                ts::dpe_tile<NS>(d, a, c, t, h, pc[p]);
#pragma unroll
                for (int r = 0; r < 16; ++r) jc[p] = fmaf(acc[t][p][r], d[r], jc[p]);
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
    TS_STAMP(20);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Backward, stage 1 (per-point cotangent streams -> operands of the weight-gradient reductions), tile-split form.  Same arithmetic and same operand
// layout as dpn_bwd_kernel (bit-identical Z0, Z1, pe6 table, gnet); decomposition as dpn_fwd_tiles_kernel: 64 points per workgroup, wave w owns tiles
// 2w, 2w+1 of both column tiles, the cotangent fragments are shared through LDS, the weights come L2 -> VGPR.
//   Z0 = g pe + sum_c gJ_c d pe / d xi_c                      -> X, K-layout rows (operand of dw1 = T1^T Z0)
//   Z1 = m1 (.) (w1 Z0 + g b1)                                -> K-layout rows (operand of S1 = M2^T Z1)
//   net 0 only: the per-point table pe6                       -> K-layout rows (from which dpn_wgrad_kernel forms G6 = g pe6 of every net: OperandView)
// Round 5: Z = w2 Z1 + Wd G6 + g (b2 + bd + e) is no longer formed.  It existed only as the Y operand of G = M2^T Z, and being linear in
// (Z1, G6, g) that product is S1 w2^T + S2 Wd^T + (M2^T g) (x) cvec: one exact-fp32 GEMM per net behind the reduction (dpn_finish_gside_kernel)
// instead of 114 688 of this kernel's 163 840 MACs per point and net, a 1-KB row written per point and net, and a fourth points-reduction product.
// LDS: the X image of 12 k-steps (48 KB in the hi+lo mode) + the b1 vector: 49 KB, <= 168 registers => THREE workgroups per CU (the kernel is a chain of
// feature evaluation, one short multiply loop and streaming stores: more waves in flight is what hides them).
template <int NS>
__global__ __launch_bounds__(256, 3) void dpn_bwd_tiles_kernel(BwdArgs a) {
    constexpr int kXBytes = 12 * 2 * NS * 1024;
    __shared__ __attribute__((aligned(16))) char lds[kXBytes + 1024];
    // a.reverse (DPN_BWD_ORDER=reverse, a probe of the memory-side cache: DESIGN.md section 4c): workgroups walk nets and tiles in the opposite order of the
    // forward launch, so stage 1 begins on the saved state the forward wrote last.  Same values either way (no sum crosses a tile).
    const int net = a.reverse ? kNets - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const unsigned bx = a.reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    TS_STAMP(0);
    const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
#if TS_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif
    float* vec = reinterpret_cast<float*>(lds + kXBytes) - kVecB1 * 256;     // only b1 is read (acc_init indexes vec + which * 256)
    char* xl = lds + lane * 16;
    {
        const float* gv = reinterpret_cast<const float*>(pk + (long)kPackKB * 1024 * NS) + kVecB1 * 256;
        reinterpret_cast<float*>(lds + kXBytes)[threadIdx.x] = gv[threadIdx.x];
    }
    auto chunk = [&](const int kb) __attribute__((always_inline)) { return pk + (long)kb * 1024 * NS; };
    ts::Head<NS, 2> H;
    ts::gemm_head<NS, 12, 2>(chunk(kS0 + 2 * w * 12), lane, H);
    const int64_t tile0 = (int64_t)bx * 2;
    const int64_t tiles32 = a.n_pad / 32;
    int64_t pc[2];
    float g[2];                                               // cotangent of the lane's point in column tile p (zero for padding points)
    const float gsc = a.g_scale ? a.g_scale[0] : 1.0f;        // an upstream cotangent on unit-cotangent streams (dpn_bwd_points_scaled)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int64_t pt = (tile0 + p) * 32 + j;
        const bool valid = pt < a.n;
        pc[p] = valid ? pt : (a.n - 1);
        g[p] = valid ? gsc * a.g_out[pc[p] * 6 + net] : 0.f;
    }
    const ts::Ident I = ts::make_ident(j, h);
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    OperandView ov = operand_view(a.operands, a.n_pad, NS);
    u32 m1w[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) m1w[p] = reinterpret_cast<const u32*>(sv.m1 + ((int64_t)net * tiles32 + tile0 + p) * 64 + lane)[w];
    if (w == 0) {                                             // lane (j, h): point j of column tile h
        const int64_t pt = (tile0 + h) * 32 + j;
        ov.gnet[(int64_t)net * a.n_pad + pt] = (pt < a.n) ? gsc * a.g_out[pt * 6 + net] : 0.f;
    }
    // ---------------- Z0 -> X (k-steps 0..11) and K-layout rows: wave w builds the (column tile of Z0, point tile) units 3w .. 3w+2.  The rows go
    // out at once (not deferred into the multiply loop as in the forward kernel): nothing of them stays live, the kernel fits 168 registers and
    // three workgroups share a CU -- with one short multiply loop per workgroup, other workgroups are what hides the stores
#pragma unroll
    for (int uu = 0; uu < 3; ++uu) {
        const int u = 3 * w + uu, ct = u >> 1, p = u & 1;     // wave-uniform
        const float gp = p ? g[1] : g[0];
        const int64_t pcp = p ? pc[1] : pc[0];
        float gjc = 0.f;
        if (a.g_jxi && ((tile0 + p) * 32 + j) < a.n) gjc = gsc * a.g_jxi[(pcp * 6 + net) * 3 + (ct >> 1)];
        Frag<NS> f0, f1;
        ts::z0_frag<NS>(f0, a, 2 * ct, h, pcp, gp, gjc);
        ts::z0_frag<NS>(f1, a, 2 * ct + 1, h, pcp, gp, gjc);
        ts::x_store<NS>(xl, 2 * ct, p, f0);
        ts::x_store<NS>(xl, 2 * ct + 1, p, f1);
        ts::save_tile_k<NS, NS>(ov.Z0, net, tile0 + p, ct, lane, I, false, f0, f1);
    }
    TS_STAMP(1);
    ts::barrier_lds();
    TS_STAMP(2);
    // ---------------- Z1 = m1 (.) (w1 Z0 + g b1) -> K-layout rows
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) ts::acc_init(acc[t][p], vec, kVecB1, h, 2 * w + t, g[p]);
    ts::gemm<NS, 12, 2, false, ts::NoSide, false>(chunk(kS0 + 2 * w * 12), xl, lane, H, acc);
    TS_STAMP(3);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            Frag<NS> F0, F1;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const u32 bits = m1w[p] >> (16 * t + r);
                frag_set2<NS>(r < 8 ? F0 : F1, (r & 7) >> 1, (bits & 1u) ? acc[t][p][r] : 0.f, (bits & 2u) ? acc[t][p][r + 1] : 0.f);
            }
            ts::save_tile_k<NS, NS>(ov.Z1, net, tile0 + p, 2 * w + t, lane, I, false, F0, F1);
        }
    TS_STAMP(4);
    if (net != 0) return;
    // ---------------- net 0: the per-point pe6 table (OperandView) -> K-layout rows, units as for Z0
#pragma unroll
    for (int uu = 0; uu < 3; ++uu) {
        const int u = 3 * w + uu, ct = u >> 1, p = u & 1;
        const int64_t pcp = p ? pc[1] : pc[0];
        Frag<NS> f0, f1;
        ts::pe6_frag<NS>(f0, a, 2 * ct, h, pcp);
        ts::pe6_frag<NS>(f1, a, 2 * ct + 1, h, pcp);
        ts::save_tile_k<NS, NS>(ov.PE6, 0, tile0 + p, ct, lane, I, false, f0, f1);
    }
    TS_STAMP(5);
}
