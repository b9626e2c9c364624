// dpn_fwd2_kernel: the fused forward + Jacobian kernel with EIGHT waves per workgroup -- two per SIMD -- as four split-K wave pairs.
// (Included by dpn_kernels.hip inside its point-kernel translation unit; uses that file's fragment / pipeline helpers.)
//
// Why.  The four-wave kernel (dpn_fwd_kernel, one 512-register wave per SIMD) runs its MFMAs at about a third of their rate:
// the per-chunk shader-clock timeline (tools/timeline_probe.py) shows 988 cycles per 16-MFMA chunk in the epilogue-free pass
// (512 would be MFMA-bound) and 2292 per 48-MFMA chunk in the hi+lo mode (1536) -- every chunk boundary (counted vmcnt wait,
// s_barrier, four to eight LDS-DMA issues at ~60 cycles each, the first fragment reads' latency) is paid with an idle matrix
// pipe, because an in-order wave that is alone on its SIMD has nobody to hand the pipe to.  A second wave per SIMD needs the
// kernel inside 256 registers, which the activations of 32 points x 256 channels (64 / 128 registers per layer in, the same out)
// do not allow.
//
// How.  Two waves share one 32-point tile and split every GEMM of the chain along K AND along the output channels:
//   role r (0 / 1) of a pair holds the input channels of the 32-row tiles T with T & 1 == r  (k-steps 2T, 2T+1: half of K),
//   multiplies every weight chunk against that half (half the A-fragment reads, half the MFMAs, half the LDS-DMA pieces per wave),
//   and OWNS the output tiles T with T & 1 == r: the other wave's partial tile travels through LDS (4 KB, published by the next
//   chunk's barrier) and the owner adds it to its own partial, then runs the bias / ReLU / split epilogue and the saved-state
//   stores of that tile in the shadow of the next chunk's MFMAs.
// Per wave: half the activation registers, half the epilogue VALU, half the feature sin/cos, half the DMA issue; per SIMD: two
// waves whose chunk-boundary stalls and epilogues hide behind each other's MFMAs.  Outputs, saved state and packed weights are
// bit-for-bit the four-wave kernel's formats (dpn_bwd / dpn_wgrad are unchanged); sums are formed as (own half) + (partner half),
// so values differ from the four-wave kernel's in the last fp32 bits only.
//
// The two roles run two compile-time specialisations of the same body (fwd2_body<NS, ROLE>), selected by one branch at kernel entry:
// inside a body tile ownership is a constant, so there is no divergent or even uniform branching around the MFMA stream and hipcc
// can schedule epilogues under MFMAs exactly as in the four-wave kernel.  Both bodies execute the same sequence of s_barriers.

// ---- chunk table: the packed per-net weight block (dpn_layout.h) walked in the forward kernel's order
DPN_HD __attribute__((always_inline)) int f2_nk(int c) { return (c < 0 || c >= 54) ? 0 : (c < 8 ? 12 : c < 16 ? 16 : c < 24 ? 12 : 16); }
DPN_HD __attribute__((always_inline)) int f2_kb(int c) {        // offset of chunk c in KB (x NS)
    return c < 8 ? kS0 + 12 * c : c < 16 ? kS1 + 16 * (c - 8) : c < 24 ? kS1 + 128 + 12 * (c - 16) : kS2 + 16 * (c - 24);
}

template <int NS>
struct Pipe8 {
    static constexpr int kRing = (NS == 1) ? 4 : 3;           // hi+lo chunks are 32 KB: three slots + vectors + exchange buffers = 134 KB
    static constexpr int kAhead = kRing - 1;                  // chunks in flight
    static constexpr int kSlotBytes = 16 * 1024 * NS;
    const char* base;
    char* lds;
    int wave, lane;
    DEV void init(const void* gsrc, char* lds_base) {
        base = reinterpret_cast<const char*>(gsrc); lds = lds_base;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); lane = threadIdx.x & 63;
    }
    // 1-KB pieces per wave: a chunk of nk k-steps is nk * NS KB, eight waves copy a contiguous eighth each.  Plain bf16, 12 k-steps:
    // 12 KB is not a multiple of 8 KB -- the waves copy 16 KB (the tail is the head of the next chunk: harmless, inside the slot).
    DPN_HD static int dmas(int nk) { return nk == 0 ? 0 : (NS == 1 ? 2 : nk / 4); }
    DEV void issue(const int c) {
        const int n = dmas(f2_nk(c));
        if (n == 0) return;
        char* slot = lds + (c % kRing) * kSlotBytes;
        const char* src = base + (long)f2_kb(c) * 1024 * NS + wave * (n * 1024) + lane * 16;
        char* dst = slot + wave * (n * 1024);
        if (n > 0) dma16_at<0>(src, dst);
        if (n > 1) dma16_at<1024>(src, dst);
        if (n > 2) dma16_at<2048>(src, dst);
        if (n > 3) dma16_at<3072>(src, dst);
    }
    DEV void prime() {
#pragma unroll
        for (int c = 0; c < kAhead; ++c) issue(c);
    }
    DEV void acquire(const int c) {                           // after this, every wave may read chunk c (and the exchange buffer written during chunk c-1)
        __builtin_amdgcn_sched_barrier(0);
        int allowed = 0;
#pragma unroll
        for (int i = 1; i < kAhead; ++i) allowed += dmas(f2_nk(c + i));
        wait_vmcnt_n(allowed);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        issue(c + kAhead);
        __builtin_amdgcn_sched_barrier(0);
    }
    DEV unsigned buf(const int c) const { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (c % kRing) * kSlotBytes; }
    DEV void drain() { wait_vmcnt<0>(); }
};

// ---- A fragments of ONE k-step (hi [+ lo]) from LDS; KS = the chunk's k-step (compile time), the role part of the address is in `addr`
template <int NS>
struct WK { u32x4 w[NS]; };
template <int NS, int KS>
DEV void lds_load_k(WK<NS>& b, unsigned addr) {
    if constexpr (NS == 1)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(b.w[0]) : "v"(addr), "n"(KS * 1024) : "memory");
    else
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(b.w[0]), "=&v"(b.w[1])
                     : "v"(addr), "n"(KS * 2048), "n"(KS * 2048 + 1024) : "memory");
}
template <int NS, int N>
DEV void lds_wait_k(WK<NS>& b) {                // at most N LDS operations issued after this k-step's reads are outstanding
    if constexpr (NS == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b.w[0]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b.w[0]), "+v"(b.w[1]) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x20 | 0x40 | 0x400);
}
template <int NS>
DEV void mma_k(const WK<NS>& b, const Frag<NS>& act, f32x16& acc) {
    const bf16x8 whi = as_bf(b.w[0]);
    if constexpr (NS == 2) {
        const bf16x8 wlo = as_bf(b.w[1]);
        acc = mfma(whi, as_bf(act.w[1]), acc);
        acc = mfma(wlo, as_bf(act.w[0]), acc);
    }
    acc = mfma(whi, as_bf(act.w[0]), acc);
}

// the role's half of a chunk: local k-step li  <->  chunk k-step 4 (li >> 1) + 2 ROLE + (li & 1); three k-steps' reads in flight
template <int NS, int NKL, int ROLE>
DEV void mma_half(unsigned slot_addr, const Frag<NS>* act, f32x16& acc) {
    static_assert(NKL == 6 || NKL == 8, "half of 12 or 16 k-steps");
    const unsigned addr = slot_addr + (threadIdx.x & 63) * 16;
#define F2_KS(LI) (4 * ((LI) >> 1) + 2 * ROLE + ((LI) & 1))
    WK<NS> b0, b1, b2;
    lds_load_k<NS, F2_KS(0)>(b0, addr);
    lds_load_k<NS, F2_KS(1)>(b1, addr);
    lds_load_k<NS, F2_KS(2)>(b2, addr);
    lds_wait_k<NS, 2 * NS>(b0); mma_k<NS>(b0, act[0], acc); lds_load_k<NS, F2_KS(3)>(b0, addr);
    lds_wait_k<NS, 2 * NS>(b1); mma_k<NS>(b1, act[1], acc); lds_load_k<NS, F2_KS(4)>(b1, addr);
    lds_wait_k<NS, 2 * NS>(b2); mma_k<NS>(b2, act[2], acc); lds_load_k<NS, F2_KS(5)>(b2, addr);
    if constexpr (NKL == 8) {
        lds_wait_k<NS, 2 * NS>(b0); mma_k<NS>(b0, act[3], acc); lds_load_k<NS, F2_KS(6)>(b0, addr);
        lds_wait_k<NS, 2 * NS>(b1); mma_k<NS>(b1, act[4], acc); lds_load_k<NS, F2_KS(7)>(b1, addr);
        lds_wait_k<NS, 2 * NS>(b2); mma_k<NS>(b2, act[5], acc);
        lds_wait_k<NS, 1 * NS>(b0); mma_k<NS>(b0, act[6], acc);
        lds_wait_k<NS, 0>(b1); mma_k<NS>(b1, act[7], acc);
    } else {
        lds_wait_k<NS, 2 * NS>(b0); mma_k<NS>(b0, act[3], acc);
        lds_wait_k<NS, 1 * NS>(b1); mma_k<NS>(b1, act[4], acc);
        lds_wait_k<NS, 0>(b2); mma_k<NS>(b2, act[5], acc);
    }
#undef F2_KS
}

// ---- partial-tile exchange between the two waves of a pair: 16 fp32 per lane = 4 x ds_write_b128 / ds_read_b128 (lane stride 16 B).
// Inline asm for the reason every other LDS access of these kernels is: an access hipcc can see is ordered behind all LDS-DMA in flight.
// The MFMA results are read by the ds_write: the wait states a compiler-visible reader would get are spelled out (s_nop).
DEV void xchg_send(const f32x16& a, unsigned addr) {
    f32x4 q0 = {a[0], a[1], a[2], a[3]}, q1 = {a[4], a[5], a[6], a[7]}, q2 = {a[8], a[9], a[10], a[11]}, q3 = {a[12], a[13], a[14], a[15]};
    asm volatile("s_nop 15\n\ts_nop 3\n\tds_write_b128 %4, %0\n\tds_write_b128 %4, %1 offset:1024\n\tds_write_b128 %4, %2 offset:2048\n\t"
                 "ds_write_b128 %4, %3 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 :: "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(addr) : "memory");
}
struct Part { f32x4 q[4]; };
DEV void xchg_recv_issue(Part& p, unsigned addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                 : "=&v"(p.q[0]), "=&v"(p.q[1]), "=&v"(p.q[2]), "=&v"(p.q[3]) : "v"(addr) : "memory");
}
DEV void xchg_recv_wait(Part& p) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3]) :: "memory");
}
DEV void acc_add_part(f32x16& acc, const Part& p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[4 * q] += p.q[q][0]; acc[4 * q + 1] += p.q[q][1]; acc[4 * q + 2] += p.q[q][2]; acc[4 * q + 3] += p.q[q][3]; }
}

// ---- per-lane context of a role: only the frequencies its k-steps use
struct Lane2 {
    int lane, j, h;
    int64_t pt;
    bool valid;
    float xi[3];
    float fr3[8];      // coordinate PE: fr3[4 x + p] = freq32[8 (2 ROLE + x) + 4 h + p]
    float fr6[8];      // data PE:       fr6[4 x + p] = freq16[8 x + 4 h + p]
    u32x4 idA, idB;
};
template <int ROLE>
DEV void lane_init2(Lane2& L, const float* x, const float* y, const float* t, int64_t n, const float* freqs, const DpnGeometry& geo, int64_t tile32) {
    L.lane = threadIdx.x & 63;
    L.j = L.lane & 31;
    L.h = L.lane >> 5;
    L.pt = tile32 * 32 + L.j;
    L.valid = L.pt < n;
    const int64_t pc = L.valid ? L.pt : (n - 1);
    L.xi[0] = x[pc] / geo.dx / geo.lon_m1;       // interface_physics.py:324-326 (two fp32 divisions, like the reference)
    L.xi[1] = y[pc] / geo.dy / geo.lat_m1;
    L.xi[2] = t[pc] / geo.pred_t_span;
#pragma unroll
    for (int m = 0; m < 8; ++m) L.fr3[m] = freqs[8 * (2 * ROLE + (m >> 2)) + 4 * L.h + (m & 3)];
#pragma unroll
    for (int m = 0; m < 8; ++m) L.fr6[m] = freqs[32 + 8 * (m >> 2) + 4 * L.h + (m & 3)];
    const int mine = (((L.j >> 3) & 1) == L.h) ? (L.j & 7) : -1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const u32 v = ((mine == 2 * p) ? 0x3F80u : 0u) | ((mine == 2 * p + 1) ? 0x3F800000u : 0u);
        L.idA[p] = (L.j < 16) ? v : 0u;
        L.idB[p] = (L.j >= 16) ? v : 0u;
    }
}
// store_tile_k takes the four-wave kernel's Lane: the fields it reads
DEV Lane lane_view(const Lane2& L2) {
    Lane L;
    L.lane = L2.lane; L.j = L2.j; L.h = L2.h; L.pt = L2.pt; L.valid = L2.valid; L.idA = L2.idA; L.idB = L2.idB;
    return L;
}

template <int NS, int ROLE>
DEV void fwd2_body(const FwdArgs& a, const int net, const int pg, const int64_t tile32, const char* pk, char* lds_w, const unsigned lds_vec,
                   const unsigned lds_x) {
#ifdef DPN_TIMELINE
    u32 tl = 0;
    DPN_STAMP(0);
#endif
    Lane2 L;
    lane_init2<ROLE>(L, a.x, a.y, a.t, a.n, a.freqs, a.geo, tile32);
    const Lane Lv = lane_view(L);
    const int h = L.h;
    const bool partial = (tile32 * 32 + 32 > a.n);
    const int64_t pc = L.valid ? L.pt : (a.n - 1);
    float cd3[3];                                   // the three data channels this role encodes: coord_data[2 g + ROLE]
#pragma unroll
    for (int g = 0; g < 3; ++g) cd3[g] = a.coord_data[pc * 6 + 2 * g + ROLE];
    const float ref_data = a.coord_data[pc * 6 + net];
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    const bool save = a.saved != nullptr;
    // exchange buffers: [2][4 point groups][4 quarters][64 lanes][16 B]; the tile with running number S goes through buffer S & 1
    const unsigned xbase = lds_x + pg * 4096 + L.lane * 16;
    const unsigned roff = 2 * ROLE * NS * 1024;      // this role's k-steps inside a chunk: 4 b + 2 ROLE + x

    Pipe8<NS> pipe;
    pipe.init(pk, lds_w);
    pipe.prime();
    DPN_STAMP(1);

    f32x16 acc[2];
    u32 m1w[4] = {0u, 0u, 0u, 0u};
    Frag<NS> actA[8], actB[8];

// One pipeline step = one weight chunk = one 32-row output tile.  C: chunk (= running tile number, the exchange buffer is C & 1);
// ACC: the tile's accumulator (acc[T & 1]); MINE / INIT: ownership and the owner's accumulator start (the partner starts from 0);
// PREV_MINE: the previous tile is mine -> fetch the partner's half (published by this step's barrier), add it, run EPI_PREV --
// in the shadow of this step's MFMAs, or (EPI_FIRST: the first step of a layer, whose MFMAs read what that epilogue writes) before them.
#define F2_STEP(C, NKL, ACT, ACC, MINE, INIT, PREV_MINE, ACC_PREV, EPI_PREV, EPI_FIRST)                             \
    do {                                                                                                            \
        DPN_STAMP(2 + (C));                                                                                         \
        pipe.acquire(C);                                                                                            \
        Part part_;                                                                                                 \
        if constexpr (PREV_MINE) xchg_recv_issue(part_, xbase + (((C) - 1) & 1) * 16384);                           \
        if constexpr (MINE) { INIT; } else { (ACC) = (f32x16)0.f; }                                                 \
        if constexpr (PREV_MINE) { xchg_recv_wait(part_); acc_add_part((ACC_PREV), part_); }                        \
        if constexpr ((PREV_MINE) && (EPI_FIRST)) { EPI_PREV; }                                                     \
        mma_half<NS, (NKL), ROLE>(pipe.buf(C) + roff, (ACT), (ACC));                                                \
        if constexpr ((PREV_MINE) && !(EPI_FIRST)) { EPI_PREV; }                                                    \
        if constexpr (!(MINE)) xchg_send((ACC), xbase + ((C) & 1) * 16384);                                         \
    } while (0)
// the common shape: tile T of a layer whose first chunk is C0; the previous tile is T - 1 of this layer or tile 7 of the one before
#define F2_TILE(C0, T, NKL, ACT, INIT, EPI_PREV, EPI_FIRST)                                                                       \
    F2_STEP((C0) + (T), NKL, ACT, acc[(T) & 1], (((T) & 1) == ROLE), INIT, ((((T) + 1) & 1) == ROLE), acc[((T) + 1) & 1], EPI_PREV, \
            EPI_FIRST)

    // ---------------- L1: pre1 = w1 . pe + b1 ; h1 = relu -> actA ; relu mask -> m1w           tiles S = 0..7
    auto epi1 = [&](const int T) __attribute__((always_inline)) {
        f32x16& v = acc[T & 1];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = v[r], p1 = v[r + 1];
            m1w[T >> 1] |= ((p0 > 0.f) ? (1u << (16 * (T & 1) + r)) : 0u) | ((p1 > 0.f) ? (2u << (16 * (T & 1) + r)) : 0u);
            frag_set2<NS>(actA[2 * (T >> 1) + (r >> 3)], (r & 7) >> 1, relu1(p0), relu1(p1));
        }
        asm volatile("" : "+v"(m1w[T >> 1]));
    };
    {
        Frag<NS> pe[6];
#pragma unroll
        for (int li = 0; li < 6; ++li) {
            const int c = li >> 1;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float s, co;
                sincos_t<NS>(L.xi[c] * L.fr3[4 * (li & 1) + p], s, co);
                frag_set2<NS>(pe[li], p, s, co);
            }
        }
        F2_STEP(0, 6, pe, acc[0], (0 == ROLE), acc_init_vec(acc[0], lds_vec, kVecB1, h, 0, 1.0f), false, acc[1], (void)0, false);
#define F2_L1(T) F2_TILE(0, T, 6, pe, acc_init_vec(acc[(T) & 1], lds_vec, kVecB1, h, (T), 1.0f), epi1((T) - 1), false)
        F2_L1(1); F2_L1(2); F2_L1(3); F2_L1(4); F2_L1(5); F2_L1(6); F2_L1(7);
#undef F2_L1
    }
    // tile 7 (role 1's) is finished inside the first step of the next layer.
    // ---------------- L2: s = w2 . h1 + (b2 + bd + e) -> actB (hi+lo)                           tiles S = 8..15
    auto epi2a = [&](const int T) __attribute__((always_inline)) {
        f32x16& v = acc[T & 1];
#pragma unroll
        for (int r = 0; r < 16; r += 2) frag_set2<NS>(actB[2 * (T >> 1) + (r >> 3)], (r & 7) >> 1, v[r], v[r + 1]);
    };
    // the step of tile T finishes tile T-1 of the same layer, or (T = 0) tile 7 of the previous layer -- BEFORE its own MFMAs then
#define F2_L2(T, EPI, FIRST) F2_TILE(8, T, 8, actA, acc_init_vec(acc[(T) & 1], lds_vec, kVecCvec, h, (T), 1.0f), EPI, FIRST)
    F2_L2(0, epi1(7), true);
    F2_L2(1, epi2a(0), false); F2_L2(2, epi2a(1), false); F2_L2(3, epi2a(2), false); F2_L2(4, epi2a(3), false);
    F2_L2(5, epi2a(4), false); F2_L2(6, epi2a(5), false); F2_L2(7, epi2a(6), false);
#undef F2_L2
    // ---------------- + Wd . pe6 -> c (actB, re-split) ; cdot = wo . c                           tiles S = 16..23
    // (the data term is added to the already split w2 term: c = (hi + lo) + Wd . pe6 -- with pe6 resident NEXT to h1 and c the hi+lo
    //  body does not fit in 256 registers; the re-rounding is one more 2^-16 on c)
    float cdot = 0.f;
    auto epi2b = [&](const int T) __attribute__((always_inline)) {
        f32x16& v = acc[T & 1];
        Vec16 wv;
        lds_read_vec16(wv, vec_addr(lds_vec, kVecWo, h, T));
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            Frag<NS>& f = actB[2 * (T >> 1) + (r >> 3)];
            const int p = (r & 7) >> 1;
            float s0 = bf_lo(f.w[0][p]), s1 = bf_hi(f.w[0][p]);
            if constexpr (NS == 2) { s0 += bf_lo(f.w[1][p]); s1 += bf_hi(f.w[1][p]); }
            const float c0 = v[r] + s0, c1 = v[r + 1] + s1;
            cdot = fmaf(wv.q[r >> 2][r & 3], c0, cdot);
            cdot = fmaf(wv.q[r >> 2][(r & 3) + 1], c1, cdot);
            frag_set2<NS>(f, p, c0, c1);
        }
    };
    {
        Frag<NS> pe6[6];
#pragma unroll
        for (int li = 0; li < 6; ++li) {
            const float v = cd3[li >> 1];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float s, co;
                sincos_t<NS>(v * L.fr6[4 * (li & 1) + p], s, co);
                frag_set2<NS>(pe6[li], p, s, co);
            }
        }
#define F2_WD(T, EPI) F2_TILE(16, T, 6, pe6, (acc[(T) & 1] = (f32x16)0.f), EPI, false)
        F2_WD(0, epi2a(7));                       // (this step's MFMAs read pe6, not actB: the epilogue may follow them)
        F2_WD(1, epi2b(0)); F2_WD(2, epi2b(1)); F2_WD(3, epi2b(2)); F2_WD(4, epi2b(3)); F2_WD(5, epi2b(4)); F2_WD(6, epi2b(5)); F2_WD(7, epi2b(6));
#undef F2_WD
    }
    // ---------------- fc1: pre2 = W1 . c + bf1 ; a = relu ; t2 = m2 (.) u -> actA ; M2 -> saved      tiles S = 24..31
    float adot = 0.f;
    auto epi3 = [&](const int T) __attribute__((always_inline)) {
        f32x16& v = acc[T & 1];
        Frag<1> mk0, mk1;
        Vec16 uv;
        lds_read_vec16(uv, vec_addr(lds_vec, kVecU, h, T));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float uu[4] = {uv.q[q][0], uv.q[q][1], uv.q[q][2], uv.q[q][3]};
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const int r = 4 * q + i;
                const float p0 = v[r], p1 = v[r + 1];
                const bool on0 = p0 > 0.f, on1 = p1 > 0.f;
                const float t0 = on0 ? uu[i] : 0.f, t1 = on1 ? uu[i + 1] : 0.f;
                adot = fmaf(p0, t0, adot);
                adot = fmaf(p1, t1, adot);
                frag_set2<NS>(actA[2 * (T >> 1) + (r >> 3)], (r & 7) >> 1, t0, t1);
                const u32 mw = (on0 ? 0x3F80u : 0u) | (on1 ? 0x3F800000u : 0u);
                if (r < 8) mk0.w[0][(r & 7) >> 1] = mw; else mk1.w[0][(r & 7) >> 1] = mw;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) asm volatile("" : "+v"(actA[2 * (T >> 1)].w[s2]), "+v"(actA[2 * (T >> 1) + 1].w[s2]));
        asm volatile("" : "+v"(mk0.w[0]), "+v"(mk1.w[0]));
        if (save) store_tile_k<1, 1>(sv.M2, net, tile32, T, Lv, mk0, mk1, partial);
    };
#define F2_FC1(T, EPI, FIRST) F2_TILE(24, T, 8, actB, acc_init_vec(acc[(T) & 1], lds_vec, kVecBf1, h, (T), 1.0f), EPI, FIRST)
    F2_FC1(0, epi2b(7), true);
    F2_FC1(1, epi3(0), false); F2_FC1(2, epi3(1), false); F2_FC1(3, epi3(2), false); F2_FC1(4, epi3(3), false);
    F2_FC1(5, epi3(4), false); F2_FC1(6, epi3(5), false); F2_FC1(7, epi3(6), false);
#undef F2_FC1
    const bool fields_only = !save && !a.jac_n;
    // ---------------- reverse sweep: v = W1^T t2 + 2 wo -> actB (+ saved V)                          tiles S = 32..39
    auto epiv = [&](const int T) __attribute__((always_inline)) {
        f32x16& v = acc[T & 1];
#pragma unroll
        for (int r = 0; r < 16; r += 2) frag_set2<NS>(actB[2 * (T >> 1) + (r >> 3)], (r & 7) >> 1, v[r], v[r + 1]);
        if (save) store_tile_k<NS, NS>(sv.V, net, tile32, T, Lv, actB[2 * (T >> 1)], actB[2 * (T >> 1) + 1], partial);
    };
    // the out value: this role's channels -> partner through the exchange buffer (role 1 sends, role 0 writes out_n)
    auto out_value = [&]() __attribute__((always_inline)) {
        float o = adot + 2.0f * cdot;
        o += __shfl_xor(o, 32);
        const unsigned oaddr = lds_x + 32768 + pg * 256 + L.lane * 4;          // 1 KB behind the exchange buffers
        if constexpr (ROLE == 1) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(oaddr), "v"(o) : "memory");
        return o;
    };
    const float const0 = lds_read_f32(lds_vec + kNumVecs * 256 * 4);
    if (fields_only) {
        // value-only call (inference / data loss without gradient): finish tile 7 of fc1, combine the halves, leave
        DPN_STAMP(2 + 32);
        pipe.acquire(32);
        if constexpr (ROLE == 1) { Part p_; xchg_recv_issue(p_, xbase + (31 & 1) * 16384); xchg_recv_wait(p_); acc_add_part(acc[1], p_); epi3(7); }
        const float o = out_value();
        pipe.drain();
        __builtin_amdgcn_s_barrier();
        if constexpr (ROLE == 0) {
            const float o1 = lds_read_f32(lds_x + 32768 + pg * 256 + L.lane * 4);
            if (L.valid && h == 0) a.out_n[L.pt * 6 + net] = o + o1 + const0 + ref_data;
        }
        return;
    }
#define F2_V(T, EPI, FIRST) F2_TILE(32, T, 8, actA, acc_init_vec(acc[(T) & 1], lds_vec, kVecWo, h, (T), 2.0f), EPI, FIRST)
    F2_V(0, epi3(7), true);
    float o_mine = out_value();                                        // adot is complete after epi3(7) (role 1) / epi3(6) (role 0)
    F2_V(1, epiv(0), false);
    if constexpr (ROLE == 0) {                                         // role 1's half was published by the barrier of step 33
        const float o1 = lds_read_f32(lds_x + 32768 + pg * 256 + L.lane * 4);
        if (L.valid && h == 0) a.out_n[L.pt * 6 + net] = o_mine + o1 + const0 + ref_data;   // + ref_data (variable_net.py:86)
    }
    F2_V(2, epiv(1), false); F2_V(3, epiv(2), false); F2_V(4, epiv(3), false); F2_V(5, epiv(4), false); F2_V(6, epiv(5), false);
    F2_V(7, epiv(6), false);
#undef F2_V
    // relu-1 mask bits of this role's tiles: half-words of the four mask words (role 0: low halves, role 1: high halves)
    if (save) {
        unsigned short* mp = reinterpret_cast<unsigned short*>(sv.m1 + ((int64_t)net * (a.n_pad / 32) + tile32) * 64 + L.lane) + ROLE;
#pragma unroll
        for (int k = 0; k < 4; ++k) mp[2 * k] = (unsigned short)(m1w[k] >> (16 * ROLE));
    }
    // ---------------- y = w2^T v ; t1 = m1 (.) y -> actA (+ saved T1)                                 tiles S = 40..47
    auto epiy = [&](const int T) __attribute__((always_inline)) {
        f32x16& v = acc[T & 1];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const u32 bits = m1w[T >> 1] >> (16 * (T & 1) + r);
            frag_set2<NS>(actA[2 * (T >> 1) + (r >> 3)], (r & 7) >> 1, (bits & 1u) ? v[r] : 0.f, (bits & 2u) ? v[r + 1] : 0.f);
        }
        if (save) store_tile_k<NS, NS>(sv.T1, net, tile32, T, Lv, actA[2 * (T >> 1)], actA[2 * (T >> 1) + 1], partial);
    };
#define F2_Y(T, EPI, FIRST) F2_TILE(40, T, 8, actB, (acc[(T) & 1] = (f32x16)0.f), EPI, FIRST)
    F2_Y(0, epiv(7), true);
    F2_Y(1, epiy(0), false); F2_Y(2, epiy(1), false); F2_Y(3, epiy(2), false); F2_Y(4, epiy(3), false); F2_Y(5, epiy(4), false);
    F2_Y(6, epiy(5), false); F2_Y(7, epiy(6), false);
#undef F2_Y
    if (!a.jac_n) {
        DPN_STAMP(2 + 48);
        pipe.acquire(48);
        if constexpr (ROLE == 1) { Part p_; xchg_recv_issue(p_, xbase + (47 & 1) * 16384); xchg_recv_wait(p_); acc_add_part(acc[1], p_); epiy(7); }
        pipe.drain();
        return;
    }
    // ---------------- gpe = w1^T t1 (6 tiles), contracted with d(pe)/d(xi) in registers               tiles S = 48..53
    float jc[3] = {0.f, 0.f, 0.f};
    auto epij = [&](const int T) __attribute__((always_inline)) {
        f32x16& v = acc[T & 1];
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const int r = 2 * rp;
            const int ks = 2 * T + (r >> 3), p = (r & 7) >> 1, c = ks >> 2;       // ks & 3 = 2 ROLE + (r >> 3) for this role's tiles
            const float fr = L.fr3[4 * (r >> 3) + p];
            float s, co;
            sincos_t<NS>(L.xi[c] * fr, s, co);
            jc[c] = fmaf(v[r], fr * co, jc[c]);
            jc[c] = fmaf(v[r + 1], -fr * s, jc[c]);
        }
    };
#define F2_G(T, EPI, FIRST) F2_TILE(48, T, 8, actA, (acc[(T) & 1] = (f32x16)0.f), EPI, FIRST)
    F2_G(0, epiy(7), true);
    F2_G(1, epij(0), false); F2_G(2, epij(1), false); F2_G(3, epij(2), false); F2_G(4, epij(3), false); F2_G(5, epij(4), false);
#undef F2_G
    pipe.drain();
    // tile 5 is role 1's: its partner half was sent at the end of step 53; one more barrier publishes it and the Jacobian halves
    __builtin_amdgcn_s_barrier();
    if constexpr (ROLE == 1) { Part p_; xchg_recv_issue(p_, xbase + (53 & 1) * 16384); xchg_recv_wait(p_); acc_add_part(acc[1], p_); epij(5); }
#pragma unroll
    for (int c = 0; c < 3; ++c) jc[c] += __shfl_xor(jc[c], 32);
    const unsigned jaddr = lds_x + 32768 + 1024 + pg * 1024 + L.lane * 16;
    if constexpr (ROLE == 1) {
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:8\n\ts_waitcnt lgkmcnt(0)"
                     :: "v"(jaddr), "v"(jc[0]), "v"(jc[1]), "v"(jc[2]) : "memory");
    }
    __builtin_amdgcn_s_barrier();
#ifdef DPN_TIMELINE
    DPN_STAMP(62);
    if (a.timeline) a.timeline[(((int64_t)blockIdx.x * kNets + net) * 8 + (2 * pg + ROLE)) * 64 + L.lane] = tl;
#endif
    if constexpr (ROLE == 0) {
        const float j0 = jc[0] + lds_read_f32(jaddr), j1 = jc[1] + lds_read_f32(jaddr + 4), j2 = jc[2] + lds_read_f32(jaddr + 8);
        if (L.valid && h == 0) {
            float* o = a.jac_n + (L.pt * 6 + net) * 3;
            o[0] = j0 / a.geo.lon_m1 / a.geo.dx;          // chain rule through x/dx/(lon-1), in the reference's backward order
            o[1] = j1 / a.geo.lat_m1 / a.geo.dy;
            o[2] = j2 / a.geo.pred_t_span;
        }
    }
#undef F2_TILE
#undef F2_STEP
}

template <int NS>
__global__ __launch_bounds__(512, 2) void dpn_fwd2_kernel(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds_w[Pipe8<NS>::kRing * Pipe8<NS>::kSlotBytes];
    __shared__ __attribute__((aligned(16))) float lds_vec_store[kNumVecs * 256 + 4];
    __shared__ __attribute__((aligned(16))) char lds_xchg[32768 + 1024 + 4096];      // exchange buffers | out halves | Jacobian halves
    const int net = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pg = wave >> 1, role = wave & 1;
    const int64_t tile32 = (int64_t)blockIdx.x * 4 + pg;
    const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
    {   // permuted fp32 vectors of this net -> LDS (published by the first step's barrier; no DMA is in flight yet)
        const float* gv = reinterpret_cast<const float*>(pk + (long)kPackKB * 1024 * NS);
        for (int i = threadIdx.x; i < kNumVecs * 256 + 4; i += 512) lds_vec_store[i] = gv[i];
    }
    __syncthreads();
    const unsigned lds_vec = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds_vec_store;
    const unsigned lds_x = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_xchg;
    if (role == 0) fwd2_body<NS, 0>(a, net, pg, tile32, pk, lds_w, lds_vec, lds_x);
    else fwd2_body<NS, 1>(a, net, pg, tile32, pk, lds_w, lds_vec, lds_x);
}
