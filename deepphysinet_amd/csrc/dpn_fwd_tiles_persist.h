// Forward + Jacobian kernel of the parity-grade (hi+lo) mode, tile-split form with PERSISTENT workgroups and the next item's coordinate features built by
// the wave that idles through the last GEMM (round 6, VERDICT r5 item 1 variant (b)).  Included by dpn_kernels.hip behind dpn_fwd_tiles.h.
//
// Same arithmetic per point and the same order of operations per output tile as dpn_fwd_tiles_kernel (the GEMM / epilogue blocks below are its text): fields,
// Jacobian and saved state are bit-identical (tools/fwd_dump.py, tests).  What changes:
//   * a workgroup walks items (net, 64-point tile) it, it + grid, ... instead of owning one: grid = two workgroups per CU;
//   * in dpn_fwd_tiles_kernel the last GEMM (gpe = w1^T t1: six output tiles) runs on waves 0..2 and wave 3 has left; every workgroup then begins with a
//     prologue in which all four waves evaluate 24 sin / cos pairs per lane before the first MFMA can issue (8.8 k of a wave's 90.8 k cycles: timeline of
//     round 6).  Here wave 3 evaluates nine of the NEXT item's twelve pe3 k-steps while waves 0..2 multiply and contract (six are held in registers until the
//     three multiplying waves have signalled, an LDS word each, that they have left the loop that reads X; three are stored as they are built), and waves
//     0..2 add one k-step each behind their contraction.  The next item starts at its first GEMM.
// Training shape only (saved state AND Jacobian); everything else stays on dpn_fwd_tiles_kernel.
#pragma once
#ifndef TSP_W3_KSTEPS
#define TSP_W3_KSTEPS 9                       // next item's k-steps 0 .. TSP_W3_KSTEPS-1 by wave 3, one each of the rest by waves 0, 1, ..
#endif

// Experiment build (-DDPN_TIMELINE -DTSP_TIMELINE, tools/persist_timeline.py): lane 0 of every wave stamps the shader clock at phase boundaries of the
// workgroup's THIRD item: a.timeline[workgroup][wave][16]
#if defined(TSP_TIMELINE) && defined(DPN_TIMELINE)
#define TSP_STAMP(I) do { if (a.timeline && lane == 0 && seq == 3) a.timeline[((size_t)blockIdx.x * 4 + w) * 16 + (I)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define TSP_STAMP(I) do { } while (0)
#endif

template <int NS>
__global__ __launch_bounds__(256, 2) void dpn_fwd_tiles_persist_kernel(FwdArgs a, int n_nets) {
    using C = ts::Cfg<NS>;
    __shared__ __attribute__((aligned(16))) char lds[C::kLdsBytes];
    __shared__ int flags[4];                 // flags[w] = sequence number of the last item whose gpe multiply loop wave w has left
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane0 = threadIdx.x & 63;
#if TS_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif
    float* vec = reinterpret_cast<float*>(lds + C::kVecOff);
    float* red = reinterpret_cast<float*>(lds + C::kRedOff);
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    const int64_t tiles32 = a.n_pad / 32;
    const int tiles64 = (int)(a.n_pad / 64);
    const int items = tiles64 * n_nets;
    if (threadIdx.x < 4) flags[threadIdx.x] = 0;
#ifdef TSP_STAGGER
    // experiment: the two workgroups of a CU start together and run the same code at the same pace -- in LOCKSTEP (both multiply, then both serve); hold the
    // second half of the grid back by TSP_STAGGER x 64 x 100 cycles
    if (blockIdx.x >= gridDim.x / 2)
        for (int d = 0; d < TSP_STAGGER; ++d) __builtin_amdgcn_s_sleep(100);
#endif
    int cur_net = -1, seq = 0;
    f32x16 acc[2][2];
    Frag<NS> F[2][2][2];                     // [tile t][column tile p][k-step of the tile's pair]: the epilogue's output fragments
    ts::Head<NS, 2> H;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        ++seq;
        // everything derived from the lane index is derived again per item: hoisted out of the loop these values (fragment offsets, identity fragments, save
        // addresses) stay live across the whole item and push the kernel over its 256 registers (34 dwords of scratch spills, reloaded inside the GEMM loops)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int j = lane & 31, h = lane >> 5;
        char* xl = lds + lane * 16;
        const ts::Ident I = ts::make_ident(j, h);
        TSP_STAMP(0);
        const int net = it / tiles64;
        const int64_t tile0 = (int64_t)(it - net * tiles64) * 2;       // first of this workgroup's two 32-point column tiles
        const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
        auto chunk = [&](const int kb) __attribute__((always_inline)) { return pk + (long)kb * 1024 * NS; };
        int64_t pc[2];
        bool zero_rows[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int64_t pt = (tile0 + p) * 32 + j;
            const bool valid = pt < a.n;
            pc[p] = valid ? pt : (a.n - 1);
            zero_rows[p] = ((tile0 + p) * 32 + 32 > a.n) && !valid;    // saved rows of padding points are zero
        }
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
        ts::gemm_head<NS, 12, 2>(chunk(kF0 + 2 * w * 12), lane, H);
        const bool first = seq == 1;
        if (net != cur_net) {   // permuted fp32 vectors of this net -> LDS (the previous item's last reader of the old ones is behind the barrier that ended it)
            const u32x4* gv = reinterpret_cast<const u32x4*>(pk + (long)kPackKB * 1024 * NS);
            const int i0 = threadIdx.x, i1 = threadIdx.x + 256;
            const u32x4 v0 = gv[i0];
            const u32x4 v1 = gv[i1 < ts::kVecFloats / 4 ? i1 : i0];
            reinterpret_cast<u32x4*>(vec)[i0] = v0;
            if (i1 < ts::kVecFloats / 4) reinterpret_cast<u32x4*>(vec)[i1] = v1;
        }
        if (first) {            // the workgroup's first item: coordinate features pe3 -> X (k-steps 0..11) by all four waves, as in dpn_fwd_tiles_kernel
            const int c0 = (3 * w) >> 2, c1 = (3 * w + 2) >> 2;    // (the *_x forms: a pointer selected by the run-time k-step puts the kernel arguments into scratch here)
            float xi0[2][2];
            f32x4 fr0[3];
#pragma unroll
            for (int p = 0; p < 2; ++p) { xi0[0][p] = ts::load_xi(a, c0, pc[p]); xi0[1][p] = ts::load_xi(a, c1, pc[p]); }
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) fr0[kk] = ts::load_fr4(a.freqs, 8 * ((3 * w + kk) & 3) + 4 * h);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    Frag<NS> f;
                    ts::pe3_frag_x<NS>(f, (((3 * w + kk) >> 2) == c0) ? xi0[0][p] : xi0[1][p], fr0[kk]);
                    ts::x_store<NS>(xl, 3 * w + kk, p, f);
                }
        }
        if (first || net != cur_net) ts::barrier_lds();     // (later items: X was published by the barrier that ended the previous item)
        cur_net = net;
        const bool save = true;
        TSP_STAMP(1);
        // ---------------- L1: pre1 = w1 . pe + b1 ; h1 = relu -> X ; relu mask bits -> m1w ; hdot = (w2^T wo) . h1 (this wave's 64 channels)
        u32 m1w[2] = {0u, 0u};
        init_all(kVecB1, 1.0f);
        ts::gemm<NS, 12, 2>(chunk(kF0 + 2 * w * 12), xl, lane, H, acc);
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
        {
#pragma unroll
            for (int p = 0; p < 2; ++p)          // word w of the lane's uint4 = tiles 2w (low half), 2w+1 (high half): the ring kernel's m1w[T >> 1]
                reinterpret_cast<u32*>(sv.m1 + ((int64_t)net * tiles32 + tile0 + p) * 64 + lane)[w] = m1w[p];
        }
        ts::barrier_lds();                       // everybody is done reading pe3
        x_store_all();
        ts::barrier_lds();
        // ---------------- pre2 = A h1 + B pe6 + (W1 cvec + bf1)
        init_all(kVecC2, 1.0f);
        ts::gemm<NS, 16, 2>(chunk(kFA + 2 * w * 16), xl, lane, H, acc);
        ts::gemm_head<NS, 12, 2>(chunk(kFB + 2 * w * 12), lane, H);
        float ddot[2] = {0.f, 0.f};              // (Wd^T wo) . pe6 over this wave's k-steps
        {   // data features pe6 (SineCosPE(6,16) of coord_data): k-steps 3w .. 3w+2 of both column tiles, built while the accumulators wait
            Frag<NS> f6[3][2];
            const float* bv = vec + kVecBv * 256;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int p = 0; p < 2; ++p) ts::pe6_frag_dot<NS>(f6[kk][p], a, 3 * w + kk, h, pc[p], bv, ddot[p]);
            ts::barrier_lds();                   // everybody is done reading h1
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int p = 0; p < 2; ++p) ts::x_store<NS>(xl, 3 * w + kk, p, f6[kk][p]);
            ts::barrier_lds();
        }
        ts::gemm<NS, 12, 2>(chunk(kFB + 2 * w * 12), xl, lane, H, acc);
        ts::gemm_head<NS, 16, 2>(chunk(kFAT + 2 * w * 16), lane, H);
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
        ts::barrier_lds();
        x_store_all();
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
        // ---------------- reverse sweep: y = A^T t2 + 2 w2^T wo ; t1 = m1 (.) y -> X (+ saved T1)
        init_all(kVecA2, 2.0f);
#if TS_DEFER_SAVES
        {
            auto side = [&](const int ks) __attribute__((always_inline)) {            // M2: four (tile, column tile) units over the 16 k-steps
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ks == 4 * u + 1) ts::save_plane_k(sv.M2, net, 1, 0, tile0 + (u & 1), 2 * w + (u >> 1), lane, I, zero_rows[u & 1], MK[u >> 1][u & 1][0].w[0], MK[u >> 1][u & 1][1].w[0]);
            };
            ts::gemm<NS, 16, 2, false>(chunk(kFAT + 2 * w * 16), xl, lane, H, acc, side);
        }
#else
        ts::gemm<NS, 16, 2>(chunk(kFAT + 2 * w * 16), xl, lane, H, acc);
#endif
        if (w < 3) ts::gemm_head<NS, 16, 2>(chunk(kF5 + 2 * w * 16), lane, H);
        // F (the t2 fragments) is rewritten by this epilogue: every wave has finished reading X(t2) only after the barrier below
        auto side_planes = [&](const KMat& m, const int ks) __attribute__((always_inline)) {       // 4 x NS (tile, column tile, plane) units over 16 k-steps
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
                if (w >= 3) ts::save_tile_k<NS, NS>(sv.T1, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], F[t][p][0], F[t][p][1]);
#else
                if (save) ts::save_tile_k<NS, NS>(sv.T1, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], F[t][p][0], F[t][p][1]);
#endif
            }
        TSP_STAMP(2);
        // ---------------- everybody has its t1 fragments: publish them
        ts::barrier_lds();
        x_store_all();
        ts::barrier_lds();
        TSP_STAMP(3);
        const int it_next = it + (int)gridDim.x;
        const bool has_next = it_next < items;
        // the next item's points (its tile only: the coordinate features do not depend on the net)
        int64_t pcn[2] = {0, 0};
        if (has_next) {
            const int netn = it_next / tiles64;
            const int64_t tile0n = (int64_t)(it_next - netn * tiles64) * 2;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int64_t pt = (tile0n + p) * 32 + j;
                pcn[p] = pt < a.n ? pt : (a.n - 1);
            }
        }
        auto wait_flags = [&]() __attribute__((always_inline)) {      // until waves 0..2 have all left the loop that reads X(t1)
#if TS_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
            while (__atomic_load_n(&flags[0], __ATOMIC_RELAXED) != seq || __atomic_load_n(&flags[1], __ATOMIC_RELAXED) != seq ||
                   __atomic_load_n(&flags[2], __ATOMIC_RELAXED) != seq)
                __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#if TS_PRIO == 2
            __builtin_amdgcn_s_setprio(1);
#endif
        };
        if (w < 3) {
            // ---------------- gpe = w1^T t1 (6 tiles: waves 0..2; both tiles of wave w belong to coordinate c = w), contracted with d(pe)/d(xi)
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc[t][0] = (f32x16)0.f; acc[t][1] = (f32x16)0.f; }
#if TS_DEFER_SAVES
            {
                auto side = [&](const int ks) __attribute__((always_inline)) { side_planes(sv.T1, ks); };
                ts::gemm<NS, 16, 2, false>(chunk(kF5 + 2 * w * 16), xl, lane, H, acc, side);
            }
#else
            ts::gemm<NS, 16, 2>(chunk(kF5 + 2 * w * 16), xl, lane, H, acc);
#endif

            TSP_STAMP(4);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_store_n(&flags[w], seq, __ATOMIC_RELAXED);      // this wave's last read of X(t1) is behind it
            {
                const int c = w;
                float jc[2] = {0.f, 0.f};
                const float xic[2] = {ts::load_xi(a, c, pc[0]), ts::load_xi(a, c, pc[1])};
                const f32x4 frc[4] = {ts::load_fr4(a.freqs, 4 * h), ts::load_fr4(a.freqs, 8 + 4 * h), ts::load_fr4(a.freqs, 16 + 4 * h), ts::load_fr4(a.freqs, 24 + 4 * h)};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f32x16 at[2] = {acc[0][p], acc[1][p]};
                    ts::jac_contract_x<NS>(jc[p], at, xic[p], frc);
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

            if (has_next && TSP_W3_KSTEPS + w < 12) {   // this wave's share of the next item's features: k-step TSP_W3_KSTEPS + w (coordinate t), both column tiles
                const f32x4 frn = ts::load_fr4(a.freqs, 8 * ((TSP_W3_KSTEPS + w) & 3) + 4 * h);
                const float xin[2] = {ts::load_xi(a, 2, pcn[0]), ts::load_xi(a, 2, pcn[1])};
                wait_flags();
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    Frag<NS> f;
                    ts::pe3_frag_x<NS>(f, xin[p], frn);
                    ts::x_store<NS>(xl, TSP_W3_KSTEPS + w, p, f);
                }
            }
        } else if (has_next) {
            // ---------------- wave 3 (it has no gpe tiles): k-steps 0..8 of the next item's features.  0..5 are built while the others multiply and wait in
            // registers for X to be free (the empty asm statement keeps hipcc from sinking the evaluation behind the wait: it did -- 12 k cycles idle, then
            // all 24 fragments: timeline), 6..8 are stored as they are built
            float xi[3][2];
            f32x4 fr[4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int p = 0; p < 2; ++p) xi[c][p] = ts::load_xi(a, c, pcn[p]);
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) fr[kq] = ts::load_fr4(a.freqs, 8 * kq + 4 * h);
            Frag<NS> NF[6][2];
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    ts::pe3_frag_x<NS>(NF[ks][p], xi[ks >> 2][p], fr[ks & 3]);
#pragma unroll
                    for (int s_ = 0; s_ < NS; ++s_) asm volatile("" : "+v"(NF[ks][p].w[s_]));
                }
            TSP_STAMP(4);
            wait_flags();
            TSP_STAMP(5);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int p = 0; p < 2; ++p) ts::x_store<NS>(xl, ks, p, NF[ks][p]);
#pragma unroll
            for (int ks = 6; ks < TSP_W3_KSTEPS; ++ks)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    Frag<NS> f;
                    ts::pe3_frag_x<NS>(f, xi[ks >> 2][p], fr[ks & 3]);
                    ts::x_store<NS>(xl, ks, p, f);
                }
        }
        TSP_STAMP(6);
        ts::barrier_lds();       // end of the item: X holds the next item's pe3 features, vec / red are free
    }
}
