// Forward + Jacobian kernel of the parity-grade (hi+lo) mode, PING-PONG form (round 6).  Included by dpn_kernels.hip (point unit) behind
// dpn_fwd_tiles.h, whose building blocks (ts::gemm, ts::gemm_head, the feature / save helpers) it reuses unchanged.
//
// Same arithmetic, same packed stream, same accumulation order per output tile as dpn_fwd_tiles_kernel: fields, Jacobian and saved state are
// BIT-IDENTICAL to it (tests/test_gpu_parity.py::test_ping_pong_forward_is_bitwise_the_tile_split_forward).  What changes is who runs beside whom:
//
//   dpn_fwd_tiles_kernel  two independent 4-wave workgroups per CU.  The two waves of a SIMD belong to different workgroups and meet in
//                         whatever phases their workgroups happen to be in: both multiplying (they share the SIMD's matrix pipe), both in
//                         an epilogue / feature / barrier phase (the pipe idles), or -- by chance -- complementary.  Timeline of round 5: a
//                         wave issues MFMAs for 27.6 k of its 90.6 k cycles; two of them keep the pipe 61 % busy.
//   dpn_fwd_pp_kernel     ONE 8-wave workgroup per CU = two groups of four waves (group g = wave >> 2; waves w and w + 4 share SIMD w), each
//                         group with its own 64 points, its own X image, vectors and field sums in LDS (2 x 71 KB of the 160 KB).  A group's
//                         work on an item is cut into ten intervals that ALTERNATE between "multiply" (one GEMM loop) and "service"
//                         (epilogue / features / packing / LDS stores):
//                               E0 M1 E1 MA P6 MB E2 My Ey Mg          (E0 = Jacobian contraction of the previous item + pe3 features of this one)
//                         and group 1 runs ONE interval behind group 0, held there by a workgroup barrier at every interval boundary.  So on
//                         every SIMD one wave multiplies while its partner does VALU / LDS work, by construction and for the whole kernel; a
//                         group's own two barriers per layer ("everybody has read X" / "X is stored") ARE those joint barriers: when a group
//                         enters a service interval all its waves have left the multiply loop.
//                         Workgroups are PERSISTENT (grid = one per CU, items = (net, 128-point pair) strided over the grid): the prologue of an
//                         item (features) and the epilogue of the previous one (Jacobian contraction) are one more service interval under the
//                         partner's last multiply instead of a ramp with an idle pipe at both ends of every workgroup.
//
// Interval pairing (group 0 | group 1): E0|Mg  M1|E0  E1|M1  MA|E1  P6|MA  MB|P6  E2|MB  My|E2  Ey|My  Mg|Ey.
#pragma once

namespace pp {
template <int NS>
struct Cfg {
    static constexpr int kGroupBytes = ((ts::Cfg<NS>::kLdsBytes + 255) / 256) * 256;
    static constexpr int kLdsBytes = 2 * kGroupBytes;
};
}  // namespace pp

// Experiment build (-DDPN_TIMELINE -DPP_TIMELINE, tools/pp_timeline.py): lane 0 of every wave writes the shader clock at the interval boundaries of the
// workgroup's SECOND item to a.timeline[workgroup][wave][stamp] (32 stamps per wave)
#if defined(PP_TIMELINE) && defined(DPN_TIMELINE)
#define PP_STAMP_AT(I, ITEM) do { if (a.timeline && lane == 0 && n_done == (ITEM)) a.timeline[((size_t)blockIdx.x * 8 + wave) * 32 + (I)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP_AT(I, ITEM) do { } while (0)
#endif
#define PP_STAMP(I) PP_STAMP_AT(I, 1)

template <int NS>
__global__ __launch_bounds__(512) void dpn_fwd_pp_kernel(FwdArgs a, int n_nets) {
    using C = ts::Cfg<NS>;
    __shared__ __attribute__((aligned(16))) char lds_all[pp::Cfg<NS>::kLdsBytes];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wave >> 2, w = wave & 3;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int tid_g = threadIdx.x & 255;                           // thread index inside the group
    char* lds = lds_all + g * pp::Cfg<NS>::kGroupBytes;
#if TS_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif
    float* vec = reinterpret_cast<float*>(lds + C::kVecOff);
    float* red = reinterpret_cast<float*>(lds + C::kRedOff);
    char* xl = lds + lane * 16;
    const ts::Ident I = ts::make_ident(j, h);
    SavedView sv = saved_view(a.saved, a.n_pad, NS);
    const int64_t tiles32 = a.n_pad / 32;
    const int64_t pairs = a.n_pad / 128;
    const int64_t items = pairs * n_nets;

    f32x16 acc[2][2];
    Frag<NS> F[2][2][2];                     // [tile t][column tile p][k-step of the tile's pair]: the epilogue's output fragments
    ts::Head<NS, 2> H;
    int cur_net = -1;
    int n_done = 0;
    // what the Jacobian contraction of the PREVIOUS item needs (it runs in this item's first service interval)
    int64_t prev_pc[2] = {0, 0}, prev_tile0 = 0;
    int prev_net = 0;

    if (g == 1) ts::barrier_lds();           // group 1 runs one interval behind group 0
    int net = 0;
    int64_t pair = blockIdx.x;
    for (int64_t it = blockIdx.x;; it += gridDim.x, pair += gridDim.x) {
        // ================================================================ E0: Jacobian of the previous item ; vectors + pe3 features of this one
        PP_STAMP(0);
        PP_STAMP_AT(20, 2);
        if (n_done > 0 && w < 3) {
            // gpe = w1^T t1 (6 tiles: waves 0..2; both tiles of wave w belong to coordinate c = w), contracted with d(pe)/d(xi)
            const int c = w;
            float jc[2] = {0.f, 0.f};
            const float xic[2] = {ts::load_xi(a, c, prev_pc[0]), ts::load_xi(a, c, prev_pc[1])};
            const f32x4 frc[4] = {ts::load_fr4(a.freqs, 4 * h), ts::load_fr4(a.freqs, 8 + 4 * h), ts::load_fr4(a.freqs, 16 + 4 * h), ts::load_fr4(a.freqs, 24 + 4 * h)};
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x16 at[2] = {acc[0][p], acc[1][p]};
                ts::jac_contract_x<NS>(jc[p], at, xic[p], frc);
                jc[p] += __shfl_xor(jc[p], 32);
            }
            // lane (j, h) stores point j of column tile h; chain rule through x / dx / (lon - 1), in the reference's backward order
            const float mine = h ? jc[1] : jc[0];
            const int64_t pt = (prev_tile0 + h) * 32 + j;
            if (pt < a.n) {
                const float g1 = (c == 0) ? a.geo.lon_m1 : (c == 1) ? a.geo.lat_m1 : a.geo.pred_t_span;
                const float g2 = (c == 0) ? a.geo.dx : (c == 1) ? a.geo.dy : 1.0f;
                a.jac_n[(pt * 6 + prev_net) * 3 + c] = mine / g1 / g2;
            }
        }
        PP_STAMP(21);
        if (it >= items) break;
        while (pair >= pairs) { pair -= pairs; ++net; }                // (no 64-bit division in the loop: it = net * pairs + pair, advanced by the grid size)
        const int64_t tile0 = (pair * 2 + g) * 2;                       // first of this group's two 32-point column tiles
        const char* pk = a.packed + (long)net * pack_bytes_per_net(NS);
        auto chunk = [&](const int kb) __attribute__((always_inline)) { return pk + (long)kb * 1024 * NS; };
        if (net != cur_net) {   // permuted fp32 vectors of this net -> the group's LDS block (published by the barrier that ends this interval; the group's last
                                // reader of the previous net's vectors was its My interval, four barriers ago)
            const u32x4* gv = reinterpret_cast<const u32x4*>(pk + (long)kPackKB * 1024 * NS);
            const int i0 = tid_g, i1 = tid_g + 256;
            const u32x4 v0 = gv[i0];
            const u32x4 v1 = gv[i1 < ts::kVecFloats / 4 ? i1 : i0];
            reinterpret_cast<u32x4*>(vec)[i0] = v0;
            if (i1 < ts::kVecFloats / 4) reinterpret_cast<u32x4*>(vec)[i1] = v1;
            cur_net = net;
        }
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
        // coordinate features pe3 -> X (k-steps 0..11): this thread builds k-steps 3w .. 3w+2 of both column tiles
        // coordinate features pe3 -> X (k-steps 0..11).  Waves 0..2 also carry the previous item's Jacobian contraction (32 angles per lane) in this
        // interval, wave 3 does not: it builds k-steps 6..11 (48 angles per lane), waves 0..2 build k-steps 2w, 2w+1 (16): 48 angles per lane for everybody
        if (w < 3) {
            const int c = (2 * w) >> 2;
            float xi[2];
            f32x4 fr[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) xi[p] = ts::load_xi(a, c, pc[p]);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fr[kk] = ts::load_fr4(a.freqs, 8 * ((2 * w + kk) & 3) + 4 * h);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    Frag<NS> f;
                    ts::pe3_frag_x<NS>(f, xi[p], fr[kk]);
                    ts::x_store<NS>(xl, 2 * w + kk, p, f);
                }
        } else {
            float xi[2][2];                     // [coordinate 1 | 2][p]
            f32x4 fr[4];
#pragma unroll
            for (int p = 0; p < 2; ++p) { xi[0][p] = ts::load_xi(a, 1, pc[p]); xi[1][p] = ts::load_xi(a, 2, pc[p]); }
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) fr[kq] = ts::load_fr4(a.freqs, 8 * kq + 4 * h);
#pragma unroll
            for (int ks = 6; ks < 12; ++ks)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    Frag<NS> f;
                    ts::pe3_frag_x<NS>(f, xi[(ks >> 2) - 1][p], fr[ks & 3]);
                    ts::x_store<NS>(xl, ks, p, f);
                }
        }
        PP_STAMP(1);
        ts::barrier_lds();
        // ================================================================ M1: pre1 = w1 . pe + b1
        PP_STAMP(2);
        u32 m1w[2] = {0u, 0u};
        init_all(kVecB1, 1.0f);
        ts::gemm<NS, 12, 2>(chunk(kF0 + 2 * w * 12), xl, lane, H, acc);
        PP_STAMP(3);
        ts::barrier_lds();
        // ================================================================ E1: h1 = relu -> X ; relu mask bits -> m1w ; hdot = (w2^T wo) . h1 (this wave's 64 channels)
        PP_STAMP(4);
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
        asm volatile("" : "+v"(hdot[0]), "+v"(hdot[1]));      // finished BEFORE the next layer's first weight fragments are requested (register pressure)
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
        asm volatile("" : "+v"(m1w[0]), "+v"(m1w[1]));        // (opaque mask words: see dpn_fwd_tiles_kernel)
#pragma unroll
        for (int p = 0; p < 2; ++p)          // word w of the lane's uint4 = tiles 2w (low half), 2w+1 (high half)
            reinterpret_cast<u32*>(sv.m1 + ((int64_t)net * tiles32 + tile0 + p) * 64 + lane)[w] = m1w[p];
        x_store_all();
        PP_STAMP(5);
        ts::barrier_lds();
        // ================================================================ MA: pre2 = A h1 + (W1 cvec + bf1) ...
        PP_STAMP(6);
        init_all(kVecC2, 1.0f);
        ts::gemm<NS, 16, 2>(chunk(kFA + 2 * w * 16), xl, lane, H, acc);
        ts::gemm_head<NS, 12, 2>(chunk(kFB + 2 * w * 12), lane, H);
        PP_STAMP(7);
        ts::barrier_lds();
        // ================================================================ P6: data features pe6 (SineCosPE(6,16) of coord_data) -> X ; ddot = (Wd^T wo) . pe6
        PP_STAMP(8);
        float ddot[2] = {0.f, 0.f};
        {
            const float* bv = vec + kVecBv * 256;
            const int j0 = (3 * w) >> 1, j1 = (3 * w + 2) >> 1;          // the (at most two) columns of coord_data behind k-steps 3w .. 3w+2
            float cdv[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p) { cdv[0][p] = a.coord_data[pc[p] * 6 + j0]; cdv[1][p] = a.coord_data[pc[p] * 6 + j1]; }
            const f32x4 fr6[2] = {ts::load_fr4(a.freqs, 32 + 4 * h), ts::load_fr4(a.freqs, 32 + 8 + 4 * h)};
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    Frag<NS> f6;
                    const int ks = 3 * w + kk;
                    ts::pe6_frag_dot_x<NS>(f6, ((ks >> 1) == j0) ? cdv[0][p] : cdv[1][p], (ks & 1) ? fr6[1] : fr6[0], bv + 16 * ks + 8 * h, ddot[p]);
                    ts::x_store<NS>(xl, 3 * w + kk, p, f6);
                }
        }
        PP_STAMP(9);
        ts::barrier_lds();
        // ================================================================ MB: ... + B pe6
        PP_STAMP(10);
        ts::gemm<NS, 12, 2>(chunk(kFB + 2 * w * 12), xl, lane, H, acc);
        ts::gemm_head<NS, 16, 2>(chunk(kFAT + 2 * w * 16), lane, H);
        PP_STAMP(11);
        ts::barrier_lds();
        // ================================================================ E2: out = u . relu(pre2) + 2 wo . c + const ; t2 = m2 (.) u -> X ; M2 mask fragments
        PP_STAMP(12);
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
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {            // this wave's share of the field: its 64 channels and its 3 k-steps of pe6, both halves of the wave
            float o = adot[p] + 2.0f * (hdot[p] + ddot[p]);
            o += __shfl_xor(o, 32);
            if (h == 0) red[w * 64 + p * 32 + j] = o;
        }
        x_store_all();
        PP_STAMP(13);
        ts::barrier_lds();
        // ================================================================ My: the field ; reverse sweep y = A^T t2 + 2 w2^T wo (M2 saved inside)
        PP_STAMP(14);
        if (w == 0) {                            // lane (j, h) finishes point j of column tile h: the four waves' shares in a fixed order
            const int64_t pt = (tile0 + h) * 32 + j;
            if (pt < a.n) {
                const float const0 = vec[kNumVecs * 256];          // wo . bf2 + bo + 2 wo . cvec
                if (vec[kNumVecs * 256 + 1] != 1.0f) __builtin_trap();     // the packed stream is not in the fused five-GEMM form
                const float o = (red[0 * 64 + h * 32 + j] + red[1 * 64 + h * 32 + j]) + (red[2 * 64 + h * 32 + j] + red[3 * 64 + h * 32 + j]);
                a.out_n[pt * 6 + net] = o + const0 + (a.ref ? a.ref : a.coord_data)[pt * 6 + net];           // + ref_data (variable_net.py:86)
            }
        }
        init_all(kVecA2, 2.0f);
        {
            auto side = [&](const int ks) __attribute__((always_inline)) {            // M2: four (tile, column tile) units over the 16 k-steps
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ks == 4 * u + 1) ts::save_plane_k(sv.M2, net, 1, 0, tile0 + (u & 1), 2 * w + (u >> 1), lane, I, zero_rows[u & 1], MK[u >> 1][u & 1][0].w[0], MK[u >> 1][u & 1][1].w[0]);
            };
            ts::gemm<NS, 16, 2, false>(chunk(kFAT + 2 * w * 16), xl, lane, H, acc, side);
        }
        if (w < 3) ts::gemm_head<NS, 16, 2>(chunk(kF5 + 2 * w * 16), lane, H);
        PP_STAMP(15);
        ts::barrier_lds();
        // ================================================================ Ey: t1 = m1 (.) y -> X
        PP_STAMP(16);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const u32 bits = m1w[p] >> (16 * t + r);
                    frag_set2<NS>(F[t][p][r >> 3], (r & 7) >> 1, (bits & 1u) ? acc[t][p][r] : 0.f, (bits & 2u) ? acc[t][p][r + 1] : 0.f);
                }
        x_store_all();
        PP_STAMP(17);
        ts::barrier_lds();
        // ================================================================ Mg: gpe = w1^T t1 (waves 0..2, T1 saved inside) ; wave 3 saves its T1 rows
        PP_STAMP(18);
        if (w < 3) {
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc[t][0] = (f32x16)0.f; acc[t][1] = (f32x16)0.f; }
            auto side = [&](const int ks) __attribute__((always_inline)) {       // 4 x NS (tile, column tile, plane) units over 16 k-steps
#pragma unroll
                for (int u = 0; u < 4 * NS; ++u) {
                    const int tp = u / NS, s_ = u % NS;
                    if (ks == (16 / (4 * NS)) * u + 1)
                        ts::save_plane_k(sv.T1, net, NS, s_, tile0 + (tp & 1), 2 * w + (tp >> 1), lane, I, zero_rows[tp & 1], F[tp >> 1][tp & 1][0].w[s_], F[tp >> 1][tp & 1][1].w[s_]);
                }
            };
            ts::gemm<NS, 16, 2, false>(chunk(kF5 + 2 * w * 16), xl, lane, H, acc, side);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) ts::save_tile_k<NS, NS>(sv.T1, net, tile0 + p, 2 * w + t, lane, I, zero_rows[p], F[t][p][0], F[t][p][1]);
        }
        PP_STAMP(19);
        ts::barrier_lds();
        prev_pc[0] = pc[0]; prev_pc[1] = pc[1]; prev_tile0 = tile0; prev_net = net;
        ++n_done;
    }
}
