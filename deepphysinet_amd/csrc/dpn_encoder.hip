// Grid-encoder ("MetaNet") kernels for MI355X: the per-FIELD part of the step that is not a plain GEMM.
//
// Reference behaviour (paths relative to /root/reference/DeepPhysiNet):
//   model/attn.py:50-68     FullAttention: softmax(q k^T / sqrt(E)) v, no mask, no dropout   -> dpn_attn_fwd / dpn_attn_bwd
//   model/transformer_net.py:28-44  x = LN1(x + attn), out = LN2(x + ffn)                    -> dpn_add_ln_fwd / dpn_add_ln_bwd
//   model/embed.py:36-64    circular token conv (as im2col), lead-time SineCosPE, token/pos/time assembly -> dpn_im2col_circ3, dpn_lead_pe,
//                           dpn_embed_assemble; dpn_sum_parts joins split-K slices / per-field gradients in a fixed order
// All kernels take a batch of field samples (attention never crosses fields).
// Sizes of the shipped config (cfg:13-24): L = 287 tokens, d_model = 256, 8 heads x 32.  Everything is exact fp32:
// the GEMM-shaped parts use v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain), reductions are fixed-order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dpn_hip.h"
#include "dpn_layout.h"

using namespace dpn;

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define DEV __device__ __forceinline__

namespace {

constexpr int kD = 256, kH = 8, kE = 32, kLmax = 288, kLp = 289;      // kLp: padded LDS row (conflict-free column walks)

DEV f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// the attention kernels run with kAT threads = kAW waves per workgroup (eight: the score / dS tiles, the softmax rows and the k-pairs of the
// reductions are spread over twice the waves of the first version -- these kernels are latency chains on 72 ... 144 workgroups)
constexpr int kAT = 512, kAW = kAT / 64;
// rows [r0, r0+nr) x 32 head columns of a [L][256] matrix -> LDS [row][33]; rows >= L are zero
// (all 16-byte loads are issued before the first LDS store: one memory round trip, not one per element)
template <int NR>
struct Rows16 { float4 v[(NR * 8 + kAT - 1) / kAT]; };
template <int NR>
DEV void rows16_fetch(Rows16<NR>& R, const float* src, int64_t row_stride, int64_t col0, int r0, int L) {
    constexpr int N = (NR * 8 + kAT - 1) / kAT;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int e = threadIdx.x + kAT * q, r = e >> 3, c4 = e & 7;
        R.v[q] = (r < NR && r0 + r < L) ? *reinterpret_cast<const float4*>(src + (int64_t)(r0 + r) * row_stride + col0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int NR>
DEV void rows16_store(float (*dst)[33], const Rows16<NR>& R) {
    constexpr int N = (NR * 8 + kAT - 1) / kAT;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int e = threadIdx.x + kAT * q, r = e >> 3, c4 = e & 7;
        if (r < NR) { dst[r][c4 * 4] = R.v[q].x; dst[r][c4 * 4 + 1] = R.v[q].y; dst[r][c4 * 4 + 2] = R.v[q].z; dst[r][c4 * 4 + 3] = R.v[q].w; }
    }
}
template <int NR>
DEV void load_rows16(float (*dst)[33], const float* src, int64_t row_stride, int64_t col0, int r0, int L) {
    Rows16<NR> R;
    rows16_fetch<NR>(R, src, row_stride, col0, r0, L);
    rows16_store<NR>(dst, R);
}
// D[r] = sum_e gO[r][e] * O[r][e] over the 32 columns of a head, rows r0 .. r0 + nrows: 8 lanes per row, one float4 of each operand per
// lane (all loads in flight at once), joined by three shuffles
DEV void head_rowdot(float* dst, const float* go, const float* o, int head, int r0, int nrows, int L) {
    for (int base = 0; base < nrows; base += kAT / 8) {
        const int r = base + (threadIdx.x >> 3), c4 = threadIdx.x & 7;
        float d = 0.f;
        if (r < nrows && r0 + r < L) {
            const float4 g4 = *reinterpret_cast<const float4*>(go + (int64_t)(r0 + r) * kD + head * kE + c4 * 4);
            const float4 o4 = *reinterpret_cast<const float4*>(o + (int64_t)(r0 + r) * kD + head * kE + c4 * 4);
            d = fmaf(g4.x, o4.x, fmaf(g4.y, o4.y, fmaf(g4.z, o4.z, g4.w * o4.w)));
        }
        d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
        if (c4 == 0 && r < nrows) dst[r] = d;
    }
}
DEV void load_head_rows(float (*dst)[33], const float* src, int head, int r0, int nr, int L) {
    if (nr == kLmax) load_rows16<kLmax>(dst, src, kD, head * kE, r0, L);
    else load_rows16<32>(dst, src, kD, head * kE, r0, L);
}

// D[32 x 32] += A[32 x K] * B[K x 32] with A(i,k) = fa(i,k), B(k,j) = fb(k,j); this wave takes the k-pairs u = wave, wave+4, ...
template <class FA, class FB>
DEV void mma_tile(f32x16& acc, int npairs, FA fa, FB fb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    for (int u = wave; u < npairs; u += kAW) acc = mfma_f32(fa(i, 2 * u + h), fb(2 * u + h, i), acc);
}
// sum the four waves' partial 32x32 tiles in a fixed order; result(r, c) handed to `sink`
template <class SINK>
DEV void reduce_tile(const f32x16& acc, float (*part)[32 * 33], SINK sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][drow32(r, h) * 33 + i] = acc[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += kAT) {
        const int r = e >> 5, c = e & 31, o = r * 33 + c;
        float v = part[0][o];
#pragma unroll
        for (int w_ = 1; w_ < kAW; ++w_) v += part[w_][o];               // fixed order (left fold over the waves)
        sink(r, c, v);
    }
}

struct AttnArgs {
    const float *q, *k, *v, *o, *go;
    float *out, *P, *dS, *Dv, *dq, *dk, *dv;
    int L;
    float scale;
    int roles;      // dpn_attn_bwd: roles per field in grid.z (3: dQ | dK | dV; 2: dQ | dK + dV)
};
// arguments of field b of a batch: rows [b*L, (b+1)*L) of every [batch*L][256] tensor, P block b of [batch][8][288][288]
DEV AttnArgs attn_field(AttnArgs a, const int b) {
    const int64_t ro = (int64_t)b * a.L * kD, po = (int64_t)b * kH * kLmax * kLmax;
    if (a.q) a.q += ro;
    if (a.k) a.k += ro;
    if (a.v) a.v += ro;
    if (a.o) a.o += ro;
    if (a.go) a.go += ro;
    if (a.out) a.out += ro;
    if (a.dq) a.dq += ro;
    if (a.dk) a.dk += ro;
    if (a.dv) a.dv += ro;
    if (a.P) a.P += po;
    return a;
}

// ---------------------------------------------------------------------------------------------------- attention forward
// grid (query tiles of 32, heads); one workgroup keeps K and V of its head in LDS, computes its 32 score rows, the row
// softmax, the probabilities (saved for the backward pass) and the 32 x 32 output tile.
__global__ __launch_bounds__(kAT) void dpn_attn_fwd_kernel(AttnArgs a0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AttnArgs a = attn_field(a0, blockIdx.z);
    float (*Ks)[33] = reinterpret_cast<float (*)[33]>(smem);                                  // [288][33]
    float (*Vs)[33] = reinterpret_cast<float (*)[33]>(smem + kLmax * 33 * 4);                 // [288][33]
    float (*Qs)[33] = reinterpret_cast<float (*)[33]>(smem + 2 * kLmax * 33 * 4);             // [32][33]
    float (*Ss)[kLp] = reinterpret_cast<float (*)[kLp]>(smem + 2 * kLmax * 33 * 4 + 32 * 33 * 4);   // [32][289]
    float (*part)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(smem + 2 * kLmax * 33 * 4 + 32 * 33 * 4 + 32 * kLp * 4);   // [kAW][1056]
    const int head = blockIdx.y, q0 = blockIdx.x * 32, L = a.L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    load_head_rows(Ks, a.k, head, 0, kLmax, L);
    load_head_rows(Vs, a.v, head, 0, kLmax, L);
    load_head_rows(Qs, a.q, head, q0, 32, L);
    __syncthreads();
#if defined(DPN_ATTN_STAGE) && DPN_ATTN_STAGE == 1
    return;
#endif
    // scores: column tile ct belongs to wave ct % 4 (whole K = 32 per tile, no cross-wave reduction)
    for (int ct = wave; ct < kLmax / 32; ct += kAW) {
        f32x16 acc = (f32x16)0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = mfma_f32(Qs[i][2 * u + h], Ks[ct * 32 + i][2 * u + h], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = ct * 32 + i;
            Ss[drow32(r, h)][col] = (col < L) ? acc[r] * a.scale : -INFINITY;
        }
    }
    __syncthreads();
#if defined(DPN_ATTN_STAGE) && DPN_ATTN_STAGE == 2
    return;
#endif
    {   // softmax: one wave per 8 rows, the 64 lanes stride the 288 columns (LDS row stride 289: consecutive lanes hit consecutive
        // banks; the probabilities leave as coalesced 256-byte stores); row max / sum by wave shuffles
        // the 8 rows of a wave are independent chains (LDS read -> shuffle max -> exp -> shuffle sum): all eight run interleaved
        constexpr int RPW = 32 / kAW;                                    // rows per wave
        float e[RPW][5], mx[RPW], sm[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            mx[rr] = -INFINITY;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int c = lane + 64 * q;
                e[rr][q] = (c < kLmax) ? Ss[wave * RPW + rr][c] : -INFINITY;
                mx[rr] = fmaxf(mx[rr], e[rr][q]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) mx[rr] = fmaxf(mx[rr], __shfl_xor(mx[rr], o));
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            sm[rr] = 0.f;
#pragma unroll
            for (int q = 0; q < 5; ++q) { e[rr][q] = __expf(e[rr][q] - mx[rr]); sm[rr] += e[rr][q]; }    // v_exp_f32 (1e-6 rel.); exp(-inf) = 0 for the padding columns
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) sm[rr] += __shfl_xor(sm[rr], o);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = wave * RPW + rr;
            const float inv = 1.f / sm[rr];
            const bool rok = q0 + row < L;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int c = lane + 64 * q;
                if (c < kLmax) {
                    const float p = rok ? e[rr][q] * inv : 0.f;
                    Ss[row][c] = p;
                    if (rok) a.P[((int64_t)head * kLmax + q0 + row) * kLmax + c] = p;
                }
            }
        }
    }
    __syncthreads();
#if defined(DPN_ATTN_STAGE) && DPN_ATTN_STAGE == 3
    return;
#endif
    f32x16 acc = (f32x16)0.f;
    mma_tile(acc, kLmax / 2, [&](int r, int k) { return Ss[r][k]; }, [&](int k, int c) { return Vs[k][c]; });
    reduce_tile(acc, part, [&](int r, int c, float v) {
        if (q0 + r < L) a.out[(int64_t)(q0 + r) * kD + head * kE + c] = v;
    });
}

// ---------------------------------------------------------------------------------------------------- attention backward
// ONE launch, grid (tiles of 32, heads, 2 roles); both roles are independent of each other (the key/value role recomputes its
// columns of dS instead of waiting for the query role to publish them), so the backward of the attention is one graph node.
// role 0, per (query tile, head): D = rowsum(gO * O), dP = gO V^T, dS = P * (dP - D) * scale, dQ = dS K.
DEV void attn_bwd_query_role(const AttnArgs& a, char* smem) {
    float (*Ks)[33] = reinterpret_cast<float (*)[33]>(smem);
    float (*Vs)[33] = reinterpret_cast<float (*)[33]>(smem + kLmax * 33 * 4);
    float (*Gs)[33] = reinterpret_cast<float (*)[33]>(smem + 2 * kLmax * 33 * 4);             // gO tile [32][33]
    float (*Ss)[kLp] = reinterpret_cast<float (*)[kLp]>(smem + 2 * kLmax * 33 * 4 + 32 * 33 * 4);
    float (*part)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(smem + 2 * kLmax * 33 * 4 + 32 * 33 * 4 + 32 * kLp * 4);
    __shared__ float Drow[32];
    const int head = blockIdx.y, q0 = blockIdx.x * 32, L = a.L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    load_head_rows(Ks, a.k, head, 0, kLmax, L);
    load_head_rows(Vs, a.v, head, 0, kLmax, L);
    load_head_rows(Gs, a.go, head, q0, 32, L);
    {   // P rows of this query tile: 32 x 288 floats = 2304 float4, nine per thread, all in flight at once
        constexpr int NP = (32 * 72 + kAT - 1) / kAT;
        float4 pv[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int e = threadIdx.x + kAT * q, r = e / 72, c4 = e % 72;
            pv[q] = (e < 32 * 72 && q0 + r < L) ? *reinterpret_cast<const float4*>(a.P + ((int64_t)head * kLmax + q0 + r) * kLmax + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int e = threadIdx.x + kAT * q, r = e / 72, c4 = e % 72;
            if (e < 32 * 72) { Ss[r][c4 * 4] = pv[q].x; Ss[r][c4 * 4 + 1] = pv[q].y; Ss[r][c4 * 4 + 2] = pv[q].z; Ss[r][c4 * 4 + 3] = pv[q].w; }
        }
    }
    head_rowdot(Drow, a.go, a.o, head, q0, 32, L);
    __syncthreads();
    for (int ct = wave; ct < kLmax / 32; ct += kAW) {
        f32x16 acc = (f32x16)0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = mfma_f32(Gs[i][2 * u + h], Vs[ct * 32 + i][2 * u + h], acc);      // dP tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = drow32(r, h), col = ct * 32 + i;
            Ss[row][col] = Ss[row][col] * (acc[r] - Drow[row]) * a.scale;
        }
    }
    __syncthreads();
    f32x16 acc = (f32x16)0.f;
    mma_tile(acc, kLmax / 2, [&](int r, int k) { return Ss[r][k]; }, [&](int k, int c) { return Ks[k][c]; });
    reduce_tile(acc, part, [&](int r, int c, float v) {
        if (q0 + r < L) a.dq[(int64_t)(q0 + r) * kD + head * kE + c] = v;
    });
}

// role 1, per (key tile, head): dV = P^T gO; then its own 288 x 32 block of dS (same expression, same order of operations as role 0);
// dK = dS^T Q.  Reductions over all queries.
template <int WHAT>      // 0: dV and dK (rounds 3-5), 1: dV only, 2: dK only (round 6: the key role was the launch's long pole, 11.4 against 8.1 us for the query role)
DEV void attn_bwd_key_role(const AttnArgs& a, char* smem) {
    float (*Rs)[33] = reinterpret_cast<float (*)[33]>(smem);                                  // [288][33]: all rows of gO, later of Q
    float (*Cs)[33] = reinterpret_cast<float (*)[33]>(smem + kLmax * 33 * 4);                 // [288][33]: P[:, key tile], later dS[:, key tile]
    float (*Vt)[33] = reinterpret_cast<float (*)[33]>(smem + 2 * kLmax * 33 * 4);             // [32][33]: V rows of the key tile
    float* Dl = reinterpret_cast<float*>(smem + 2 * kLmax * 33 * 4 + 32 * 33 * 4);            // [288]
    float (*part)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(smem + 2 * kLmax * 33 * 4 + 32 * 33 * 4 + kLmax * 4);
    const int head = blockIdx.y, j0 = blockIdx.x * 32, L = a.L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    load_head_rows(Rs, a.go, head, 0, kLmax, L);
    load_rows16<kLmax>(Cs, a.P + (int64_t)head * kLmax * kLmax, kLmax, j0, 0, L);
    if constexpr (WHAT != 1) {
        load_head_rows(Vt, a.v, head, j0, 32, L);
        head_rowdot(Dl, a.go, a.o, head, 0, kLmax, L);
    }
    __syncthreads();
    Rows16<kLmax> qreg;                                                                      // Q is needed last: its loads fly under dV and dS
    if constexpr (WHAT != 1) rows16_fetch<kLmax>(qreg, a.q, kD, head * kE, 0, L);
    f32x16 acc = (f32x16)0.f;
    if constexpr (WHAT != 2) {
        mma_tile(acc, kLmax / 2, [&](int r, int k) { return Cs[k][r]; }, [&](int k, int c) { return Rs[k][c]; });       // dV[j][e] = sum_i P[i][j] gO[i][e]
        reduce_tile(acc, part, [&](int r, int c, float v) {
            if (j0 + r < L) a.dv[(int64_t)(j0 + r) * kD + head * kE + c] = v;
        });
        if constexpr (WHAT == 1) return;
        __syncthreads();
    }
    for (int rt = wave; rt < kLmax / 32; rt += kAW) {                                            // dS[rows of tile rt][key tile]
        f32x16 dp = (f32x16)0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) dp = mfma_f32(Rs[rt * 32 + i][2 * u + h], Vt[i][2 * u + h], dp);              // dP[row][j] = sum_e gO[row][e] V[j][e]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + drow32(r, h);
            Cs[row][i] = Cs[row][i] * (dp[r] - Dl[row]) * a.scale;
        }
    }
    __syncthreads();
    rows16_store<kLmax>(Rs, qreg);
    __syncthreads();
    acc = (f32x16)0.f;
    mma_tile(acc, kLmax / 2, [&](int r, int k) { return Cs[k][r]; }, [&](int k, int c) { return Rs[k][c]; });       // dK[j][e] = sum_i dS[i][j] Q[i][e]
    reduce_tile(acc, part, [&](int r, int c, float v) {
        if (j0 + r < L) a.dk[(int64_t)(j0 + r) * kD + head * kE + c] = v;
    });
}

__global__ __launch_bounds__(kAT) void dpn_attn_bwd_kernel(AttnArgs a0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // grid.z = 3 * field + role (round 6: the key role's dV and dK are two roles; DPN_ATTN_BWD_ROLES=2 launches rounds 3-5's two: a0.roles)
    const int roles = a0.roles, role = blockIdx.z % roles;
    const AttnArgs a = attn_field(a0, blockIdx.z / roles);
    // (ablation builds, wrong results on purpose: tools/variant_build.py --unit=2 -DATTN_ABL_NOQ | -DATTN_ABL_NOK time one role alone)
    if (role == 0) {
#ifndef ATTN_ABL_NOQ
        attn_bwd_query_role(a, smem);
#endif
    } else {
#ifndef ATTN_ABL_NOK
        if (roles == 2) attn_bwd_key_role<0>(a, smem);
        else if (role == 1) attn_bwd_key_role<2>(a, smem);     // (the longer one first)
        else attn_bwd_key_role<1>(a, smem);
#endif
    }
}

// ---------------------------------------------------------------------------------------------------- add + LayerNorm
// out = LN(x + r) * gamma + beta over 256 columns (eps 1e-5, biased variance: nn.LayerNorm); one wave per row.
struct LnArgs {
    const float *x, *r, *gamma, *beta, *g, *xhat, *rstd;
    float *out, *xhat_out, *rstd_out, *gx, *dgamma, *dbeta;
    int rows;
};
DEV float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void dpn_add_ln_fwd_kernel(LnArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.rows) return;
    const float4 xv = reinterpret_cast<const float4*>(a.x + (int64_t)row * kD)[lane];
    float v[4] = {xv.x, xv.y, xv.z, xv.w};
    if (a.r) {
        const float4 rv = reinterpret_cast<const float4*>(a.r + (int64_t)row * kD)[lane];
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
    }
    const float mean = wave_sum64((v[0] + v[1]) + (v[2] + v[3])) * (1.f / kD);
    float d[4], sq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { d[c] = v[c] - mean; sq = fmaf(d[c], d[c], sq); }
    const float rstd = rsqrtf(wave_sum64(sq) * (1.f / kD) + 1e-5f);
    const float4 gm = reinterpret_cast<const float4*>(a.gamma)[lane], bt = reinterpret_cast<const float4*>(a.beta)[lane];
    float4 xh, o;
    xh.x = d[0] * rstd; xh.y = d[1] * rstd; xh.z = d[2] * rstd; xh.w = d[3] * rstd;
    o.x = fmaf(xh.x, gm.x, bt.x); o.y = fmaf(xh.y, gm.y, bt.y); o.z = fmaf(xh.z, gm.z, bt.z); o.w = fmaf(xh.w, gm.w, bt.w);
    reinterpret_cast<float4*>(a.out + (int64_t)row * kD)[lane] = o;
    if (a.xhat_out) reinterpret_cast<float4*>(a.xhat_out + (int64_t)row * kD)[lane] = xh;
    if (a.rstd_out && lane == 0) a.rstd_out[row] = rstd;
}
// gx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)); each block (4 rows) also leaves its partial column sums of
// g*xhat and g in `partial` [blocks][2][256]; dpn_ln_colsum_kernel adds the blocks up in a fixed order -> dgamma, dbeta.
__global__ __launch_bounds__(256) void dpn_add_ln_bwd_kernel(LnArgs a, float* partial) {
    __shared__ float4 pg[4][64], pb[4][64];
    const int w = threadIdx.x >> 6, row = blockIdx.x * 4 + w, lane = threadIdx.x & 63;
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    if (row < a.rows) {
        const float4 gv = reinterpret_cast<const float4*>(a.g + (int64_t)row * kD)[lane];
        const float4 xh = reinterpret_cast<const float4*>(a.xhat + (int64_t)row * kD)[lane];
        const float4 gm = reinterpret_cast<const float4*>(a.gamma)[lane];
        const float t[4] = {gv.x * gm.x, gv.y * gm.y, gv.z * gm.z, gv.w * gm.w};
        const float xs[4] = {xh.x, xh.y, xh.z, xh.w};
        const float m1 = wave_sum64((t[0] + t[1]) + (t[2] + t[3])) * (1.f / kD);
        const float m2 = wave_sum64(fmaf(t[0], xs[0], t[1] * xs[1]) + fmaf(t[2], xs[2], t[3] * xs[3])) * (1.f / kD);
        const float rstd = a.rstd[row];
        float4 o;
        o.x = rstd * (t[0] - m1 - xs[0] * m2); o.y = rstd * (t[1] - m1 - xs[1] * m2);
        o.z = rstd * (t[2] - m1 - xs[2] * m2); o.w = rstd * (t[3] - m1 - xs[3] * m2);
        reinterpret_cast<float4*>(a.gx + (int64_t)row * kD)[lane] = o;
        dg = make_float4(gv.x * xh.x, gv.y * xh.y, gv.z * xh.z, gv.w * xh.w);
        db = gv;
    }
    pg[w][lane] = dg; pb[w][lane] = db;
    __syncthreads();
    if (w == 0) {
        float4 s1 = pg[0][lane], s2 = pb[0][lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            s1.x += pg[k][lane].x; s1.y += pg[k][lane].y; s1.z += pg[k][lane].z; s1.w += pg[k][lane].w;
            s2.x += pb[k][lane].x; s2.y += pb[k][lane].y; s2.z += pb[k][lane].z; s2.w += pb[k][lane].w;
        }
        reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * 512)[lane] = s1;
        reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * 512 + 256)[lane] = s2;
    }
}
__global__ __launch_bounds__(256) void dpn_ln_colsum_kernel(const float* partial, int nblocks, float* dgamma, float* dbeta) {
    const int c = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int b = 0; b < nblocks; ++b) { s1 += partial[(int64_t)b * 512 + c]; s2 += partial[(int64_t)b * 512 + 256 + c]; }
    dgamma[c] = s1;
    dbeta[c] = s2;
}


// ---------------------------------------------------------------- data embedding pieces (model/embed.py:36-64)
// SineCosPE of a scalar per field (include_input=False): out[b][2f] = sin(h[b] * freq[f]), out[b][2f+1] = cos(h[b] * freq[f])
// (position_encoding.py:35-50), for one or two frequency tables.
__global__ void dpn_lead_pe_kernel(const float* h, int batch, const float* fa, int na, float* oa, const float* fb, int nb, float* ob) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * (na + nb)) return;
    const int b = i / (na + nb), j = i - b * (na + nb);
    const float hv = h[b];
    if (j < na) { const float s = hv * fa[j]; oa[(int64_t)b * 2 * na + 2 * j] = sinf(s); oa[(int64_t)b * 2 * na + 2 * j + 1] = cosf(s); }
    else { const int jj = j - na; const float s = hv * fb[jj]; ob[(int64_t)b * 2 * nb + 2 * jj] = sinf(s); ob[(int64_t)b * 2 * nb + 2 * jj + 1] = cosf(s); }
}
// im2col of the circular k=3 convolution along the token axis of every field: out[b*T + t][c*3 + tap] = x[b*T + (t + tap - 1) mod T][c],
// so that the conv is out . W^T with the Conv1d weight [d_model][C][3] read in place as [d_model][3C] (no permuted copy, and the weight
// gradient of that GEMM is already in the parameter's layout).
__global__ __launch_bounds__(256) void dpn_im2col_circ3_kernel(const float* x, int T, int C, int64_t total, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t tg = i / (3 * C);
    const int r = (int)(i - tg * 3 * C), c = r / 3, tap = r - 3 * c;
    const int64_t b = tg / T;
    int ts = (int)(tg - b * T) + tap - 1;
    ts = ts < 0 ? ts + T : (ts >= T ? ts - T : ts);
    out[i] = x[(b * T + ts) * C + c];
}
// x0 = cat(learnable_token, value_embedding) + positional table + lead-time embedding (embed.py:60-64, transformer_net.py:124-126) for every
// field of the batch.  The value embedding arrives as n_parts split-K partial products [n_parts][batch*n_emb][256] (added here in a fixed
// order) plus the conv bias; te is [batch][256].
__global__ __launch_bounds__(256) void dpn_embed_assemble_kernel(const float* token, int n_tok, const float* emb_parts, int n_parts, int n_emb,
                                                                  int batch, const float* bias, const float* pos, const float* te, float* out) {
    const int L = n_tok + n_emb, b = blockIdx.x / L, row = blockIdx.x - b * L, c = threadIdx.x;
    float v;
    if (row < n_tok) v = token[(int64_t)row * kD + c];
    else {
        const int64_t o = ((int64_t)b * n_emb + row - n_tok) * kD + c;
        v = emb_parts[o];
        for (int p = 1; p < n_parts; ++p) v += emb_parts[(int64_t)p * batch * n_emb * kD + o];
        v += bias ? bias[c] : 0.f;
    }
    out[(int64_t)blockIdx.x * kD + c] = (v + pos[(int64_t)row * kD + c]) + te[(int64_t)b * kD + c];
}

__global__ __launch_bounds__(256) void dpn_sum_parts_kernel(const float* parts, int n_parts, int64_t count, int64_t zero_tail, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) {
        float v = parts[i];
        for (int p = 1; p < n_parts; ++p) v += parts[(int64_t)p * count + i];
        out[i] = v;
    } else if (i < count + zero_tail) out[i] = 0.f;
}

constexpr int kAttnLds = 2 * kLmax * 33 * 4 + 32 * 33 * 4 + 32 * kLp * 4 + kAW * 32 * 33 * 4;     // 151,040 B with eight waves
static_assert(2 * kLmax * 33 * 4 + 32 * 33 * 4 + kLmax * 4 + kAW * 32 * 33 * 4 <= kAttnLds, "the key role fits in the query role's LDS");
static_assert(kAttnLds <= 160 * 1024, "one workgroup per CU");

}  // namespace

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: the "already set" mark is a bit per device (atomic: two host threads may launch
// first at the same time; setting the attribute twice is harmless, skipping it on a second GPU is a launch failure)
#include <atomic>
static inline bool dpn_first_use_on_device(std::atomic<unsigned long long>& done, unsigned long long& bit) {
    int d = 0;
    (void)hipGetDevice(&d);
    bit = 1ull << (d & 63);
    return (done.load(std::memory_order_acquire) & bit) == 0;
}

extern "C" {

int dpn_attn_fwd(const float* q, const float* k, const float* v, int L, int batch, float* out, float* P, void* stream) {
    if (!q || !k || !v || !out || !P || L <= 0 || L > kLmax || batch <= 0 || batch > 32767) return -1;
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.P = P; a.L = L; a.scale = 1.0f / sqrtf((float)kE);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static std::atomic<unsigned long long> done{0};
    unsigned long long bit;
    if (dpn_first_use_on_device(done, bit)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dpn_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kAttnLds); done.fetch_or(bit, std::memory_order_release); }
    hipLaunchKernelGGL(dpn_attn_fwd_kernel, dim3((L + 31) / 32, kH, batch), dim3(kAT), kAttnLds, s, a);
    return (int)hipGetLastError();
}

int dpn_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* P, const float* go, int L, int batch,
                 float* dq, float* dk, float* dv, void* stream) {
    if (!q || !k || !v || !o || !P || !go || !dq || !dk || !dv || L <= 0 || L > kLmax || batch <= 0 || batch > 16383) return -1;      // (grid.z = 3 * batch <= 65535)
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.o = o; a.go = go; a.P = const_cast<float*>(P); a.dq = dq; a.dk = dk; a.dv = dv; a.L = L;
    a.scale = 1.0f / sqrtf((float)kE);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static std::atomic<unsigned long long> done{0};
    unsigned long long bit;
    if (dpn_first_use_on_device(done, bit)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dpn_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kAttnLds); done.fetch_or(bit, std::memory_order_release); }
    // Round 6: three roles per field (dQ | dK | dV) while the launch does not fill the chip -- one field: 216 workgroups, 9.2 against 11.0 us per launch, the step
    // -7.7 us (profiles/round6_attn_bwd_roles.txt); lead batches keep two (61 fields: 56.4 against 56.5 ms).  DPN_ATTN_BWD_ROLES=2|3 overrides.
    const char* er = getenv("DPN_ATTN_BWD_ROLES");
    a.roles = er ? ((er[0] == '2') ? 2 : 3) : (batch <= 2 ? 3 : 2);
    hipLaunchKernelGGL(dpn_attn_bwd_kernel, dim3((L + 31) / 32, kH, a.roles * batch), dim3(kAT), kAttnLds, s, a);
    return (int)hipGetLastError();
}

int dpn_add_ln_fwd(const float* x, const float* r, const float* gamma, const float* beta, int rows, float* out, float* xhat, float* rstd,
                   void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0) return -1;
    LnArgs a{};
    a.x = x; a.r = r; a.gamma = gamma; a.beta = beta; a.out = out; a.xhat_out = xhat; a.rstd_out = rstd; a.rows = rows;
    hipLaunchKernelGGL(dpn_add_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

int dpn_add_ln_bwd(const float* g, const float* xhat, const float* rstd, const float* gamma, int rows, float* gx, float* dgamma, float* dbeta,
                   float* scratch, void* stream) {
    if (!g || !xhat || !rstd || !gamma || !gx || !scratch || rows <= 0 || ((dgamma == nullptr) != (dbeta == nullptr))) return -1;
    LnArgs a{};
    a.g = g; a.xhat = xhat; a.rstd = rstd; a.gamma = gamma; a.gx = gx; a.dgamma = dgamma; a.dbeta = dbeta; a.rows = rows;
    const int nb = (rows + 3) / 4;
    hipLaunchKernelGGL(dpn_add_ln_bwd_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, scratch);
    if (dgamma)   // NULL: the caller reduces `scratch` itself (dpn_sgemm_batch_jobs)
        hipLaunchKernelGGL(dpn_ln_colsum_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const float*)scratch, nb, dgamma, dbeta);
    return (int)hipGetLastError();
}

int dpn_sum_parts(const float* parts, int n_parts, int64_t count, int64_t zero_tail, float* out, void* stream) {
    if (!parts || !out || n_parts <= 0 || count <= 0 || zero_tail < 0) return -1;
    hipLaunchKernelGGL(dpn_sum_parts_kernel, dim3((unsigned)((count + zero_tail + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), parts,
                       n_parts, count, zero_tail, out);
    return (int)hipGetLastError();
}

int dpn_lead_pe(const float* h_dev, int batch, const float* freqs_a, int n_a, float* out_a, const float* freqs_b, int n_b, float* out_b, void* stream) {
    if (!h_dev || batch <= 0 || !freqs_a || !out_a || n_a <= 0 || n_b < 0 || (n_b > 0 && (!freqs_b || !out_b))) return -1;
    hipLaunchKernelGGL(dpn_lead_pe_kernel, dim3((batch * (n_a + n_b) + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), h_dev, batch,
                       freqs_a, n_a, out_a, freqs_b, n_b, out_b);
    return (int)hipGetLastError();
}

int dpn_im2col_circ3(const float* x, int T, int C, int batch, float* out, void* stream) {
    if (!x || !out || T <= 0 || C <= 0 || batch <= 0) return -1;
    const int64_t total = (int64_t)batch * T * C * 3;
    hipLaunchKernelGGL(dpn_im2col_circ3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, T, C, total, out);
    return (int)hipGetLastError();
}

int dpn_embed_assemble(const float* token, int n_tok, const float* emb_parts, int n_parts, int n_emb, int batch, const float* bias, const float* pos,
                       const float* te, float* out, void* stream) {
    if (!token || !emb_parts || !pos || !te || !out || n_tok < 0 || n_emb <= 0 || n_parts <= 0 || batch <= 0) return -1;
    hipLaunchKernelGGL(dpn_embed_assemble_kernel, dim3((unsigned)(batch * (n_tok + n_emb))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), token,
                       n_tok, emb_parts, n_parts, n_emb, batch, bias, pos, te, out);
    return (int)hipGetLastError();
}

}  // extern "C"
