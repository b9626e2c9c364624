// Layout algebra shared by the pack kernel, the point kernels and the host-side unit tests.
//
// Everything the point kernels multiply goes through v_mfma_f32_32x32x16_bf16 in the
// "points-as-columns" orientation:  Out[channel][point] = sum_k Wt[channel][k] * Act[k][point]
//   A operand (weights)     : lane l holds row  i = l&31, k-slots (h = l>>5, e = 0..7)
//   B operand (activations) : lane l holds col  j = l&31 (= the lane's collocation point), k-slots (h, e)
//   D accumulator           : lane l holds col  j = l&31, rows (r&3) + 8*(r>>2) + 4*h, r = 0..15
// Because the hardware pairs A slot (h,e) with B slot (h,e), the meaning of a k-slot is ours to
// choose.  We choose it so that the D registers of one layer ARE the B operand of the next
// (no LDS round trip, no cross-lane traffic): k-step ks, slot (h,e)  <->  channel chain_ch(ks,h,e).
#pragma once

#if defined(__HIPCC__)
#define DPN_HD __host__ __device__ constexpr
#else
#define DPN_HD constexpr
#endif

namespace dpn {

constexpr int kNets = 6;        // u, v, P, T, q, rho  (model/physics_net.py:49-54)
constexpr int kHidden = 256;    // net_cfg.hidden_channels
constexpr int kPe = 192;        // net_cfg.in_channels
constexpr int kW1Stride = 193;  // coord_input_fc output row: [w1 (192) | b1]
constexpr int kW2Stride = 257;  // coord_hidden_fc output row: [w2 (256) | b2]

// row of a 32-row D tile held in accumulator register r by a lane of half h
DPN_HD int drow32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// channel fed by chained k-slot (ks, h, e): D tile T = ks/2, register r = 8*(ks&1)+e
DPN_HD int chain_ch(int ks, int h, int e) { return 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3); }

// coordinate PE (SineCosPE(3, N_freqs=32), utils/position_encoding.py:35-50): original channel = f*6 + fn*3 + c.
// k-slot (ks,h,e): angle a = 8*ks + 4*h + (e>>1) in [0,96): c = a>>5, f = a&31; fn = e&1 (0 sin, 1 cos).
DPN_HD int pe3_angle(int ks, int h, int e) { return 8 * ks + 4 * h + (e >> 1); }
DPN_HD int pe3_ch(int ks, int h, int e) { return (pe3_angle(ks, h, e) & 31) * 6 + (e & 1) * 3 + (pe3_angle(ks, h, e) >> 5); }
// data PE (SineCosPE(6, N_freqs=16), model/variable_net.py:45): original channel = f*12 + fn*6 + c6; a = c6*16 + f.
DPN_HD int pe6_ch(int ks, int h, int e) { return (pe3_angle(ks, h, e) & 15) * 12 + (e & 1) * 6 + (pe3_angle(ks, h, e) >> 4); }

// The reverse sweep ends with gpe = w1^T t1 laid out so that D tile T, register r of a lane of half h is the
// cotangent of the lane's OWN q-th PE feature, q = 16*T + r = 8*ks + e  (ks = 2T + (r>>3), e = r&7).
// Row rho = 32*T + i of that GEMM's A operand therefore is PE slot:
DPN_HD int gpe_row_to_pe3_ch(int rho) {
    const int T = rho >> 5, i = rho & 31;
    const int r = (i & 3) + 4 * (i >> 3), h = (i >> 2) & 1;
    return pe3_ch(2 * T + (r >> 3), h, r & 7);
}

// ------------------------------------------------------------------ packed per-net weight block
// Stream order (KB = 1024 B units for NS = 1; multiply by NS for the hi/lo split):
//   S0  w1          8 tiles x 12 k-steps   (A rows o,   K = PE3 slots)
//   S1  w2 , Wd     8 x 16 then 8 x 12     (A rows o,   K = chain(h1) / PE6 slots): all w2 tiles, then all Wd tiles
//   S2  W1          8 x 16                 (A rows o,   K = chain(c))
//   S3  W1^T        8 x 16                 (A rows i,   K = chain(t2))
//   S4  w2^T        8 x 16                 (A rows i,   K = chain(v))
//   S5  w1^T        6 x 16                 (A rows rho, K = chain(t1))
// One k-step fragment = 64 lanes x 8 bf16 = 1 KB; with NS = 2 the hi KB is followed by the lo KB.
constexpr int kS0 = 0;
constexpr int kS1 = kS0 + 8 * 12;
constexpr int kS2 = kS1 + 8 * 28;
constexpr int kS3 = kS2 + 8 * 16;
constexpr int kS4 = kS3 + 8 * 16;
constexpr int kS5 = kS4 + 8 * 16;
constexpr int kPackKB = kS5 + 6 * 16;      // 800 KB per net (x NS)
// fp32 vectors appended after the matrices, each 256 floats in [h][T][r] order (index h*128 + T*16 + r):
enum { kVecB1 = 0, kVecCvec, kVecBf1, kVecU, kVecWo, kVecB2BdE_unused, kNumVecs = 6 };
// ... followed by 4 floats: const0 = wo.bf2 + bo, 0, 0, 0
DPN_HD long pack_bytes_per_net(int ns) { return (long)kPackKB * 1024 * ns + kNumVecs * 1024 + 16; }

// ------------------------------------------------------------------ FUSED form of the block (round 5; dpn_fwd_tiles_kernel)
// With A = W1 w2 and B = W1 Wd formed once per net (exact fp32) the forward + Jacobian pass is five GEMMs instead of seven
// (dpn_fwd_tiles.h; dpn_pack_fused_kernel forms the products and writes their fragments in one launch).  Same block size and the same offsets for what the other kernels read (S0 = w1 for the backward stage-1 kernels, S5 = w1^T,
// the vector block), so that one buffer serves either form:
//   F0  = S0  w1     8 x 12        FA  A      8 x 16  (A rows o, K = chain(h1))       FB  B    8 x 12  (A rows o, K = PE6 slots)
//   FAT       A^T    8 x 16  (A rows j, K = chain(t2))                                  F5  = S5  w1^T  6 x 16;   [kS3, kS5) stays unwritten
// vectors: B1 = b1 | C2 = W1 cvec + bf1 | A2 = w2^T wo | U = W2^T wo | Wo | Bv = Wd^T wo in PE6 slot order (192 used);
//          const0 = wo.bf2 + bo + 2 wo.cvec
constexpr int kF0 = kS0, kFA = kS1, kFB = kS1 + 8 * 16, kFAT = kS2, kF5 = kS5;
static_assert(kFB + 8 * 12 == kS2 && kFAT + 8 * 16 == kS3, "fused form fits the block");
enum { kVecC2 = kVecCvec, kVecA2 = kVecBf1, kVecBv = kVecB2BdE_unused };

}  // namespace dpn
