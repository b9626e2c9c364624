/* libdpn_hip.so -- C ABI of the MI355X (gfx950) point path of the DeepPhysiNet physics-informed step.
 *
 * The reference has no FFI layer: its hot path is Python calling torch ops
 * (DeepPhysiNet/model/variable_net.py:49-87, model/physics_net.py:41-55,
 * interface/interface_physics.py:90-185,232-332).  These entry points are what a
 * ctypes binding on the reference side would call instead;
 * plain device pointers and sizes, no torch types.  All pointers are DEVICE pointers
 * unless said otherwise; `stream` is a hipStream_t.  Every call only enqueues work on
 * `stream` (no allocation, no synchronisation: hipGraph-capturable).  Return value:
 * 0 on success, otherwise a hipError_t (launch/configuration errors) or -1 (bad argument).
 *
 * Precision modes (`prec`): 1 = bf16 MFMA operands, fp32 accumulate ("bf16");
 *                           2 = bf16 hi+lo split operands, 3 MFMAs per product ("bf16x2", fp32-class).
 * Everything that is not a GEMM operand (coordinates, positional features, de-normalisation,
 * residuals, reductions) is fp32 (loss sums fp64) in both modes.
 */
#ifndef DPN_HIP_H
#define DPN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DPN_NETS 6          /* u, v, P, T, q, rho   (physics_net.py:49-54) */
#define DPN_HIDDEN 256
#define DPN_PE 192

/* Per-VariableNet fp32 tensors of ONE field sample (device pointers).
 * w1b1 / w2b2 are the hyper-network heads' outputs (variable_net.py:59-65), evec is
 * fore_h_fc(PE(forecast_h)) (variable_net.py:75-78); the rest are the module's own parameters. */
typedef struct DpnNetPtrs {
    const float* w1b1;   /* [256][193]  rows: [w1 (192) | b1]            */
    const float* w2b2;   /* [256][257]  rows: [w2 (256) | b2]            */
    const float* evec;   /* [256]                                         */
    const float* Wd;     /* [256][192]  data_input_fc.weight             */
    const float* bd;     /* [256]       data_input_fc.bias               */
    const float* W1;     /* [256][256]  cat_fc1.fc.0.weight              */
    const float* bf1;    /* [256]       cat_fc1.fc.0.bias                */
    const float* W2;     /* [256][256]  cat_fc1.fc.2.weight              */
    const float* bf2;    /* [256]       cat_fc1.fc.2.bias                */
    const float* wo;     /* [256]       out_fc.weight                    */
    const float* bo;     /* [1]         out_fc.bias                      */
    int64_t ld_w1b1;     /* row stride of w1b1 in floats (>= 193): lets the six heads live in one GEMM output */
    int64_t ld_w2b2;     /* row stride of w2b2 in floats (>= 257)                                             */
} DpnNetPtrs;

/* Gradient destinations, same shapes as DpnNetPtrs (written, not accumulated). */
typedef struct DpnNetGradPtrs {
    float* w1b1; float* w2b2; float* evec; float* Wd; float* bd; float* W1; float* bf1; float* W2; float* bf2; float* wo; float* bo;
    int64_t ld_w1b1, ld_w2b2;   /* row strides (floats) of the w1b1 / w2b2 gradient destinations */
} DpnNetGradPtrs;

/* Grid constants of encoding_coord (interface_physics.py:322-332). */
typedef struct DpnGeometry {
    float dx, dy;            /* metres                                     */
    float lon_m1, lat_m1;    /* lon_size-1, lat_size-1                     */
    float pred_t_span;       /* seconds                                    */
} DpnGeometry;

/* De-normalisation, clip bounds and loss factors (configs/DeepPhysiNet_NCEP_cfg.py:64-76,139-148). */
typedef struct DpnPhysics {
    float mean[DPN_NETS], std[DPN_NETS];
    float clip_lo[DPN_NETS], clip_hi[DPN_NETS];
    int   clip_on[DPN_NETS];              /* with_clip && bounds apply to this field (u,v: never)  */
    float factor[DPN_NETS];               /* motion_u, motion_v, continuous, energy, vapor, gas     */
    int   criterion;                      /* the PDE criterion `loss(residual, 0)` (train_cfg.losses.pde_loss, interface_physics.py:384; the reference's
                                           * losses/builder.py offers these three): DPN_CRIT_MSE nn.MSELoss, DPN_CRIT_L1 nn.L1Loss, DPN_CRIT_SMOOTH_L1
                                           * WeightSmoothL1Loss(beta) = mean of nn.SmoothL1Loss(beta, reduction='none') (weights_loss.py:12-21)        */
    float beta;                           /* DPN_CRIT_SMOOTH_L1 only (> 0)                                                                        */
    int   sq_on[DPN_NETS];                /* inverse_norm's three-factor min_max form (interface_physics.py:244-247): v = (out * std + mean)^2 + sq_add,
                                           * std = nf[1] - nf[0], mean = nf[0], sq_add = nf[2]; 0: the affine forms                                 */
    float sq_add[DPN_NETS];
    int   reduce_sum;                     /* the criterion's reduction: 0 "mean" (shipped), 1 "sum" (`pde_loss=dict(name='MSELoss', reduction='sum')`)       */
} DpnPhysics;
enum { DPN_CRIT_MSE = 0, DPN_CRIT_L1 = 1, DPN_CRIT_SMOOTH_L1 = 2 };

/* Byte sizes of the caller-allocated work buffers for n_points collocation points. */
typedef struct DpnSizes {
    int64_t n_pad;           /* points padded to the tile size                              */
    int64_t packed;          /* packed bf16 weight fragments + fp32 vectors, all 6 nets     */
    int64_t saved;           /* per-point state saved by dpn_fwd for the backward pass      */
    int64_t operands;        /* per-point operands written by dpn_bwd_points                */
    int64_t partials;        /* split-K partial weight gradients                            */
    int32_t k_splits;        /* number of point ranges dpn_wgrad splits the reduction into  */
} DpnSizes;

int dpn_version(void);
int dpn_sizes(int64_t n_points, int prec, DpnSizes* out);

/* Measurement aid (bench.py): one-thread kernel that appends the device's constant-rate clock (wall_clock64) to ring[cursor++ % cap].  Capturable:
 * it brackets a launch INSIDE a replayed hipGraph, where a HIP event pair cannot be used (event records in a capture are dependencies, not timers).
 * dpn_clock_rate_khz: the counter's rate (hipDeviceAttributeWallClockRate). */
int dpn_clock_stamp(unsigned long long* ring, unsigned int* cursor, unsigned int cap, void* stream);
int dpn_clock_rate_khz(int* khz);

/* fp32 tables: freq32 = 2**linspace(0,4,32), freq16 = 2**linspace(0,4,16) formed in fp32 exactly as
 * utils/position_encoding.py:27 does.  Copied to the device by the caller (8 + 4 = 48 floats: [32 | 16]). */

/* Weight preparation: fp32 tensors -> MFMA-fragment-ordered bf16 (hi/lo) + permuted fp32 vectors + u = W2^T wo.
 * Two forms of the packed block (csrc/dpn_layout.h): form 0 = the matrices of variable_net.py:49-87 as they stand (w1, w2, Wd, W1 and their
 * transposes: seven GEMMs per point and net); form 1 = FUSED: A = cat_fc1.fc.0.weight . w2 and B = cat_fc1.fc.0.weight . data_input_fc.weight
 * are formed once per net in exact fp32 (variable_net.py:81-84 applies fc.0 to c = w2 h1 + Wd pe6 + ..., so pre2 = A h1 + B pe6 + const, and the
 * reverse sweep needs only A^T): five GEMMs per point and net, c and d out / d c never formed.  dpn_fwd_form says which form the forward launch of
 * a precision mode reads (1: hi+lo mode with raw coordinates; 0: plain bf16, or caller-encoded coordinates); dpn_pack_weights packs that form
 * for raw coordinates.  The backward entry points read w1 and the vectors, which sit at the same offsets in both forms. */
int dpn_fwd_form(int prec, int has_pe_in);
int dpn_pack_weights_form(const DpnNetPtrs nets[DPN_NETS], int prec, int form, void* packed, void* stream);
int dpn_pack_weights(const DpnNetPtrs nets[DPN_NETS], int prec, void* packed, void* stream);
/* The same for n_fields field samples in ONE launch (BASELINE configs[2]: a batch of forecast leads, each with its own hyper-network output): nets = the
 * pointer table of field 0; field f reads w1b1 / w2b2 moved by f * heads_stride floats and evec by f * evec_stride floats (the static tensors are
 * shared) and writes its packed block at packed + f * packed_stride bytes (>= DpnSizes.packed, a multiple of 16).  n_fields = 1 ignores the strides. */
int dpn_pack_weights_batch(const DpnNetPtrs nets[DPN_NETS], int n_fields, int64_t heads_stride, int64_t evec_stride, int prec, int form, void* packed,
                           int64_t packed_stride, void* stream);

/* Forward + coordinate Jacobian (replaces PhysicsNet.forward's six VariableNet calls, physics_net.py:49-54,
 * and the 18 unique autograd.grad derivatives of interface_physics.py:90-95).
 *   x,y,t [N] raw coordinates (metres, metres, seconds), OR pe_in [N][192]: coordinates already encoded by the
 *   caller in the reference's SineCosPE channel order (then x,y,t and jac_n must be NULL);
 *   coord_data [N][6]; freqs [48]; out_n [N][6] normalised fields;
 *   jac_n [N][6][3] = d out_n / d(x,y,t)  (may be NULL: value-only); with pe_in it is [N][6][192] = d out_n / d pe_in instead
 *         (the caller chains it through its own coordinate encoding; dpn_contract_gpe contracts it with a cotangent of out_n);
 *   saved: state for the backward pass (may be NULL: inference only). */
int dpn_fwd(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n_points,
            const float* freqs, const DpnGeometry* geo, const void* packed, int prec,
            float* out_n, float* jac_n, void* saved, void* stream);
/* dpn_fwd with the reference's separate ref_data argument (model/variable_net.py:49, :86: `out_fc(y) + ref_data`): ref_data [N][6] is added to
 * out_n in place of coord_data (NULL: coord_data, i.e. dpn_fwd -- PhysicsNet.forward passes coord_data[:, k] as net k's ref_data,
 * physics_net.py:49-54).  VariableNet.forward standalone uses it, so that its output is not rebuilt by subtraction. */
int dpn_fwd_ref(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, const float* ref_data,
                int64_t n_points, const float* freqs, const DpnGeometry* geo, const void* packed, int prec,
                float* out_n, float* jac_n, void* saved, void* stream);

/* dpn_fwd_ref for the first n_nets VariableNets only (1..6), inference only (nothing is saved for a backward pass): VariableNet.forward
 * standalone (model/variable_net.py:49-87) evaluates ONE net, not six.  Columns >= n_nets of out_n / jac_n are left untouched. */
int dpn_fwd_ref_nets(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, const float* ref_data,
                     int64_t n_points, const float* freqs, const DpnGeometry* geo, const void* packed, int prec, int n_nets,
                     float* out_n, float* jac_n, void* stream);

/* g_pe[N][192] = sum_k g_out[N][k] * gpe[N][k][192]: backward of PhysicsNet.forward w.r.t. its encoded-coordinate input. */
int dpn_contract_gpe(const float* g_out, const float* gpe, int64_t n_points, float* g_pe, void* stream);

/* inverse_norm + six residual losses (interface_physics.py:97-185,232-262).
 *   loss_sums [ceil(N/256)][6] fp64: per-block sums over points of residual^2 (written, not accumulated: no zeroing, no atomics;
 *             dpn_residual_finish adds the rows in a fixed order -> deterministic losses);
 *   losses    [7] fp32: [0..5] = factor_i * sum_i / N, [6] = their sum in the reference's order (:301); dpn_residual_finish;
 *   when g_out != NULL also writes d(sum_i w_i * loss_i)/d out_n  [N][6] and d(...)/d J_xi [N][6][3] (cotangent of the
 *   Jacobian w.r.t. the NORMALISED coordinates xi), with w_i = gl[i] + gtot[0] (either may be NULL; both NULL: w_i = 1). */
int dpn_residual(const float* out_n, const float* jac_n, const float* f, int64_t n_points, const DpnGeometry* geo,
                 const DpnPhysics* phys, const float* gl /*[6] or NULL*/, const float* gtot /*[1] or NULL*/, double* loss_sums,
                 float* g_out, float* g_jxi, void* stream);
int dpn_residual_finish(const double* loss_sums, int64_t n_points, const DpnPhysics* phys, float* losses, void* stream);
/* n_fields fields of n_points each in one launch: loss_sums [n_fields][ceil(n_points / 256)][6], losses [n_fields][7]. */
int dpn_residual_finish_batch(const double* loss_sums, int64_t n_points, int n_fields, const DpnPhysics* phys, float* losses, void* stream);

/* Backward, stage 1: per-point cotangent streams -> the operands of the weight-gradient reductions.
 *   g_out [N][6]; g_jxi [N][6][3] or NULL (value-only loss, e.g. the data loss). */
int dpn_bwd_points(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n_points,
                   const float* freqs, const DpnGeometry* geo, const void* packed, int prec,
                   const float* g_out, const float* g_jxi, const void* saved, void* operands, void* stream);
/* The same with a device scalar multiplied into both cotangent streams as they are read: the caller has g_out / g_jxi for a UNIT cotangent of the total
 * loss (dpn_residual evaluated once, losses and cotangents in one pass) and the upstream cotangent arrives later, as a tensor (loss.backward(seed)). */
int dpn_bwd_points_scaled(const float* x, const float* y, const float* t, const float* pe_in, const float* coord_data, int64_t n_points,
                          const float* freqs, const DpnGeometry* geo, const void* packed, int prec,
                          const float* g_out, const float* g_jxi, const float* g_scale /* [1] device, or NULL = 1 */, const void* saved, void* operands,
                          void* stream);

/* Backward, stage 2: weight-gradient reductions over points (split-K partial sums). */
int dpn_wgrad(int64_t n_points, int prec, const float* g_out, const void* saved, const void* operands, void* partials, void* stream);

/* Backward, stage 3: reduce the partial sums, undo the fragment permutations and assemble every gradient of DpnNetPtrs from the three
 * points-reduction products S1 = M2^T Z1, S2 = M2^T G6, dw1 = T1^T Z0 and their vector sums (interface_physics.py:506, the backward of
 * variable_net.py:49-87):  d w2b2 / d Wd = W1^T diag(u) S + 2 wo (x) colsum (csrc SavedView);  d cat_fc1.fc.0.weight = diag(u) G with
 * G = M2^T Z = S1 w2^T + S2 Wd^T + (M2^T g) (x) (b2 + bd + e) -- Z itself is never formed per point (csrc dpn_finish_gside_kernel);
 * the rank-1 cat_fc1.fc.2 / out_fc gradients follow from r = rowdot(W1, G) + bf1 (.) M2^T g. */
int dpn_wgrad_finish(const DpnNetPtrs nets[DPN_NETS], const void* packed, int64_t n_points, int prec,
                     const void* partials, const DpnNetGradPtrs grads[DPN_NETS], void* stream);
/* The same in two halves.  parts bit 0: what the hyper-network's backward waits for (d w1b1, d w2b2, d evec; d Wd and d bd ride along);
 * parts bit 1: the gradients of cat_fc1.fc.0 / fc.2 and out_fc (static tensors only).  Half 1 reads what half 0 left in the scratch tail of
 * `partials`: issue it on the same stream, or on another one ordered behind half 0 (a branch beside the encoder's backward chain). */
int dpn_wgrad_finish_parts(const DpnNetPtrs nets[DPN_NETS], const void* packed, int64_t n_points, int prec,
                           const void* partials, const DpnNetGradPtrs grads[DPN_NETS], int parts, void* stream);

/* SmoothL1(beta) data loss on the normalised fields (losses/weights_loss.py:17-20): per-block sums -> loss_sum[ceil(6N/256)] (fp64,
 * written not accumulated; the caller adds them in a fixed order), g_out = scale * dSmoothL1 (may be NULL). */
int dpn_smooth_l1(const float* out_n, const float* labels, int64_t n_points, float beta, float scale,
                  double* loss_sum, float* g_out, int accumulate /* g_out += instead of = */,
                  const float* scale_dev /* optional device scalar multiplied into scale (an upstream cotangent) */, void* stream);

/* Small fp32 GEMM for the per-field tensors (encoder linears of model/attn.py:177-196 and transformer_net.py:28-44, the
 * hyper-network heads of variable_net.py:59-65):  C[M][N] = op(A)[M][K] op(B)[K][N] (+ bias[N]) (+ C if accumulate);
 * ta/tb = 1 reads A as [K][M] / B as [N][K] (row-major, leading dimensions lda/ldb/ldc);
 * asum (may be NULL) receives sum_k op(A)[m][k] -- the bias gradient when op(A) = grad_out^T.
 * workspace (may be NULL): scratch for the deterministic two-pass split-K used when K is long and the output small. */
int dpn_sgemm(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
              const float* bias, float* asum, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* Up to 26 independent small fp32 GEMMs in one launch (exact-fp32 MFMA, fixed reduction order); each
 * C[M][N] = epilogue(sum_{t<nterms} op(A_t)[M][K_t] op(B_t)[K_t][N] + bias[N]); asum as in dpn_sgemm (single-term problems only).
 * Used for the q/k/v projections (attn.py:183-185), the paired input-/weight-gradient GEMMs of every encoder linear, and the
 * twelve hyper-network heads + six lead-time embeddings of the VariableNets (variable_net.py:59-65,75-78) and their backward. */
#define DPN_EPI_NONE 0
#define DPN_EPI_GELU 1            /* C = gelu(v) (exact erf, transformer_net.py:26,41); aux_out (optional) receives v                    */
#define DPN_EPI_MUL_GELU_GRAD 2   /* C = v * gelu'(aux): backward of the activation folded into the GEMM that feeds it                 */
#define DPN_EPI_ADD 3             /* C = v + aux: the residual-branch gradient joins the input gradient without a separate add kernel  */
#define DPN_GEMM_MAX_TERMS 12      /* per problem */
#define DPN_GEMM_MAX_PROBLEMS 26   /* per launch; at most 32 terms per launch in total */
#define DPN_GEMM_MAX_JOBS 10       /* ride-along column-sum jobs per launch */
typedef struct DpnGemmProblem {
    const float* A[DPN_GEMM_MAX_TERMS]; const float* B[DPN_GEMM_MAX_TERMS]; int32_t lda[DPN_GEMM_MAX_TERMS], ldb[DPN_GEMM_MAX_TERMS];
    int32_t k_term[DPN_GEMM_MAX_TERMS];   /* reduction length of term t; 0 = K */
    const float* bias; float* C; float* asum;
    int32_t M, N, K, ldc, ta, tb, nterms;
    const float* aux; float* aux_out;   /* [M][ldc], see DPN_EPI_* */
    int32_t epi;
} DpnGemmProblem;
int dpn_sgemm_batch(int n_problems, const DpnGemmProblem* problems /* host array */, void* stream);
/* The same launch with up to DPN_GEMM_MAX_JOBS ride-along column-sum jobs: out_a[c] = sum_b partial[b][c], out_b[c] = sum_b partial[b][256 + c],
 * c < 256, b < n_blocks (fixed order) -- the LayerNorm parameter gradients from dpn_add_ln_bwd's scratch (n_blocks = ceil(rows/4)),
 * finished inside a GEMM launch that follows it instead of in a launch of their own. */
typedef struct DpnColsumJob { const float* partial; float* out_a; float* out_b; int32_t n_blocks; } DpnColsumJob;
int dpn_sgemm_batch_jobs(int n_problems, const DpnGemmProblem* problems, int n_jobs, const DpnColsumJob* jobs /* host array */, void* stream);

/* LayerNorm folded into the A operand of the GEMM that consumes it (d_model = 256: the GEMM's K is the LayerNorm's row):
 *   mode 1:  C = epilogue((LN(x + r) * gamma + beta) . op(B) + bias); also writes y = LN(..)*gamma+beta [M][256], xhat [M][256], rstd [M]
 *            (EncoderLayer: x1 = norm1(x + attn) feeding conv1, transformer_net.py:37-41);
 *   mode 2:  C = epilogue(gs . op(B) + bias) with gs = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)) the LayerNorm input
 *            gradient (x = g, r = xhat); also writes gs [M][256] (y_out) and partial[ceil(M/32)][2][256] = per-row-block sums of g*xhat
 *            and g, to be reduced by a DpnColsumJob (n_blocks = ceil(M/32)) of a later dpn_sgemm_batch_jobs launch.
 * op(B) = B^T with B stored [N][256] (tb = 1) or B stored [256][N] (tb = 0).  One launch instead of LayerNorm + GEMM. */
typedef struct DpnLnGemm {
    int32_t mode, M, N, tb, ldb, ldc, epi;
    const float *x, *r, *gamma, *beta, *rstd_in;
    float *y_out, *xhat_out, *rstd_out, *partial;
    const float *B, *bias;
    float* C;
    const float* aux;
    float* aux_out;
} DpnLnGemm;
int dpn_sgemm_ln(const DpnLnGemm* problem, void* stream);

/* FullAttention of the encoder (model/attn.py:50-68): o = softmax(q k^T / sqrt(32)) v for 8 heads x 32 over L <= 288 tokens.
 * q,k,v,o,go,dq,dk,dv: [batch*L][256] fp32 row-major (field b = rows b*L.., head h = columns 32h..32h+31); attention never crosses
 * fields.  P (saved probabilities): [batch][8][288][288] fp32.
 * The backward is one launch: query-tile blocks produce dq, key-tile blocks produce dk and dv (recomputing their columns of dS). */
int dpn_attn_fwd(const float* q, const float* k, const float* v, int L, int batch, float* out, float* P, void* stream);
int dpn_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* P, const float* go, int L, int batch,
                 float* dq, float* dk, float* dv, void* stream);
/* dpn_attn_fwd on the f16 hi+lo matrix cores with 16-row query tiles (same arguments, same P): q, k, v never pass through LDS -- a lane's
 * eight consecutive head channels of one token row are an MFMA fragment slot.  The encoder's fused forward uses this one. */
int dpn_attn16_fwd(const float* q, const float* k, const float* v, int L, int batch, float* out, float* P, void* stream);

/* ---------------------------------------------------------------- row-local fused encoder nodes (csrc/dpn_encoder_chain.hip)
 * Everything of an EncoderLayer except the attention itself is row-wise (attn.py:183-185,196; transformer_net.py:33-44, :68, :129), so the
 * whole stretch between two attention kernels is ONE launch (a workgroup carries 16 or 32 token rows through up to six 256 x 256 GEMMs).
 * All matrices are [256][256] fp32 as the reference stores them ([out][in]; Conv1d k=1 weights with the last axis squeezed), all row
 * tensors [rows][256] fp32, rows = batch * L.  The GEMMs run on f16 hi+lo split operands (three products, fp32 accumulate: fp32-class).
 *
 * dpn_enc_pack: MFMA-fragment images of n_mats matrices (dpn_enc_pack_bytes(n_mats) bytes: per matrix one image for x W^T and one for
 * g W).  Once per step, after the optimiser changed the weights.  *status_dev (may be NULL) gets bit 0 set when an |entry| >= 32768. */
#define DPN_ENC_MAX_MATS 32
int64_t dpn_enc_pack_bytes(int n_mats);
int dpn_enc_pack(int n_mats, const float* const* weights /* host array of device pointers */, void* packed, int* status_dev, void* stream);

/* dpn_enc_prep: everything of the encoder forward that depends on the step's inputs only, in ONE launch -- the weight images (dpn_enc_pack),
 * the im2col rows of the circular token convolution (dpn_im2col_circ3: x [batch*T][C] -> xu [batch*T][3C]) and the lead-time SineCosPE for one
 * or two frequency tables (dpn_lead_pe: h [batch] -> out_a [batch][2 n_a], out_b [batch][2 n_b]).  Each part is optional (n_mats = 0 / x = NULL
 * / h = NULL). */
typedef struct DpnEncPrep {
    int32_t n_mats; const float* const* weights /* host array of device pointers */; void* packed; int* status_dev;
    const float* x; int32_t T, C, batch; float* xu;
    const float* h; const float* freqs_a; int32_t n_a; float* out_a; const float* freqs_b; int32_t n_b; float* out_b;
} DpnEncPrep;
int dpn_enc_prep(const DpnEncPrep* p, void* stream);

/* Forward.  tail = 1:  x1 = norm1(x + o Wo^T + bo);  pre = x1 Wc1^T + bc1;  act = gelu(pre);  x2 = norm2(x1 + act Wc2^T + bc2)
 *                      (o = the attention output, x = the layer input; x1, xhat1, rstd1, pre, act, x2, xhat2, rstd2 are written);
 *           next = 1:  y0, y1, y2 = t Wn0^T + bn0, ...   the NEXT layer's q / k / v projections of t = x2 (tail = 1) or t = xin (tail = 0);
 *           next = 2:  xf = encoder.norm(x2) (xf, xhatf, rstdf written);  y0 = xf Wn0^T + bn0   (the output projection);
 *           next = 0:  nothing behind norm2.
 * m_*: indices of the matrices in `wpack`.  row_tiles: 1 or 2 sixteen-row tiles per workgroup (2 for batches of fields). */
typedef struct DpnEncFwd {
    const void* wpack; int32_t n_mats, rows, row_tiles, tail, next;
    int32_t m_o, m_c1, m_c2, m_n0, m_n1, m_n2;
    const float *o, *x, *xin;
    const float *bo, *g1, *be1, *bc1, *bc2, *g2, *be2, *gf, *bef, *bn0, *bn1, *bn2;
    float *x1, *xhat1, *rstd1, *pre, *act, *x2, *xhat2, *rstd2, *xf, *xhatf, *rstdf, *y0, *y1, *y2;
    /* tail = 0, one field (round 6): emb_parts != NULL makes this launch ASSEMBLE its input instead of reading xin -- what dpn_embed_assemble does in a
     * launch of its own (embed.py:60-64, transformer_net.py:124-126), same order of additions:
     *   row <  emb_n_tok:  x0[row] = (emb_token[row] + emb_pos[row]) + emb_te
     *   row >= emb_n_tok:  x0[row] = ((sum_p emb_parts[p][row - emb_n_tok], p = 0 .. emb_n_parts - 1, in order) + emb_bias + emb_pos[row]) + emb_te
     * emb_parts [emb_n_parts][emb_part_stride floats] (the token convolution's split-K slices, rows of 256), emb_te [256]; x0 is also written to emb_out [rows][256]. */
    const float *emb_parts, *emb_bias, *emb_pos, *emb_te, *emb_token;
    float* emb_out;
    int64_t emb_part_stride;
    int32_t emb_n_parts, emb_n_tok;
} DpnEncFwd;
int dpn_enc_fwd(const DpnEncFwd* p, void* stream);

/* Backward of the same stretch, in the order the cotangent travels.
 *   head = 1:  g = res + dq Wh0 + dk Wh1 + dv Wh2      (the q / k / v projections of the layer ABOVE: its attention backward's outputs and
 *                                                      its residual-branch cotangent gs1)
 *   head = 2:  g = encoder.norm backward of (dmeta Wh0) (output projection; partial_f receives the norm's parameter partial sums)
 *   head = 0:  g = gin
 *   body = 1:  gs2 = norm2 backward of g;  dpre = (gs2 Wc2) * gelu'(pre);  gs1 = norm1 backward of (dpre Wc1 + gs2);  dout = gs1 Wo
 *              (gs2, dpre, gs1: the operands of the weight-gradient GEMMs dWc2 = gs2^T act, dWc1 = dpre^T x1, dWo = gs1^T o; dout: the
 *              attention backward's input; gs1 is also the residual-branch cotangent the next launch takes as `res`)
 *   body = 0:  gx = g   (the cotangent of the first layer's input); its first gx_head_rows rows are written to gx_head as well when that is
 *              given (the learnable tokens in front of the field tokens, model/transformer_net.py:123-126: their gradient lands in its own
 *              tensor without a copy)
 * partial_f / partial2 / partial1: [workgroups][512] = per-workgroup sums of (g * xhat | g) of the three LayerNorms, reduced in a fixed order by a
 * DpnColsumJob with n_blocks = ceil(rows / (16 row_tiles)). */
typedef struct DpnEncBwd {
    const void* wpack; int32_t n_mats, rows, row_tiles, head, body;
    int32_t m_h0, m_h1, m_h2, m_c2, m_c1, m_o;
    const float *res, *dq, *dk, *dv, *dmeta, *xhatf, *rstdf, *gin;
    const float *xhat2, *rstd2, *pre, *xhat1, *rstd1, *g2, *g1, *gf;
    float *gs2, *dpre, *gs1, *dout, *gx, *partial_f, *partial2, *partial1;
    float* gx_head; int32_t gx_head_rows;
} DpnEncBwd;
int dpn_enc_bwd(const DpnEncBwd* p, void* stream);

/* Weight gradients of linears over token rows, up to DPN_WGRAD_MAX_PROBLEMS in one launch:  dW[M][N] = G^T X,  db[M] = column sums of G
 * (db may be NULL), G [rows][M] (row stride ldg) the cotangent of the linear's output, X [rows][N] (ldx) its input, dW row stride ldw.
 * f16 hi+lo split operands with a running power-of-two scale per operand strip (any magnitude), fp32 accumulate, fixed order.
 * slices > 1 cuts the row reduction into that many slices (batches of fields): `partials` then needs dpn_wgrad16_partial_floats() floats and
 * a second launch adds the slices in order.  Up to DPN_GEMM_MAX_JOBS DpnColsumJob ride along (the LayerNorm parameter sums of dpn_enc_bwd).
 * Replaces dpn_sgemm_batch for the weight gradients of the encoder stack (attn.py:183-196, transformer_net.py:38-42,129, embed.py:45-47). */
#define DPN_WGRAD_MAX_PROBLEMS 32
typedef struct DpnWgradProblem { const float* G; const float* X; float* dW; float* db; int32_t M, N, rows, ldg, ldx, ldw; } DpnWgradProblem;
int64_t dpn_wgrad16_partial_floats(int n, const DpnWgradProblem* problems, int slices);
int dpn_wgrad16(int n, const DpnWgradProblem* problems /* host array */, int n_jobs, const DpnColsumJob* jobs, int slices, float* partials,
                void* stream);

/* out = LayerNorm_256(x + r) * gamma + beta (eps 1e-5), r may be NULL (transformer_net.py:37,44,68); saves xhat [rows][256] and rstd [rows].
 * Backward: gx = d/d(x + r) (the same tensor is the gradient of x and of r), dgamma, dbeta [256].  With dgamma = dbeta = NULL only gx and the
 * per-block partial sums in `scratch` ([ceil(rows/4)][512]) are produced, to be reduced by a DpnColsumJob of dpn_sgemm_batch_jobs. */
int dpn_add_ln_fwd(const float* x, const float* r, const float* gamma, const float* beta, int rows, float* out, float* xhat, float* rstd,
                   void* stream);
int dpn_add_ln_bwd(const float* g, const float* xhat, const float* rstd, const float* gamma, int rows, float* gx, float* dgamma, float* dbeta,
                   float* scratch /* ceil(rows/4) * 512 floats */, void* stream);

/* Data embedding of the encoder (model/embed.py:36-64, transformer_net.py:124-126), d_model = 256:
 *   dpn_lead_pe        SineCosPE of the scalar lead time for one or two frequency tables: out[2f] = sin(h f), out[2f+1] = cos(h f)
 *                      (encoder time embedding, N_freqs = 128, embed.py:58; VariableNet lead-time PE, N_freqs = 96, variable_net.py:46);
 *   dpn_im2col_circ3   out[T][3C], out[t][3c + tap] = x[(t + tap - 1) mod T][c]: the circular k=3 TokenEmbedding conv becomes
 *                      out . W^T with the Conv1d weight [256][C][3] read in place;
 *   dpn_embed_assemble out[n_tok + n_emb][256] = cat(learnable_token, value_embedding) + positional table + lead-time embedding, where the
 *                      value embedding is given as n_parts split-K partial products [n_parts][n_emb][256] (+ bias[256], may be NULL). */
/* out[0 .. count) = sum_p parts[p][0 .. count) (fixed order), out[count .. count + zero_tail) = 0: joins split-K partial products. */
int dpn_sum_parts(const float* parts, int n_parts, int64_t count, int64_t zero_tail, float* out, void* stream);
int dpn_lead_pe(const float* h_dev /* [batch] */, int batch, const float* freqs_a, int n_a, float* out_a /* [batch][2 n_a] */,
                const float* freqs_b, int n_b, float* out_b, void* stream);
int dpn_im2col_circ3(const float* x /* [batch*T][C] */, int T, int C, int batch, float* out /* [batch*T][3C] */, void* stream);
int dpn_embed_assemble(const float* token, int n_tok, const float* emb_parts /* [n_parts][batch*n_emb][256] */, int n_parts, int n_emb, int batch,
                       const float* bias, const float* pos, const float* te /* [batch][256] */, float* out /* [batch*(n_tok+n_emb)][256] */,
                       void* stream);

/* clip_grad_norm_(max_norm) + torch.optim.Adam step (interface_physics.py:514-515; cfg:151-155: L2-in-gradient weight decay)
 * over a list of fp32 tensors.  The pointer arrays and `numel` are HOST arrays of length n_tensors (device pointers inside);
 * scratch_dev: dpn_clip_adam_scratch_doubles(n_tensors, numel) doubles on the device ([0] receives the sum of squares of all
 * gradients, the rest are per-block partials added in a fixed order: the norm is deterministic and there are no atomics);
 * step_dev (int, the Adam step counter, incremented on the device) and out_norm_dev (float, may be NULL) are device scalars.
 * Seven launches for the 155 tensors of a PhysicsNet. */
int64_t dpn_clip_adam_scratch_doubles(int n_tensors, const int64_t* numel);
int dpn_clip_adam(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const int64_t* numel, double* scratch_dev, int* step_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float max_norm, float* out_norm_dev, void* stream);

/* ---------------------------------------------------------------- collocation sampler + full-grid gather
 * Geometry of the fine (label) grid and the coarse forecast cube (dataset/physics_dataset.py:104-126): the fine grid has
 * lon x lat nodes, dlat degrees apart, starting at begin_lat; one fine cell is cells_x / cells_y coarse cells wide
 * (0.25 deg / 1 deg = 0.25 for the shipped data); the coarse cube has t_in time slices, t_step_hours apart
 * (input_time_step = 6, input_time_step_nums + 1 = 5), and collocation times are whole hours in [0, t_hours]. */
typedef struct DpnSampler {
    int32_t lon, lat;                 /* fine grid (label_lon_size, label_lat_size) = (257, 145)     */
    int32_t lon_in, lat_in, t_in;     /* coarse cube (65, 37, 5)                                       */
    int32_t t_hours;                  /* input_time_step * input_time_step_nums = 24                   */
    double cells_x, cells_y;          /* coarse cells per fine cell                                    */
    double t_step_hours;              /* 6                                                             */
    double begin_lat, dlat;           /* degrees                                                       */
    float dx, dy;                     /* metres per fine cell                                          */
} DpnSampler;
#define DPN_SAMPLE_INTERIOR 0   /* get_inter_data       (physics_dataset.py:431-499): x, y continuous uniform, t integer hours */
#define DPN_SAMPLE_MARGIN   1   /* get_item_label_data  (:323-429): x, y integer grid nodes, + label gather                    */
#define DPN_SAMPLE_EXPLICIT 2   /* get_margin_grid      (:528-587): node indices xi, yi and hour ti given by the caller         */
/* Draws n collocation points (Philox-4x32-10, counter = offset + point index, key = seed), interpolates the coarse cube
 * cube[6][lat_in][lon_in][t_in] (fp32, variable order u10,v10,pres,t2,q2,rio) tri-linearly at them (fp64 weights, result
 * cast to fp32: what xarray's .interp + .float() produce) and writes x, y (metres), t (seconds), f (Coriolis),
 * coord_data[n][6].  labels[t_hours+1][6][lat][lon] / label_out[n][6] (node modes only) and raw[n][3] (the draws as
 * doubles: x index, y index, hour) are optional (NULL). */
int dpn_sample_points(const DpnSampler* s, const float* cube, const float* labels, int mode, const int32_t* xi, const int32_t* yi,
                      const int32_t* ti, int64_t n, uint64_t seed, uint64_t offset, float* x, float* y, float* t, float* f,
                      float* coord_data, float* label_out, double* raw, void* stream);
/* The same draw with the Philox counter advanced on the DEVICE: counter = offset + *step_dev * stride + point index.  step_dev is the
 * optimiser's device-side step counter (dpn_clip_adam_flat's step_dev, bumped once per step), stride the points drawn per step, so a
 * sampler launch captured in a hipGraph draws fresh points on every replay (a captured host-side offset would freeze them).
 * step_dev == NULL is dpn_sample_points.  Replaces the per-__getitem__ np.random draws of physics_dataset.py:334-338,442-446 inside a
 * captured training step. */
int dpn_sample_points_replay(const DpnSampler* s, const float* cube, const float* labels, int mode, const int32_t* xi, const int32_t* yi,
                             const int32_t* ti, int64_t n, uint64_t seed, uint64_t offset, const int32_t* step_dev, uint64_t stride,
                             float* x, float* y, float* t, float* f, float* coord_data, float* label_out, double* raw, void* stream);
/* Normalised fields of all lon*lat nodes in the reference's node order (x outer, y inner; interface_physics.py:538-543),
 * out_n[lon*lat][6] -> de-normalised maps[6][lat][lon] (inverse_norm :232-262 + the scatter loop :583-591). */
int dpn_grid_maps(const float* out_n, int lon, int lat, const DpnPhysics* phys, int with_clip, float* maps, void* stream);

/* The same step with the optimiser state held as ONE flat fp32 buffer per moment: tensor i lives at offset (sum of ceil(numel_j / 2048)
 * over j < i) * 2048, i.e. every tensor is padded to whole 2048-element chunks; dpn_clip_adam_flat_floats gives the buffer length.
 * No per-tensor state pointers travel in the kernel arguments, so up to 160 tensors are ONE launch per pass: three launches for a
 * PhysicsNet (gradient norm, its fixed-order reduction, update). */
int64_t dpn_clip_adam_flat_floats(int n_tensors, const int64_t* numel);
int dpn_clip_adam_flat(int n_tensors, float* const* params, const float* const* grads, const int64_t* numel, float* exp_avg_flat,
                       float* exp_avg_sq_flat, double* scratch_dev, int* step_dev, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float max_norm, float* out_norm_dev, void* stream);

/* The same with the hyper-parameters read from DEVICE memory at run time: hyper_dev[7] = {lr, beta1, beta2, eps, weight_decay, max_norm,
 * grad_scale}.  A step captured in a hipGraph then follows a learning-rate schedule (CosineAnnealingLR stepped per epoch,
 * interface_physics.py:831-833) -- the host rewrites hyper_dev[0] between replays; grad_scale multiplies every gradient before the norm and
 * the update (1 / world_size behind a SUM all-reduce of the flat gradient buffer, replacing DistributedDataParallel's averaging, :903-907). */
int dpn_clip_adam_flat_dev(int n_tensors, float* const* params, const float* const* grads, const int64_t* numel, float* exp_avg_flat,
                           float* exp_avg_sq_flat, double* scratch_dev, int* step_dev, const float* hyper_dev, float* out_norm_dev,
                           void* stream);

/* BASELINE configs[4] (OFF by default; `bench.py --encoder-fp8` / DPN_ENCODER_FP8=mx routes the encoder layers' forward GEMMs here): C[M][N] =
 * epilogue(A[M][K] . W[N][K]^T + bias[N]) on the block-scaled (MX) fp8 instruction v_mfma_scale_f32_32x32x64_f8f6f4: OCP e4m3 operands quantised in
 * the kernel, one E8M0 power-of-two scale per 32 consecutive k of a row (OCP MX) applied by the hardware, fp32 accumulate; K % 64 == 0; epi =
 * DPN_EPI_NONE or DPN_EPI_GELU (aux_out receives the pre-activation).  Replaces nothing of the reference by default: its measured parity error is why
 * (profiles/round3_fp8_mx_encoder.json).  (The non-scaled fp8 form with per-row scales is a shelved experiment: dpn_hip_experiments.h.) */
int dpn_gemm_fp8_mx(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int epi,
                 float* aux_out, void* stream);

/* Self-test of the MFMA fragment-layout assumptions in dpn_layout.h (A = I against an asymmetric B). Returns 0 if they hold. */
int dpn_selftest(void* scratch_dev /* >= 64 KiB */, void* stream);

#ifdef __cplusplus
}
#endif
#endif
