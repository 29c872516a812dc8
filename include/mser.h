/* libmser -- C-ABI of the MI355X-native speaker-aware LSTHM hot path.
 *
 * The reference (MallVilliers/Multimodal-Framework-for-speaker-emotion-recognition) is pure Python/PyTorch and has NO
 * FFI of its own; the drop-in boundary is its nn.Module surface.  This header is the boundary the
 * replacement puts *under* that surface: every entry point takes raw device pointers owned by the caller (PyTorch
 * tensors on the Python side), explicit sizes/strides and a hipStream_t; it never allocates, frees or synchronises,
 * returns 0 on success and a non-zero code otherwise (message via mser_last_error()).  No torch types cross it.
 *
 * Each entry names the reference code it replaces (path:line relative to the reference checkout).
 * Layout conventions: activations are TIME-MAJOR row matrices, row r = t*B + b (the reference's [L,B,D]),
 * fp32, innermost dimension contiguous unless a leading dimension ("ld", in elements) is given.
 */
#ifndef MSER_H_
#define MSER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mser_stream_t; /* hipStream_t */

#define MSER_VERSION 120   /* 110: + encoder layer, grouped GEMM, head tail, ingest, confusion, cell desc addends */

int mser_version(void);
const char* mser_last_error(void);

/* Bits of the caller-owned sticky fault word (uint32 in device memory) that kernels OR into when something went wrong that
 * cannot be reported through a return code (the launch is asynchronous): see mser_cell_desc::fault, mser_gru_speaker_desc::status,
 * mser_masked_loss_fwd, mser_adam_flat_dev.  The reference raises in the corresponding situations (an out-of-range label makes
 * NLLLoss / CrossEntropyLoss raise, loss.py:19-24); a persistent chain has no counterpart there. */
enum { MSER_FAULT_CHAIN_TIMEOUT = 1, MSER_FAULT_LINK_TIMEOUT = 2, MSER_FAULT_BAD_LABEL = 4 };

/* ------------------------------------------------------------------------------------------------
 * Generic strided batched fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains).
 *   C[z1,z2][m,n] (+)= epilogue( alpha * (*alpha_dev) * sum_k A[z1,z2][m,k] * B[z1,z2][k,n] )
 *   epilogue(v) = relu?( v + bias[n] ) + R1[m,n] + R2[m,n]
 * Replaces every nn.Linear / torch.matmul on the path (model/lsthm_sps.py:29-32,64-70,93-99,121-127,
 * 316,321,324,353; model/encoder.py:37-39,53,74,84,105).
 * ------------------------------------------------------------------------------------------------ */
enum { MSER_GEMM_RELU = 1, MSER_GEMM_ACCUM = 2 };

typedef struct mser_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  int32_t M, N, K;
  int64_t sAm, sAk;       /* element strides of A[m,k] */
  int64_t sBk, sBn;       /* element strides of B[k,n] */
  int64_t ldc;            /* row stride of C (n contiguous) */
  int32_t batch1, batch2; /* >= 1 */
  int64_t sA1, sA2, sB1, sB2, sC1, sC2;
  const float* bias;      /* [N] or NULL */
  const float* alpha_dev; /* device scalar or NULL */
  float alpha;
  int32_t flags;          /* MSER_GEMM_* */
  int32_t splitk;         /* >= 1; > 1 accumulates with float atomics (C must be pre-initialised) */
  const float* R1;        /* residuals, same indexing as C with row stride ldr1/ldr2; or NULL */
  const float* R2;
  int64_t ldr1, ldr2;
  int64_t sR1_1, sR1_2;   /* batch strides of R1 (R2 shares them) */
} mser_gemm_desc;

int mser_gemm(const mser_gemm_desc* d, mser_stream_t stream);
/* n independent products in as few launches as their load-mode / tile classes allow (one for the weight gradients of a
 * training step: dW = dY^T X of every nn.Linear on the path -- what autograd computes per module, model_trainer.py:113).
 * Results are identical to n mser_gemm calls up to the order of the split-K float atomics. */
int mser_gemm_grouped(const mser_gemm_desc* d, int32_t n, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row kernels.
 * ------------------------------------------------------------------------------------------------ */
/* in-place softmax over the last dim of S[rows,n] (row stride ld); optional multiplicative weights `mul`
 * and byte mask `mask` (same indexing as S): where mask[i]==mask_on the logit is replaced by `fill`
 * BEFORE the softmax (model/encoder.py:75-80 uses mask==0 -> -1e9; attention:/SelfAttention.py:67-71
 * uses weights then mask==1 -> -inf). */
int mser_softmax_rows(float* S, int64_t rows, int32_t n, int64_t ld, const float* mul, const uint8_t* mask,
                      int32_t mask_on, float fill, mser_stream_t stream);
/* dS = P * (dP - sum_j P*dP) (* mul), written over dP. */
int mser_softmax_bwd_rows(const float* P, float* dP, int64_t rows, int32_t n, int64_t ld, const float* mul,
                          mser_stream_t stream);
/* y = LayerNorm(x + res) * gamma + beta over D (model/encoder.py:56-58, 109-111); saves mean/rstd [rows].
 * `sum_out` (optional) receives x + res (needed by the backward). res may be NULL. */
int mser_add_layernorm_fwd(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* gamma,
                           const float* beta, float* y, float* sum_out, float* mean, float* rstd, int64_t rows,
                           int32_t D, float eps, mser_stream_t stream);
/* dx = LN backward wrt the summed input; dgamma/dbeta are ACCUMULATED (float atomics). */
int mser_layernorm_bwd(const float* dy, const float* xsum, const float* mean, const float* rstd,
                       const float* gamma, float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t D,
                       mser_stream_t stream);
/* out[n] += sum_m X[m,n] */
int mser_colsum_acc(const float* X, int64_t rows, int32_t n, int64_t ld, float* out, mser_stream_t stream);
/* dY *= (Y > 0) */
int mser_relu_bwd(float* dY, const float* Y, int64_t count, mser_stream_t stream);
/* out[r, :D] = a[r, :D] (+ b[r, :D]); b may be NULL */
int mser_add_rows(float* out, int64_t ldo, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows,
                  int32_t D, mser_stream_t stream);
/* acc[r,:D] += (*s_dev) * t[r,:D];  *ds += <t, x>   (gradient of y = s * x w.r.t. x and the scalar s;
 * replaces autograd through `self.w * x_l` etc., model/lsthm_sps.py:377-383). s_dev/ds may be NULL (s = 1). */
int mser_scale_acc_dot(float* acc, int64_t ldacc, const float* t, int64_t ldt, const float* x, int64_t ldx,
                       const float* s_dev, float* ds, int64_t rows, int32_t D, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * EncoderLayer (model/encoder.py:116-133: MultiHeadAttention :7-60, ScaledDotProductAttention :63-86,
 * PositionwiseFeedForward :89-113) as one call: forward = projection GEMM + one attention launch per
 * (dialogue, head) + one row-tiled launch (fc, residual, LayerNorm, FFN, residual, LayerNorm);
 * backward = row-tiled launch + attention launch + input-gradient GEMM (MSER_ENC_BWD_ACT) and the four
 * weight-gradient GEMMs (MSER_ENC_BWD_WGRAD, any stream, any time after ACT).
 * Dropout sites (:54,:83,:106): see the rng fields at the end of the descriptor.  Rows: row(b, l) = b*sb + l*sl.
 * The caller owns every buffer; `mser_encoder_layer_supported` tells whether the fused kernels cover the
 * shape (L <= 128, d_k == d_v <= 64 and % 8 == 0, D <= 128, ...); otherwise compose the layer from mser_gemm + row kernels.
 * ------------------------------------------------------------------------------------------------ */
enum { MSER_ENC_BWD_ACT = 1, MSER_ENC_BWD_WGRAD = 2 };

typedef struct mser_encoder_desc {
  int32_t nb, nl;              /* dialogues, positions */
  int64_t sb, sl;              /* row(b, l) = b*sb + l*sl */
  int32_t D, nh, dk, dv, dff;  /* d_model, heads, d_k, d_v, d_inner */
  float eps;                   /* LayerNorm eps (1e-6, :24,:97) */
  const float* x;              /* [rows, D] contiguous layer input */
  const uint8_t* mask;         /* [nb, nh, L, L] 0 = masked_fill(-1e9) (:75-77) or NULL */
  /* parameters */
  const float* w_qs; const float* w_ks; const float* w_vs;   /* [nh*dk, D] */
  const float* fc;                                           /* [D, nh*dv] */
  const float* ln1_g; const float* ln1_b;                    /* slf_attn.layer_norm */
  const float* w1; const float* b1;                          /* [dff, D], [dff] */
  const float* w2; const float* b2;                          /* [D, dff], [D] */
  const float* ln2_g; const float* ln2_b;                    /* pos_ffn.layer_norm */
  /* saved by the forward for the backward (caller-allocated) */
  float* qkv;                  /* [rows, 2*nh*dk + nh*dv]  q | k | v */
  float* P;                    /* [nb, nh, L, L] attention (the module's second return value) */
  float* O;                    /* [rows, nh*dv] */
  float* y1; float* mean1; float* rstd1;    /* pre-norm sum and statistics of LayerNorm 1 */
  float* e1;                   /* [rows, D] MultiHeadAttention output */
  float* hdn;                  /* [rows, dff] relu(w_1 e1) */
  float* y2; float* mean2; float* rstd2;
  float* out;                  /* [rows, D] */
  /* backward only */
  const float* dout;           /* [rows, D] */
  float* dy2; float* dh; float* dy1; float* dO; float* dqkv;   /* scratch the weight-gradient phase reads */
  float* dx;                   /* [rows, D] gradient of the layer input (written) */
  float* g_w_qs; float* g_w_ks; float* g_w_vs; float* g_fc; float* g_ln1_g; float* g_ln1_b;   /* ACCUMULATED */
  float* g_w1; float* g_b1; float* g_w2; float* g_b2; float* g_ln2_g; float* g_ln2_b;
  /* Dropout (see "Dropout"; rng == NULL: identity): sites drop_site (attention :83, element = flat index of P), drop_site+1
   * (after fc :54, element row*D + col), drop_site+2 (after w_2 :106, same indexing).  With p_fc or p_ffn > 0 the backward
   * needs dt1 [rows, D] (gradient at the fc output; dy1 stays the residual's).  P keeps the plain softmax. */
  const uint32_t* rng;
  uint32_t drop_site;
  float p_attn, p_fc, p_ffn;
  float* dt1;
} mser_encoder_desc;

int mser_encoder_layer_supported(const mser_encoder_desc* d);
int mser_encoder_layer_fwd(const mser_encoder_desc* d, mser_stream_t stream);
int mser_encoder_layer_bwd(const mser_encoder_desc* d, int32_t phases, mser_stream_t stream);
/* The MSER_ENC_BWD_WGRAD products as descriptors (at most 6) for a caller that batches them with other weight gradients
 * into one mser_gemm_grouped launch; returns the number written (< 0: error).  Valid once MSER_ENC_BWD_ACT is enqueued. */
int mser_encoder_layer_wgrad_descs(const mser_encoder_desc* d, mser_gemm_desc* out, int32_t cap);

/* ------------------------------------------------------------------------------------------------
 * Sequence-level cross-modal attention core (model/lsthm_sps.py:88-101 CrossAttention2, :116-129 CrossAttention3, after the three
 * projection products): O = dropout(softmax(scale Q K^T)) V per (dialogue, head), fused: one workgroup per (32-query tile, head,
 * dialogue), scores and probabilities never leave LDS; the forward saves only the row statistics (max of the scaled logits,
 * 1 / sum).  The backward recomputes P from them; dq is written.  dk_ / dv receive contributions from every query tile of a dialogue:
 * part_stride == 0: ACCUMULATED with float atomics (zero them first; order-dependent rounding); part_stride != 0 (floats, a multiple
 * of 4): tile t STORES its contribution at dk_ / dv + t * part_stride (the caller provides ceil(Lq / 32) slabs, no zeroing) and a second
 * launch adds slabs 1.. onto slab 0 in a fixed order: bit-reproducible gradients.
 * q / k / v / o / dO / dq / dk_ / dv are row views [rows, ld] with head h in columns h*dk .. h*dk+dk-1 (dk == dv);
 * row(b, l) = b*sb + l*sl on the query side (sbq, slq) and on the key side (sbk, slk).  Supported: Lk <= 128, dk % 8 == 0,
 * dk <= 128 (mser_xattn_seq_supported); otherwise compose it from mser_gemm + mser_softmax_rows.
 * Dropout (:98,:126): element ((b*nh + h)*Lq + i)*Lk + j of site `site`; rng == NULL: identity.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mser_xattn_desc {
  int32_t nb, nh, Lq, Lk, dk;
  const float* q; int64_t ldq; const float* k; int64_t ldk; const float* v; int64_t ldv;
  int64_t sbq, slq, sbk, slk;
  float* o; int64_t ldo;                 /* forward: written; backward: read (delta = <dO, O>) */
  float* stats;                          /* [nb, nh, Lq, 2] */
  float scale;
  const uint32_t* rng; uint32_t site; float p;
  const float* dO; int64_t lddo;         /* backward */
  float* dq; int64_t lddq; float* dk_; int64_t lddk; float* dv; int64_t lddv;
  int64_t part_stride;                   /* backward: see above (0 = atomic accumulation) */
} mser_xattn_desc;
int mser_xattn_seq_supported(const mser_xattn_desc* d);
int mser_xattn_seq_fwd(const mser_xattn_desc* d, mser_stream_t stream);
int mser_xattn_seq_bwd(const mser_xattn_desc* d, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Classifier tail of the fusion head (model/lsthm_sps.py:390-393 after `self.fc`): y1r = y1 + x_l + x_a,
 * y2 = relu(nn_out.0(y1r)), y3 = nn_out.3(y2), lp[b*L+t] = log_softmax(y3[t*B+b]) -- one row-tiled launch, and its
 * whole backward (log_softmax, both Linear layers, both ReLUs incl. the one of `fc`, the residual fan-out) as one launch.
 * Dropout (:318,:323): identity when rng == NULL.  Otherwise the nn_out site is drawn here (site_out, p_out, element index
 * row*F + n) and the backward scales dy1 by 1/(1-p_fc): the caller has applied the fc site to y1 in place (mser_dropout_apply)
 * before the forward, so a dropped unit reads 0 and fails the ReLU test by itself.
 * Weight gradients are left to the caller (mser_gemm_grouped over dy3 / dy2 / dy1).
 * ------------------------------------------------------------------------------------------------ */
typedef struct mser_head_tail_desc {
  int32_t L, B, D, F, C;       /* rows = L*B time-major; D = d_l (100); F = nn_out hidden (32); C = classes */
  const float* y1;             /* [rows, D] relu(fc(h)) */
  const float* x_l; const float* x_a;            /* [rows, D] encoder outputs (residual, :390) */
  const float* w0; const float* b0;              /* nn_out.0 [F, D], [F] */
  const float* w3; const float* b3;              /* nn_out.3 [C, F], [C] */
  float* y1r; float* y2;       /* saved: [rows, D], [rows, F] */
  float* lp;                   /* [B*L, C] batch-major log-probabilities */
  /* backward */
  const float* dlp;            /* [B*L, C] */
  const float* dx_l_in; const float* dx_a_in;    /* optional [rows, D]: gradients of the returned x_l / x_a */
  float* dy3; float* dy2; float* dy1;            /* [rows, C], [rows, F], [rows, D] (dy1 = gradient at the fc output) */
  float* dx_l; float* dx_a;    /* [rows, D] written: d(y1r) (+ dx_*_in) */
  float* g_b0; float* g_b3; float* g_bfc;        /* ACCUMULATED bias gradients of nn_out.0, nn_out.3, fc.0 */
  const uint32_t* rng;         /* dropout state {seed, step} in device memory, or NULL */
  uint32_t site_out;
  float p_out, p_fc;
} mser_head_tail_desc;

int mser_head_tail_fwd(const mser_head_tail_desc* d, mser_stream_t stream);
int mser_head_tail_bwd(const mser_head_tail_desc* d, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sequence bookkeeping (model/lsthm_sps.py:396-409 _reverse_seq; :177 argmax; :238-259 _select_parties).
 * ------------------------------------------------------------------------------------------------ */
/* lens[b] = sum_t umask[b,t];  rev[t,b] = lens[b]-1-t if t < lens[b] else -1   (rev is int32 [L,B]) */
int mser_build_reverse_index(const float* umask, int32_t B, int32_t L, int32_t* lens, int32_t* rev,
                             mser_stream_t stream);
/* out[t,b,:D] = rev[t,b] >= 0 ? X[rev[t,b], b, :D] : 0 */
int mser_reverse_by_length(const float* X, int64_t ldx, const int32_t* rev, float* out, int64_t ldo, int32_t L,
                           int32_t B, int32_t D, mser_stream_t stream);
/* Slot tables for the speaker recurrence from qmask[T,B,2] (optionally read through `rev`, i.e. the reversed
 * dialogue): party[t,b] (argmax, ties -> 0), perm[t,r] = dialogue landing in row r of the (party-0 || party-1)
 * ordering, n0[t].  qm_out[t,b,2] receives the (possibly reversed) mask values used by the blend (:204-207). */
int mser_build_slot_tables(const float* qmask, const int32_t* rev, int32_t T, int32_t B, int32_t* party,
                           int32_t* perm, int32_t* n0, float* qm_out, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MARN_cell (model/lsthm_sps.py:132-221): speaker recurrence + LSTHM recurrence + per-step rank-1 attention.
 * One call runs `ndir` independent cells (forward / backward direction) in the same launches.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mser_cell_params {      /* one direction; pointers into the parameter (or gradient) storage */
  float* lsthm_W[2];  float* lsthm_Wb[2];   /* [4H,D], [4H]   index 0 = lsthm_l, 1 = lsthm_a  (:16) */
  float* lsthm_U[2];  float* lsthm_Ub[2];   /* [4H,H]                                          (:17) */
  float* lsthm_V[2];  float* lsthm_Vb[2];   /* [4H,H]                                          (:18) */
  float* lsthm_S[2];  float* lsthm_Sb[2];   /* [4H,Hs]                                         (:19) */
  float* q_Wih[2];    float* q_Whh[2];      /* [4Hs,Hs] nn.LSTMCell lstm_q0 / lstm_q1          (:147-148) */
  float* q_bih[2];    float* q_bhh[2];      /* [4Hs] */
  float* att_Wq;      float* att_Wk;        /* [H] crossatt_l2a.Wq / .Wk                       (:53-54) */
} mser_cell_params;

typedef struct mser_cell_dir {
  mser_cell_params p;          /* parameters */
  mser_cell_params g;          /* gradients (accumulated); used by the backward only */
  const float* qmask;          /* [T,B,2] natural time order */
  const int32_t* rev;          /* NULL for the forward direction; rev[t,b] (mser_build_reverse_index) for the backward */
  float* out;                  /* h rows written at natural time position: out[(tau*B+b)*ldo + 0..3H+Hs) = h_l|h_a|z|h_q */
  float* dout;                 /* gradient of `out`, same indexing (backward only) */
} mser_cell_dir;

typedef struct mser_cell_desc {
  int32_t T, B, D, H;          /* Hs == H (the reference feeds zeros(N,dh_q=dh_l) to LSTMCell(dh_s,dh_s), :147,:162) */
  int32_t ndir;                /* 1 or 2 */
  const float* x_l; int64_t ldxl;   /* [T*B, D] natural order, shared by the directions */
  const float* x_a; int64_t ldxa;
  float* dx_l; float* dx_a;    /* [T*B, D] contiguous, ACCUMULATED (backward only) */
  int64_t ldo;                 /* row stride of out / dout */
  mser_cell_dir dir[2];
  void* workspace;             /* mser_marn_cell_workspace_bytes(); holds everything saved for the backward */
  size_t workspace_bytes;
  /* optional (may be NULL): further contiguous [T*B, D] addends that MSER_PHASE_LSTHM_BWD_DX folds into dx_l / dx_a in the same
   * launch that adds the cell's own input gradients (the caller's partial sums from other branches of the backward graph) */
  const float* dx_l_add[2];
  const float* dx_a_add[2];
  /* Dropout inside the cell (see "Dropout" below; rng == NULL: identity).  Per direction i: site drop_site[i] = h_q0 / h_q1
   * (:183,:188; element ((t*2 + cell)*B + slot)*H + unit), drop_site[i]+1 = h_l / h_a (:211,:213; ((t*2 + stream)*B + b)*H + unit),
   * both with p_state[i]; drop_site[i]+2 = the rank-1 attention (:69; ((t*B + b)*H + i)*H + j) with p_attn[i].  t is the
   * direction's own time index.  The same rng words must be passed to every phase of the forward and the backward of one step. */
  const uint32_t* rng;
  uint32_t drop_site[2];
  float p_state[2];
  float p_attn[2];
  /* External speaker state (the GRU-speaker variants; NULL: the cell's own LSTM speaker chain).  ext_hq[i]
   * [T*B, H] (direction i's time order, complete before MSER_PHASE_LSTHM_FWD; e.g. mser_gru_speaker_fwd's hs, which can also fill
   * the h_q quarter of `out`) replaces h_q[t] as the LSTHM streams' speaker input (copied into the workspace by
   * MSER_PHASE_LSTHM_FWD, so the buffer may be reused afterwards): the speaker phases launch nothing, the speaker
   * parameters of `p` / `g` may be NULL and the p_state dropout of the h_q site is the caller's.  The backward's
   * MSER_PHASE_SPEAKER_BWD then writes ext_dhq[i] [T*B, H] = the total gradient at ext_hq[i] (output quarter included). */
  const float* ext_hq[2];
  float* ext_dhq[2];
  /* ext_linked != 0 (forward, persistent launch only): the rows are not complete when MSER_PHASE_LSTHM_FWD is issued -- a producer
   * kernel of the caller, already enqueued on ANOTHER stream and therefore running concurrently, writes them into the workspace
   * rows and advances the step counter that mser_marn_cell_ext_link returns (MSER_PHASE_FWD_PREP must precede that producer: it
   * zeroes the counter).  Never inside stream capture (a graph executor may start the consumer first; its waits are bounded but it
   * would give up).  ext_hq must still be non-NULL (it selects the mode); it is not read. */
  int32_t ext_linked;
  /* Sticky fault word in device memory (may be NULL): a persistent launch that gives up at a bounded wait ORs
   * MSER_FAULT_CHAIN_TIMEOUT into it.  Unlike the abort word inside the workspace (cleared by the next MSER_PHASE_FWD_PREP) it
   * stays set until the caller clears it; mser_adam_flat_dev skips its update while it is non-zero. */
  uint32_t* fault;
} mser_cell_desc;

size_t mser_marn_cell_workspace_bytes(int32_t T, int32_t B, int32_t D, int32_t H, int32_t ndir);
int mser_marn_cell_fwd(const mser_cell_desc* d, mser_stream_t stream);
int mser_marn_cell_bwd(const mser_cell_desc* d, mser_stream_t stream);
/* The same work in separately schedulable phases, so that the caller can overlap independent parts on different streams.
 *   forward : FWD_PREP (tables, initial states, counters) -> SPEAKER_FWD -> LSTHM_FWD
 *   backward: BWD_PREP -> LSTHM_BWD (BPTT chains) -> LSTHM_BWD_DX (dx_l, dx_a [, dHQ]) -> SPEAKER_BWD ;
 *             LSTHM_WGRAD any time after LSTHM_BWD
 * In persistent mode (MSER_OPT_PERSISTENT, H in {128,256}, all workgroups co-resident) both chains of a pass run inside ONE
 * fused launch issued by LSTHM_FWD / LSTHM_BWD (speaker and LSTHM workgroups linked by device-side step counters); SPEAKER_FWD
 * is then empty and SPEAKER_BWD only computes the speaker-cell parameter gradients.  Otherwise the chains are per-step launches
 * and SPEAKER_FWD (which needs only qmask) can overlap the encoders on another stream. */
enum { MSER_PHASE_SPEAKER_FWD = 1, MSER_PHASE_LSTHM_FWD = 2, MSER_PHASE_LSTHM_BWD = 4, MSER_PHASE_SPEAKER_BWD = 8,
       MSER_PHASE_LSTHM_BWD_DX = 16, MSER_PHASE_LSTHM_WGRAD = 32, MSER_PHASE_FWD_PREP = 64, MSER_PHASE_BWD_PREP = 128,
       /* modifier for SPEAKER_FWD and LSTHM_FWD (pass it to both): keep the forward chains as two persistent launches so that the
        * speaker chain, issued on another REAL stream, starts before the encoders finish.  Never inside stream capture. */
       MSER_PHASE_SEPARATE_SPEAKER = 256,
       /* modifier for FWD_PREP: also do the work of BWD_PREP (zeroed carries, accumulators and BPTT counters), so that ONE backward
        * over this forward may leave BWD_PREP out -- it otherwise sits between the head's backward and the BPTT launch. */
       MSER_PHASE_PREP_BOTH = 512,
       /* the hoisted input products x W^T of the LSTHM streams alone (they need only x_l / x_a, not FWD_PREP: a caller can run the
        * preparation on another stream beside them); LSTHM_FWD | PRE_DONE then goes straight to the chains. */
       MSER_PHASE_LSTHM_PRE = 1024, MSER_PHASE_PRE_DONE = 2048,
       /* the same, one input stream at a time: the products over x_l / over x_a alone (each can follow its own encoder branch) */
       MSER_PHASE_LSTHM_PRE_L = 4096, MSER_PHASE_LSTHM_PRE_A = 8192 };
/* Where a linked producer publishes direction `dir`'s speaker rows (hq_rows [T*B, H], inside the workspace) and the counter it
 * advances by per_step after each step (replicas x replica_stride words).  partner_wgs = the workgroups of the producer launch:
 * both kernels must be co-resident for the hand-off to progress.  Returns 1 if the persistent LSTHM launch will be used for these
 * sizes and partner_wgs further workgroups fit on the chip beside it (a link is possible), 0 if not, < 0 on error. */
int mser_marn_cell_ext_link(const mser_cell_desc* d, int32_t dir, int32_t partner_wgs, float** hq_rows, uint32_t** counter,
                            int32_t* replicas, int32_t* replica_stride, uint32_t* per_step);
/* The backward counterpart: where the BPTT launch leaves the gradient at the external speaker state while it runs -- *dhq [T*B, H]
 * (output quarter) plus *n_parts arrays at *dhq_parts, *part_stride floats apart -- and the counter that reaches
 * per_step * (T - t) when step t's rows are complete.  Returns 1 if the persistent BPTT launch will be used and partner_wgs further
 * workgroups (the consumer launch) fit beside it, 0 if not, < 0 on error.
 * A linked consumer must be enqueued on another stream after MSER_PHASE_BWD_PREP (which zeroes the counter); never inside capture. */
int mser_marn_cell_ext_link_bwd(const mser_cell_desc* d, int32_t dir, int32_t partner_wgs, const float** dhq, const float** dhq_parts,
                                int32_t* n_parts, int64_t* part_stride, uint32_t** counter, int32_t* replicas, int32_t* replica_stride,
                                uint32_t* per_step);
int mser_marn_cell_pipelined(int32_t B, int32_t H, int32_t ndir);
int mser_marn_cell_run(const mser_cell_desc* d, int32_t phases, mser_stream_t stream);

/* Launch mode of the recurrent chains.  MSER_OPT_PERSISTENT = 1 (default): each chain is ONE persistent launch with the time
 * loop inside (weights in registers, per-direction counter barriers, write-through hand-offs) whenever every workgroup can be
 * co-resident (H in {128,256}, workgroups <= CUs); 0: one launch per time step (always valid; used as the cross-check). */
/* MSER_OPT_WGRAD_INKERNEL = 1 (default): in persistent mode at H = 128 the weight gradients of the LSTHM streams and the speaker
 * cells (dW, dU, dV, dS, dW_ih, dW_hh) are accumulated by extra workgroups of the fused BPTT launch while the chains run
 * (registers, no atomics: deterministic); MSER_PHASE_LSTHM_WGRAD / SPEAKER_BWD then only add the bias sums.  The gradient
 * pointers of the descriptor must be the same in every phase of one backward pass.  0: grouped split-K GEMMs after the chains.
 * MSER_OPT_BPTT_KSPLIT = 1 (default): at H = 128 every matvec product of a BPTT step is reduced by two workgroups (K halves,
 * the consumers add the partials) when the doubled grid still fits the chip.
 * MSER_OPT_XCD_PLACEMENT = 0 (default; 1 = experimental): the fused persistent launches cover every CU and each 32-workgroup
 * chain group is formed by workgroups that are physically resident on two XCDs (cheaper counter barriers; the hand-off protocol
 * itself does not depend on the placement, unneeded workgroups leave at once).  Takes precedence over the K-split.  Measured
 * slower end to end (the concentrated groups starve the kernels that run beside the chains), hence off.
 * MSER_OPT_FWD_STATS_ROLES = 1 (default): in the fused forward launch at H = 128 the softmax statistics of the rank-1 attention
 * rows that the BPTT consumes are computed by extra workgroups following the chain's step counter instead of inside the row
 * phase of the chain.
 * MSER_OPT_FWD_SENTINEL = 1 (default): the persistent forward chains hand their state from workgroup to workgroup through
 * self-validating payload (the state arrays start as a sentinel bit pattern, consumers re-load until their words are final)
 * instead of counter barriers: one store->load trip per seam, no store drain, no atomics.  0: the counter barriers.
 * MSER_OPT_BWD_SENTINEL = 2 (default): both seams of the LSTHM BPTT step (gate gradients -> matvec roles, carry products -> next row
 * phase) and the hand-off to the speaker BPTT (speaker-state gradients in step-indexed arrays) self-validating as well; the chain's
 * counter then only advances (the weight-gradient roles and a linked consumer follow it).  1: the first seam keeps its counter
 * barrier.  0: counter barriers on both seams.
 * MSER_OPT_H256_SPLIT = 1 (default): persistent chains at H = 256 share every dialogue row's rank-1 attention between two workgroups
 * (forward: 128 query units each, backward: 128 keys of the transposed pass each) and K-split the BPTT products; the two dx products
 * then run as GEMMs after the chain (their workgroups are what the K-split needs).  0: one workgroup per row, no K-split.
 * MSER_OPT_SPK_BWD_KSPLIT = 1 (default): the speaker BPTT roles of the persistent launch split each step's product over its REDUCTION
 * index (a workgroup owns 16 hidden units of a cell: gate gradients and the dc carry local, partial products summed in a fixed order
 * by the next step's owner of each column) instead of over its output columns (every workgroup rebuilding the whole gate-gradient
 * tile).  0: the output-split form. */
enum { MSER_OPT_PERSISTENT = 1, MSER_OPT_WGRAD_INKERNEL = 2, MSER_OPT_BPTT_KSPLIT = 3, MSER_OPT_XCD_PLACEMENT = 4,
       MSER_OPT_FWD_STATS_ROLES = 5, MSER_OPT_FWD_SENTINEL = 6, MSER_OPT_BWD_SENTINEL = 7, MSER_OPT_H256_SPLIT = 8,
       MSER_OPT_SPK_BWD_KSPLIT = 9,
       /* value = n: with both seams of the LSTHM BPTT self-validating, n x 64 clocks pass between a workgroup's arrive and its first look
        * at the other workgroups' gate gradients (measured: they are usually visible at the first look already; default 0) */
       MSER_OPT_BWD_POLL_DELAY = 10,
       /* 1: cells wider than 512 (BASELINE configs[4]: hid = 1024) run every time step of a chain inside ONE launch per chain and pass (one
        * workgroup per CU, the per-step phases separated by counter barriers, hand-offs stored write-through and loaded through the
        * caches) instead of 3 + 4 launches per time step; 0 (default; the persistent launches hold every CU and serialise the GEMMs the
        * per-step launches overlap: measured slower end to end): one launch per phase and step */
       MSER_OPT_WIDE_PERSISTENT = 11,
       /* 1 (default): mser_drnn_fwd / mser_drnn_bwd run the whole time loop of both directions inside ONE launch per pass (one workgroup
        * per CU; the phases of a step are lists of 32-row tiles dealt to the workgroups and separated by grid barriers; the weights
        * stream from L2 / the infinity cache; hand-offs are stored write-through and loaded L2-bypassing); 0: the 9 + 13 launches per time
        * step of the first version (kept as the cross-check and as the fallback after a MSER_FAULT_CHAIN_TIMEOUT) */
       MSER_OPT_DRNN_PERSISTENT = 12 };
int mser_set_option(int32_t key, int32_t value);
/* Synchronises `stream` and reports whether a persistent kernel of the last fwd/bwd call on this workspace gave up at a
 * barrier (bounded spins; returns -2 and a message in that case).  Diagnostic; not needed on the hot path. */
int mser_marn_cell_status(const mser_cell_desc* d, mser_stream_t stream);

/* Single LSTHM1 step (model/lsthm_sps.py:28-44) and single rank-1 CrossAttention (:59-72) for the module-level API. */
int mser_lsthm_step_fwd(const float* x, const float* c, const float* h, const float* z, const float* s,
                        const float* W, const float* Wb, const float* U, const float* Ub, const float* V,
                        const float* Vb, const float* S, const float* Sb, float* c_out, float* h_out, float* gates,
                        int32_t B, int32_t D, int32_t H, int32_t Hz, int32_t Hs, mser_stream_t stream);
/* rng != NULL: the attention's Dropout (:69) with site `site`, probability p, element index (b*H + i)*H + j (see "Dropout"). */
int mser_rank1_attention_fwd(const float* x1, const float* x2, const float* Wq, const float* Wk, float* out,
                             int32_t B, int32_t H, const uint32_t* rng, uint32_t site, float p, mser_stream_t stream);
/* Their backward for the module-level API (training runs through the fused BPTT of mser_marn_cell_bwd instead).
 * lsthm_step_bwd: from d(c_t) / d(h_t) (either may be NULL) and the saved gates to the pre-activation gradients dgates [B,4H]
 * (order f,i,o,c~) and d(c_{t-1}); the products with W, U, V, S and the bias sums are mser_gemm / mser_colsum_acc calls.
 * rank1_attention_bwd: dx1, dx2 written, gWq / gWk [H] ACCUMULATED over the rows. */
int mser_lsthm_step_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dc_new, const float* dh_new,
                        float* dgates, float* dc_prev, int32_t B, int32_t H, mser_stream_t stream);
int mser_rank1_attention_bwd(const float* x1, const float* x2, const float* Wq, const float* Wk, const float* dout, float* dx1,
                             float* dx2, float* gWq, float* gWk, int32_t B, int32_t H, const uint32_t* rng, uint32_t site, float p,
                             mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Head: log_softmax + permute to batch-major (model/lsthm_sps.py:391-393) and MaskedLoss (loss.py:13-21).
 * ------------------------------------------------------------------------------------------------ */
/* lp[b*L+t, :C] = log_softmax(y[t*B+b, :C]) */
int mser_logsoftmax_tb_fwd(const float* y, float* lp, int32_t L, int32_t B, int32_t C, mser_stream_t stream);
/* dy[t*B+b,:] = dlp[b*L+t,:] - exp(lp[b*L+t,:]) * sum_c dlp[b*L+t,c] */
int mser_logsoftmax_tb_bwd(const float* dlp, const float* lp, float* dy, int32_t L, int32_t B, int32_t C,
                           mser_stream_t stream);
/* loss = -sum_r mask[r]*pred[r,target[r]] / sum(mask);  loss_out[0] = loss, loss_out[1] = sum(mask) */
int mser_masked_nll_fwd(const float* pred, const int64_t* target, const float* mask, int64_t rows, int32_t C,
                        float* loss_out, mser_stream_t stream);
/* The full MaskedLoss (loss.py:13-25): class weights (weight [C] or NULL) and both lossers of model_trainer.py:74-77.
 *   is_ce = 0: NLLLoss(weight, 'sum')(pred*mask, target) / D ;  is_ce = 1: CrossEntropyLoss(weight, 'sum')(pred*mask, target) / D
 *   D = sum(mask) (weight NULL) or sum(weight[target]*mask).  CrossEntropyLoss re-applies log_softmax to pred*mask, so a masked
 *   row adds weight[y]*log(C) to the numerator (the reference's default --loss CrossEntropy reports that on padded batches).
 * loss_out[0] = loss, loss_out[1] = D.  bwd: dpred = (*gscale_dev) * d loss / d pred (all columns written). */
int mser_masked_loss_fwd(const float* pred, const int64_t* target, const float* mask, const float* weight, int32_t is_ce,
                         int64_t rows, int32_t C, float* loss_out, uint32_t* fault, mser_stream_t stream);
/* (fault: optional sticky fault word.  A target of -100 without class weights is torch's ignore_index: no term, no gradient, the
 * row still counts in sum(mask).  Any other target outside [0, C) -- where NLLLoss / CrossEntropyLoss raise -- is skipped and ORs
 * MSER_FAULT_BAD_LABEL into *fault.  bwd: an empty shard (D == 0) yields a zero gradient, not NaN.) */
int mser_masked_loss_bwd(const float* pred, const int64_t* target, const float* mask, const float* weight, int32_t is_ce,
                         const float* loss_out, const float* gscale_dev, float* dpred, int64_t rows, int32_t C,
                         mser_stream_t stream);
/* dpred[r,c] = -(*gscale_dev) * mask[r] / sum(mask) * (c == target[r]) */
int mser_masked_nll_bwd(const int64_t* target, const float* mask, const float* loss_out, const float* gscale_dev,
                        float* dpred, int64_t rows, int32_t C, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The steps either side of the model in the trainer's loops.
 * ------------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------------
 * Speaker state of the GRU-speaker variants (model/lsthm_onlysp.py:170-181, the reference CLI's default
 * model): per dialogue b and step t (the direction's own time order)
 *   qs0 = q[b, argmax(qmask[t, b])] ; h_s = dropout(GRUCell(U_t, qs0)) ; q[b, p] = q[b, p] (1 - qmask[t,b,p]) + h_s qmask[t,b,p]
 * with gi = U W_ih^T + b_ih supplied by the caller (one mser_gemm over all steps).  One workgroup carries a 32-dialogue block
 * through the whole sequence (W_hh in registers, party states in LDS, no inter-workgroup hand-off).  H = 128.
 * Backward: dhs (+ up to two addends) is the gradient at hs from every consumer outside this recurrence; dgi / dgh are the
 * gradients at gi and at gh = qs0 W_hh^T + b_hh, from which the caller forms dW_ih = dgi^T U, db_ih = colsum(dgi),
 * dW_hh = dgh^T qs0 (qs0 = save[:, 0:H], row stride 5H), db_hh = colsum(dgh), dU = dgi W_ih with GEMMs.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mser_gru_speaker_desc {
  int32_t T, B, H;
  const float* gi;             /* [T*B, 3H] gates r | z | n */
  const float* w_hh;           /* [3H, H] */
  const float* b_hh;           /* [3H] */
  const float* qmask;          /* [T, B, 2] */
  float* hs;                   /* [T*B, H] written: the (dropped) speaker state fed to the LSTHM streams */
  float* out; int64_t ldo;     /* optional: h_s also goes to out[tau*B + b, 0:H], tau = rev ? rev[t*B+b] : t (skipped when < 0) */
  const int32_t* rev;
  float* save;                 /* mser_gru_speaker_save_bytes(): [T*B, 5H] = qs0 | r | z | n | W_hn qs0 + b_hn */
  /* backward */
  const float* dhs;            /* [T*B, H] */
  const float* dhs_add[2];     /* optional further addends of the same shape */
  float* dgi; float* dgh;      /* [T*B, 3H] written */
  const uint32_t* rng; uint32_t drop_site; float p;      /* dropout on h_s (:177), element (t*B + b)*H + u; NULL: identity */
  /* Forward link to a consumer that runs CONCURRENTLY (mser_cell_desc::ext_linked; values from mser_marn_cell_ext_link, hs then
   * being that call's hq_rows): after every step, once the step's hs rows are visible device-wide, each of pub_replicas counters
   * (pub_replica_stride words apart) is raised to pub_per_step * (number of steps that EVERY workgroup of the chain has
   * published); pub_progress: ceil(B / 16) words of scratch per chain, zeroed by the caller before the launch.  NULL: no link. */
  uint32_t* pub_counter; uint32_t pub_per_step; int32_t pub_replicas; int32_t pub_replica_stride; uint32_t* pub_progress;
  /* Backward link to a producer that runs CONCURRENTLY (the cell's BPTT launch; values from mser_marn_cell_ext_link_bwd): step t
   * starts once *sub_counter >= sub_per_step * (T - t); its incoming gradient is then dhs + the sub_nparts arrays at sub_parts
   * (sub_part_stride floats apart), read with device-coherent loads; dhs_add is ignored.  The wait is bounded: on a time-out the
   * kernel ORs MSER_FAULT_LINK_TIMEOUT into *status (optional sticky fault word in device memory) and carries on.  NULL: no link
   * (dhs / dhs_add are complete at launch). */
  const uint32_t* sub_counter; uint32_t sub_per_step; const float* sub_parts; int32_t sub_nparts; int64_t sub_part_stride;
  uint32_t* status;
  /* listener_blend != 0: the state update of model/lsthm_nsps.py:188-191 (MARN1_nsps / MARN1_no_en): both party states become
   * ql_0 (1 - qmask[t,b,p]) + h_s qmask[t,b,p] with ql_0 = q[b, 1 - argmax(qmask[t,b])], the state of the party NOT speaking --
   * identical to the default for a one-hot qmask row, different on padded (all-zero) rows.  hli (optional, [T*B, H]) receives the
   * ql_0 rows (the cell's h_li output), dhli (optional) is their incoming gradient in the backward. */
  int32_t listener_blend;
  float* hli;
  const float* dhli;
} mser_gru_speaker_desc;

size_t mser_gru_speaker_save_bytes(int32_t T, int32_t B, int32_t H);
/* d: array of n (1 or 2) descriptors with the same T and B -- the two directions of a bidirectional cell share one launch */
int mser_gru_speaker_fwd(const mser_gru_speaker_desc* d, int32_t n, mser_stream_t stream);
int mser_gru_speaker_bwd(const mser_gru_speaker_desc* d, int32_t n, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DialogueRNN (BASELINE configs[3]; model/DialogueRNN.py:80-198 as BiModel :201-277 and model_trainer.py:35-47 use
 * it: listener_state=True, context_attention='general').  One call runs BOTH DialogueRNNs of a BiModel (dialog_rnn_f on the input
 * as given, dialog_rnn_r on the per-dialogue reversed input, rev from mser_build_reverse_index) through shared launches and writes
 * the emotion states e[t] at their natural time rows: out[(tau*B + b)*ldo + dir*De ...], tau = t (dir 0) or rev[t,b] (dir 1, rows
 * beyond a dialogue's length are not written: the caller zeroes `out` first, pad_sequence semantics).
 * Per step: global GRU on [U_t | q[b,s_b]], 'general' MatchingAttention of W_att U_t over the history of global states, party GRU
 * for both parties on [U_t | c], listener GRU on [U_t | qs[b,s_b]], blend by qmask, emotion GRU on q[b,s_b].
 * ------------------------------------------------------------------------------------------------ */
typedef struct mser_drnn_params {      /* one DialogueRNN.dialogue_cell; pointers into the parameter (or gradient) storage */
  float* g_wih; float* g_whh; float* g_bih; float* g_bhh;     /* g_cell GRUCell(Dm+Dp, Dg): [3Dg, Dm+Dp], [3Dg, Dg], [3Dg], [3Dg]  (:92) */
  float* p_wih; float* p_whh; float* p_bih; float* p_bhh;     /* p_cell GRUCell(Dm+Dg, Dp)                                       (:93) */
  float* e_wih; float* e_whh; float* e_bih; float* e_bhh;     /* e_cell GRUCell(Dp, De)                                          (:94) */
  float* l_wih; float* l_whh; float* l_bih; float* l_bhh;     /* l_cell GRUCell(Dm+Dp, Dp)                                       (:96) */
  float* att_w;                                               /* attention.transform.weight [Dg, Dm] (no bias, :35)              */
} mser_drnn_params;

typedef struct mser_drnn_desc {
  int32_t T, B, Dm, Dg, Dp, De;
  const float* U; int64_t ldu;      /* [T*B, Dm] natural time order */
  const float* qmask;               /* [T, B, 2] */
  const int32_t* rev;               /* [T, B] (mser_build_reverse_index) */
  mser_drnn_params p[2];            /* dialog_rnn_f, dialog_rnn_r.  Corresponding tensors of the two directions may lie anywhere; */
  mser_drnn_params g[2];            /* gradients (ACCUMULATED), backward only                                                      */
  float* out; int64_t ldo;          /* [T*B, ldo >= 2 De] */
  const float* dout;                /* gradient of `out`, same indexing (backward only) */
  void* workspace; size_t workspace_bytes;      /* mser_drnn_workspace_bytes(); holds everything saved for the backward */
  /* the cell's nn.Dropout (:98; identity when rng == NULL): direction i draws sites drop_site[i] + {0: g (:136), 1: qs (:146),
   * 2: ql (:153), 3: e (:161)} with element indices (t*B + b)*Dg + u, ((t*B + b)*2 + party)*Dp + u (qs and ql), (t*B + b)*De + u,
   * t = the direction's own time index */
  const uint32_t* rng; uint32_t drop_site[2]; float p_drop;
  /* Sticky fault word in device memory (may be NULL): a persistent launch (MSER_OPT_DRNN_PERSISTENT) that gives up at a bounded
   * grid barrier ORs MSER_FAULT_CHAIN_TIMEOUT into it; its outputs and gradients are then invalid. */
  uint32_t* fault;
} mser_drnn_desc;

size_t mser_drnn_workspace_bytes(int32_t T, int32_t B, int32_t Dm, int32_t Dg, int32_t Dp, int32_t De);
int mser_drnn_fwd(const mser_drnn_desc* d, mser_stream_t stream);
int mser_drnn_bwd(const mser_drnn_desc* d, mser_stream_t stream);
/* alpha_f / alpha_b of BiModel.forward (:196,:240,:250) after mser_drnn_fwd: *alpha = [T][B][T] inside the workspace, row (t, b)
 * valid in its first t entries (direction `dir`'s own time order). */
int mser_drnn_alpha(const mser_drnn_desc* d, int32_t dir, const float** alpha);
/* MatchingAttention(att_type='general2') rows (:61-68): S [rows, n] holds <W x_t + b, M_s>; mask [rows / L, n] (row r uses mask row
 * r / L): in place  a_ = softmax(S * mask); alpha = a_ mask / sum(a_ mask).  bwd: dS from d alpha (written over dA), given the saved
 * alpha and the UNNORMALISED a_ is not needed: a_ = alpha * Z with Z recomputed from S -- the backward therefore takes the original
 * scores S0 as well. */
int mser_general2_rows_fwd(const float* S0, float* alpha, const float* mask, int64_t rows, int32_t n, int32_t L, mser_stream_t stream);
int mser_general2_rows_bwd(const float* S0, const float* mask, float* dA, int64_t rows, int32_t n, int32_t L, mser_stream_t stream);

/* Dropout.  A site's mask is a pure function of (rng[0] = seed, rng[1] = step, site, element index): nothing is stored between
 * the forward and the backward, both evaluate keep(idx) = mix32(idx ^ key(seed, step, site)) >= p * 2^32 (a full-avalanche
 * 32-bit mix; the streams of torch's CPU generators cannot be reproduced on a GPU, so train-mode parity is defined mask for
 * mask: mser_dropout_scale hands the factors to the checker).  The rank-1 attention sites (mser_cell_desc drop_site+2,
 * mser_rank1_attention_*) draw 16 bits per element -- elements 2w, 2w+1 share one mixed word, p resolved to 1/65536 -- because
 * their masks are evaluated inside the recurrent chains: read them back with draw_bits = 16, every other site with 32.  rng lives in device memory; mser_rng_advance increments the step
 * word on the device, so a captured graph draws fresh masks at every replay.
 *   mser_dropout_apply : x[r, c] *= keep(idx0 + r*cols + c) ? 1/(1-p) : 0   (in place; activations and gradients alike)
 *   mser_dropout_scale : out[e]   = keep(idx0 + e)          ? 1/(1-p) : 0 */
int mser_dropout_apply(float* x, int64_t rows, int32_t cols, int64_t ld, const uint32_t* rng, uint32_t site, float p,
                       uint32_t idx0, mser_stream_t stream);
int mser_dropout_scale(float* out, int64_t n, const uint32_t* rng, uint32_t site, float p, uint32_t idx0, int32_t draw_bits,
                       mser_stream_t stream);
int mser_rng_advance(uint32_t* rng, mser_stream_t stream);

/* Batch ingest (model_trainer.py:104-105, :138-139): x[r, :d_r] = (((r1+r2)+r3)+r4)/4, x[r, d_r:d_r+d_a] = acouf[r, :]; all
 * inputs contiguous [rows, d_r] / [rows, d_a], x contiguous [rows, d_r+d_a].  Bit-identical to the reference expression. */
int mser_ingest_features(const float* r1, const float* r2, const float* r3, const float* r4, const float* acouf, float* x,
                         int64_t rows, int32_t d_r, int32_t d_a, mser_stream_t stream);
/* Evaluation bookkeeping (model_trainer.py:142-156): pred[r] = argmax_c lp[r, c] (first maximum) and
 * conf[label[r]*C + pred[r]] += mask[r] (float64, accumulated across calls; the caller zeroes it).  accuracy_score and the
 * weighted f1_score with sample_weight = mask are functions of this C x C matrix.  pred_out (int64 [rows]) may be NULL. */
int mser_confusion_update(const float* lp, const int64_t* label, const float* mask, int64_t rows, int32_t C, double* conf,
                          int64_t* pred_out, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser over the flat parameter buffer: torch.optim.Adam(lr, weight_decay=wd) (model_trainer.py:82).
 * `live` (optional, uint8 per element) marks elements that own a gradient; dead parameters are skipped like
 * torch skips tensors whose .grad is None.  gscale multiplies the gradient first (data-parallel mean).
 * ------------------------------------------------------------------------------------------------ */
int mser_adam_flat(float* p, const float* g, float* m, float* v, const uint8_t* live, int64_t n, int32_t step,
                   float lr, float beta1, float beta2, float eps, float wd, float gscale, mser_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hook (bench.py): bracket every launch of ONE recurrent kernel with HIP events on its launch stream.
 * mser_prof_enable(id, max) arms it (id 0 disarms); mser_prof_collect returns the summed elapsed time and the number of
 * launches seen since the last collect.  Not for use inside hipGraph capture.
 * ------------------------------------------------------------------------------------------------ */
enum { MSER_PROF_SPK_FWD = 1, MSER_PROF_LSTHM_FWD_GATES = 2, MSER_PROF_LSTHM_FWD_Z = 3, MSER_PROF_LSTHM_BWD_ROW = 4,
       MSER_PROF_LSTHM_BWD_MAT = 5, MSER_PROF_SPK_BWD = 6,
       MSER_PROF_XATTN_FWD = 7, MSER_PROF_XATTN_BWD = 8,           /* mser_xattn_seq_fwd / bwd (csrc/xattn.hip) */
       MSER_PROF_ENC_ATTN_FWD = 9, MSER_PROF_ENC_ATTN_BWD = 10 };  /* the encoder's per-head attention launches (csrc/encoder.hip) */
int mser_prof_enable(int32_t kernel_id, int32_t max_launches);
int mser_prof_collect(float* total_ms, int32_t* launches);

/* Same update with the step counter, {lr, beta1, beta2} and the bias-correction scratch (2 floats) on the device, so the two
 * launches can be captured into a hipGraph and replayed.  The gradient is multiplied by gscale / (*gscale_div_dev)
 * (gscale_div_dev may be NULL): after the data-parallel all-reduce *gscale_div_dev is the global mask count. */
int mser_adam_flat_dev(float* p, const float* g, float* m, float* v, const uint8_t* live, int64_t n, int32_t* step_dev,
                       const float* hp_dev, float* sched_dev, float eps, float wd, const float* gscale_div_dev, float gscale,
                       const uint32_t* fault, const float* gfault, mser_stream_t stream);
/* (fault, gfault: optional.  While the sticky fault word *fault is non-zero, or the all-reduced fault flag *gfault is, the update
 * -- step counter included -- is skipped: a training step whose kernels reported a fault never reaches the weights.) */
/* Pack for the single data-parallel all-reduce (new: the reference has no distributed code):
 * buf[0..n) = g * (*cnt_dev), buf[n] = *cnt_dev, buf[n+1] = (*fault != 0) (fault may be NULL), so that after a SUM all-reduce
 * buf[0..n)/buf[n] is the gradient of the globally mask-weighted loss (loss.py:21 divides by the local mask count) and buf[n+1]
 * counts the ranks whose step faulted (every rank then skips the update together). */
int mser_dp_pack(float* buf, const float* g, const float* cnt_dev, int64_t n, const uint32_t* fault, mser_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSER_H_ */
