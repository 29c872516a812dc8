// DialogueRNN (reference model/DialogueRNN.py:80-198; SURVEY 8(f) row f2, BASELINE configs[3]) on gfx950: the two directions of a
// BiModel share every launch.  Per time step and direction (listener_state=True, 'general' attention over the growing history):
//   g  = dropout(GRU_g([U_t | q[b,s_b]], g_{t-1}))                      s_b = argmax(qmask[t,b])               (:129-136)
//   c  = sum_{s<t} softmax_s(<W_att U_t, g_s>) g_s   (zeros at t = 0)                                            (:137-141, :56-59,:75)
//   qs = dropout(GRU_p([U_t | c], q[b,p])), p = 0,1 ;  ql = dropout(GRU_l([U_t | qs[b,s_b]], q[b,p]))            (:144-153)
//   q  = ql (1 - qmask) + qs qmask ;  e = dropout(GRU_e(q[b,s_b], e_{t-1}))                                      (:156-161)
// Structure: the U-dependent halves of the four input products and W_att U are ONE GEMM each over all steps (hoisted); every weight
// gradient is a reduction over all (t, b) rows and runs as a few large GEMMs after the time loop.  The time loop itself has two forms:
//   * persistent (MSER_OPT_DRNN_PERSISTENT = 1, default; widths up to 512): ONE launch per pass runs all steps of both directions --
//     see "Persistent form" below (drnn_fwd_persist / drnn_bwd_persist).  Widths at the reference's configuration (D_g = D_p = 500,
//     21 MB of fp32 weights per direction) do not fit a register-resident chain the way the LSTHM cell does: the weights stream from
//     L2 / the infinity cache in MFMA fragment order, the fp32 MFMA (exact fmaf chains: the gate is 1e-4 on log-probs after 200
//     dependent steps) does 2.1 GFLOP per step and pass (12 us per step at the chip's rate).
//   * per step (option = 0; the cross-check and the fallback): a step is 8 direction-batched GEMMs + 4 gate epilogues + 1 history-
//     attention launch issued from the host loop (no Python between launches; capturable); the BPTT mirrors it (8 GEMMs + 5 launches).
#include "common.h"
#include "../../include/mser.h"
#include <cstring>

namespace mser {

int g_opt_drnn_persist = 1;                         // MSER_OPT_DRNN_PERSISTENT (set through mser_set_option, recurrent.hip)

int gemm(const mser_gemm_desc& d, hipStream_t s);   // gemm.hip
int gemm_group(const mser_gemm_desc* d, int n, hipStream_t s);

namespace {

struct Gate { float r, z, n, h; };
// gi: the input product (b_ih included unless `bih` is given), gh: the hidden product WITHOUT its bias
__device__ __forceinline__ Gate gru_gate(const float* gi, const float* gh, const float* bhh, int H, int u, float hprev, float& ghn_out,
                                         const float* bih = nullptr) {
  Gate g;
  const float b0 = bih ? bih[u] : 0.f, b1 = bih ? bih[H + u] : 0.f, b2 = bih ? bih[2 * H + u] : 0.f;
  g.r = sigmoidf_(gi[u] + b0 + gh[u] + bhh[u]);
  g.z = sigmoidf_(gi[H + u] + b1 + gh[H + u] + bhh[H + u]);
  const float ghn = gh[2 * H + u] + bhh[2 * H + u];
  g.n = tanhf(gi[2 * H + u] + b2 + g.r * ghn);
  g.h = (1.f - g.z) * g.n + g.z * hprev;
  ghn_out = ghn;
  return g;
}
// The hidden products gh (and the e cell's input product) are split-K accumulation targets of the next step's GEMMs: the one thread that
// reads an element clears it (no memset node per product and step).
__device__ __forceinline__ void clear3(float* g3, int H, int u) { g3[u] = 0.f; g3[H + u] = 0.f; g3[2 * H + u] = 0.f; }
// dh: gradient at h' (after undoing the dropout factor).  Returns the gate pre-activation gradients and the direct path to hprev.
struct GateGrad { float dar, daz, dan, danr, dhp; };
__device__ __forceinline__ GateGrad gru_gate_bwd(float dh, float r, float z, float n, float ghn, float hprev) {
  GateGrad o;
  o.dan = dh * (1.f - z) * (1.f - n * n);
  o.daz = dh * (hprev - n) * z * (1.f - z);
  o.dar = o.dan * ghn * r * (1.f - r);
  o.danr = o.dan * r;
  o.dhp = dh * z;
  return o;
}

struct Dims { int T, B, Dm, Dg, Dp, De; };
// fragment-ordered ("packed") operands of the persistent launches: K padded to a multiple of 16, rows to a multiple of 32
__host__ __device__ __forceinline__ int k8p(int K) { return ((K + 15) / 16) * 2; }
__host__ __device__ __forceinline__ long pack_rows(int rows, int K) { return (long)((rows + 31) / 32) * k8p(K) * 256; }

// Workspace, direction outermost ([2][...]): dir stride = the array's size / 2, so a step's slice of both directions is a
// batch-2 GEMM operand and the (t, b) rows of one direction have ONE stride (the weight-gradient reductions after the loop).
struct WS {
  float *Ud, *qm; int* idx;
  float *GIg, *GIp, *GIl, *Xatt;                 // hoisted products [2][T*B][3Dg | 3Dp | 3Dp | Dg]
  float *Gh, *Q, *Eh;                            // states [2][T+1][B][Dg], [2][T+1][B][2][Dp], [2][T+1][B][De]
  float *sv_g, *sv_p, *sv_l, *sv_e;              // gate saves [2][T][rows][4H]: r z n ghn
  float *q0sel, *ss, *qsel, *cvec, *alpha;       // [2][T][B][Dp] x3, [2][T][B][Dg], [2][T][B][T]
  float *gi_g, *gh_g, *gi_p, *gh_p, *gi_l, *gh_l, *gi_e, *gh_e;      // per-step scratch [2][rows][3H]
  // backward
  float *dgi_g, *dgh_g, *dgi_p, *dgh_p, *dgi_l, *dgh_l, *dgi_e, *dgh_e, *dXatt;   // [2][T][rows][3H] (dgi_p / dgi_l summed over parties)
  unsigned* sync;                                // the persistent launches' barrier counters (2 directions x 8 replicas, one line each) and abort word
  void* pk_dev;                                  // the persistent launches' parameter block (struct PK) in device memory
  // persistent forward: packed weights [2 dirs][8 products: g_in g_h p_in p_h l_in l_h e_in e_h] and packed states (offsets in floats
  // into apk): q0p [dir] | Ghp [parity][dir] | cvp [dir] | Qp [parity][dir][party] | ssp [dir] | qselp [parity][dir] | Ehp [parity][dir]
  float *wpk, *apk; long wpk_off[8], wpk_dir; long o_q0p, o_Ghp, o_cvp, o_Qp, o_ssp, o_qselp, o_Ehp; size_t apk_floats;
  // persistent backward: transposed weight packs (they reuse wpk: [2 dirs][8 products: e_ih e_hh l_ih l_hh p_ih p_hh g_ih g_hh], each
  // [column tile of 32][k / 8][32][8]) and ONE zero-initialised region bk per call, per direction (stride bk_dir):
  //   row-major [rows][N] exchange buffers: dEdir PEh | Pqsel | QdirL QdirP PQl PQp | Pq0 Pss dqs | Pc Ddir_g PGh
  //   packed gate gradients (the products' A operands): gi_l gh_l gi_e gh_e gi_p gh_p gi_g gh_g
  float* bk; long wtk_off[8], wtk_dir, bk_dir; size_t bk_floats;
  long b_dEdir, b_PEh, b_Pqsel, b_QdirL, b_QdirP, b_PQl, b_PQp, b_Pq0, b_Pss, b_dqs, b_Pc, b_Ddirg, b_PGh;
  long b_gil, b_ghl, b_gie, b_ghe, b_gip, b_ghp, b_gig, b_ghg;
  float *dGh, *dQ, *dEc, *dqsel, *dss, *dq0sel, *dc, *dqs;   // dGh [2][T+1][B][Dg]; dQ [2 ping-pong][2][B][2][Dp]; dEc [2][B][De]; ...
  size_t bytes;
};

struct Carver {
  char* base; size_t off;
  template <class T> T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

WS carve(char* base, const Dims& d) {
  Carver cv{base, 0};
  WS w;
  const size_t T = d.T, B = d.B, TB = T * B;
  w.Ud = cv.take<float>(2 * TB * d.Dm); w.qm = cv.take<float>(2 * TB * 2); w.idx = cv.take<int>(2 * (T + 1) * B);
  w.GIg = cv.take<float>(2 * TB * 3 * d.Dg); w.GIp = cv.take<float>(2 * TB * 3 * d.Dp); w.GIl = cv.take<float>(2 * TB * 3 * d.Dp);
  w.Xatt = cv.take<float>(2 * TB * d.Dg);
  w.Gh = cv.take<float>(2 * (T + 1) * B * d.Dg); w.Q = cv.take<float>(2 * (T + 1) * B * 2 * d.Dp); w.Eh = cv.take<float>(2 * (T + 1) * B * d.De);
  w.sv_g = cv.take<float>(2 * TB * 4 * d.Dg); w.sv_p = cv.take<float>(2 * TB * 2 * 4 * d.Dp); w.sv_l = cv.take<float>(2 * TB * 2 * 4 * d.Dp);
  w.sv_e = cv.take<float>(2 * TB * 4 * d.De);
  w.q0sel = cv.take<float>(2 * TB * d.Dp); w.ss = cv.take<float>(2 * TB * d.Dp); w.qsel = cv.take<float>(2 * TB * d.Dp);
  w.cvec = cv.take<float>(2 * TB * d.Dg); w.alpha = cv.take<float>(2 * TB * T);
  w.gi_g = cv.take<float>(2 * B * 3 * d.Dg); w.gh_g = cv.take<float>(2 * B * 3 * d.Dg);
  w.gi_p = cv.take<float>(2 * B * 3 * d.Dp); w.gh_p = cv.take<float>(2 * 2 * B * 3 * d.Dp);
  w.gi_l = cv.take<float>(2 * B * 3 * d.Dp); w.gh_l = cv.take<float>(2 * 2 * B * 3 * d.Dp);
  w.gi_e = cv.take<float>(2 * B * 3 * d.De); w.gh_e = cv.take<float>(2 * B * 3 * d.De);
  w.dgi_g = cv.take<float>(2 * TB * 3 * d.Dg); w.dgh_g = cv.take<float>(2 * TB * 3 * d.Dg);
  w.dgi_p = cv.take<float>(2 * TB * 3 * d.Dp); w.dgh_p = cv.take<float>(2 * TB * 2 * 3 * d.Dp);
  w.dgi_l = cv.take<float>(2 * TB * 3 * d.Dp); w.dgh_l = cv.take<float>(2 * TB * 2 * 3 * d.Dp);
  w.dgi_e = cv.take<float>(2 * TB * 3 * d.De); w.dgh_e = cv.take<float>(2 * TB * 3 * d.De);
  w.dXatt = cv.take<float>(2 * TB * d.Dg);
  w.dGh = cv.take<float>(2 * (T + 1) * B * d.Dg); w.dQ = cv.take<float>(2 * 2 * B * 2 * d.Dp); w.dEc = cv.take<float>(2 * B * d.De);
  w.dqsel = cv.take<float>(2 * B * d.Dp); w.dss = cv.take<float>(2 * B * d.Dp); w.dq0sel = cv.take<float>(2 * 2 * B * d.Dp);
  w.dc = cv.take<float>(2 * B * d.Dg); w.dqs = cv.take<float>(2 * B * 2 * d.Dp);
  w.sync = cv.take<unsigned>(4096);
  w.pk_dev = cv.take<char>(4096);
  {
    const int UWh = 10;                            // = UW (unit slab of the persistent cell tiles)
    const int Hs[4] = {d.Dg, d.Dp, d.Dp, d.De}, Kin[4] = {d.Dp, d.Dg, d.Dp, d.Dp};
    long o = 0;
    for (int c = 0; c < 4; ++c) {
      const long ns = (Hs[c] + UWh - 1) / UWh;
      w.wpk_off[2 * c] = o; o += ns * k8p(Kin[c]) * 256;
      w.wpk_off[2 * c + 1] = o; o += ns * k8p(Hs[c]) * 256;
    }
    w.wpk_dir = o;
    // transposed packs of the backward (same storage): product i = [K3 = 3 H rows][N columns]
    const int K3[8] = {3 * d.De, 3 * d.De, 3 * d.Dp, 3 * d.Dp, 3 * d.Dp, 3 * d.Dp, 3 * d.Dg, 3 * d.Dg};
    const int Nn[8] = {d.Dp, d.De, d.Dp, d.Dp, d.Dg, d.Dp, d.Dp, d.Dg};
    long ot = 0;
    for (int i = 0; i < 8; ++i) { w.wtk_off[i] = ot; ot += (long)((Nn[i] + 31) / 32) * k8p(K3[i]) * 256; }
    w.wtk_dir = ot;
    w.wpk = cv.take<float>((size_t)2 * (o > ot ? o : ot));
    {
      const long Bq = d.B;
      long a = 0;
      w.b_dEdir = a; a += Bq * d.De; w.b_PEh = a; a += Bq * d.De; w.b_Pqsel = a; a += Bq * d.Dp;
      w.b_QdirL = a; a += Bq * 2 * d.Dp; w.b_QdirP = a; a += Bq * 2 * d.Dp; w.b_PQl = a; a += Bq * 2 * d.Dp; w.b_PQp = a; a += Bq * 2 * d.Dp;
      w.b_Pq0 = a; a += Bq * d.Dp; w.b_Pss = a; a += Bq * d.Dp; w.b_dqs = a; a += Bq * 2 * d.Dp;
      w.b_Pc = a; a += Bq * d.Dg; w.b_Ddirg = a; a += Bq * d.Dg; w.b_PGh = a; a += Bq * d.Dg;
      a = (a + 63) & ~63L;
      w.b_gil = a; a += pack_rows(d.B, 3 * d.Dp); w.b_ghl = a; a += pack_rows(2 * d.B, 3 * d.Dp);
      w.b_gie = a; a += pack_rows(d.B, 3 * d.De); w.b_ghe = a; a += pack_rows(d.B, 3 * d.De);
      w.b_gip = a; a += pack_rows(d.B, 3 * d.Dp); w.b_ghp = a; a += pack_rows(2 * d.B, 3 * d.Dp);
      w.b_gig = a; a += pack_rows(d.B, 3 * d.Dg); w.b_ghg = a; a += pack_rows(d.B, 3 * d.Dg);
      w.bk_dir = a;
      w.bk_floats = (size_t)2 * a;
      w.bk = cv.take<float>(w.bk_floats);
    }
    const long prg = pack_rows(d.B, d.Dg), prp = pack_rows(d.B, d.Dp), pre = pack_rows(d.B, d.De);
    long a = 0;
    w.o_q0p = a; a += 2 * prp; w.o_Ghp = a; a += 4 * prg; w.o_cvp = a; a += 2 * prg; w.o_Qp = a; a += 8 * prp;
    w.o_ssp = a; a += 2 * prp; w.o_qselp = a; a += 4 * prp; w.o_Ehp = a; a += 4 * pre;
    w.apk_floats = (size_t)a;
    w.apk = cv.take<float>((size_t)a);
  }
  w.bytes = (cv.off + 255) & ~size_t(255);
  return w;
}

// ---- prep: party index and mask values in each direction's own time order ----------------------------------------------------------
// idx[d][t][b] = argmax(qmask_d[t,b]) (ties and padded rows -> 0); idx[d][T][b] = 0 (the "next step" of the last one)
__global__ void drnn_prep_kernel(const float* qmask, const int* rev, float* qm, int* idx, int T, int B) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long TB = (long)T * B;
  if (i >= 2 * (TB + B)) return;
  const int d = (int)(i / (TB + B));
  const long r = i % (TB + B);
  if (r >= TB) { idx[(long)d * (TB + B) + r] = 0; return; }
  float m0, m1;
  if (d == 0) { m0 = qmask[r * 2]; m1 = qmask[r * 2 + 1]; }
  else {
    const int src = rev[r];               // rev[t,b] = len_b - 1 - t, or -1 beyond the dialogue: zero rows (pad_sequence)
    const long b = r % B;
    m0 = src >= 0 ? qmask[((long)src * B + b) * 2] : 0.f;
    m1 = src >= 0 ? qmask[((long)src * B + b) * 2 + 1] : 0.f;
  }
  qm[((long)d * TB + r) * 2] = m0; qm[((long)d * TB + r) * 2 + 1] = m1;
  idx[(long)d * (TB + B) + r] = m1 > m0 ? 1 : 0;
}

__global__ void drnn_gather_rows_kernel(const float* U, long ldu, const int* rev, float* out, int T, int B, int D) {
  // out[t*B+b, :] = rev[t,b] >= 0 ? U[rev[t,b]*B + b, :] : 0      (rev == null: plain copy)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)T * B * D;
  if (i >= n) return;
  const long row = i / D; const int c = (int)(i % D);
  long src = row;
  if (rev) { const int s = rev[row]; src = s >= 0 ? (long)s * B + (row % B) : -1; }
  out[i] = src >= 0 ? U[src * ldu + c] : 0.f;
}

// q0sel[d][b] = Q[d][0][b][idx[d][0][b]] = 0 at t = 0: handled by zeroing.  Later steps' q0sel come out of the l epilogue.

struct StepF {
  Dims d; int t;
  // parameters with direction strides
  const float* bhh; long bhh_ds;
  const uint32_t* rng; uint32_t site[2]; float p;
};

// ---- g cell epilogue: (dir, b, u) ---------------------------------------------------------------------------------------------------
__global__ void drnn_g_fwd_kernel(int B, int H, const float* gi, long gi_ds, float* gh, const float* bhh, long bhh_ds, const float* hprev,
                                  float* hnew, long st_ds, float* save, long sv_ds, const uint32_t* rng, uint32_t site0, uint32_t site1,
                                  float p, uint32_t idx0) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  const long row = (long)dir * B + b;
  float ghn;
  const float hp = hprev[(long)dir * st_ds + (long)b * H + u];
  const Gate g = gru_gate(gi + (long)dir * gi_ds + (long)b * 3 * H, gh + row * 3 * H, bhh + dir * bhh_ds, H, u, hp, ghn);
  clear3(gh + row * 3 * H, H, u);
  float h = g.h;
  if (rng) h *= drop_scale(drop_key(rng, dir ? site1 : site0, p), idx0 + (uint32_t)e);
  hnew[(long)dir * st_ds + (long)b * H + u] = h;
  float* sv = save + (long)dir * sv_ds + (long)b * 4 * H + u;
  sv[0] = g.r; sv[H] = g.z; sv[2 * H] = g.n; sv[3 * H] = ghn;
}

// ---- p cell epilogue: (dir, b, u), both parties; also ss[d][b] = qs[b, idx] ------------------------------------------------------------
__global__ void drnn_p_fwd_kernel(int B, int H, const float* gi, long gi_ds, float* gh, const float* bhh, long bhh_ds, const float* Qt, long q_ds,
                                  float* qs, float* save, long sv_ds, const int* idx, long idx_ds, float* ss, long ss_ds,
                                  const uint32_t* rng, uint32_t site0, uint32_t site1, float p, uint32_t idx0) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  const int sp = idx[dir * idx_ds + b];
  DropKey dk;
  if (rng) dk = drop_key(rng, dir ? site1 : site0, p);
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const long row = ((long)dir * B + b) * 2 + pt;
    float ghn;
    const float hp = Qt[(long)dir * q_ds + ((long)b * 2 + pt) * H + u];
    const Gate g = gru_gate(gi + (long)dir * gi_ds + (long)b * 3 * H, gh + row * 3 * H, bhh + dir * bhh_ds, H, u, hp, ghn);
    clear3(gh + row * 3 * H, H, u);
    float h = g.h;
    if (rng) h *= drop_scale(dk, idx0 + (uint32_t)(((long)b * 2 + pt) * H + u));
    qs[row * H + u] = h;
    float* sv = save + (long)dir * sv_ds + ((long)b * 2 + pt) * 4 * H + u;
    sv[0] = g.r; sv[H] = g.z; sv[2 * H] = g.n; sv[3 * H] = ghn;
    if (pt == sp) ss[(long)dir * ss_ds + (long)b * H + u] = h;
  }
}

// ---- l cell epilogue + blend: Q[t+1] = ql (1 - m) + qs m ; qsel = Q[t+1][b, idx_t] ; q0next = Q[t+1][b, idx_{t+1}] -----------------------
__global__ void drnn_l_fwd_kernel(int B, int H, const float* gi, long gi_ds, float* gh, const float* bhh, long bhh_ds, const float* Qt, float* Qn,
                                  long q_ds, const float* qs, float* save, long sv_ds, const float* qm, long qm_ds, const int* idx,
                                  const int* idx_next, long idx_ds, float* qsel, long sel_ds, float* q0next, long q0n_ds, const uint32_t* rng,
                                  uint32_t site0, uint32_t site1, float p, uint32_t idx0) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  const int sp = idx[dir * idx_ds + b], sn = idx_next[dir * idx_ds + b];
  DropKey dk;
  if (rng) dk = drop_key(rng, dir ? site1 : site0, p);
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const long row = ((long)dir * B + b) * 2 + pt;
    float ghn;
    const float hp = Qt[(long)dir * q_ds + ((long)b * 2 + pt) * H + u];
    const Gate g = gru_gate(gi + (long)dir * gi_ds + (long)b * 3 * H, gh + row * 3 * H, bhh + dir * bhh_ds, H, u, hp, ghn);
    clear3(gh + row * 3 * H, H, u);
    float h = g.h;
    if (rng) h *= drop_scale(dk, idx0 + (uint32_t)(((long)b * 2 + pt) * H + u));
    float* sv = save + (long)dir * sv_ds + ((long)b * 2 + pt) * 4 * H + u;
    sv[0] = g.r; sv[H] = g.z; sv[2 * H] = g.n; sv[3 * H] = ghn;
    const float m = qm[(long)dir * qm_ds + (long)b * 2 + pt];
    const float qn = h * (1.f - m) + qs[row * H + u] * m;
    Qn[(long)dir * q_ds + ((long)b * 2 + pt) * H + u] = qn;
    if (pt == sp) qsel[(long)dir * sel_ds + (long)b * H + u] = qn;
    if (pt == sn) q0next[(long)dir * q0n_ds + (long)b * H + u] = qn;
  }
}

// ---- e cell epilogue: also writes the emotion row at its natural time position ---------------------------------------------------------
__global__ void drnn_e_fwd_kernel(int B, int H, float* gi, float* gh, const float* bih, const float* bhh, long bhh_ds, const float* hprev,
                                  float* hnew, long st_ds, float* save, long sv_ds, float* out, long ldo, const int* rev, int t,
                                  const uint32_t* rng, uint32_t site0, uint32_t site1, float p, uint32_t idx0) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  const long row = (long)dir * B + b;
  float ghn;
  const float hp = hprev[(long)dir * st_ds + (long)b * H + u];
  const Gate g = gru_gate(gi + row * 3 * H, gh + row * 3 * H, bhh + dir * bhh_ds, H, u, hp, ghn, bih + dir * bhh_ds);
  clear3(gi + row * 3 * H, H, u);
  clear3(gh + row * 3 * H, H, u);
  float h = g.h;
  if (rng) h *= drop_scale(drop_key(rng, dir ? site1 : site0, p), idx0 + (uint32_t)e);
  hnew[(long)dir * st_ds + (long)b * H + u] = h;
  float* sv = save + (long)dir * sv_ds + (long)b * 4 * H + u;
  sv[0] = g.r; sv[H] = g.z; sv[2 * H] = g.n; sv[3 * H] = ghn;
  const int tau = dir ? rev[(long)t * B + b] : t;
  if (tau >= 0) out[((long)tau * B + b) * ldo + (long)dir * H + u] = h;
}

// ---- attention over the history (:56-59,:75): one workgroup per (b, dir) ---------------------------------------------------------------
// scores_s = <x, g_s>, s = 0..t-1 (g_s = Gh[s+1]); alpha = softmax; c = sum alpha_s g_s.  LDS: x[Dg] | sc[t]
constexpr int ATT_NT = 1024;       // 16 waves: the score pass is a chain of dependent global-load round trips per wave, so many short chains
__global__ __launch_bounds__(ATT_NT) void drnn_attn_fwd_kernel(int B, int Dg, int T, int t, const float* Xatt_t, long x_ds, const float* Gh,
                                                               long gh_ds, float* alpha_t, long al_ds, float* c_t, long c_ds) {
  extern __shared__ float sm[];
  float* x = sm;
  float* sc = sm + Dg;
  __shared__ float red[ATT_NT / 64];
  const int b = blockIdx.x, dir = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int u = tid; u < Dg; u += ATT_NT) x[u] = Xatt_t[(long)dir * x_ds + (long)b * Dg + u];
  __syncthreads();
  const float* G = Gh + (long)dir * gh_ds + (long)b * Dg;               // g_s at G + (s + 1) * B * Dg
  const long gs = (long)B * Dg;
  // four history rows per wave and trip: their loads are independent and in flight together (one row per trip made the score pass a
  // chain of t/16 dependent round trips)
  for (int s = wave * 4; s < t; s += (ATT_NT / 64) * 4) {
    const float* g0 = G + (long)(s + 1) * gs;
    const float* g1 = G + (long)(min(s + 1, t - 1) + 1) * gs;
    const float* g2 = G + (long)(min(s + 2, t - 1) + 1) * gs;
    const float* g3 = G + (long)(min(s + 3, t - 1) + 1) * gs;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int u = lane; u < Dg; u += 64) {
      const float xv = x[u];
      a0 = fmaf(xv, g0[u], a0); a1 = fmaf(xv, g1[u], a1); a2 = fmaf(xv, g2[u], a2); a3 = fmaf(xv, g3[u], a3);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    if (lane == 0) {
      sc[s] = a0;
      if (s + 1 < t) sc[s + 1] = a1;
      if (s + 2 < t) sc[s + 2] = a2;
      if (s + 3 < t) sc[s + 3] = a3;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int s = tid; s < t; s += ATT_NT) mx = fmaxf(mx, sc[s]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int k = 1; k < ATT_NT / 64; ++k) mx = fmaxf(mx, red[k]);
  __syncthreads();
  float z = 0.f;
  for (int s = tid; s < t; s += ATT_NT) { const float ev = expf(sc[s] - mx); sc[s] = ev; z += ev; }
  z = wave_sum(z);
  if (lane == 0) red[wave] = z;
  __syncthreads();
  z = 0.f;
#pragma unroll
  for (int k = 0; k < ATT_NT / 64; ++k) z += red[k];
  const float rz = 1.f / z;
  for (int s = tid; s < t; s += ATT_NT) { const float a = sc[s] * rz; sc[s] = a; alpha_t[(long)dir * al_ds + (long)b * T + s] = a; }
  __syncthreads();
  // pooled vector: thread (u, part) sums its quarter of the history (independent loads, unrolled), the parts meet through atomics
  // on LDS-resident accumulators (x[] is free now)
  for (int u = tid; u < Dg; u += ATT_NT) x[u] = 0.f;
  __syncthreads();
  {
    const int NP = ATT_NT / 256;                          // history parts
    const int part = tid / 256, lu = tid % 256;
    const int s0 = (int)((long)t * part / NP), s1 = (int)((long)t * (part + 1) / NP);
    for (int u = lu; u < Dg; u += 256) {
      float acc = 0.f;
#pragma unroll 8
      for (int s = s0; s < s1; ++s) acc = fmaf(sc[s], G[(long)(s + 1) * gs + u], acc);
      atomicAdd(&x[u], acc);
    }
  }
  __syncthreads();
  for (int u = tid; u < Dg; u += ATT_NT) c_t[(long)dir * c_ds + (long)b * Dg + u] = x[u];
}
// backward: dalpha_s = <dc, g_s>; ds = alpha (dalpha - sum alpha dalpha); dx = sum ds_s g_s; dGh[s+1] += alpha_s dc + ds_s x
__global__ __launch_bounds__(ATT_NT) void drnn_attn_bwd_kernel(int B, int Dg, int T, int t, const float* Xatt_t, long x_ds, const float* Gh,
                                                               float* dGh, long gh_ds, const float* alpha_t, long al_ds, float* dc,
                                                               long c_ds, float* dX_t) {
  extern __shared__ float sm[];
  float* x = sm; float* dcv = sm + Dg; float* ds = sm + 2 * Dg; float* al = ds + T;
  __shared__ float red[ATT_NT / 64];
  const int b = blockIdx.x, dir = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (dc, like dqsel / dss / dq0sel, is a split-K accumulation target of the step's data-gradient GEMMs: its one reader clears it for the
  // next step, which saves a memset node per product and step)
  for (int u = tid; u < Dg; u += ATT_NT) {
    x[u] = Xatt_t[(long)dir * x_ds + (long)b * Dg + u];
    dcv[u] = dc[(long)dir * c_ds + (long)b * Dg + u];
    dc[(long)dir * c_ds + (long)b * Dg + u] = 0.f;
  }
  for (int s = tid; s < t; s += ATT_NT) al[s] = alpha_t[(long)dir * al_ds + (long)b * T + s];
  __syncthreads();
  const float* G = Gh + (long)dir * gh_ds + (long)b * Dg;
  float* dG = dGh + (long)dir * gh_ds + (long)b * Dg;
  const long gs = (long)B * Dg;
  for (int s = wave * 4; s < t; s += (ATT_NT / 64) * 4) {          // (four rows per trip, as in the forward)
    const float* g0 = G + (long)(s + 1) * gs;
    const float* g1 = G + (long)(min(s + 1, t - 1) + 1) * gs;
    const float* g2 = G + (long)(min(s + 2, t - 1) + 1) * gs;
    const float* g3 = G + (long)(min(s + 3, t - 1) + 1) * gs;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int u = lane; u < Dg; u += 64) {
      const float dv = dcv[u];
      a0 = fmaf(dv, g0[u], a0); a1 = fmaf(dv, g1[u], a1); a2 = fmaf(dv, g2[u], a2); a3 = fmaf(dv, g3[u], a3);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    if (lane == 0) {                          // dalpha_s
      ds[s] = a0;
      if (s + 1 < t) ds[s + 1] = a1;
      if (s + 2 < t) ds[s + 2] = a2;
      if (s + 3 < t) ds[s + 3] = a3;
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int s = tid; s < t; s += ATT_NT) dot = fmaf(al[s], ds[s], dot);
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = 0.f;
#pragma unroll
  for (int k = 0; k < ATT_NT / 64; ++k) dot += red[k];
  __syncthreads();
  for (int s = tid; s < t; s += ATT_NT) ds[s] = al[s] * (ds[s] - dot);
  __syncthreads();
  float* dxa = al + T;                          // [Dg] accumulator of dx
  for (int u = tid; u < Dg; u += ATT_NT) dxa[u] = 0.f;
  __syncthreads();
  {
    const int NP = ATT_NT / 256;
    const int part = tid / 256, lu = tid % 256;
    const int s0 = (int)((long)t * part / NP), s1 = (int)((long)t * (part + 1) / NP);
    for (int u = lu; u < Dg; u += 256) {
      float acc = 0.f;
      const float dcu = dcv[u], xu = x[u];
#pragma unroll 4
      for (int s = s0; s < s1; ++s) {
        const long o = (long)(s + 1) * gs + u;
        acc = fmaf(ds[s], G[o], acc);
        dG[o] += al[s] * dcu + ds[s] * xu;     // this workgroup owns every (s, b, dir) row it touches; each (s, u) has one thread
      }
      atomicAdd(&dxa[u], acc);
    }
  }
  __syncthreads();
  for (int u = tid; u < Dg; u += ATT_NT) dX_t[(long)dir * x_ds + (long)b * Dg + u] = dxa[u];
}

// ---- backward epilogues ------------------------------------------------------------------------------------------------------------------
// e cell: dh' = dout row (natural position) + carry; writes dgi / dgh rows of step t and the direct path into the carry.
__global__ void drnn_e_bwd_kernel(int B, int H, const float* dout, long ldo, const int* rev, int t, float* dEc, long ec_ds, const float* save,
                                  long sv_ds, const float* hprev, long st_ds, float* dgi, float* dgh, long dg_ds, const uint32_t* rng,
                                  uint32_t site0, uint32_t site1, float p, uint32_t idx0) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  const int tau = dir ? rev[(long)t * B + b] : t;
  float dh = dEc[(long)dir * ec_ds + (long)b * H + u];
  if (tau >= 0) dh += dout[((long)tau * B + b) * ldo + (long)dir * H + u];
  if (rng) dh *= drop_scale(drop_key(rng, dir ? site1 : site0, p), idx0 + (uint32_t)e);
  const float* sv = save + (long)dir * sv_ds + (long)b * 4 * H + u;
  const GateGrad g = gru_gate_bwd(dh, sv[0], sv[H], sv[2 * H], sv[3 * H], hprev[(long)dir * st_ds + (long)b * H + u]);
  float* o = dgi + (long)dir * dg_ds + (long)b * 3 * H + u;
  o[0] = g.dar; o[H] = g.daz; o[2 * H] = g.dan;
  float* o2 = dgh + (long)dir * dg_ds + (long)b * 3 * H + u;
  o2[0] = g.dar; o2[H] = g.daz; o2[2 * H] = g.danr;
  dEc[(long)dir * ec_ds + (long)b * H + u] = g.dhp;         // + dgh W_hh_e by the GEMM that follows
}

// l cell + blend: dQn (gradient at Q[t+1]) = dQnext + [p == idx_t] dqsel + [p == idx_{t+1}] dq0sel_next.
// d ql = dQn (1 - m), d qs_blend = dQn m (-> dqs); GRU backward of the l cell; dQcur = d ql z (direct path; the GEMM adds dgh W_hh).
__global__ void drnn_l_bwd_kernel(int B, int H, const float* dQn, float* dQc, long q_ds, float* dqsel, float* dq0n, long sel_ds,
                                  const int* idx, const int* idx_next, long idx_ds, const float* qm, long qm_ds, const float* save,
                                  long sv_ds, const float* Qt, long qh_ds, float* dgi, long dgi_ds, float* dgh, long dgh_ds, float* dqs,
                                  const uint32_t* rng, uint32_t site0, uint32_t site1, float p, uint32_t idx0, int has_next) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  const int sp = idx[dir * idx_ds + b], sn = idx_next[dir * idx_ds + b];
  DropKey dk;
  if (rng) dk = drop_key(rng, dir ? site1 : site0, p);
  float sr = 0.f, sz = 0.f, sna = 0.f;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const long qi = (long)dir * q_ds + ((long)b * 2 + pt) * H + u;
    float d = has_next ? dQn[qi] : 0.f;
    if (pt == sp) d += dqsel[(long)dir * sel_ds + (long)b * H + u];
    if (has_next && pt == sn) d += dq0n[(long)dir * sel_ds + (long)b * H + u];
    const float m = qm[(long)dir * qm_ds + (long)b * 2 + pt];
    dqs[((long)dir * B + b) * 2 * H + (long)pt * H + u] = d * m;
    float dh = d * (1.f - m);
    if (rng) dh *= drop_scale(dk, idx0 + (uint32_t)(((long)b * 2 + pt) * H + u));
    const float* sv = save + (long)dir * sv_ds + ((long)b * 2 + pt) * 4 * H + u;
    const GateGrad g = gru_gate_bwd(dh, sv[0], sv[H], sv[2 * H], sv[3 * H], Qt[(long)dir * qh_ds + ((long)b * 2 + pt) * H + u]);
    float* o2 = dgh + (long)dir * dgh_ds + ((long)b * 2 + pt) * 3 * H + u;
    o2[0] = g.dar; o2[H] = g.daz; o2[2 * H] = g.danr;
    sr += g.dar; sz += g.daz; sna += g.dan;
    dQc[qi] = g.dhp;
  }
  float* o = dgi + (long)dir * dgi_ds + (long)b * 3 * H + u;     // both parties share the input row [U_t | ss]
  o[0] = sr; o[H] = sz; o[2 * H] = sna;
  dqsel[(long)dir * sel_ds + (long)b * H + u] = 0.f;             // read-and-clear (this thread is the only reader of the element)
  dq0n[(long)dir * sel_ds + (long)b * H + u] = 0.f;
}

// p cell: d qs = dqs (blend) + [p == idx_t] dss; GRU backward; dQcur += d qs z
__global__ void drnn_p_bwd_kernel(int B, int H, const float* dqs, float* dss, long sel_ds, const int* idx, long idx_ds, float* dQc,
                                  long q_ds, const float* save, long sv_ds, const float* Qt, long qh_ds, float* dgi, long dgi_ds, float* dgh,
                                  long dgh_ds, const uint32_t* rng, uint32_t site0, uint32_t site1, float p, uint32_t idx0) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  const int sp = idx[dir * idx_ds + b];
  DropKey dk;
  if (rng) dk = drop_key(rng, dir ? site1 : site0, p);
  float sr = 0.f, sz = 0.f, sna = 0.f;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    float dh = dqs[((long)dir * B + b) * 2 * H + (long)pt * H + u];
    if (pt == sp) dh += dss[(long)dir * sel_ds + (long)b * H + u];
    if (rng) dh *= drop_scale(dk, idx0 + (uint32_t)(((long)b * 2 + pt) * H + u));
    const float* sv = save + (long)dir * sv_ds + ((long)b * 2 + pt) * 4 * H + u;
    const GateGrad g = gru_gate_bwd(dh, sv[0], sv[H], sv[2 * H], sv[3 * H], Qt[(long)dir * qh_ds + ((long)b * 2 + pt) * H + u]);
    float* o2 = dgh + (long)dir * dgh_ds + ((long)b * 2 + pt) * 3 * H + u;
    o2[0] = g.dar; o2[H] = g.daz; o2[2 * H] = g.danr;
    sr += g.dar; sz += g.daz; sna += g.dan;
    dQc[(long)dir * q_ds + ((long)b * 2 + pt) * H + u] += g.dhp;
  }
  float* o = dgi + (long)dir * dgi_ds + (long)b * 3 * H + u;
  o[0] = sr; o[H] = sz; o[2 * H] = sna;
  dss[(long)dir * sel_ds + (long)b * H + u] = 0.f;               // read-and-clear
}

// g cell: dh' = dGh[t+1]; direct path dGh[t] += dh' z
__global__ void drnn_g_bwd_kernel(int B, int H, const float* dGn, float* dGc, long g_ds, const float* save, long sv_ds, const float* hprev,
                                  float* dgi, float* dgh, long dg_ds, const uint32_t* rng, uint32_t site0, uint32_t site1, float p,
                                  uint32_t idx0) {
  const int dir = blockIdx.z;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * H) return;
  const int b = (int)(e / H), u = (int)(e % H);
  float dh = dGn[(long)dir * g_ds + (long)b * H + u];
  if (rng) dh *= drop_scale(drop_key(rng, dir ? site1 : site0, p), idx0 + (uint32_t)e);
  const float* sv = save + (long)dir * sv_ds + (long)b * 4 * H + u;
  const GateGrad g = gru_gate_bwd(dh, sv[0], sv[H], sv[2 * H], sv[3 * H], hprev[(long)dir * g_ds + (long)b * H + u]);
  float* o = dgi + (long)dir * dg_ds + (long)b * 3 * H + u;
  o[0] = g.dar; o[H] = g.daz; o[2 * H] = g.dan;
  float* o2 = dgh + (long)dir * dg_ds + (long)b * 3 * H + u;
  o2[0] = g.dar; o2[H] = g.daz; o2[2 * H] = g.danr;
  dGc[(long)dir * g_ds + (long)b * H + u] += g.dhp;
}

// ---- MatchingAttention 'general2' rows (:61-68): one wave per row ----------------------------------------------------------------------
constexpr int G2_WPB = 4;
__global__ __launch_bounds__(64 * G2_WPB) void general2_fwd_kernel(const float* S0, float* alpha, const float* mask, long rows, int n, int L) {
  const long row = (long)blockIdx.x * G2_WPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* sp = S0 + row * n;
  const float* mk = mask + (row / L) * n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, sp[j] * mk[j]);
  mx = wave_max(mx);
  float sum = 0.f, zm = 0.f;
  for (int j = lane; j < n; j += 64) { const float e = expf(sp[j] * mk[j] - mx); sum += e; zm += e * mk[j]; }
  sum = wave_sum(sum); zm = wave_sum(zm);
  // alpha = (e / sum) m / (zm / sum) = e m / zm
  const float inv = 1.f / zm;
  for (int j = lane; j < n; j += 64) alpha[row * n + j] = expf(sp[j] * mk[j] - mx) * mk[j] * inv;
}
// dA (in: d alpha, out: d S0)
__global__ __launch_bounds__(64 * G2_WPB) void general2_bwd_kernel(const float* S0, const float* mask, float* dA, long rows, int n, int L) {
  const long row = (long)blockIdx.x * G2_WPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* sp = S0 + row * n;
  const float* mk = mask + (row / L) * n;
  float* da = dA + row * n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, sp[j] * mk[j]);
  mx = wave_max(mx);
  float sum = 0.f, zm = 0.f;
  for (int j = lane; j < n; j += 64) { const float e = expf(sp[j] * mk[j] - mx); sum += e; zm += e * mk[j]; }
  sum = wave_sum(sum); zm = wave_sum(zm);
  const float Z = zm / sum;                    // sum_j a_j m_j
  float dot = 0.f;                             // sum_k dA_k alpha_k
  for (int j = lane; j < n; j += 64) dot = fmaf(da[j], expf(sp[j] * mk[j] - mx) * mk[j] / zm, dot);
  dot = wave_sum(dot);
  float dot2 = 0.f;                            // sum_k a_k da_k,  da_k = m_k (dA_k - dot) / Z
  for (int j = lane; j < n; j += 64) dot2 = fmaf(expf(sp[j] * mk[j] - mx) / sum, mk[j] * (da[j] - dot) / Z, dot2);
  dot2 = wave_sum(dot2);
  for (int j = lane; j < n; j += 64) {
    const float a = expf(sp[j] * mk[j] - mx) / sum;
    const float daj = mk[j] * (da[j] - dot) / Z;
    da[j] = a * (daj - dot2) * mk[j];
  }
}

// =====================================================================================================================================
// Persistent form (MSER_OPT_DRNN_PERSISTENT): ONE launch per pass runs the whole time loop of both directions.  One workgroup per CU;
// a step of a direction is a short list of PHASES, a phase is a list of independent tasks dealt to the workgroups, consecutive phases of
// a direction are separated by a grid barrier (one monotonic counter per direction, split into arrive / wait: every workgroup alternates
// between the two directions, so one direction's barrier completes while the workgroup works for the other).  A GRU-cell task owns
// 32 dialogue rows x UW hidden units x 3 gates: it streams its slab of W_ih / W_hh (L2 / infinity cache; the two row blocks of a slab
// run on the same XCD), multiplies on the fp32 MFMA with the 8 waves splitting K, and applies the gate math itself -- no split-K
// atomics, no gate-product scratch, no epilogue launch.  Everything one workgroup hands to another inside the launch is stored
// write-through and loaded L2-bypassing (`sc1` buffer accesses through one descriptor per array), the counter add sits behind
// s_waitcnt vmcnt(0) + a workgroup barrier (MI355X_MICROARCH.md "valid forms").  Every wait is bounded: a workgroup that gives up sets
// the abort word and MSER_FAULT_CHAIN_TIMEOUT, and the whole grid drains.
// Measured (configs[3], B = 64, T = 200, rocprofv3): forward 11.6 ms, backward 13.4 ms per launch (58 / 67 us per step, of which the MFMA
// chains are 12 us per CU: a task is a latency chain -- L2-bypassing operand loads 3 us, MFMA 5 us, cross-wave reduction 1-2 us, gate
// math 1.5 us, store drain + arrive 1.5 us -- with one task in flight per CU); the per-step launches took 20 + 40 ms.  DESIGN.md 7 (f2)
// lists what each step of the way was worth.
#ifndef MSER_DRNN_WAVES
#define MSER_DRNN_WAVES 8
#endif
// 8 waves: one workgroup per CU.  (4 waves = TWO workgroups per CU, one per direction chain, so that one chain's MFMA work could fill the
// other's load / reduction / barrier time: measured SLOWER, forward 26 ms and backward 31 ms against 18 ms each -- a wave's K share
// and with it every task's latency chain doubles, and the step is a chain of task latencies, not a throughput problem.)
// UW units x 3 gates = 30 of a product tile's 32 columns
constexpr int PNW = MSER_DRNN_WAVES, PNT = 64 * PNW, UW = 10, P_WGS_PER_CU = 8 / PNW;
constexpr int P_NIT = (32 * 2 * UW + PNT - 1) / PNT;      // epilogue items per thread (32 rows x 2 parties x UW units)
constexpr int P_SC1 = 16;
constexpr int P_TS = 36, P_T1 = 32 * P_TS;                  // a product tile in LDS: [column][row], row stride padded to 36 (16-byte vector accesses)
constexpr int P_RED1 = PNW * P_T1, P_TILES = 3 * P_T1;
// dynamic LDS (float offsets): 3 x P_RED1 (the waves' partial tiles of up to 3 products) | P_TILES | per-kernel rest (attention)
constexpr int P_OFF_TILES = 3 * P_RED1, P_OFF_ATT = 3 * P_RED1 + P_TILES;
constexpr int B_RED = (P_RED1 > PNW * 512 ? P_RED1 : PNW * 512), B_OFF_REDW = B_RED, B_OFF_ATT = B_RED + 64;     // the backward's: red (>= the attention's [PNW][512] sums) | 64 | attention state
constexpr unsigned P_SPIN_LIMIT = 1u << 21;
// true: each half of the grid runs ONE direction's chain (the two chains advance concurrently, a phase's tasks take up to two rounds on
// 128 workgroups); false: every workgroup alternates between the directions (a phase is one round, the chains' phases are serialised)
constexpr bool P_SPLIT_DIRS = false;           // (measured equal at configs[3]: 47.0 ms per step either way)
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned int pgu32;

struct XB { __amdgpu_buffer_rsrc_t r; const void* p; unsigned bytes; };
__device__ __forceinline__ XB xb_make(const void* p, size_t elems) {
  XB b;
  b.p = p; b.bytes = (unsigned)(elems * sizeof(float));
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, b.bytes, 0x00020000);
  return b;
}
// inside a non-inlined task function the descriptor arrives in vector registers (the compiler would wrap every access in a
// readfirstlane loop): rebuild it from values declared wave-uniform
__device__ __forceinline__ XB xb_uni(const XB& x) {
  const unsigned long long a = (unsigned long long)x.p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned nb = __builtin_amdgcn_readfirstlane(x.bytes);
  XB b;
  b.p = (const void*)(((unsigned long long)hi << 32) | lo); b.bytes = nb;
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(b.p), 0, nb, 0x00020000);
  return b;
}
__device__ __forceinline__ float xb_ld(const XB& b, long e) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.r, (int)(e * 4), 0, P_SC1));
}
__device__ __forceinline__ void xb_st(const XB& b, long e, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), b.r, (int)(e * 4), 0, P_SC1);
}
__device__ __forceinline__ void pzero8(float* a) {
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
}
#ifdef MSER_STAMPS
__shared__ unsigned long long pst_acc[16];
__shared__ unsigned long long pst_last;
#define PST_INIT() do { if (threadIdx.x == 0) { for (int _i = 0; _i < 16; ++_i) pst_acc[_i] = 0; pst_last = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define PST(k) do { if (threadIdx.x == 0) { const unsigned long long _n = __builtin_amdgcn_s_memrealtime(); pst_acc[k] += _n - pst_last; pst_last = _n; } } while (0)
#ifdef MSER_STAMPS_FINE
#define PSTC(k) PST(k)
#else
#define PSTC(k)
#endif
#define PST_DUMP(name, T) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 200 || blockIdx.x == 255 || blockIdx.x == 256)) \
  printf("[drnn stamps %s wg %3d hwid %08x | 10 ns ticks per step]  %llu %llu %llu %llu %llu %llu %llu %llu | %llu %llu %llu %llu %llu %llu %llu %llu\n", name, (int)blockIdx.x, (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)), pst_acc[0] / (T), pst_acc[1] / (T), \
         pst_acc[2] / (T), pst_acc[3] / (T), pst_acc[4] / (T), pst_acc[5] / (T), pst_acc[6] / (T), pst_acc[7] / (T), pst_acc[8] / (T), pst_acc[9] / (T), pst_acc[10] / (T), pst_acc[11] / (T), \
         pst_acc[12] / (T), pst_acc[13] / (T), pst_acc[14] / (T), pst_acc[15] / (T)); } while (0)
#else
#define PST_INIT()
#define PST(k)
#define PSTC(k)
#define PST_DUMP(name, T)
#endif

// Grid barrier in two halves, one counter per direction chain (monotonic; `target` counts the arrivals expected so far).  arrive():
// the workgroup's stores are complete (s_waitcnt vmcnt(0) by every thread, then a workgroup barrier), one relaxed agent-scope add.
// wait(): bounded poll.  Between the two a workgroup works for the OTHER direction, so a barrier's latency is not idle time.
// wait() returns false when some workgroup gave up (the caller returns; every workgroup does, so the grid drains).
#ifndef MSER_DRNN_BAR_REP
#define MSER_DRNN_BAR_REP 8
#endif
#ifndef MSER_DRNN_BAR_SLEEP
#define MSER_DRNN_BAR_SLEEP 1
#endif
constexpr int P_BAR_REP = MSER_DRNN_BAR_REP;
struct GridBar { unsigned* cnt; unsigned* abortw; uint32_t* fault; unsigned target, G; };
__device__ __forceinline__ void bar_arrive(GridBar& gb) {
  gb.target += gb.G;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // P_BAR_REP replicas of the counter, one cache line each: an arrival adds to all of them with ONE wave instruction (P_BAR_REP active
  // lanes), a waiting workgroup polls only replica (workgroup % P_BAR_REP) -- 256 pollers on one line queue the arrivals behind their
  // loads (measured: configs[3] step 43.5 -> 40.7 ms)
  if (threadIdx.x < P_BAR_REP) __hip_atomic_fetch_add((pgu32*)(gb.cnt + threadIdx.x * 32), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool bar_wait(const GridBar& gb, int* ok_lds) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    int ok = 1;
    const pgu32* mine = (const pgu32*)(gb.cnt + (blockIdx.x % P_BAR_REP) * 32);
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gb.target) {
      __builtin_amdgcn_s_sleep(MSER_DRNN_BAR_SLEEP);
      ++spins;
      if ((spins & 1023u) == 0u && __hip_atomic_load((const pgu32*)gb.abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      if (spins > P_SPIN_LIMIT) {
        __hip_atomic_store((pgu32*)gb.abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gb.fault) atomicOr(gb.fault, (uint32_t)MSER_FAULT_CHAIN_TIMEOUT);
        ok = 0;
        break;
      }
    }
    *ok_lds = ok;
  }
  __syncthreads();
  return *ok_lds != 0;
}

struct PK {
  Dims d;
  mser_drnn_params p[2];       // parameters
  WS w;
  const uint32_t* rng; uint32_t site[2]; float pdrop;
  float* out; const float* dout; long ldo; const int* rev;
  unsigned* sync; uint32_t* fault;
  // backward: host-computed task schedule (make_bwd_sched).  sched[ph][w][k]: the (up to two) tasks of phase ph (0 = C, 1 = D, 2 = F) that
  // the workgroup with direction-relative index w runs, as indices into the phase's full task list; 255 = none.  sched_ok = 0: deal the
  // lists round-robin instead.
  unsigned char sched[3][256][2]; int sched_ok;
  unsigned char sched_f[2][256][2]; int sched_f_ok;      // forward: phases B, C (make_fwd_sched)
};
static_assert(sizeof(PK) + 16 <= 4096, "PK travels as a by-value kernel argument of drnn_store_pk_kernel");
// The launch parameters live in device memory (drnn_store_pk_kernel writes the by-value argument there -- capturable, no host copy) and are
// read through the constant address space: scalar loads wherever a field is needed, instead of ~100 preloaded SGPRs spilled all over the
// kernel, and a plain pointer to hand to the non-inlined task functions.
typedef const __attribute__((address_space(4))) PK CPK;
__global__ void drnn_store_pk_kernel(const PK k, PK* dst) {
  static_assert(sizeof(PK) % 4 == 0, "copied word by word");
  const unsigned* src = reinterpret_cast<const unsigned*>(&k);
  unsigned* d = reinterpret_cast<unsigned*>(dst);
  for (unsigned i = threadIdx.x; i < sizeof(PK) / 4; i += blockDim.x) d[i] = src[i];
}
__device__ __forceinline__ const CPK& pk_uni(const CPK* p) {        // (inside a non-inlined function the pointer arrives in vector registers)
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return *(const CPK*)(((unsigned long long)hi << 32) | lo);
}

// ---- packed operands -------------------------------------------------------------------------------------------------------------------
// An MFMA fragment is 8 consecutive k of one row; lane r of a wave holds row r.  Read straight from a row-major matrix a wave's load
// touches 32 rows x 32 bytes (32 cache lines for 1 KB of payload: measured 20 us per p-cell tile).  Both operands of the per-step products
// are therefore kept in FRAGMENT ORDER  [row block of 32][k / 8][row % 32][8]:  a wave's load is 1 KB contiguous, K is zero-padded to a
// multiple of 16 (no tail masks).  Weights are packed once per call (drnn_pack_w_kernel: [unit slab][k / 8][gate x UW + unit][8], columns
// 30, 31 zero); the states are written in this order by the epilogue that produces them (and row-major as well, for everything that
// reads them after the launch).
__device__ __forceinline__ long pk_off(int b, int k, int K) { return (((long)(b >> 5) * k8p(K) + (k >> 3)) * 32 + (b & 31)) * 8 + (k & 7); }

__global__ void drnn_pack_w_kernel(const float* W, long ldw, int H, int K, float* out, long n_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int j = (int)(i & 7), n = (int)((i >> 3) & 31);
  const long q = i >> 8;
  const int k8n = k8p(K);
  const int k = (int)(q % k8n) * 8 + j, slab = (int)(q / k8n);
  const int gate = n / UW, u = slab * UW + n - gate * UW;
  out[i] = (n < 3 * UW && u < H && k < K) ? W[(long)(gate * H + u) * ldw + k] : 0.f;
}

template <int NA, int UN, class AL, class BL>
__device__ __forceinline__ void pmm(int Kp, AL aload, BL bload, f32x16* acc) {        // Kp: padded K (multiple of 16)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int KC = ((Kp + PNW * 16 - 1) / (PNW * 16)) * 16;
  const int kbeg = wave * KC, kend = min(Kp, kbeg + KC);
  for (int kb = kbeg; kb < kend; kb += 16 * UN) {
    float a[NA][UN][8], b[UN][8];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k8 = ((kb + 16 * u) >> 3) + half;
      if (kb + 16 * u < kend) {
        bload(r, k8, b[u]);
#pragma unroll
        for (int i = 0; i < NA; ++i) aload(i, r, k8, a[i][u]);
      } else {
        pzero8(b[u]);
#pragma unroll
        for (int i = 0; i < NA; ++i) pzero8(a[i][u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u][j], b[u][j], acc[i], 0, 0, 0);
  }
}
// fragment loads through a descriptor: e = element offset of the fragment (multiple of 8: the packs are 256-byte aligned), or < 0 for a
// pass this wave does not have -- the offset then lies beyond the descriptor's range and the load returns zeros (no branch).  SC1 = true:
// hand-off data (L2-bypassing); false: weights (cached).
template <bool SC1>
__device__ __forceinline__ void xb_frag(const XB& b, long e, float* a) {
  const int off = e >= 0 ? (int)(e * 4) : (int)0xFFFFFF00;
  const pu32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(b.r, off, 0, SC1 ? P_SC1 : 0);
  const pu32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(b.r, off, 16, SC1 ? P_SC1 : 0);
  a[0] = __uint_as_float(v0.x); a[1] = __uint_as_float(v0.y); a[2] = __uint_as_float(v0.z); a[3] = __uint_as_float(v0.w);
  a[4] = __uint_as_float(v1.x); a[5] = __uint_as_float(v1.y); a[6] = __uint_as_float(v1.z); a[7] = __uint_as_float(v1.w);
}

// a phase's tile list of one direction: (row block, unit slab); the row blocks of one slab are 8 apart in the list, so that with a grid
// that is a multiple of 8 they run on the same XCD and share the weight slab in its L2
__device__ __forceinline__ bool tile_decode(int j, int NRB, int NS, int& rb, int& slab) {
  const int q = j / (8 * NRB), rem = j - q * 8 * NRB;
  rb = rem >> 3;
  slab = q * 8 + (rem & 7);
  return slab < NS;
}
__host__ __device__ __forceinline__ int tile_count(int rows, int H, int uw) {
  const int NRB = (rows + 31) / 32, NS = (H + uw - 1) / uw;
  return ((NS + 7) / 8) * 8 * NRB;
}

struct FB { XB Gh, Q, Eh, qs, apk, wpk; };     // the forward's hand-off arrays (row-major states + the packed operand region)

// ---- a GRU-cell tile: 32 dialogue rows x UW units x 3 gates of one direction -------------------------------------------------------------
// CELL 0 = g (rows b), 1 = p, 2 = l (rows b, both parties), 3 = e (rows b).  A task is split in two halves around the grid barrier that
// precedes its phase: `pre` issues the loads that do not depend on the phase just before it (the weight fragments of both products and the
// epilogue's operands: 78 registers) so that their latency passes while the workgroup waits at the barrier; `post` loads the state
// fragments, runs the MFMA chains, reduces the waves' partial tiles and applies the gate math.  Widths up to 512: a wave's K share of a
// product is at most 4 passes of 16, all held in registers (8 waves per workgroup: 256 registers per lane).
struct Task { int kind, t, dir, rb, slab, b; };      // kind 0..3 = CELL, 4 = attention (b, dir), -1 = none
struct TRegs {
  float bi[4][8], bh[4][8];          // weight fragments: input product, hidden product
};
struct ERegs {
  float hp[P_NIT], gi[P_NIT][3], bb[P_NIT][3], qs[P_NIT];   // epilogue operands of this thread's items (qs: the l cell's blend partner)
  float qm[P_NIT]; int sp[P_NIT], sn[P_NIT];                // (p, l cells) the item's party mask value and the speaker indices of steps t, t + 1
};
struct CellGeom {
  int H, Kin, NPT;
  const float *bhh, *bih, *GI;
  long Wi, Wh;                        // element offsets into the packed weights
  const XB* bh;
  long h0, ain, ah, ah_pt;
  float* save;
  uint32_t site, idx0;
};
template <int CELL>
__device__ __forceinline__ CellGeom cell_geom(const CPK& P, const FB& F, const Task& k) {
  const int B = P.d.B, T = P.d.T, Dg = P.d.Dg, Dp = P.d.Dp, De = P.d.De;
  const long TB = (long)T * B;
  const int t = k.t, dir = k.dir;
  const auto& W = P.p[dir];
  const auto& w = P.w;
  const long prg = pack_rows(B, Dg), prp = pack_rows(B, Dp), pre = pack_rows(B, De);
  const int par = t & 1;
  CellGeom g;
  g.NPT = (CELL == 1 || CELL == 2) ? 2 : 1;
  g.H = CELL == 0 ? Dg : (CELL == 3 ? De : Dp);
  g.Kin = CELL == 1 ? Dg : Dp;
  g.bih = nullptr; g.GI = nullptr; g.ah_pt = 0; g.site = P.site[dir] + CELL;
  const int H = g.H;
  if constexpr (CELL == 0) {
    g.bhh = W.g_bhh; g.GI = w.GIg + ((long)dir * TB + (long)t * B) * 3 * H;
    g.bh = &F.Gh; g.h0 = ((long)dir * (T + 1) + t) * B * H;
    g.ain = w.o_q0p + dir * prp; g.ah = w.o_Ghp + (par * 2 + dir) * prg;
    g.save = w.sv_g + ((long)dir * TB + (long)t * B) * 4 * H; g.idx0 = (uint32_t)((long)t * B * H);
  } else if constexpr (CELL == 1) {
    g.bhh = W.p_bhh; g.GI = w.GIp + ((long)dir * TB + (long)t * B) * 3 * H;
    g.bh = &F.Q; g.h0 = ((long)dir * (T + 1) + t) * B * 2 * H;
    g.ain = w.o_cvp + dir * prg; g.ah = w.o_Qp + (long)(par * 2 + dir) * 2 * prp; g.ah_pt = prp;
    g.save = w.sv_p + ((long)dir * TB + (long)t * B) * 2 * 4 * H; g.idx0 = (uint32_t)((long)t * B * 2 * H);
  } else if constexpr (CELL == 2) {
    g.bhh = W.l_bhh; g.GI = w.GIl + ((long)dir * TB + (long)t * B) * 3 * H;
    g.bh = &F.Q; g.h0 = ((long)dir * (T + 1) + t) * B * 2 * H;
    g.ain = w.o_ssp + dir * prp; g.ah = w.o_Qp + (long)(par * 2 + dir) * 2 * prp; g.ah_pt = prp;
    g.save = w.sv_l + ((long)dir * TB + (long)t * B) * 2 * 4 * H; g.idx0 = (uint32_t)((long)t * B * 2 * H);
  } else {
    g.bhh = W.e_bhh; g.bih = W.e_bih;
    g.bh = &F.Eh; g.h0 = ((long)dir * (T + 1) + t) * B * H;
    g.ain = w.o_qselp + (par * 2 + dir) * prp; g.ah = w.o_Ehp + (par * 2 + dir) * pre;
    g.save = w.sv_e + ((long)dir * TB + (long)t * B) * 4 * H; g.idx0 = (uint32_t)((long)t * B * H);
  }
  const int k8i = k8p(g.Kin), k8h = k8p(H);
  g.Wi = (long)dir * w.wpk_dir + w.wpk_off[2 * CELL] + (long)k.slab * k8i * 256;
  g.Wh = (long)dir * w.wpk_dir + w.wpk_off[2 * CELL + 1] + (long)k.slab * k8h * 256;
  g.ain += (long)k.rb * k8i * 256;
  g.ah += (long)k.rb * k8h * 256;
  return g;
}
// this wave's passes (up to 8 = two chunks of 4) of a product with padded reduction length Kp (<= 512): fragment index k8 of pass p, or -1
__device__ __forceinline__ int pass_k8(int Kp, int p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int KC = ((Kp + PNW * 16 - 1) / (PNW * 16)) * 16;
  const int kb = wave * KC + 16 * p;
  return (p * 16 < KC && kb < Kp) ? (kb >> 3) + (lane >> 5) : -1;
}
template <bool SC1>
__device__ __forceinline__ void load_chunk(const XB& x, long base, int Kp, int c, float (*f)[8]) {
  const int r = threadIdx.x & 31;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int k8 = pass_k8(Kp, 4 * c + p);
    xb_frag<SC1>(x, k8 >= 0 ? base + ((long)k8 * 32 + r) * 8 : -1, f[p]);
  }
}
__device__ __forceinline__ void mma_chunk(const float (*a)[8], const float (*b)[8], f32x16& acc) {
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][j], b[p][j], acc, 0, 0, 0);
}
template <int CELL>
__device__ __forceinline__ void cell_pre(const CPK& P, const FB& F, const Task& k, TRegs& R) {
  const CellGeom g = cell_geom<CELL>(P, F, k);
  const int Kpi = k8p(g.Kin) * 8, Kph = k8p(g.H) * 8;
  load_chunk<false>(F.wpk, g.Wi, Kpi, 0, R.bi);
  load_chunk<false>(F.wpk, g.Wh, Kph, 0, R.bh);
}
template <int CELL>
__device__ __forceinline__ void cell_post(const CPK& P, const FB& F, const Task& k, TRegs& R, float* red, float* tiles) {
  const CellGeom g = cell_geom<CELL>(P, F, k);
  constexpr int NPT = (CELL == 1 || CELL == 2) ? 2 : 1;
  const int B = P.d.B, T = P.d.T, H = g.H, t = k.t, dir = k.dir;
  const long TB = (long)T * B;
  const auto& w = P.w;
  const int r = threadIdx.x & 31;
  const int Kpi = k8p(g.Kin) * 8, Kph = k8p(H) * 8;
  ERegs E;
  auto load_epi = [&]() {                       // requested between the two chunks: in flight under the second chunk's MFMA chains
#pragma unroll
    for (int j = 0; j < P_NIT; ++j) {
      const int it = threadIdx.x + j * PNT;
      const int uu = it % UW, pt = (it / UW) % NPT, rr = it / (UW * NPT);
      const int b = k.rb * 32 + rr, u = k.slab * UW + uu;
      E.qs[j] = 0.f;
      if (it < 32 * NPT * UW && b < B && u < H) {
        if constexpr (CELL == 3) { E.gi[j][0] = g.bih[u]; E.gi[j][1] = g.bih[H + u]; E.gi[j][2] = g.bih[2 * H + u]; }
        else { const float* q = g.GI + (long)b * 3 * H; E.gi[j][0] = q[u]; E.gi[j][1] = q[H + u]; E.gi[j][2] = q[2 * H + u]; }
        E.bb[j][0] = g.bhh[u]; E.bb[j][1] = g.bhh[H + u]; E.bb[j][2] = g.bhh[2 * H + u];
        E.hp[j] = xb_ld(*g.bh, g.h0 + ((long)b * NPT + pt) * H + u);
        if constexpr (CELL == 1 || CELL == 2) {
          const int* ix = w.idx + (long)dir * (TB + B) + (long)t * B + b;
          E.sp[j] = ix[0]; E.sn[j] = ix[B];
          E.qm[j] = w.qm[((long)dir * TB + (long)t * B + b) * 2 + pt];
        }
        if constexpr (CELL == 2) E.qs[j] = xb_ld(F.qs, ((long)dir * B + b) * 2 * H + (long)pt * H + u);
      }
    }
  };
  f32x16 acc[1 + NPT];
#pragma unroll
  for (int i = 0; i < 1 + NPT; ++i) acc[i] = f32x16{0};
  // a wave's K share is two chunks of four passes.  Register budget (256 per lane with two workgroups per CU): one chunk's operands at a
  // time, the second party's state fragments are requested when the input product's operands are dead and arrive under the first
  // party's MFMA chain; the latencies that stay exposed are covered by the CU's other workgroup (the other direction's chain)
  constexpr bool two = PNW < 8;                 // (8 waves: a wave's share of K <= 512 is one chunk)
  {
    float ai[4][8], ah[NPT][4][8];
    load_chunk<true>(F.apk, g.ain, Kpi, 0, ai);
    load_chunk<true>(F.apk, g.ah, Kph, 0, ah[0]);
    if (!two) load_epi();
    if (CELL == 1 || CELL == 2) PSTC(8);
    mma_chunk(ai, R.bi, acc[0]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NPT == 2) load_chunk<true>(F.apk, g.ah + g.ah_pt, Kph, 0, ah[1]);
    __builtin_amdgcn_sched_barrier(0);          // (all of them requested before the first party's chain starts, not one per MFMA group)
#pragma unroll
    for (int i = 0; i < NPT; ++i) mma_chunk(ah[i], R.bh, acc[1 + i]);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (two) {
    float ai[4][8], ah[NPT][4][8];
    load_chunk<false>(F.wpk, g.Wi, Kpi, 1, R.bi);
    load_chunk<true>(F.apk, g.ain, Kpi, 1, ai);
    load_chunk<false>(F.wpk, g.Wh, Kph, 1, R.bh);
    load_chunk<true>(F.apk, g.ah, Kph, 1, ah[0]);
    load_epi();
    mma_chunk(ai, R.bi, acc[0]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NPT == 2) load_chunk<true>(F.apk, g.ah + g.ah_pt, Kph, 1, ah[1]);
#pragma unroll
    for (int i = 0; i < NPT; ++i) mma_chunk(ah[i], R.bh, acc[1 + i]);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (CELL == 1 || CELL == 2) PSTC(9);
  {   // the waves' partial tiles meet in LDS (one exchange for all products), summed in a fixed order; kept [column][row] so that an
      // accumulator's four consecutive rows are one 16-byte access
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 1 + NPT; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(red + i * P_RED1 + wave * P_T1 + r * P_TS + 8 * g4 + 4 * half) =
            make_float4(acc[i][4 * g4], acc[i][4 * g4 + 1], acc[i][4 * g4 + 2], acc[i][4 * g4 + 3]);
    if (CELL == 1 || CELL == 2) PSTC(14);
    __syncthreads();
    if (CELL == 1 || CELL == 2) PSTC(15);
    for (int e = tid; e < (1 + NPT) * 256; e += PNT) {      // (product, column, group of four rows)
      const int i = e >> 8, c = (e >> 3) & 31, rg = e & 7;
      const float* src = red + i * P_RED1 + c * P_TS + rg * 4;
      float4 sum = *reinterpret_cast<const float4*>(src);
#pragma unroll
      for (int wv = 1; wv < PNW; ++wv) {
        const float4 v = *reinterpret_cast<const float4*>(src + wv * P_T1);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      *reinterpret_cast<float4*>(tiles + i * P_T1 + c * P_TS + rg * 4) = sum;
    }
    __syncthreads();
  }
  if (CELL == 1 || CELL == 2) PSTC(10);
  DropKey dk{};
  const bool drop = P.rng != nullptr;
  if (drop) dk = drop_key(P.rng, g.site, P.pdrop);
  const long prg = pack_rows(B, P.d.Dg), prp = pack_rows(B, P.d.Dp), pre = pack_rows(B, P.d.De);
  const int par = t & 1, b0 = k.rb * 32, u0 = k.slab * UW;
#pragma unroll
  for (int j = 0; j < P_NIT; ++j) {
    const int it = threadIdx.x + j * PNT;
    const int uu = it % UW, pt = (it / UW) % NPT, rr = it / (UW * NPT);
    const int b = b0 + rr, u = u0 + uu;
    if (!(it < 32 * NPT * UW && b < B && u < H)) continue;
    const float* ti = tiles + rr;                            // element (row rr, column n) at [n * P_TS]
    const float* th = tiles + P_T1 * (1 + pt) + rr;
    const float hp = E.hp[j];
    const float rg = sigmoidf_(E.gi[j][0] + ti[uu * P_TS] + th[uu * P_TS] + E.bb[j][0]);
    const float z = sigmoidf_(E.gi[j][1] + ti[(UW + uu) * P_TS] + th[(UW + uu) * P_TS] + E.bb[j][1]);
    const float ghn = th[(2 * UW + uu) * P_TS] + E.bb[j][2];
    const float n = tanhf(E.gi[j][2] + ti[(2 * UW + uu) * P_TS] + rg * ghn);
    float h = (1.f - z) * n + z * hp;
    const long e = ((long)b * NPT + pt) * H + u;            // element index inside the step (rows (b, party))
    if (drop) h *= drop_scale(dk, g.idx0 + (uint32_t)e);
    float* sv = g.save + ((long)b * NPT + pt) * 4 * H + u;
    sv[0] = rg; sv[H] = z; sv[2 * H] = n; sv[3 * H] = ghn;
    if constexpr (CELL == 0) {
      xb_st(F.Gh, g.h0 + (long)B * H + (long)b * H + u, h);                  // Gh[t + 1]: the attention's history and the next step's h'
      xb_st(F.apk, w.o_Ghp + ((par ^ 1) * 2 + dir) * prg + pk_off(b, u, H), h);
    } else if constexpr (CELL == 1) {
      xb_st(F.qs, ((long)dir * B + b) * 2 * H + (long)pt * H + u, h);
      if (pt == E.sp[j]) {
        w.ss[((long)dir * TB + (long)t * B + b) * H + u] = h;
        xb_st(F.apk, w.o_ssp + dir * prp + pk_off(b, u, H), h);
      }
    } else if constexpr (CELL == 2) {
      const int sp = E.sp[j], sn = E.sn[j];
      const float m = E.qm[j];
      const float qn = h * (1.f - m) + E.qs[j] * m;
      xb_st(F.Q, g.h0 + (long)B * 2 * H + e, qn);                             // Q[t + 1]
      xb_st(F.apk, w.o_Qp + ((long)((par ^ 1) * 2 + dir) * 2 + pt) * prp + pk_off(b, u, H), qn);
      if (pt == sp) {
        w.qsel[((long)dir * TB + (long)t * B + b) * H + u] = qn;
        xb_st(F.apk, w.o_qselp + (par * 2 + dir) * prp + pk_off(b, u, H), qn);
      }
      if (pt == sn && t + 1 < T) {
        w.q0sel[((long)dir * TB + (long)(t + 1) * B + b) * H + u] = qn;
        xb_st(F.apk, w.o_q0p + dir * prp + pk_off(b, u, H), qn);
      }
    } else {
      xb_st(F.Eh, g.h0 + (long)B * H + (long)b * H + u, h);                  // Eh[t + 1]
      xb_st(F.apk, w.o_Ehp + ((par ^ 1) * 2 + dir) * pre + pk_off(b, u, H), h);
      const int tau = dir ? P.rev[(long)t * B + b] : t;
      if (tau >= 0) P.out[((long)tau * B + b) * P.ldo + (long)dir * H + u] = h;
    }
  }
  __syncthreads();            // tiles / red are free for the next task
  if (CELL == 1 || CELL == 2) PSTC(11);
}

// attention over the history g_0 .. g_{t-1} of (b, dir) (DialogueRNN.py:56-59,:75) in ONE pass over the history: wave w takes rows
// s = w, w + 8, ...; a row stays in registers between its score and its share of the pooled vector (running maximum / running sum per
// wave, merged at the end).  Lane l holds u = 4 l + 256 i .. + 3, i < ATT_V.  LDS: sc[T] raw scores | 16 merge words; wacc = [8][ATT_V * 256]
constexpr int ATT_V = 2;                         // D_g <= 512
struct Row4 { float v[ATT_V][4]; };
__device__ __forceinline__ void att_row_load(const XB& G, long off, int Dg, bool v4, int lane, bool valid, Row4& o) {
#pragma unroll
  for (int i = 0; i < ATT_V; ++i) {
    const int u = 4 * lane + 256 * i;
    if (valid && v4 && u < Dg) {
      const pu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(G.r, (int)((off + u) * 4), 0, P_SC1);
      o.v[i][0] = __uint_as_float(q.x); o.v[i][1] = __uint_as_float(q.y); o.v[i][2] = __uint_as_float(q.z); o.v[i][3] = __uint_as_float(q.w);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) o.v[i][c] = (valid && !v4 && u + c < Dg) ? xb_ld(G, off + u + c) : 0.f;
    }
  }
}
__device__ __forceinline__ void att_fwd_task(const CPK& P, const FB& F, int t, int b, int dir, float* sc, float* wacc) {
  const int B = P.d.B, T = P.d.T, Dg = P.d.Dg;
  const long TB = (long)T * B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool v4 = (Dg & 3) == 0;
  float* mrg = sc + T;                                        // [0..8) wave maxima, [8..16) wave sums
  const float* xs = P.w.Xatt + ((long)dir * TB + (long)t * B + b) * Dg;
  Row4 x;
#pragma unroll
  for (int i = 0; i < ATT_V; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int u = 4 * lane + 256 * i + c; x.v[i][c] = u < Dg ? xs[u] : 0.f; }
  const long g0 = ((long)dir * (T + 1) * B + b) * Dg;        // g_s = Gh[s + 1] at g0 + (s + 1) gs
  const long gs = (long)B * Dg;
  float m = -INFINITY, z = 0.f;
  Row4 acc;
#pragma unroll
  for (int i = 0; i < ATT_V; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc.v[i][c] = 0.f;
  for (int s0 = wave; s0 < t; s0 += PNW * 4) {
    Row4 rw[4];
    float scv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) att_row_load(F.Gh, g0 + (long)(s0 + PNW * j + 1) * gs, Dg, v4, lane, s0 + PNW * j < t, rw[j]);
    float mn = m;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < ATT_V; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) d = fmaf(x.v[i][c], rw[j].v[i][c], d);
      d = wave_sum(d);
      const bool ok = s0 + PNW * j < t;
      scv[j] = ok ? d : -INFINITY;
      if (ok && lane == 0) sc[s0 + PNW * j] = d;
      mn = fmaxf(mn, scv[j]);
    }
    const float scale = expf(m - mn);                          // (m = -inf at the first group: 0)
    z *= scale;
#pragma unroll
    for (int i = 0; i < ATT_V; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc.v[i][c] *= scale;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float e = expf(scv[j] - mn);                       // 0 for the rows beyond the history
      z += e;
#pragma unroll
      for (int i = 0; i < ATT_V; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc.v[i][c] = fmaf(e, rw[j].v[i][c], acc.v[i][c]);
    }
    m = mn;
  }
  if (lane == 0) { mrg[wave] = m; mrg[8 + wave] = z; }
#pragma unroll
  for (int i = 0; i < ATT_V; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) wacc[wave * (ATT_V * 256) + 4 * lane + 256 * i + c] = acc.v[i][c];
  __syncthreads();
  float M = mrg[0];
#pragma unroll
  for (int k = 1; k < PNW; ++k) M = fmaxf(M, mrg[k]);
  float Z = 0.f, f[PNW];
#pragma unroll
  for (int k = 0; k < PNW; ++k) { f[k] = expf(mrg[k] - M); Z = fmaf(mrg[8 + k], f[k], Z); }
  const float rz = 1.f / Z;
  float* al = P.w.alpha + ((long)dir * TB + (long)t * B + b) * T;
  for (int s = tid; s < t; s += PNT) al[s] = expf(sc[s] - M) * rz;
  const long prg = pack_rows(B, Dg);
  for (int u = tid; u < Dg; u += PNT) {
    float c = 0.f;
#pragma unroll
    for (int k = 0; k < PNW; ++k) c = fmaf(wacc[k * (ATT_V * 256) + u], f[k], c);
    c *= rz;
    P.w.cvec[((long)dir * TB + (long)t * B + b) * Dg + u] = c;
    xb_st(F.apk, P.w.o_cvp + dir * prg + pk_off(b, u, Dg), c);
  }
  __syncthreads();
}

__device__ __forceinline__ FB fb_uni(const FB& x) {
  FB f;
  f.Gh = xb_uni(x.Gh); f.Q = xb_uni(x.Q); f.Eh = xb_uni(x.Eh); f.qs = xb_uni(x.qs); f.apk = xb_uni(x.apk); f.wpk = xb_uni(x.wpk);
  return f;
}
// (a non-inlined function sees pointer arguments as generic addresses -- LDS through flat instructions: such tasks take nothing but the
// kernel's fixed LDS layout and address it through the dynamic-LDS symbol themselves)
__device__ __noinline__ void att_task(const CPK* Pp, const FB& Fv, const Task k) {
  const CPK& P = pk_uni(Pp);
  extern __shared__ __attribute__((aligned(16))) float psm[];
  float* red = psm;
  float* attx = psm + P_OFF_ATT;
  const FB F = fb_uni(Fv);
  att_fwd_task(P, F, k.t, k.b, k.dir, attx, red);
}
// the two halves of a task around its phase's barrier wait: `pre` requests the weight fragments (they do not depend on the phase before)
__device__ __forceinline__ void task_pre(const CPK& P, const FB& F, const Task& k, TRegs& R) {
  switch (k.kind) {
    case 0: cell_pre<0>(P, F, k, R); break;
    case 1: cell_pre<1>(P, F, k, R); break;
    case 2: cell_pre<2>(P, F, k, R); break;
    case 3: cell_pre<3>(P, F, k, R); break;
    default: break;
  }
}
__device__ __forceinline__ void task_post(const CPK& P, const FB& F, const Task& k, TRegs& R) {
  extern __shared__ __attribute__((aligned(16))) float psm[];
  switch (k.kind) {
    case 0: cell_post<0>(P, F, k, R, psm, psm + P_OFF_TILES); break;
    case 1: cell_post<1>(P, F, k, R, psm, psm + P_OFF_TILES); break;
    case 2: cell_post<2>(P, F, k, R, psm, psm + P_OFF_TILES); break;
    case 3: cell_post<3>(P, F, k, R, psm, psm + P_OFF_TILES); break;
    case 4: att_task(&P, F, k); break;
    default: break;
  }
}

// Per step and direction two phases, two grid barriers (the recurrence l(t-1) -> p(t) -> l(t) needs both exchanges; the rest rides along):
//   phase B(t): p cell of step t | g cell of step t                                       (read what phase C(t-1) left)
//   phase C(t): l cell of step t (+ blend) | history attention of step t+1 | e cell of step t-1
// The two directions are independent chains with a barrier counter each; every workgroup alternates between them
// (B0 B1 C0 C1 B0 ...): while direction 0's barrier completes it works for direction 1.  A phase of one direction is at most one task
// per workgroup at the reference's widths (208 / 232 tasks on 256 CUs).
__global__ __launch_bounds__(PNT, 2) void drnn_fwd_persist(const PK* __restrict__ pkp) {
  const CPK& P = *(const CPK*)pkp;
  __shared__ int bar_ok;
  const int B = P.d.B, T = P.d.T, Dg = P.d.Dg, Dp = P.d.Dp, De = P.d.De;
  FB F;
  F.Gh = xb_make(P.w.Gh, (size_t)2 * (T + 1) * B * Dg); F.Q = xb_make(P.w.Q, (size_t)2 * (T + 1) * B * 2 * Dp);
  F.Eh = xb_make(P.w.Eh, (size_t)2 * (T + 1) * B * De);
  F.qs = xb_make(P.w.dqs, (size_t)2 * B * 2 * Dp);
  F.apk = xb_make(P.w.apk, P.w.apk_floats);
  F.wpk = xb_make(P.w.wpk, (size_t)2 * P.w.wpk_dir);
  const int G = P_SPLIT_DIRS ? (int)(gridDim.x >> 1) : (int)gridDim.x;       // workgroups per direction chain
  const int wg = P_SPLIT_DIRS ? (int)(blockIdx.x % (unsigned)G) : (int)blockIdx.x;
  const int dir_lo = P_SPLIT_DIRS ? (int)(blockIdx.x / (unsigned)G) : 0, dir_hi = P_SPLIT_DIRS ? dir_lo + 1 : 2;
  GridBar gb[2] = {{P.sync, P.sync + 2 * P_BAR_REP * 32, P.fault, 0u, (unsigned)G}, {P.sync + P_BAR_REP * 32, P.sync + 2 * P_BAR_REP * 32, P.fault, 0u, (unsigned)G}};
  const int NRB = (B + 31) / 32;
  const int ntg = tile_count(B, Dg, UW), ntp = tile_count(B, Dp, UW), nte = tile_count(B, De, UW);
  const int nsg = (Dg + UW - 1) / UW, nsp = (Dp + UW - 1) / UW, nse = (De + UW - 1) / UW;
  // task v of phase ph (0 = B, 1 = C) of step t, direction dir
  auto n_tasks = [&](int ph, int t) { return ph == 0 ? (t < T ? ntp + ntg : 0) : (t < T ? ntp : 0) + (t + 1 < T ? B : 0) + (t > 0 ? nte : 0); };
  // task v of phase ph: v indexes the phase's FULL list (the same at every step; a kind that does not exist at step t gives kind -1)
  auto task_of = [&](int ph, int t, int dir, int v) {
    Task k{-1, t, dir, 0, 0, 0};
    if (ph == 0) {
      if (t >= T) return k;
      if (v < ntp) { if (tile_decode(v, NRB, nsp, k.rb, k.slab)) k.kind = 1; }
      else if (v < ntp + ntg) { if (tile_decode(v - ntp, NRB, nsg, k.rb, k.slab)) k.kind = 0; }
    } else {
      if (v < ntp) { if (t < T && tile_decode(v, NRB, nsp, k.rb, k.slab)) k.kind = 2; }
      else if (v < ntp + B) { if (t + 1 < T) { k.kind = 4; k.t = t + 1; k.b = v - ntp; } }
      else if (v < ntp + B + nte) { if (t > 0 && tile_decode(v - ntp - B, NRB, nse, k.rb, k.slab)) { k.kind = 3; k.t = t - 1; } }
    }
    return k;
  };
  const int n_full[2] = {ntp + ntg, ntp + B + nte};
  const bool use_sched = P.sched_f_ok != 0 && !P_SPLIT_DIRS;
  TRegs R;
  PST_INIT();
  bool first[2] = {true, true};
  for (int t = 0; t <= T; ++t) {
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
      if (n_tasks(ph, t) == 0) continue;
#pragma unroll 1
      for (int dir = dir_lo; dir < dir_hi; ++dir) {
        // a workgroup's index relative to the direction (direction 1's lists are dealt from the middle of the grid) and its tasks of this
        // phase: from the host's schedule (make_fwd_sched: at most two, placed so that every physical workgroup carries about the same
        // work per step), or the list dealt round-robin
        const int wrel = P_SPLIT_DIRS ? wg : (int)((blockIdx.x + (unsigned)dir * (G / 2)) % (unsigned)G);
        int vs[2] = {-1, -1};
        if (use_sched) {
          const int a = P.sched_f[ph][wrel][0], b2 = P.sched_f[ph][wrel][1];
          vs[0] = a != 255 ? a : -1; vs[1] = b2 != 255 ? b2 : -1;
        } else {
          vs[0] = wrel < n_full[ph] ? wrel : -1; vs[1] = wrel + G < n_full[ph] ? wrel + G : -1;       // (more than two rounds: see persist_ok)
        }
        Task k = task_of(ph, t, dir, vs[0] >= 0 ? vs[0] : 0);
        if (vs[0] < 0) k.kind = -1;
        task_pre(P, F, k, R);
        PST(4 * ph + 2 * dir);
        if (!first[dir]) { if (!bar_wait(gb[dir], &bar_ok)) return; }
        first[dir] = false;
        PST(4 * ph + 2 * dir + 1);
        task_post(P, F, k, R);
        if (vs[1] >= 0) { k = task_of(ph, t, dir, vs[1]); task_pre(P, F, k, R); task_post(P, F, k, R); }
        PSTC(12);
        bar_arrive(gb[dir]);
        PSTC(13);
      }
    }
  }
  PST_DUMP("fwd: (work wait) x B0 B1 C0 C1 | loads mfma reduce epilogue | - arrive", (unsigned long long)T);
}

// ======================================================================================================================================
// Persistent backward: the BPTT of both directions in ONE launch, same machinery (packed operands, per-direction barrier counters, every
// workgroup alternating between the two direction chains).  Per step (descending) and direction four phases:
//   C(t): l cell backward of step t | e cell backward of step t-1 | history-attention backward of step t+1      (element-wise / per row)
//   D(t): data-gradient products of the gate gradients C(t) left:  dss, dQ (l) ; dqsel, dE (e, step t-1)
//   E(t): p cell backward | g cell backward of step t
//   F(t): products: dc, dQ (p) ; dq0sel, dG (g)
// (The e chain depends on nothing else of its step, so it runs one step ahead of the rest: one phase pair less per step.)  Nothing is
// accumulated atomically: every product has its own output buffer and the element-wise consumer adds the parts (direct path + products).
// A product tile = 32 rows x 32 output columns, K = 3 H gate columns split over the 8 waves (chunks of 4 passes, double-buffered); its B
// operand is the weight matrix in transposed fragment order (drnn_pack_wt_kernel).
__global__ void drnn_pack_wt_kernel(const float* W, long ldw, int K3, int N, float* out, long n_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (n fastest: coalesced reads of W's rows)
  if (i >= n_out) return;
  const int n = (int)(i & 31), j = (int)((i >> 5) & 7);
  const long q = i >> 8;
  const int k8n = k8p(K3);
  const int k = (int)(q % k8n) * 8 + j, col = (int)(q / k8n) * 32 + n;
  out[(q * 32 + n) * 8 + j] = (k < K3 && col < N) ? W[(long)k * ldw + col] : 0.f;
}

struct BB { XB bk, wt, dGh; };
__device__ __forceinline__ BB bb_uni(const BB& x) { BB b; b.bk = xb_uni(x.bk); b.wt = xb_uni(x.wt); b.dGh = xb_uni(x.dGh); return b; }
__device__ __forceinline__ void pk_store3(const XB& bk, long pack, int row, int H, int u, float a, float b, float c) {
  xb_st(bk, pack + pk_off(row, u, 3 * H), a);
  xb_st(bk, pack + pk_off(row, H + u, 3 * H), b);
  xb_st(bk, pack + pk_off(row, 2 * H + u, 3 * H), c);
}
__device__ __forceinline__ void row_store3(float* o, int H, float a, float b, float c) { o[0] = a; o[H] = b; o[2 * H] = c; }

// e cell backward of step t: elements (b, u) of chunk `ch`
__device__ __noinline__ void e_bwd_task(const CPK* Pp, const BB& Xv, int t, int dir, int ch) {
  const CPK& P = pk_uni(Pp);
  const BB X = bb_uni(Xv);
  const int B = P.d.B, T = P.d.T, H = P.d.De;
  const long TB = (long)T * B, base = (long)dir * P.w.bk_dir;
  const int it = ch * PNT + threadIdx.x;
  if (it >= B * H) return;
  const int b = it / H, u = it - b * H;
  const int tau = dir ? P.rev[(long)t * B + b] : t;
  float dh = xb_ld(X.bk, base + P.w.b_dEdir + it) + xb_ld(X.bk, base + P.w.b_PEh + it);
  if (tau >= 0) dh += P.dout[((long)tau * B + b) * P.ldo + (long)dir * H + u];
  if (P.rng) dh *= drop_scale(drop_key(P.rng, P.site[dir] + 3, P.pdrop), (uint32_t)((long)t * B * H) + (uint32_t)it);
  const float* sv = P.w.sv_e + ((long)dir * TB + (long)t * B + b) * 4 * H + u;
  const GateGrad g = gru_gate_bwd(dh, sv[0], sv[H], sv[2 * H], sv[3 * H], P.w.Eh[((long)dir * (T + 1) + t) * B * H + it]);
  const long row = ((long)dir * TB + (long)t * B + b) * 3 * H + u;
  row_store3(P.w.dgi_e + row, H, g.dar, g.daz, g.dan);
  row_store3(P.w.dgh_e + row, H, g.dar, g.daz, g.danr);
  pk_store3(X.bk, base + P.w.b_gie, b, H, u, g.dar, g.daz, g.dan);
  pk_store3(X.bk, base + P.w.b_ghe, b, H, u, g.dar, g.daz, g.danr);
  xb_st(X.bk, base + P.w.b_dEdir + it, g.dhp);
}
// g cell backward of step t, element (b, u): dh' = dGh[t+1] (the attention's accumulations) + the direct path and the hidden product of
// step t+1
__device__ __forceinline__ void g_elem_bwd(const CPK& P, const BB& X, int t, int dir, int b, int u) {
  const int B = P.d.B, T = P.d.T, H = P.d.Dg;
  const long TB = (long)T * B, base = (long)dir * P.w.bk_dir;
  const long it = (long)b * H + u;
  float dh = xb_ld(X.dGh, ((long)dir * (T + 1) + t + 1) * B * H + it) + xb_ld(X.bk, base + P.w.b_Ddirg + it) + xb_ld(X.bk, base + P.w.b_PGh + it);
  if (P.rng) dh *= drop_scale(drop_key(P.rng, P.site[dir], P.pdrop), (uint32_t)((long)t * B * H) + (uint32_t)it);
  const float* sv = P.w.sv_g + ((long)dir * TB + (long)t * B + b) * 4 * H + u;
  const GateGrad g = gru_gate_bwd(dh, sv[0], sv[H], sv[2 * H], sv[3 * H], P.w.Gh[((long)dir * (T + 1) + t) * B * H + it]);
  const long row = ((long)dir * TB + (long)t * B + b) * 3 * H + u;
  row_store3(P.w.dgi_g + row, H, g.dar, g.daz, g.dan);
  row_store3(P.w.dgh_g + row, H, g.dar, g.daz, g.danr);
  pk_store3(X.bk, base + P.w.b_gig, b, H, u, g.dar, g.daz, g.dan);
  pk_store3(X.bk, base + P.w.b_ghg, b, H, u, g.dar, g.daz, g.danr);
  xb_st(X.bk, base + P.w.b_Ddirg + it, g.dhp);
}
// (stand-alone: only the last time step, whose successor has no attention task to ride on)
__device__ __noinline__ void g_bwd_task(const CPK* Pp, const BB& Xv, int t, int dir, int ch) {
  const CPK& P = pk_uni(Pp);
  const BB X = bb_uni(Xv);
  const int it = ch * PNT + threadIdx.x;
  if (it >= P.d.B * P.d.Dg) return;
  g_elem_bwd(P, X, t, dir, it / P.d.Dg, it % P.d.Dg);
}
// l cell backward + blend (LCELL) or p cell backward of step t, both parties of element (b, u)
template <bool LCELL>
__device__ __noinline__ void lp_bwd_task(const CPK* Pp, const BB& Xv, int t, int dir, int ch) {
  const CPK& P = pk_uni(Pp);
  const BB X = bb_uni(Xv);
  const int B = P.d.B, T = P.d.T, H = P.d.Dp;
  const long TB = (long)T * B, base = (long)dir * P.w.bk_dir;
  const auto& w = P.w;
  const int it = ch * PNT + threadIdx.x;
  if (it >= B * H) return;
  const int b = it / H, u = it - b * H;
  const int* ix = w.idx + (long)dir * (TB + B) + (long)t * B + b;
  const int sp = ix[0], sn = ix[B];
  const bool has_next = t + 1 < T;
  DropKey dk{};
  if (P.rng) dk = drop_key(P.rng, P.site[dir] + (LCELL ? 2 : 1), P.pdrop);
  const float* save = (LCELL ? w.sv_l : w.sv_p) + ((long)dir * TB + (long)t * B) * 2 * 4 * H;
  float* dgh = (LCELL ? w.dgh_l : w.dgh_p) + ((long)dir * TB + (long)t * B) * 2 * 3 * H;
  float* dgi = (LCELL ? w.dgi_l : w.dgi_p) + ((long)dir * TB + (long)t * B + b) * 3 * H + u;
  const long pk_gh = base + (LCELL ? w.b_ghl : w.b_ghp), pk_gi = base + (LCELL ? w.b_gil : w.b_gip);
  // every load of the element first (one round trip), then the arithmetic and the stores
  float dp[2][4], svv[2][4], hq[2], mm[2];
  const float sel = xb_ld(X.bk, base + (LCELL ? w.b_Pqsel : w.b_Pss) + it);
  const float q0n = (LCELL && has_next) ? xb_ld(X.bk, base + w.b_Pq0 + it) : 0.f;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const long qe = ((long)b * 2 + pt) * H + u;
    if constexpr (LCELL) {
      dp[pt][0] = xb_ld(X.bk, base + w.b_QdirL + qe); dp[pt][1] = xb_ld(X.bk, base + w.b_QdirP + qe);
      dp[pt][2] = xb_ld(X.bk, base + w.b_PQl + qe); dp[pt][3] = xb_ld(X.bk, base + w.b_PQp + qe);      // gradient at Q[t+1] (zeros at the last step)
      mm[pt] = w.qm[((long)dir * TB + (long)t * B + b) * 2 + pt];
    } else {
      dp[pt][0] = xb_ld(X.bk, base + w.b_dqs + qe);
    }
    const float* sv = save + ((long)b * 2 + pt) * 4 * H + u;
    svv[pt][0] = sv[0]; svv[pt][1] = sv[H]; svv[pt][2] = sv[2 * H]; svv[pt][3] = sv[3 * H];
    hq[pt] = w.Q[((long)dir * (T + 1) + t) * B * 2 * H + qe];
  }
  float sr = 0.f, sz = 0.f, sna = 0.f;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const long qe = ((long)b * 2 + pt) * H + u;
    float dh;
    if constexpr (LCELL) {
      float d = dp[pt][0] + dp[pt][1] + dp[pt][2] + dp[pt][3];
      if (pt == sp) d += sel;
      if (has_next && pt == sn) d += q0n;
      xb_st(X.bk, base + w.b_dqs + qe, d * mm[pt]);
      dh = d * (1.f - mm[pt]);
    } else {
      dh = dp[pt][0];
      if (pt == sp) dh += sel;
    }
    if (P.rng) dh *= drop_scale(dk, (uint32_t)((long)t * B * 2 * H) + (uint32_t)qe);
    const GateGrad g = gru_gate_bwd(dh, svv[pt][0], svv[pt][1], svv[pt][2], svv[pt][3], hq[pt]);
    row_store3(dgh + ((long)b * 2 + pt) * 3 * H + u, H, g.dar, g.daz, g.danr);
    pk_store3(X.bk, pk_gh, b * 2 + pt, H, u, g.dar, g.daz, g.danr);
    sr += g.dar; sz += g.daz; sna += g.dan;
    xb_st(X.bk, base + (LCELL ? w.b_QdirL : w.b_QdirP) + qe, g.dhp);
  }
  row_store3(dgi, H, sr, sz, sna);                     // both parties share the input row
  pk_store3(X.bk, pk_gi, b, H, u, sr, sz, sna);
}
// history-attention backward of step t, row (b, dir), in two halves that run in consecutive phases on the SAME workgroup (its LDS carries
// x, dc, alpha and ds from one to the other):
//   att_bwd_a (phase C): ONE pass over the history g_0 .. g_{t-1}: dalpha_s = <dc, g_s> and, with the row still in registers,
//     acc += alpha_s dalpha_s g_s.  Then dot = sum alpha dalpha, ds_s = alpha_s (dalpha_s - dot) and
//     dx = sum_s ds_s g_s = acc - dot c_t   (c_t = sum alpha_s g_s is the forward's pooled vector)                       -> dXatt[t]
//   att_bwd_b (phase D, beside the products): dGh[s+1] += alpha_s dc + ds_s x for s < t -- independent read-modify-writes, NB rows'
//     loads in flight per thread.  (dGh[t] is complete after this: the g cell backward of step t-1 reads it in phase E.)
// LDS state per direction: x[Dg] | dc[Dg] | ds[T] | alpha[T]
__device__ __noinline__ void att_bwd_a(const CPK* Pp, const BB& Xv, int t, int b, int dir, int st_off) {
  const CPK& P = pk_uni(Pp);
  extern __shared__ __attribute__((aligned(16))) float psm[];
  float* st = psm + B_OFF_ATT + st_off;
  float* wacc = psm;
  float* redw = psm + B_OFF_REDW;
  const BB X = bb_uni(Xv);
  const int B = P.d.B, T = P.d.T, Dg = P.d.Dg;
  const long TB = (long)T * B, base = (long)dir * P.w.bk_dir;
  float* x = st; float* dcv = st + Dg; float* ds = st + 2 * Dg; float* al = ds + T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xs = P.w.Xatt + ((long)dir * TB + (long)t * B + b) * Dg;
  const float* as = P.w.alpha + ((long)dir * TB + (long)t * B + b) * T;
  for (int u = tid; u < Dg; u += PNT) { x[u] = xs[u]; dcv[u] = xb_ld(X.bk, base + P.w.b_Pc + (long)b * Dg + u); }
  for (int s = tid; s < t; s += PNT) al[s] = as[s];
  __syncthreads();
  const float* G = P.w.Gh + ((long)dir * (T + 1) * B + b) * Dg;          // g_s at G + (s + 1) gs (written by the forward launch)
  const long gs = (long)B * Dg;
  Row4 dc4, acc;
#pragma unroll
  for (int i = 0; i < ATT_V; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int u = 4 * lane + 256 * i + c; dc4.v[i][c] = u < Dg ? dcv[u] : 0.f; acc.v[i][c] = 0.f; }
  const bool v4 = (Dg & 3) == 0;
  float dotp = 0.f;
  for (int s0 = wave; s0 < t; s0 += PNW * 4) {
    Row4 rw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = s0 + PNW * j;
      const float* row = G + (long)(s + 1) * gs;
#pragma unroll
      for (int i = 0; i < ATT_V; ++i) {
        const int u = 4 * lane + 256 * i;
        if (s < t && v4 && u < Dg) {
          const float4 q = *reinterpret_cast<const float4*>(row + u);
          rw[j].v[i][0] = q.x; rw[j].v[i][1] = q.y; rw[j].v[i][2] = q.z; rw[j].v[i][3] = q.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) rw[j].v[i][c] = (s < t && !v4 && u + c < Dg) ? row[u + c] : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = s0 + PNW * j;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < ATT_V; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) d = fmaf(dc4.v[i][c], rw[j].v[i][c], d);
      d = wave_sum(d);
      if (s < t) {
        const float a = al[s];
        if (lane == 0) ds[s] = d;                      // dalpha_s for now
        dotp = fmaf(a, d, dotp);
        const float f = a * d;
#pragma unroll
        for (int i = 0; i < ATT_V; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc.v[i][c] = fmaf(f, rw[j].v[i][c], acc.v[i][c]);
      }
    }
  }
  if (lane == 0) redw[wave] = dotp;                    // (every lane of the wave holds the wave's partial sum)
#pragma unroll
  for (int i = 0; i < ATT_V; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) wacc[wave * (ATT_V * 256) + 4 * lane + 256 * i + c] = acc.v[i][c];
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < PNW; ++k) dot += redw[k];
  const float* ct = P.w.cvec + ((long)dir * TB + (long)t * B + b) * Dg;
  float* dX = P.w.dXatt + ((long)dir * TB + (long)t * B + b) * Dg;
  for (int u = tid; u < Dg; u += PNT) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < PNW; ++k) a += wacc[k * (ATT_V * 256) + u];
    dX[u] = a - dot * ct[u];
  }
  __syncthreads();                                     // (every thread has read redw / ds before they change)
  for (int s = tid; s < t; s += PNT) ds[s] = al[s] * (ds[s] - dot);
  __syncthreads();
}
__device__ __noinline__ void att_bwd_b(const CPK* Pp, const BB& Xv, int t, int b, int dir, int st_off) {
  const CPK& P = pk_uni(Pp);
  extern __shared__ __attribute__((aligned(16))) float psm[];
  const float* st = psm + B_OFF_ATT + st_off;
  const BB X = bb_uni(Xv);
  const int B = P.d.B, T = P.d.T, Dg = P.d.Dg;
  const float* x = st; const float* dcv = st + Dg; const float* ds = st + 2 * Dg; const float* al = ds + T;
  const int tid = threadIdx.x;
  const long g0 = ((long)dir * (T + 1) * B + b) * Dg, gs = (long)B * Dg;
  if ((Dg & 3) == 0) {
    // PNT / 128 row groups x 128 column quads: thread (grp, q) updates u = 4 q .. 4 q + 3 of the rows s = grp, grp + NG, ...
    constexpr int NB = 32, NG = PNT / 128;
    const int q = tid & 127, grp = tid >> 7;
    for (int u = 4 * q; u < Dg; u += 512) {
      const float4 dc4 = *reinterpret_cast<const float4*>(dcv + u), x4 = *reinterpret_cast<const float4*>(x + u);
      for (int sb = grp; sb < t; sb += NG * NB) {
        pu32x4 ov[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int sj = sb + NG * j;
          if (sj < t) ov[j] = __builtin_amdgcn_raw_buffer_load_b128(X.dGh.r, (int)((g0 + (long)(sj + 1) * gs + u) * 4), 0, P_SC1);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int sj = sb + NG * j;
          if (sj < t) {
            const float a = al[sj], dsv = ds[sj];
            pu32x4 nv;
            nv.x = __float_as_uint(__uint_as_float(ov[j].x) + a * dc4.x + dsv * x4.x);
            nv.y = __float_as_uint(__uint_as_float(ov[j].y) + a * dc4.y + dsv * x4.y);
            nv.z = __float_as_uint(__uint_as_float(ov[j].z) + a * dc4.z + dsv * x4.z);
            nv.w = __float_as_uint(__uint_as_float(ov[j].w) + a * dc4.w + dsv * x4.w);
            __builtin_amdgcn_raw_buffer_store_b128(nv, X.dGh.r, (int)((g0 + (long)(sj + 1) * gs + u) * 4), 0, P_SC1);
          }
        }
      }
    }
  } else {
    constexpr int NB = 8;
    for (int u = tid % 256; u < Dg; u += 256) {
      const float dcu = dcv[u], xu = x[u];
      constexpr int NP = PNT / 256;
      const int part = tid / 256;
      const int s0 = (int)((long)t * part / NP), s1 = (int)((long)t * (part + 1) / NP);
      for (int sb = s0; sb < s1; sb += NB) {
        float ov[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) if (sb + j < s1) ov[j] = xb_ld(X.dGh, g0 + (long)(sb + j + 1) * gs + u);
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (sb + j < s1) xb_st(X.dGh, g0 + (long)(sb + j + 1) * gs + u, ov[j] + al[sb + j] * dcu + ds[sb + j] * xu);
      }
    }
  }
  // dGh[t] of this row is complete now (steps > t have added theirs before): the g cell backward of step t-1 follows at once, by the
  // workgroup that holds the row -- no element-wise phase (and grid barrier) of its own
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int u = tid; u < Dg; u += PNT) g_elem_bwd(P, X, t - 1, dir, b, u);
  __syncthreads();
}
// product i of direction dir, tile (rb, ct): out[rows, N] tile = dgates[rows, K3] W^T-pack
struct BProd { long a_off, w_off, o_off; int K3, N, rows; };
template <int I>
__device__ __forceinline__ BProd bprod(const CPK& P, int dir) {
  const auto& w = P.w;
  const int B = P.d.B, Dg = P.d.Dg, Dp = P.d.Dp, De = P.d.De;
  const long a = I == 0 ? w.b_gie : I == 1 ? w.b_ghe : I == 2 ? w.b_gil : I == 3 ? w.b_ghl : I == 4 ? w.b_gip : I == 5 ? w.b_ghp : I == 6 ? w.b_gig : w.b_ghg;
  const long o = I == 0 ? w.b_Pqsel : I == 1 ? w.b_PEh : I == 2 ? w.b_Pss : I == 3 ? w.b_PQl : I == 4 ? w.b_Pc : I == 5 ? w.b_PQp : I == 6 ? w.b_Pq0 : w.b_PGh;
  BProd d;
  d.a_off = (long)dir * w.bk_dir + a; d.o_off = (long)dir * w.bk_dir + o; d.w_off = (long)dir * w.wtk_dir + w.wtk_off[I];
  d.K3 = 3 * (I < 2 ? De : I < 6 ? Dp : Dg);
  d.N = (I == 1) ? De : (I == 4 || I == 7) ? Dg : Dp;
  d.rows = (I == 3 || I == 5) ? 2 * B : B;
  return d;
}
// operands of the p cell backward of element (b, u), both parties (everything but the dss value that arrives through the product)
struct PElem { float dqs[2], sv[2][4], hq[2]; int sp; bool ok; };
__device__ __forceinline__ PElem p_elem_load(const CPK& P, const BB& X, int t, int dir, int b, int u) {
  const int B = P.d.B, T = P.d.T, H = P.d.Dp;
  const long TB = (long)T * B, base = (long)dir * P.w.bk_dir;
  PElem e;
  e.ok = b < B && u < H;
  if (e.ok) {
    e.sp = P.w.idx[(long)dir * (TB + B) + (long)t * B + b];
    const float* save = P.w.sv_p + ((long)dir * TB + (long)t * B) * 2 * 4 * H;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const long qe = ((long)b * 2 + pt) * H + u;
      e.dqs[pt] = xb_ld(X.bk, base + P.w.b_dqs + qe);
      const float* sv = save + ((long)b * 2 + pt) * 4 * H + u;
      e.sv[pt][0] = sv[0]; e.sv[pt][1] = sv[H]; e.sv[pt][2] = sv[2 * H]; e.sv[pt][3] = sv[3 * H];
      e.hq[pt] = P.w.Q[((long)dir * (T + 1) + t) * B * 2 * H + qe];
    }
  }
  return e;
}
// p cell backward of element (b, u) given dss = the l cell's input-product gradient (DialogueRNN.py:144-153 backwards): gate gradients
// of both parties (row-major for the weight-gradient GEMMs, fragment order for this step's products), the direct path into dQ
__device__ __forceinline__ void p_elem_bwd(const CPK& P, const BB& X, int t, int dir, int b, int u, const PElem& e, float dss) {
  const int B = P.d.B, T = P.d.T, H = P.d.Dp;
  const long TB = (long)T * B, base = (long)dir * P.w.bk_dir;
  DropKey dk{};
  if (P.rng) dk = drop_key(P.rng, P.site[dir] + 1, P.pdrop);
  float* dgh = P.w.dgh_p + ((long)dir * TB + (long)t * B) * 2 * 3 * H;
  float sr = 0.f, sz = 0.f, sna = 0.f;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const long qe = ((long)b * 2 + pt) * H + u;
    float dh = e.dqs[pt];
    if (pt == e.sp) dh += dss;
    if (P.rng) dh *= drop_scale(dk, (uint32_t)((long)t * B * 2 * H) + (uint32_t)qe);
    const GateGrad g = gru_gate_bwd(dh, e.sv[pt][0], e.sv[pt][1], e.sv[pt][2], e.sv[pt][3], e.hq[pt]);
    row_store3(dgh + ((long)b * 2 + pt) * 3 * H + u, H, g.dar, g.daz, g.danr);
    pk_store3(X.bk, base + P.w.b_ghp, b * 2 + pt, H, u, g.dar, g.daz, g.danr);
    sr += g.dar; sz += g.daz; sna += g.dan;
    xb_st(X.bk, base + P.w.b_QdirP + qe, g.dhp);
  }
  row_store3(P.w.dgi_p + ((long)dir * TB + (long)t * B + b) * 3 * H + u, H, sr, sz, sna);
  pk_store3(X.bk, base + P.w.b_gip, b, H, u, sr, sz, sna);
}
// FUSE_P: the tile is dss (the l cell's input-product gradient) and its only reader is the p cell backward of the same elements: the
// tile's owner applies it to them right away (their other operands were requested before the MFMA chain) instead of storing dss for an
// element-wise phase of its own -- one grid barrier less per step
template <bool FUSE_P>
__device__ __noinline__ void bwd_prod_task(const CPK* Pp, const BB& Xv, const BProd d, int rb, int ct, int t, int dir) {
  extern __shared__ __attribute__((aligned(16))) float psm[];
  float* red = psm;
  const BB X = bb_uni(Xv);
  PElem pe[1024 / PNT];
  if constexpr (FUSE_P) {
    const CPK& P = pk_uni(Pp);
#pragma unroll
    for (int j = 0; j < 1024 / PNT; ++j) {
      const int e = threadIdx.x + j * PNT;
      pe[j] = p_elem_load(P, X, t, dir, rb * 32 + (e >> 5), ct * 32 + (e & 31));
    }
  }
  const int k8n = k8p(d.K3), Kp = k8n * 8;
  const long ab = d.a_off + (long)rb * k8n * 256, wb = d.w_off + (long)ct * k8n * 256;
  const int KC = ((Kp + PNW * 16 - 1) / (PNW * 16)) * 16;
  const int nch = (KC / 16 + 3) / 4;
  f32x16 acc = {0};
  if constexpr (PNW >= 8) {
    // the gate gradients (written this phase pair by other workgroups: L2-bypassing loads, the long latency) are requested for the wave's
    // whole K share at once -- at most 3 chunks of 4 passes (K3 <= 1536), 96 registers; the weight fragments (cached) are double-buffered
    float a[3][4][8], b0[4][8], b1[4][8];
#pragma unroll
    for (int c = 0; c < 3; ++c) load_chunk<true>(X.bk, ab, Kp, c, a[c]);          // (passes beyond the wave's share read zeros)
    load_chunk<false>(X.wt, wb, Kp, 0, b0);
    if (nch > 1) load_chunk<false>(X.wt, wb, Kp, 1, b1);
    mma_chunk(a[0], b0, acc);
    if (nch > 1) {
      if (nch > 2) load_chunk<false>(X.wt, wb, Kp, 2, b0);
      mma_chunk(a[1], b1, acc);
      if (nch > 2) mma_chunk(a[2], b0, acc);
    }
  } else {
    // (fewer waves: a wave's K share is up to 6 chunks -- both operands double-buffered, two chunks in flight)
    float a0[4][8], b0[4][8], a1[4][8], b1[4][8];
    load_chunk<true>(X.bk, ab, Kp, 0, a0);
    load_chunk<false>(X.wt, wb, Kp, 0, b0);
    for (int c = 0; c < nch; c += 2) {
      if (c + 1 < nch) { load_chunk<true>(X.bk, ab, Kp, c + 1, a1); load_chunk<false>(X.wt, wb, Kp, c + 1, b1); }
      mma_chunk(a0, b0, acc);
      if (c + 1 < nch) {
        if (c + 2 < nch) { load_chunk<true>(X.bk, ab, Kp, c + 2, a0); load_chunk<false>(X.wt, wb, Kp, c + 2, b0); }
        mma_chunk(a1, b1, acc);
      }
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
    *reinterpret_cast<float4*>(red + wave * P_T1 + r * P_TS + 8 * g4 + 4 * half) = make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
  __syncthreads();
  PSTC(13);
#pragma unroll
  for (int j = 0; j < 1024 / PNT; ++j) {               // (output column fastest: coalesced stores)
    const int e = tid + j * PNT;
    const int n = e & 31, rr = e >> 5;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < PNW; ++wv) s += red[wv * P_T1 + n * P_TS + rr];
    const int row = rb * 32 + rr, col = ct * 32 + n;
    if constexpr (FUSE_P) {
      if (pe[j].ok) p_elem_bwd(pk_uni(Pp), X, t, dir, row, col, pe[j], s);
    } else {
      if (row < d.rows && col < d.N) xb_st(X.bk, d.o_off + (long)row * d.N + col, s);
    }
  }
  __syncthreads();
  PSTC(14);
}

__global__ __launch_bounds__(PNT, 2) void drnn_bwd_persist(const PK* __restrict__ pkp) {
  const CPK& P = *(const CPK*)pkp;
  __shared__ int bar_ok;
  const int B = P.d.B, T = P.d.T, Dg = P.d.Dg, Dp = P.d.Dp, De = P.d.De;
  BB X;
  X.bk = xb_make(P.w.bk, P.w.bk_floats);
  X.wt = xb_make(P.w.wpk, (size_t)2 * P.w.wtk_dir);
  X.dGh = xb_make(P.w.dGh, (size_t)2 * (T + 1) * B * Dg);
  const int G = P_SPLIT_DIRS ? (int)(gridDim.x >> 1) : (int)gridDim.x;       // workgroups per direction chain
  const int wg = P_SPLIT_DIRS ? (int)(blockIdx.x % (unsigned)G) : (int)blockIdx.x;
  const int dir_lo = P_SPLIT_DIRS ? (int)(blockIdx.x / (unsigned)G) : 0, dir_hi = P_SPLIT_DIRS ? dir_lo + 1 : 2;
  GridBar gb[2] = {{P.sync, P.sync + 2 * P_BAR_REP * 32, P.fault, 0u, (unsigned)G}, {P.sync + P_BAR_REP * 32, P.sync + 2 * P_BAR_REP * 32, P.fault, 0u, (unsigned)G}};
  const int nrb1 = (B + 31) / 32, nrb2 = (2 * B + 31) / 32;
  const int ctg = (Dg + 31) / 32, ctp = (Dp + 31) / 32, cte = (De + 31) / 32;
  const int nel = (B * Dp + PNT - 1) / PNT, neg = (B * Dg + PNT - 1) / PNT, nee = (B * De + PNT - 1) / PNT;
  // tiles of the eight products (rows x column tiles of 32, padded for the XCD pairing)
  const int nt_e_ih = tile_count(B, Dp, 32), nt_e_hh = tile_count(B, De, 32), nt_l_ih = tile_count(B, Dp, 32), nt_l_hh = tile_count(2 * B, Dp, 32);
  const int nt_p_ih = tile_count(B, Dg, 32), nt_p_hh = tile_count(2 * B, Dp, 32), nt_g_ih = tile_count(B, Dp, 32), nt_g_hh = tile_count(B, Dg, 32);
  (void)ctg; (void)ctp; (void)cte;
  // phase ph (0 = C, 1 = D, 2 = E, 3 = F) of step t: number of tasks, and task v
  auto n_tasks = [&](int ph, int t) {
    switch (ph) {
      case 0: return (t + 1 < T ? B : 0) + (t < T ? nel : 0) + (t >= 1 ? nee : 0);
      case 1: return (t + 1 < T ? B : 0) + (t < T ? nt_l_ih + (t > 0 ? nt_l_hh : 0) : 0) + (t >= 1 ? nt_e_ih + nt_e_hh : 0);
      case 2: return t == T - 1 ? neg : 0;                 // (only the last step's g cell: the others ride on the attention rows, the p cell on dss)
      default: return (t < T && t > 0) ? nt_p_ih + nt_p_hh + nt_g_ih + nt_g_hh : 0;
    }
  };
  const int att_st = 2 * Dg + 2 * T;                   // LDS state of an attention row between its two halves, per direction
  // task v of phase ph: v indexes the phase's FULL list (the same at every step; a kind that does not exist at step t is skipped)
  auto run_task = [&](int ph, int t, int dir, int v) {
    int rb, ct;
    if (ph == 0) {                                       // C: attention first halves | l cell backward | e cell backward of step t-1
      if (v < B) { if (t + 1 < T) att_bwd_a(&P, X, t + 1, v, dir, dir * att_st); }
      else if (v < B + nel) { if (t < T) lp_bwd_task<true>(&P, X, t, dir, v - B); }
      else if (v < B + nel + nee) { if (t >= 1) e_bwd_task(&P, X, t - 1, dir, v - B - nel); }
    } else if (ph == 1) {                                // D: attention second halves (same workgroup as the first) | products
      if (v < B) { if (t + 1 < T) att_bwd_b(&P, X, t + 1, v, dir, dir * att_st); return; }
      v -= B;
      if (v < nt_l_hh) { if (t < T && t > 0 && tile_decode(v, nrb2, ctp, rb, ct)) bwd_prod_task<false>(&P, X, bprod<3>(P, dir), rb, ct, t, dir); return; }
      v -= nt_l_hh;
      if (v < nt_l_ih) { if (t < T && tile_decode(v, nrb1, ctp, rb, ct)) bwd_prod_task<true>(&P, X, bprod<2>(P, dir), rb, ct, t, dir); return; }
      v -= nt_l_ih;
      if (v < nt_e_ih) { if (t >= 1 && tile_decode(v, nrb1, ctp, rb, ct)) bwd_prod_task<false>(&P, X, bprod<0>(P, dir), rb, ct, t, dir); return; }
      v -= nt_e_ih;
      if (v < nt_e_hh && t >= 1 && tile_decode(v, nrb1, cte, rb, ct)) bwd_prod_task<false>(&P, X, bprod<1>(P, dir), rb, ct, t, dir);
    } else if (ph == 2) {                                // E: only the last step's g cell backward
      g_bwd_task(&P, X, t, dir, v);
    } else {                                             // F: products of the p and g cells
      if (!(t < T && t > 0)) return;
      if (v < nt_p_hh) { if (tile_decode(v, nrb2, ctp, rb, ct)) bwd_prod_task<false>(&P, X, bprod<5>(P, dir), rb, ct, t, dir); return; }
      v -= nt_p_hh;
      if (v < nt_p_ih) { if (tile_decode(v, nrb1, ctg, rb, ct)) bwd_prod_task<false>(&P, X, bprod<4>(P, dir), rb, ct, t, dir); return; }
      v -= nt_p_ih;
      if (v < nt_g_ih) { if (tile_decode(v, nrb1, ctp, rb, ct)) bwd_prod_task<false>(&P, X, bprod<6>(P, dir), rb, ct, t, dir); return; }
      v -= nt_g_ih;
      if (v < nt_g_hh && tile_decode(v, nrb1, ctg, rb, ct)) bwd_prod_task<false>(&P, X, bprod<7>(P, dir), rb, ct, t, dir);
    }
  };
  const int n_full[4] = {B + nel + nee, B + nt_l_hh + nt_l_ih + nt_e_ih + nt_e_hh, neg, nt_p_hh + nt_p_ih + nt_g_ih + nt_g_hh};
  const bool use_sched = P.sched_ok != 0 && !P_SPLIT_DIRS;
  PST_INIT();
  bool first[2] = {true, true};
  for (int t = T; t >= 0; --t) {
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
      if (n_tasks(ph, t) == 0) continue;
#pragma unroll 1
      for (int dir = dir_lo; dir < dir_hi; ++dir) {
        // a workgroup's index relative to the direction: direction 1's lists are dealt from the middle of the grid, so that the two
        // directions' heavy tasks (attention rows: workgroups 0 .. B-1 of a direction) land on different workgroups
        const int wrel = P_SPLIT_DIRS ? wg : (int)((blockIdx.x + (unsigned)dir * (G / 2)) % (unsigned)G);
        PST(2 * ph);
        if (!first[dir]) { if (!bar_wait(gb[dir], &bar_ok)) return; }
        first[dir] = false;
        PST(2 * ph + 1);
        if (use_sched && ph != 2) {
          const int pi = ph == 3 ? 2 : ph;
#pragma unroll 1
          for (int k2 = 0; k2 < 2; ++k2) {
            const int v = P.sched[pi][wrel][k2];
            if (v != 255) run_task(ph, t, dir, v);
          }
        } else {
          for (int v = wrel; v < n_full[ph]; v += G) run_task(ph, t, dir, v);
        }
        PSTC(10);
        bar_arrive(gb[dir]);
        PSTC(12);
      }
    }
  }
  PST_DUMP("bwd: (work wait) x C D E F", (unsigned long long)T);
}

// ---- host helpers -----------------------------------------------------------------------------------------------------------------------
mser_gemm_desc gd() {
  mser_gemm_desc g;
  memset(&g, 0, sizeof(g));
  g.batch1 = g.batch2 = 1; g.alpha = 1.f; g.splitk = 1;
  return g;
}
// C[dir][M, N] (+)= A[dir][M, K] W[dir][N, K]^T (+ R1[dir][M, N])       -- nn.Linear orientation, both directions in one launch
mser_gemm_desc mm_nt_desc(const float* A, long lda, long a_ds, const float* W, long ldw, long w_ds, float* C, long ldc, long c_ds, int M, int N,
                          int K, bool accum, const float* R1 = nullptr, long ldr = 0, long r_ds = 0) {
  // accum: C is initialised (hoisted product or cleared by its reader); the product is added with split-K float atomics (mser::gemm picks the split)
  mser_gemm_desc g = gd();
  g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K;
  g.sAm = lda; g.sAk = 1; g.sBk = 1; g.sBn = ldw; g.ldc = ldc;
  g.batch1 = 2; g.sA1 = a_ds; g.sB1 = w_ds; g.sC1 = c_ds;
  if (accum) { g.flags |= MSER_GEMM_ACCUM; g.splitk = 2; }
  g.R1 = R1; g.ldr1 = ldr; g.sR1_1 = r_ds;
  return g;
}
int mm_nt(hipStream_t s, const float* A, long lda, long a_ds, const float* W, long ldw, long w_ds, float* C, long ldc, long c_ds, int M, int N,
          int K, bool accum, const float* R1 = nullptr, long ldr = 0, long r_ds = 0) {
  return gemm(mm_nt_desc(A, lda, a_ds, W, ldw, w_ds, C, ldc, c_ds, M, N, K, accum, R1, ldr, r_ds), s);
}
// the accumulating form of mm_nn as a descriptor (for a grouped launch; C must be live)
mser_gemm_desc mm_nn_acc_desc(const float* A, long lda, long a_ds, const float* W, long ldw, long w_ds, float* C, long ldc, long c_ds, int M,
                              int N, int K) {
  mser_gemm_desc g = gd();
  g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K;
  g.sAm = lda; g.sAk = 1; g.sBk = ldw; g.sBn = 1; g.ldc = ldc;
  g.batch1 = 2; g.sA1 = a_ds; g.sB1 = w_ds; g.sC1 = c_ds;
  g.flags |= MSER_GEMM_ACCUM;
  g.splitk = 2;
  return g;
}
// C[dir][M, N] (+)= A[dir][M, K] W[dir][K, N]        -- backward data gradient through an nn.Linear weight [K, N] (row-major, ld ldw)
// The reduction runs over the 3H gate columns (K = 1500 at the reference's widths) while the output is only N <= 500 wide: without a
// split the grid would be 16 column tiles x 2 directions = 32 workgroups with 24 serial k-tiles each (33 us per product, measured);
// every such product therefore ACCUMULATES with split-K float atomics into a buffer that is either live (accum) or zeroed first.
int mm_nn(hipStream_t s, const float* A, long lda, long a_ds, const float* W, long ldw, long w_ds, float* C, long ldc, long c_ds, int M, int N,
          int K, bool accum) {
  if (!accum) {
    if (c_ds == (long)M * ldc) MSER_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)2 * M * ldc * sizeof(float), s));
    else for (int dir = 0; dir < 2; ++dir) MSER_CHECK_HIP(hipMemsetAsync(C + dir * c_ds, 0, (size_t)M * ldc * sizeof(float), s));
  }
  mser_gemm_desc g = gd();
  g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K;
  g.sAm = lda; g.sAk = 1; g.sBk = ldw; g.sBn = 1; g.ldc = ldc;
  g.batch1 = 2; g.sA1 = a_ds; g.sB1 = w_ds; g.sC1 = c_ds;
  g.flags |= MSER_GEMM_ACCUM;
  g.splitk = 2;                      // "C is initialised, accumulate atomically": mser::gemm picks the split that fills the chip
  return gemm(g, s);
}
// dW[N, K] += dY[rows, N]^T X[rows, K]   (one direction; split-K over the rows, float atomics)
mser_gemm_desc wgrad_desc(const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, int rows, int N, int K) {
  mser_gemm_desc g = gd();
  g.A = dY; g.B = X; g.C = dW; g.M = N; g.N = K; g.K = rows;
  g.sAm = 1; g.sAk = ldy; g.sBk = ldx; g.sBn = 1; g.ldc = ldw;
  g.flags = MSER_GEMM_ACCUM;
  g.splitk = 2;                      // "C is initialised, accumulate atomically"; the split itself is chosen by mser::gemm
  return g;
}
int wgrad(hipStream_t s, const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, int rows, int N, int K) {
  return gemm(wgrad_desc(dY, ldy, X, ldx, dW, ldw, rows, N, K), s);
}

int validate(const mser_drnn_desc& d, bool bwd) {
  MSER_REQUIRE(d.T > 0 && d.B > 0 && d.Dm > 0 && d.Dg > 0 && d.Dp > 0 && d.De > 0, "mser_drnn: bad sizes");
  MSER_REQUIRE(d.U && d.qmask && d.rev && d.out && d.workspace, "mser_drnn: null pointer");
  MSER_REQUIRE(d.ldu >= d.Dm && d.ldo >= 2 * d.De, "mser_drnn: leading dimension too small");
  MSER_REQUIRE(((uintptr_t)d.workspace & 255) == 0, "mser_drnn: workspace must be 256-byte aligned");
  MSER_REQUIRE(d.workspace_bytes >= mser_drnn_workspace_bytes(d.T, d.B, d.Dm, d.Dg, d.Dp, d.De), "mser_drnn: workspace too small");
  MSER_REQUIRE((3 * (size_t)d.Dg + 2 * (size_t)d.T) * sizeof(float) <= 64 * 1024, "mser_drnn: D_g / T too large for the attention kernel's LDS");
  for (int i = 0; i < 2; ++i) {
    const mser_drnn_params& p = d.p[i];
    MSER_REQUIRE(p.g_wih && p.g_whh && p.g_bih && p.g_bhh && p.p_wih && p.p_whh && p.p_bih && p.p_bhh && p.e_wih && p.e_whh && p.e_bih &&
                 p.e_bhh && p.l_wih && p.l_whh && p.l_bih && p.l_bhh && p.att_w, "mser_drnn: null parameter (direction %d)", i);
    if (bwd) {
      const mser_drnn_params& g = d.g[i];
      MSER_REQUIRE(g.g_wih && g.g_whh && g.g_bih && g.g_bhh && g.p_wih && g.p_whh && g.p_bih && g.p_bhh && g.e_wih && g.e_whh && g.e_bih &&
                   g.e_bhh && g.l_wih && g.l_whh && g.l_bih && g.l_bhh && g.att_w, "mser_drnn_bwd: null gradient (direction %d)", i);
    }
  }
  MSER_REQUIRE(d.p[1].e_bih - d.p[0].e_bih == d.p[1].e_bhh - d.p[0].e_bhh, "mser_drnn: e_cell bias_ih / bias_hh must be equally spaced in both directions");
  if (bwd) MSER_REQUIRE(d.dout, "mser_drnn_bwd: null dout");
  MSER_REQUIRE(!d.rng || (d.p_drop >= 0.f && d.p_drop < 1.f), "mser_drnn: dropout p=%f", d.p_drop);
  return 0;
}

#define DS(field) ((long)(d.p[1].field - d.p[0].field))

int persist_grid() {
  static int g = 0;
  if (g == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) g = n;
    if (g <= 0) g = 1;
    if (g >= 8) g &= ~7;            // a multiple of the 8 XCDs (tile_decode)
  }
  return g;
}
size_t persist_lds(const Dims& d, bool bwd) {
  return (size_t)(bwd ? B_OFF_ATT + 2 * (2 * d.Dg + 2 * d.T) + 16 : P_OFF_ATT + d.T + 32 + 16) * sizeof(float);
}
// every hand-off array is addressed through a 32-bit byte offset (buffer descriptor); one workgroup per CU must fit
bool persist_ok(const Dims& d, bool bwd) {
  if (!g_opt_drnn_persist || persist_grid() < 8 || d.Dg > 512 || d.Dp > 512 || d.De > 512) return false;
  if (bwd && d.B > persist_grid() / 2) return false;
  if (tile_count(d.B, d.Dp, UW) + (d.Dg > d.De ? tile_count(d.B, d.Dg, UW) : d.B + tile_count(d.B, d.De, UW)) > 2 * persist_grid()) return false;    // (a forward phase is at most two rounds)        // (an attention row's two halves meet in its workgroup's LDS: one row per workgroup and direction)      // (a wave's K share in registers; ATT_V)
  const size_t big = (size_t)2 * ((size_t)d.T + 1) * d.B * 2 * (size_t)(d.Dp > d.Dg ? d.Dp : d.Dg) * 3 * sizeof(float);   // >= the largest of them
  return big < ((size_t)1 << 31) && persist_lds(d, bwd) <= (size_t)160 * 1024 / P_WGS_PER_CU;
}
// Backward task schedule: which workgroup (index relative to its direction) runs which task of the phases C, D, F.  Greedy, heaviest
// first: the 2 B attention halves are fixed on workgroups 0 .. B-1 (10 us each), then the product tiles (11 us) and the element-wise chunks
// (3 us) go to the workgroup whose PAIR (w, w + G/2: the two directions' roles of one physical workgroup) carries the least so far, at
// most one task per workgroup and phase where the counts allow.  Round-robin lists left the attention workgroups with three products
// per step on top of their rows (56 us of work per step against 39 on average).
void make_bwd_sched(PK& K, int G) {
  const Dims& d = K.d;
  K.sched_ok = 0;
  memset(K.sched, 255, sizeof(K.sched));
  const int nel = cdiv((long)d.B * d.Dp, PNT), nee = cdiv((long)d.B * d.De, PNT);
  const int nD = tile_count(2 * d.B, d.Dp, 32) + tile_count(d.B, d.Dp, 32) + tile_count(d.B, d.Dp, 32) + tile_count(d.B, d.De, 32);
  const int nF = tile_count(2 * d.B, d.Dp, 32) + tile_count(d.B, d.Dg, 32) + tile_count(d.B, d.Dp, 32) + tile_count(d.B, d.Dg, 32);
  if (G != 256 || d.B > G / 2 || d.B + nel + nee >= 255 || d.B + nD >= 255 || nF >= 255) return;
  float load[256];
  int used[3][256];
  for (int w = 0; w < G; ++w) { load[w] = w < d.B ? 20.f : 0.f; used[0][w] = used[1][w] = w < d.B ? 1 : 0; used[2][w] = 0; }
  for (int w = 0; w < d.B; ++w) { K.sched[0][w][0] = (unsigned char)w; K.sched[1][w][0] = (unsigned char)w; }
  // xcd: a tile task prefers a workgroup of XCD (list index % 8) -- the row blocks of one weight slab / column tile have the same index
  // modulo 8 (tile_decode), so they share the slab in that XCD's L2
  auto place = [&](int pi, int first, int count, float cost, bool xcd) {
    for (int j = 0; j < count; ++j) {
      int best = -1; float bl = 0.f; int bu = 0;
      for (int w = 0; w < G; ++w) {
        if (used[pi][w] >= 2) continue;
        const float pl = load[w] + load[(w + G / 2) % G] + ((xcd && (w & 7) != ((first + j) & 7)) ? 1000.f : 0.f);
        if (best < 0 || used[pi][w] < bu || (used[pi][w] == bu && pl < bl)) { best = w; bl = pl; bu = used[pi][w]; }
      }
      if (best < 0) { K.sched_ok = -1; return; }
      K.sched[pi][best][used[pi][best]++] = (unsigned char)(first + j);
      load[best] += cost;
    }
  };
  place(1, d.B, nD, 11.f, (d.B & 7) == 0);
  place(2, 0, nF, 11.f, true);
  place(0, d.B, nel + nee, 3.f, false);
  K.sched_ok = K.sched_ok == 0 ? 1 : 0;
}
// Forward task schedule (phases B: p | g tiles; C: l tiles | attention rows | e tiles), same greedy as make_bwd_sched without fixed tasks.
void make_fwd_sched(PK& K, int G) {
  const Dims& d = K.d;
  K.sched_f_ok = 0;
  memset(K.sched_f, 255, sizeof(K.sched_f));
  const int ntp = tile_count(d.B, d.Dp, UW), ntg = tile_count(d.B, d.Dg, UW), nte = tile_count(d.B, d.De, UW);
  if (G != 256 || ntp + ntg >= 255 || ntp + d.B + nte >= 255) return;
  float load[256];
  int used[2][256];
  for (int w = 0; w < G; ++w) { load[w] = 0.f; used[0][w] = used[1][w] = 0; }
  bool ok = true;
  auto place = [&](int pi, int first, int count, float cost, int xcd_off) {      // xcd_off >= 0: tile list starting at that index
    for (int j = 0; j < count; ++j) {
      int best = -1; float bl = 0.f; int bu = 0;
      for (int w = 0; w < G; ++w) {
        if (used[pi][w] >= 2) continue;
        const float pl = load[w] + load[(w + G / 2) % G] + ((xcd_off >= 0 && (w & 7) != (j & 7)) ? 1000.f : 0.f);
        if (best < 0 || used[pi][w] < bu || (used[pi][w] == bu && pl < bl)) { best = w; bl = pl; bu = used[pi][w]; }
      }
      if (best < 0) { ok = false; return; }
      K.sched_f[pi][best][used[pi][best]++] = (unsigned char)(first + j);
      load[best] += cost;
    }
  };
  place(0, 0, ntp, 11.f, 0);            // p tiles (three MFMA chains)
  place(1, 0, ntp, 11.f, 0);            // l tiles
  place(0, ntp, ntg, 8.f, 0);           // g tiles
  place(1, ntp, d.B, 8.f, -1);          // attention rows
  place(1, ntp + d.B, nte, 7.f, 0);     // e tiles
  K.sched_f_ok = ok ? 1 : 0;
}
PK make_pk(const mser_drnn_desc& d, const WS& w) {
  PK K;
  K.d = Dims{d.T, d.B, d.Dm, d.Dg, d.Dp, d.De};
  K.p[0] = d.p[0]; K.p[1] = d.p[1];
  K.w = w;
  K.rng = (d.rng && d.p_drop > 0.f) ? d.rng : nullptr; K.site[0] = d.drop_site[0]; K.site[1] = d.drop_site[1]; K.pdrop = d.p_drop;
  K.out = d.out; K.dout = d.dout; K.ldo = d.ldo; K.rev = d.rev;
  K.sync = w.sync; K.fault = d.fault;
  K.sched_ok = 0; K.sched_f_ok = 0;
  return K;
}

}  // namespace
}  // namespace mser

using namespace mser;

extern "C" {

size_t mser_drnn_workspace_bytes(int32_t T, int32_t B, int32_t Dm, int32_t Dg, int32_t Dp, int32_t De) {
  Dims d{T, B, Dm, Dg, Dp, De};
  return carve(nullptr, d).bytes;
}

int mser_drnn_fwd(const mser_drnn_desc* dp, mser_stream_t stream) {
  if (!dp) { set_error("mser_drnn_fwd: null descriptor"); return -1; }
  const mser_drnn_desc& d = *dp;
  MSER_TRY(validate(d, false));
  hipStream_t s = (hipStream_t)stream;
  const Dims dm{d.T, d.B, d.Dm, d.Dg, d.Dp, d.De};
  const WS w = carve((char*)d.workspace, dm);
  const int T = d.T, B = d.B, Dm = d.Dm, Dg = d.Dg, Dp = d.Dp, De = d.De;
  const long TB = (long)T * B;
  const uint32_t* rng = (d.rng && d.p_drop > 0.f) ? d.rng : nullptr;
  const float p = d.p_drop;
  // ---- prep: direction-ordered inputs, party tables, zero initial states (index 0 of the T+1 long state arrays)
  hipLaunchKernelGGL(drnn_prep_kernel, dim3(cdiv(2 * (TB + B), 256)), dim3(256), 0, s, d.qmask, d.rev, w.qm, w.idx, T, B);
  hipLaunchKernelGGL(drnn_gather_rows_kernel, dim3(cdiv(TB * Dm, 256)), dim3(256), 0, s, d.U, (long)d.ldu, (const int*)nullptr, w.Ud, T, B, Dm);
  hipLaunchKernelGGL(drnn_gather_rows_kernel, dim3(cdiv(TB * Dm, 256)), dim3(256), 0, s, d.U, (long)d.ldu, d.rev, w.Ud + TB * Dm, T, B, Dm);
  MSER_TRY(check_launch("drnn_prep"));
  for (int dir = 0; dir < 2; ++dir) {
    MSER_CHECK_HIP(hipMemsetAsync(w.Gh + (long)dir * (T + 1) * B * Dg, 0, (size_t)B * Dg * sizeof(float), s));
    MSER_CHECK_HIP(hipMemsetAsync(w.Q + (long)dir * (T + 1) * B * 2 * Dp, 0, (size_t)B * 2 * Dp * sizeof(float), s));
    MSER_CHECK_HIP(hipMemsetAsync(w.Eh + (long)dir * (T + 1) * B * De, 0, (size_t)B * De * sizeof(float), s));
    MSER_CHECK_HIP(hipMemsetAsync(w.q0sel + (long)dir * TB * Dp, 0, (size_t)B * Dp * sizeof(float), s));      // q[b, s] of the zero state
    MSER_CHECK_HIP(hipMemsetAsync(w.cvec + (long)dir * TB * Dg, 0, (size_t)B * Dg * sizeof(float), s));       // c_0 = 0 (:137-139)
  }
  // ---- hoisted: the U halves of the three input products (+ b_ih) and W_att U, all steps, per direction
  for (int dir = 0; dir < 2; ++dir) {
    const mser_drnn_params& P = d.p[dir];
    const float* Ud = w.Ud + (long)dir * TB * Dm;
    struct { const float* W; long ld; const float* b; float* C; int N; } hs[4] = {
      {P.g_wih, (long)Dm + Dp, P.g_bih, w.GIg + (long)dir * TB * 3 * Dg, 3 * Dg},
      {P.p_wih, (long)Dm + Dg, P.p_bih, w.GIp + (long)dir * TB * 3 * Dp, 3 * Dp},
      {P.l_wih, (long)Dm + Dp, P.l_bih, w.GIl + (long)dir * TB * 3 * Dp, 3 * Dp},
      {P.att_w, (long)Dm, nullptr, w.Xatt + (long)dir * TB * Dg, Dg}};
    mser_gemm_desc hg[4];                 // one grouped launch per direction
    for (int i = 0; i < 4; ++i) {
      const auto& h = hs[i];
      mser_gemm_desc g = gd();
      g.A = Ud; g.B = h.W; g.C = h.C; g.M = (int)TB; g.N = h.N; g.K = Dm;
      g.sAm = Dm; g.sAk = 1; g.sBk = 1; g.sBn = h.ld; g.ldc = h.N; g.bias = h.b;
      hg[i] = g;
    }
    MSER_TRY(gemm_group(hg, 4, s));
  }
  if (persist_ok(dm, false)) {
    PK K = make_pk(d, w);
    make_fwd_sched(K, persist_grid());
    const size_t lds = persist_lds(dm, false);
    MSER_CHECK_HIP(hipMemsetAsync(w.sync, 0, 4096 * sizeof(unsigned), s));
    MSER_CHECK_HIP(hipMemsetAsync(w.apk, 0, w.apk_floats * sizeof(float), s));
    for (int dir = 0; dir < 2; ++dir) {
      const mser_drnn_params& P = d.p[dir];
      struct { const float* W; long ld; int H, K; } pk[8] = {
        {P.g_wih + Dm, (long)Dm + Dp, Dg, Dp}, {P.g_whh, (long)Dg, Dg, Dg}, {P.p_wih + Dm, (long)Dm + Dg, Dp, Dg}, {P.p_whh, (long)Dp, Dp, Dp},
        {P.l_wih + Dm, (long)Dm + Dp, Dp, Dp}, {P.l_whh, (long)Dp, Dp, Dp}, {P.e_wih, (long)Dp, De, Dp}, {P.e_whh, (long)De, De, De}};
      for (int i = 0; i < 8; ++i) {
        const long n_out = (long)((pk[i].H + UW - 1) / UW) * k8p(pk[i].K) * 256;
        hipLaunchKernelGGL(drnn_pack_w_kernel, dim3(cdiv(n_out, 256)), dim3(256), 0, s, pk[i].W, pk[i].ld, pk[i].H, pk[i].K,
                           w.wpk + (long)dir * w.wpk_dir + w.wpk_off[i], n_out);
      }
    }
    MSER_CHECK_HIP(hipFuncSetAttribute((const void*)drnn_fwd_persist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(drnn_store_pk_kernel, dim3(1), dim3(256), 0, s, K, (PK*)w.pk_dev);
    hipLaunchKernelGGL(drnn_fwd_persist, dim3(P_WGS_PER_CU * persist_grid()), dim3(PNT), lds, s, (const PK*)w.pk_dev);
    return check_launch("drnn_fwd_persist");
  }
  // the per-step hidden products (and gi_e) are accumulation targets cleared by their readers: zero them once (carved back to back)
  MSER_CHECK_HIP(hipMemsetAsync(w.gi_g, 0, (size_t)((char*)(w.gh_e + (size_t)2 * B * 3 * De) - (char*)w.gi_g), s));
  const dim3 blk(256);
  const long idx_ds = TB + B;
  for (int t = 0; t < T; ++t) {
    const float* Ght = w.Gh + (long)t * B * Dg;             float* Ghn = w.Gh + (long)(t + 1) * B * Dg;
    const float* Qt = w.Q + (long)t * B * 2 * Dp;           float* Qn = w.Q + (long)(t + 1) * B * 2 * Dp;
    const float* Et = w.Eh + (long)t * B * De;              float* En = w.Eh + (long)(t + 1) * B * De;
    const long g_ds = (long)(T + 1) * B * Dg, q_ds = (long)(T + 1) * B * 2 * Dp, e_ds = (long)(T + 1) * B * De;
    const float* q0s = w.q0sel + (long)t * B * Dp;
    // -- the five products that read only the state left by step t-1 (q[b,s_b], g_{t-1}, q, q, e_{t-1}), both directions each: ONE
    //    grouped launch (ten members) fills the chip where a product of its own keeps 47-94 workgroups busy for 18 us
    {
      mser_gemm_desc grp[5] = {
        mm_nt_desc(q0s, Dp, TB * Dp, d.p[0].g_wih + Dm, Dm + Dp, DS(g_wih), w.GIg + (long)t * B * 3 * Dg, 3 * Dg, TB * 3 * Dg, B, 3 * Dg, Dp, true),
        mm_nt_desc(Ght, Dg, g_ds, d.p[0].g_whh, Dg, DS(g_whh), w.gh_g, 3 * Dg, (long)B * 3 * Dg, B, 3 * Dg, Dg, true),
        mm_nt_desc(Qt, Dp, q_ds, d.p[0].p_whh, Dp, DS(p_whh), w.gh_p, 3 * Dp, (long)2 * B * 3 * Dp, 2 * B, 3 * Dp, Dp, true),
        mm_nt_desc(Qt, Dp, q_ds, d.p[0].l_whh, Dp, DS(l_whh), w.gh_l, 3 * Dp, (long)2 * B * 3 * Dp, 2 * B, 3 * Dp, Dp, true),
        mm_nt_desc(Et, De, e_ds, d.p[0].e_whh, De, DS(e_whh), w.gh_e, 3 * De, (long)B * 3 * De, B, 3 * De, De, true)};
      MSER_TRY(gemm_group(grp, 5, s));
    }
    // -- attention over g_0 .. g_{t-1}, then the p cell's input product (the chain of the step: attention -> p -> l -> e)
    if (t > 0) {
      hipLaunchKernelGGL(drnn_attn_fwd_kernel, dim3(B, 2), dim3(ATT_NT), (size_t)(Dg + T) * sizeof(float), s, B, Dg, T, t,
                         w.Xatt + (long)t * B * Dg, TB * Dg, w.Gh, g_ds, w.alpha + (long)t * B * T, TB * T, w.cvec + (long)t * B * Dg, TB * Dg);
    }
    MSER_TRY(mm_nt(s, w.cvec + (long)t * B * Dg, Dg, TB * Dg, d.p[0].p_wih + Dm, Dm + Dg, DS(p_wih), w.GIp + (long)t * B * 3 * Dp, 3 * Dp, TB * 3 * Dp, B,
                   3 * Dp, Dg, true));                  // (in place on the hoisted U part of this step's input product)
    // -- g cell epilogue (g_t is first read by step t+1)
    hipLaunchKernelGGL(drnn_g_fwd_kernel, dim3(cdiv((long)B * Dg, 256), 1, 2), blk, 0, s, B, Dg, w.GIg + (long)t * B * 3 * Dg, TB * 3 * Dg, w.gh_g, d.p[0].g_bhh, DS(g_bhh), Ght, Ghn,
                       g_ds, w.sv_g + (long)t * B * 4 * Dg, TB * 4 * Dg, rng, d.drop_site[0], d.drop_site[1], p, (uint32_t)((long)t * B * Dg));
    // -- p cell (both parties)
    float* qs = w.dqs;            // (forward: scratch for the p cell's dropped output; the backward reuses the buffer)
    hipLaunchKernelGGL(drnn_p_fwd_kernel, dim3(cdiv((long)B * Dp, 256), 1, 2), blk, 0, s, B, Dp, w.GIp + (long)t * B * 3 * Dp, TB * 3 * Dp, w.gh_p, d.p[0].p_bhh, DS(p_bhh), Qt, q_ds,
                       qs, w.sv_p + (long)t * B * 2 * 4 * Dp, TB * 2 * 4 * Dp, w.idx + (long)t * B, idx_ds, w.ss + (long)t * B * Dp, TB * Dp, rng,
                       d.drop_site[0] + 1, d.drop_site[1] + 1, p, (uint32_t)((long)t * B * 2 * Dp));
    // -- l cell + blend
    MSER_TRY(mm_nt(s, w.ss + (long)t * B * Dp, Dp, TB * Dp, d.p[0].l_wih + Dm, Dm + Dp, DS(l_wih), w.GIl + (long)t * B * 3 * Dp, 3 * Dp, TB * 3 * Dp, B,
                   3 * Dp, Dp, true));
    float* q0n = (t + 1 < T) ? w.q0sel + (long)(t + 1) * B * Dp : w.dss;         // (last step: a scratch target)
    hipLaunchKernelGGL(drnn_l_fwd_kernel, dim3(cdiv((long)B * Dp, 256), 1, 2), blk, 0, s, B, Dp, w.GIl + (long)t * B * 3 * Dp, TB * 3 * Dp, w.gh_l, d.p[0].l_bhh, DS(l_bhh), Qt, Qn, q_ds,
                       qs, w.sv_l + (long)t * B * 2 * 4 * Dp, TB * 2 * 4 * Dp, w.qm + (long)t * B * 2, TB * 2, w.idx + (long)t * B,
                       w.idx + (long)(t + 1) * B, idx_ds, w.qsel + (long)t * B * Dp, TB * Dp, q0n, (t + 1 < T) ? TB * Dp : (long)B * Dp, rng,
                       d.drop_site[0] + 2, d.drop_site[1] + 2, p, (uint32_t)((long)t * B * 2 * Dp));
    // -- e cell (its input is not U, so b_ih is not part of a hoisted product: the epilogue adds it)
    MSER_TRY(mm_nt(s, w.qsel + (long)t * B * Dp, Dp, TB * Dp, d.p[0].e_wih, Dp, DS(e_wih), w.gi_e, 3 * De, (long)B * 3 * De, B, 3 * De, Dp, true));
    hipLaunchKernelGGL(drnn_e_fwd_kernel, dim3(cdiv((long)B * De, 256), 1, 2), blk, 0, s, B, De, w.gi_e, w.gh_e, d.p[0].e_bih, d.p[0].e_bhh, DS(e_bhh), Et, En, e_ds,
                       w.sv_e + (long)t * B * 4 * De, TB * 4 * De, d.out, (long)d.ldo, d.rev, t, rng, d.drop_site[0] + 3, d.drop_site[1] + 3, p,
                       (uint32_t)((long)t * B * De));
    MSER_TRY(check_launch("drnn_fwd step"));
  }
  return 0;
}

int mser_drnn_bwd(const mser_drnn_desc* dp, mser_stream_t stream) {
  if (!dp) { set_error("mser_drnn_bwd: null descriptor"); return -1; }
  const mser_drnn_desc& d = *dp;
  MSER_TRY(validate(d, true));
  hipStream_t s = (hipStream_t)stream;
  const Dims dm{d.T, d.B, d.Dm, d.Dg, d.Dp, d.De};
  const WS w = carve((char*)d.workspace, dm);
  const int T = d.T, B = d.B, Dm = d.Dm, Dg = d.Dg, Dp = d.Dp, De = d.De;
  const long TB = (long)T * B;
  const uint32_t* rng = (d.rng && d.p_drop > 0.f) ? d.rng : nullptr;
  const float p = d.p_drop;
  const long g_ds = (long)(T + 1) * B * Dg, q_ds = (long)(T + 1) * B * 2 * Dp, e_ds = (long)(T + 1) * B * De;
  const long idx_ds = TB + B;
  MSER_CHECK_HIP(hipMemsetAsync(w.dGh, 0, (size_t)2 * (T + 1) * B * Dg * sizeof(float), s));
  if (persist_ok(dm, true)) {
    // ---- persistent BPTT: one launch (drnn_bwd_persist)
    MSER_CHECK_HIP(hipMemsetAsync(w.bk, 0, w.bk_floats * sizeof(float), s));
    MSER_CHECK_HIP(hipMemsetAsync(w.sync, 0, 4096 * sizeof(unsigned), s));
    for (int dir = 0; dir < 2; ++dir) {
      MSER_CHECK_HIP(hipMemsetAsync(w.dXatt + (long)dir * TB * Dg, 0, (size_t)B * Dg * sizeof(float), s));       // step 0 has no attention
      const mser_drnn_params& P = d.p[dir];
      struct { const float* W; long ld; int K3, N; } pk[8] = {
        {P.e_wih, (long)Dp, 3 * De, Dp}, {P.e_whh, (long)De, 3 * De, De}, {P.l_wih + Dm, (long)Dm + Dp, 3 * Dp, Dp}, {P.l_whh, (long)Dp, 3 * Dp, Dp},
        {P.p_wih + Dm, (long)Dm + Dg, 3 * Dp, Dg}, {P.p_whh, (long)Dp, 3 * Dp, Dp}, {P.g_wih + Dm, (long)Dm + Dp, 3 * Dg, Dp}, {P.g_whh, (long)Dg, 3 * Dg, Dg}};
      for (int i = 0; i < 8; ++i) {
        const long n_out = (long)((pk[i].N + 31) / 32) * k8p(pk[i].K3) * 256;
        hipLaunchKernelGGL(drnn_pack_wt_kernel, dim3(cdiv(n_out, 256)), dim3(256), 0, s, pk[i].W, pk[i].ld, pk[i].K3, pk[i].N,
                           w.wpk + (long)dir * w.wtk_dir + w.wtk_off[i], n_out);
      }
    }
    PK K = make_pk(d, w);
    make_bwd_sched(K, persist_grid());
    const size_t lds = persist_lds(dm, true);
    MSER_CHECK_HIP(hipFuncSetAttribute((const void*)drnn_bwd_persist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(drnn_store_pk_kernel, dim3(1), dim3(256), 0, s, K, (PK*)w.pk_dev);
    hipLaunchKernelGGL(drnn_bwd_persist, dim3(P_WGS_PER_CU * persist_grid()), dim3(PNT), lds, s, (const PK*)w.pk_dev);
    MSER_TRY(check_launch("drnn_bwd_persist"));
  } else {
  MSER_CHECK_HIP(hipMemsetAsync(w.dQ, 0, (size_t)2 * 2 * B * 2 * Dp * sizeof(float), s));
  MSER_CHECK_HIP(hipMemsetAsync(w.dEc, 0, (size_t)2 * B * De * sizeof(float), s));
  MSER_CHECK_HIP(hipMemsetAsync(w.dq0sel, 0, (size_t)2 * 2 * B * Dp * sizeof(float), s));
  // accumulation targets of the per-step data-gradient products: zero once, their readers clear them again (see drnn_attn_bwd_kernel)
  MSER_CHECK_HIP(hipMemsetAsync(w.dqsel, 0, (size_t)2 * B * Dp * sizeof(float), s));
  MSER_CHECK_HIP(hipMemsetAsync(w.dss, 0, (size_t)2 * B * Dp * sizeof(float), s));
  MSER_CHECK_HIP(hipMemsetAsync(w.dc, 0, (size_t)2 * B * Dg * sizeof(float), s));
  const dim3 blk(256);
  const long dq_ds = (long)B * 2 * Dp;                     // dir stride inside one ping-pong half of dQ
  int pp = 0;
  for (int t = T - 1; t >= 0; --t) {
    float* dQn = w.dQ + (long)pp * 2 * dq_ds;               // gradient at Q[t+1] left by step t+1
    float* dQc = w.dQ + (long)(1 - pp) * 2 * dq_ds;         // gradient at Q[t] built by this step
    float* dq0n = w.dq0sel + (long)((t + 1) & 1) * 2 * B * Dp;
    float* dq0c = w.dq0sel + (long)(t & 1) * 2 * B * Dp;
    // -- e cell
    hipLaunchKernelGGL(drnn_e_bwd_kernel, dim3(cdiv((long)B * De, 256), 1, 2), blk, 0, s, B, De, d.dout, (long)d.ldo, d.rev, t, w.dEc, (long)B * De,
                       w.sv_e + (long)t * B * 4 * De, TB * 4 * De, w.Eh + (long)t * B * De, e_ds, w.dgi_e + (long)t * B * 3 * De,
                       w.dgh_e + (long)t * B * 3 * De, TB * 3 * De, rng, d.drop_site[0] + 3, d.drop_site[1] + 3, p, (uint32_t)((long)t * B * De));
    MSER_TRY(mm_nn(s, w.dgi_e + (long)t * B * 3 * De, 3 * De, TB * 3 * De, d.p[0].e_wih, Dp, DS(e_wih), w.dqsel, Dp, (long)B * Dp, B, Dp, 3 * De, true));
    // -- l cell + blend
    hipLaunchKernelGGL(drnn_l_bwd_kernel, dim3(cdiv((long)B * Dp, 256), 1, 2), blk, 0, s, B, Dp, dQn, dQc, dq_ds, w.dqsel, dq0n, (long)B * Dp,
                       w.idx + (long)t * B, w.idx + (long)(t + 1) * B, idx_ds, w.qm + (long)t * B * 2, TB * 2, w.sv_l + (long)t * B * 2 * 4 * Dp,
                       TB * 2 * 4 * Dp, w.Q + (long)t * B * 2 * Dp, q_ds, w.dgi_l + (long)t * B * 3 * Dp, TB * 3 * Dp,
                       w.dgh_l + (long)t * 2 * B * 3 * Dp, TB * 2 * 3 * Dp, w.dqs, rng, d.drop_site[0] + 2, d.drop_site[1] + 2, p,
                       (uint32_t)((long)t * B * 2 * Dp), t + 1 < T ? 1 : 0);
    MSER_TRY(mm_nn(s, w.dgi_l + (long)t * B * 3 * Dp, 3 * Dp, TB * 3 * Dp, d.p[0].l_wih + Dm, Dm + Dp, DS(l_wih), w.dss, Dp, (long)B * Dp, B, Dp, 3 * Dp, true));
    // -- p cell
    hipLaunchKernelGGL(drnn_p_bwd_kernel, dim3(cdiv((long)B * Dp, 256), 1, 2), blk, 0, s, B, Dp, w.dqs, w.dss, (long)B * Dp, w.idx + (long)t * B, idx_ds,
                       dQc, dq_ds, w.sv_p + (long)t * B * 2 * 4 * Dp, TB * 2 * 4 * Dp, w.Q + (long)t * B * 2 * Dp, q_ds,
                       w.dgi_p + (long)t * B * 3 * Dp, TB * 3 * Dp, w.dgh_p + (long)t * 2 * B * 3 * Dp, TB * 2 * 3 * Dp, rng, d.drop_site[0] + 1,
                       d.drop_site[1] + 1, p, (uint32_t)((long)t * B * 2 * Dp));
    MSER_TRY(mm_nn(s, w.dgi_p + (long)t * B * 3 * Dp, 3 * Dp, TB * 3 * Dp, d.p[0].p_wih + Dm, Dm + Dg, DS(p_wih), w.dc, Dg, (long)B * Dg, B, Dg, 3 * Dp, true));
    // -- attention over the history
    if (t > 0) {
      hipLaunchKernelGGL(drnn_attn_bwd_kernel, dim3(B, 2), dim3(ATT_NT), (size_t)(3 * Dg + 2 * T) * sizeof(float), s, B, Dg, T, t,
                         w.Xatt + (long)t * B * Dg, TB * Dg, w.Gh, w.dGh, g_ds, w.alpha + (long)t * B * T, TB * T, w.dc, (long)B * Dg,
                         w.dXatt + (long)t * B * Dg);
    } else {
      for (int dir = 0; dir < 2; ++dir) MSER_CHECK_HIP(hipMemsetAsync(w.dXatt + (long)dir * TB * Dg, 0, (size_t)B * Dg * sizeof(float), s));
    }
    // -- g cell
    hipLaunchKernelGGL(drnn_g_bwd_kernel, dim3(cdiv((long)B * Dg, 256), 1, 2), blk, 0, s, B, Dg, w.dGh + (long)(t + 1) * B * Dg, w.dGh + (long)t * B * Dg,
                       g_ds, w.sv_g + (long)t * B * 4 * Dg, TB * 4 * Dg, w.Gh + (long)t * B * Dg, w.dgi_g + (long)t * B * 3 * Dg,
                       w.dgh_g + (long)t * B * 3 * Dg, TB * 3 * Dg, rng, d.drop_site[0], d.drop_site[1], p, (uint32_t)((long)t * B * Dg));
    MSER_TRY(mm_nn(s, w.dgi_g + (long)t * B * 3 * Dg, 3 * Dg, TB * 3 * Dg, d.p[0].g_wih + Dm, Dm + Dp, DS(g_wih), dq0c, Dp, (long)B * Dp, B, Dp, 3 * Dg, true));
    // -- the four hidden-path products of the step (into the state gradients that step t-1 reads): one grouped launch, split-K atomics
    {
      mser_gemm_desc grp[4] = {
        mm_nn_acc_desc(w.dgh_e + (long)t * B * 3 * De, 3 * De, TB * 3 * De, d.p[0].e_whh, De, DS(e_whh), w.dEc, De, (long)B * De, B, De, 3 * De),
        mm_nn_acc_desc(w.dgh_l + (long)t * 2 * B * 3 * Dp, 3 * Dp, TB * 2 * 3 * Dp, d.p[0].l_whh, Dp, DS(l_whh), dQc, Dp, dq_ds, 2 * B, Dp, 3 * Dp),
        mm_nn_acc_desc(w.dgh_p + (long)t * 2 * B * 3 * Dp, 3 * Dp, TB * 2 * 3 * Dp, d.p[0].p_whh, Dp, DS(p_whh), dQc, Dp, dq_ds, 2 * B, Dp, 3 * Dp),
        mm_nn_acc_desc(w.dgh_g + (long)t * B * 3 * Dg, 3 * Dg, TB * 3 * Dg, d.p[0].g_whh, Dg, DS(g_whh), w.dGh + (long)t * B * Dg, Dg, g_ds, B, Dg, 3 * Dg)};
      MSER_TRY(gemm_group(grp, 4, s));
    }
    MSER_TRY(check_launch("drnn_bwd step"));
    pp ^= 1;
  }
  }
  // ---- parameter gradients: reductions over all (t, b) rows of one direction, a few large GEMMs each
  for (int dir = 0; dir < 2; ++dir) {
    const mser_drnn_params& G = d.g[dir];
    const float* Ud = w.Ud + (long)dir * TB * Dm;
    const float* dgi_g = w.dgi_g + (long)dir * TB * 3 * Dg; const float* dgh_g = w.dgh_g + (long)dir * TB * 3 * Dg;
    const float* dgi_p = w.dgi_p + (long)dir * TB * 3 * Dp; const float* dgh_p = w.dgh_p + (long)dir * TB * 2 * 3 * Dp;
    const float* dgi_l = w.dgi_l + (long)dir * TB * 3 * Dp; const float* dgh_l = w.dgh_l + (long)dir * TB * 2 * 3 * Dp;
    const float* dgi_e = w.dgi_e + (long)dir * TB * 3 * De; const float* dgh_e = w.dgh_e + (long)dir * TB * 3 * De;
    // the twelve weight-gradient products of the direction as ONE grouped launch (no tail / launch gap between them), the bias gradients
    // (column sums of the gate gradients) after it
    const mser_gemm_desc grp[12] = {
      wgrad_desc(dgi_g, 3 * Dg, Ud, Dm, G.g_wih, Dm + Dp, (int)TB, 3 * Dg, Dm),
      wgrad_desc(dgi_g, 3 * Dg, w.q0sel + (long)dir * TB * Dp, Dp, G.g_wih + Dm, Dm + Dp, (int)TB, 3 * Dg, Dp),
      wgrad_desc(dgh_g, 3 * Dg, w.Gh + (long)dir * g_ds, Dg, G.g_whh, Dg, (int)TB, 3 * Dg, Dg),
      wgrad_desc(dgi_p, 3 * Dp, Ud, Dm, G.p_wih, Dm + Dg, (int)TB, 3 * Dp, Dm),
      wgrad_desc(dgi_p, 3 * Dp, w.cvec + (long)dir * TB * Dg, Dg, G.p_wih + Dm, Dm + Dg, (int)TB, 3 * Dp, Dg),
      wgrad_desc(dgh_p, 3 * Dp, w.Q + (long)dir * q_ds, Dp, G.p_whh, Dp, (int)(2 * TB), 3 * Dp, Dp),
      wgrad_desc(dgi_l, 3 * Dp, Ud, Dm, G.l_wih, Dm + Dp, (int)TB, 3 * Dp, Dm),
      wgrad_desc(dgi_l, 3 * Dp, w.ss + (long)dir * TB * Dp, Dp, G.l_wih + Dm, Dm + Dp, (int)TB, 3 * Dp, Dp),
      wgrad_desc(dgh_l, 3 * Dp, w.Q + (long)dir * q_ds, Dp, G.l_whh, Dp, (int)(2 * TB), 3 * Dp, Dp),
      wgrad_desc(dgi_e, 3 * De, w.qsel + (long)dir * TB * Dp, Dp, G.e_wih, Dp, (int)TB, 3 * De, Dp),
      wgrad_desc(dgh_e, 3 * De, w.Eh + (long)dir * e_ds, De, G.e_whh, De, (int)TB, 3 * De, De),
      wgrad_desc(w.dXatt + (long)dir * TB * Dg, Dg, Ud, Dm, G.att_w, Dm, (int)TB, Dg, Dm)};
    MSER_TRY(gemm_group(grp, 12, s));
    MSER_TRY(mser_colsum_acc(dgi_g, TB, 3 * Dg, 3 * Dg, G.g_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_g, TB, 3 * Dg, 3 * Dg, G.g_bhh, s));
    MSER_TRY(mser_colsum_acc(dgi_p, TB, 3 * Dp, 3 * Dp, G.p_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_p, 2 * TB, 3 * Dp, 3 * Dp, G.p_bhh, s));
    MSER_TRY(mser_colsum_acc(dgi_l, TB, 3 * Dp, 3 * Dp, G.l_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_l, 2 * TB, 3 * Dp, 3 * Dp, G.l_bhh, s));
    MSER_TRY(mser_colsum_acc(dgi_e, TB, 3 * De, 3 * De, G.e_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_e, TB, 3 * De, 3 * De, G.e_bhh, s));
  }
  return 0;
}

int mser_general2_rows_fwd(const float* S0, float* alpha, const float* mask, int64_t rows, int32_t n, int32_t L, mser_stream_t stream) {
  MSER_REQUIRE(S0 && alpha && mask && n > 0 && L > 0, "mser_general2_rows_fwd: bad arguments");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(general2_fwd_kernel, dim3(cdiv(rows, G2_WPB)), dim3(64 * G2_WPB), 0, (hipStream_t)stream, S0, alpha, mask, (long)rows, n, L);
  return check_launch("mser_general2_rows_fwd");
}
int mser_general2_rows_bwd(const float* S0, const float* mask, float* dA, int64_t rows, int32_t n, int32_t L, mser_stream_t stream) {
  MSER_REQUIRE(S0 && dA && mask && n > 0 && L > 0, "mser_general2_rows_bwd: bad arguments");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(general2_bwd_kernel, dim3(cdiv(rows, G2_WPB)), dim3(64 * G2_WPB), 0, (hipStream_t)stream, S0, mask, dA, (long)rows, n, L);
  return check_launch("mser_general2_rows_bwd");
}

/* alpha_f / alpha_b of BiModel.forward (model/DialogueRNN.py:196,:240,:250): the attention map of direction `dir` after mser_drnn_fwd,
 * [T][B][T] with row (t, b) valid in its first t entries. */
int mser_drnn_alpha(const mser_drnn_desc* dp, int32_t dir, const float** alpha) {
  if (!dp || !alpha || dir < 0 || dir > 1) { set_error("mser_drnn_alpha: bad arguments"); return -1; }
  const Dims dm{dp->T, dp->B, dp->Dm, dp->Dg, dp->Dp, dp->De};
  const WS w = carve((char*)dp->workspace, dm);
  *alpha = w.alpha + (long)dir * dp->T * dp->B * dp->T;
  return 0;
}

}  // extern "C"
