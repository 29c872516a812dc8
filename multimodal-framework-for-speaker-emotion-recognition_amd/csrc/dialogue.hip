// DialogueRNN (reference model/DialogueRNN.py:80-198; SURVEY 8(f) row f2, BASELINE configs[3]) on gfx950: the two directions of a
// BiModel share every launch.  Per time step and direction (listener_state=True, 'general' attention over the growing history):
//   g  = dropout(GRU_g([U_t | q[b,s_b]], g_{t-1}))                      s_b = argmax(qmask[t,b])               (:129-136)
//   c  = sum_{s<t} softmax_s(<W_att U_t, g_s>) g_s   (zeros at t = 0)                                            (:137-141, :56-59,:75)
//   qs = dropout(GRU_p([U_t | c], q[b,p])), p = 0,1 ;  ql = dropout(GRU_l([U_t | qs[b,s_b]], q[b,p]))            (:144-153)
//   q  = ql (1 - qmask) + qs qmask ;  e = dropout(GRU_e(q[b,s_b], e_{t-1}))                                      (:156-161)
// Structure of this first version: the U-dependent halves of the four input products and W_att U are ONE GEMM each over all
// steps (hoisted); inside the loop a step is 8 direction-batched GEMMs on the fp32 MFMA (exact fmaf chains: the gate is 1e-4 on
// log-probs after 200 dependent steps) + 4 gate epilogues + 1 history-attention launch, issued from this host loop (no Python
// between launches; capturable).  The BPTT mirrors it (8 GEMMs + 5 launches per step); every weight gradient is a reduction over
// all (t, b) rows and runs as a few large GEMMs after the loop.  Widths at the reference's configuration (D_g = D_p = 500, 21 MB
// of fp32 weights per direction) do not fit a register-resident persistent chain the way the LSTHM cell does; the step is
// MFMA-rate bound (0.9 GFLOP per step and direction at B = 64).
#include "common.h"
#include "../../include/mser.h"
#include <cstring>

namespace mser {

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
int wgrad(hipStream_t s, const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, int rows, int N, int K) {
  mser_gemm_desc g = gd();
  g.A = dY; g.B = X; g.C = dW; g.M = N; g.N = K; g.K = rows;
  g.sAm = 1; g.sAk = ldy; g.sBk = ldx; g.sBn = 1; g.ldc = ldw;
  g.flags = MSER_GEMM_ACCUM;
  g.splitk = 2;                      // "C is initialised, accumulate atomically"; the split itself is chosen by mser::gemm
  return gemm(g, s);
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
    for (auto& h : hs) {
      mser_gemm_desc g = gd();
      g.A = Ud; g.B = h.W; g.C = h.C; g.M = (int)TB; g.N = h.N; g.K = Dm;
      g.sAm = Dm; g.sAk = 1; g.sBk = 1; g.sBn = h.ld; g.ldc = h.N; g.bias = h.b;
      MSER_TRY(gemm(g, s));
    }
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
  // ---- parameter gradients: reductions over all (t, b) rows of one direction, a few large GEMMs each
  for (int dir = 0; dir < 2; ++dir) {
    const mser_drnn_params& G = d.g[dir];
    const float* Ud = w.Ud + (long)dir * TB * Dm;
    const float* dgi_g = w.dgi_g + (long)dir * TB * 3 * Dg; const float* dgh_g = w.dgh_g + (long)dir * TB * 3 * Dg;
    const float* dgi_p = w.dgi_p + (long)dir * TB * 3 * Dp; const float* dgh_p = w.dgh_p + (long)dir * TB * 2 * 3 * Dp;
    const float* dgi_l = w.dgi_l + (long)dir * TB * 3 * Dp; const float* dgh_l = w.dgh_l + (long)dir * TB * 2 * 3 * Dp;
    const float* dgi_e = w.dgi_e + (long)dir * TB * 3 * De; const float* dgh_e = w.dgh_e + (long)dir * TB * 3 * De;
    // g cell
    MSER_TRY(wgrad(s, dgi_g, 3 * Dg, Ud, Dm, G.g_wih, Dm + Dp, (int)TB, 3 * Dg, Dm));
    MSER_TRY(wgrad(s, dgi_g, 3 * Dg, w.q0sel + (long)dir * TB * Dp, Dp, G.g_wih + Dm, Dm + Dp, (int)TB, 3 * Dg, Dp));
    MSER_TRY(wgrad(s, dgh_g, 3 * Dg, w.Gh + (long)dir * g_ds, Dg, G.g_whh, Dg, (int)TB, 3 * Dg, Dg));
    MSER_TRY(mser_colsum_acc(dgi_g, TB, 3 * Dg, 3 * Dg, G.g_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_g, TB, 3 * Dg, 3 * Dg, G.g_bhh, s));
    // p cell
    MSER_TRY(wgrad(s, dgi_p, 3 * Dp, Ud, Dm, G.p_wih, Dm + Dg, (int)TB, 3 * Dp, Dm));
    MSER_TRY(wgrad(s, dgi_p, 3 * Dp, w.cvec + (long)dir * TB * Dg, Dg, G.p_wih + Dm, Dm + Dg, (int)TB, 3 * Dp, Dg));
    MSER_TRY(wgrad(s, dgh_p, 3 * Dp, w.Q + (long)dir * q_ds, Dp, G.p_whh, Dp, (int)(2 * TB), 3 * Dp, Dp));
    MSER_TRY(mser_colsum_acc(dgi_p, TB, 3 * Dp, 3 * Dp, G.p_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_p, 2 * TB, 3 * Dp, 3 * Dp, G.p_bhh, s));
    // l cell
    MSER_TRY(wgrad(s, dgi_l, 3 * Dp, Ud, Dm, G.l_wih, Dm + Dp, (int)TB, 3 * Dp, Dm));
    MSER_TRY(wgrad(s, dgi_l, 3 * Dp, w.ss + (long)dir * TB * Dp, Dp, G.l_wih + Dm, Dm + Dp, (int)TB, 3 * Dp, Dp));
    MSER_TRY(wgrad(s, dgh_l, 3 * Dp, w.Q + (long)dir * q_ds, Dp, G.l_whh, Dp, (int)(2 * TB), 3 * Dp, Dp));
    MSER_TRY(mser_colsum_acc(dgi_l, TB, 3 * Dp, 3 * Dp, G.l_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_l, 2 * TB, 3 * Dp, 3 * Dp, G.l_bhh, s));
    // e cell
    MSER_TRY(wgrad(s, dgi_e, 3 * De, w.qsel + (long)dir * TB * Dp, Dp, G.e_wih, Dp, (int)TB, 3 * De, Dp));
    MSER_TRY(wgrad(s, dgh_e, 3 * De, w.Eh + (long)dir * e_ds, De, G.e_whh, De, (int)TB, 3 * De, De));
    MSER_TRY(mser_colsum_acc(dgi_e, TB, 3 * De, 3 * De, G.e_bih, s));
    MSER_TRY(mser_colsum_acc(dgh_e, TB, 3 * De, 3 * De, G.e_bhh, s));
    // attention transform
    MSER_TRY(wgrad(s, w.dXatt + (long)dir * TB * Dg, Dg, Ud, Dm, G.att_w, Dm, (int)TB, Dg, Dm));
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
