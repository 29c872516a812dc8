// Speaker state of the GRU-speaker variants (reference model/lsthm_onlysp.py:170-181, the CLI's default model):
//   U_t = [x_l[t] | x_a[t]],  qs0 = q[b, party_t[b]],  h_s = dropout(GRUCell(U_t, qs0)),  q[b, p] = q[b, p] (1 - qmask_t[b, p]) + h_s qmask_t[b, p]
// There is no slot compaction: every dialogue carries its own two party states, so the recurrence is batch-independent and ONE
// workgroup owns a 16-dialogue block for the whole sequence -- no inter-workgroup hand-off at all.  (16, not 32: the step is bound
// by the CU's fp32 MFMA rate, and with v_mfma_f32_16x16x4_f32 a block of 16 rows costs half the MFMA cycles of a 32-row block, so
// a batch of 32 dialogues runs on two CUs per direction at half the step time.)  The input product
// gi = U W_ih^T + b_ih is one GEMM before the chain (caller); only W_hh [3H, H] sits in the loop, register-resident: 12 waves,
// wave w = (gate g = w / 4, unit slice s = w % 4) owns the 16 x 32 piece gh[:, g*H + s*32 ..] = qs0 W_hh[g*H + s*32 .., :]^T as two
// 16 x 16 tiles (2 x 32 MFMA 16x16x4 per step, its 64 B values per lane loaded once).  The party states live in LDS.
// The backward is the same structure in reverse time: dqs0 = dgh W_hh (wave = (unit slice, gate third of K)), the three partial
// tiles meet in LDS.  Weight gradients (dW_ih = dgi^T U, dW_hh = dgh^T qs0), bias sums and dU = dgi W_ih are GEMMs after the chain
// (caller).  gfx950 only; H = 128.
#include "common.h"
#include "../../include/mser.h"

namespace mser {
namespace {

constexpr int GH = 128;            // the reference's dh_s
constexpr int GNT = 768;           // 12 waves
constexpr int QS = 2 * GH + 4;     // LDS row stride of the party states [RB][2][H]: +4 words, so that the 16-byte A reads of eight
constexpr int TS = 3 * GH + 4;     // consecutive rows cover all 32 banks; likewise for the gate tiles [RB][3H]
constexpr int RB = 16;             // dialogues per workgroup
constexpr int NEL = (RB * GH + GNT - 1) / GNT;     // (row, unit) elements per thread and step

struct GruArgs {
  int T, B;
  const float* gi; const float* w_hh; const float* b_hh; const float* qmask;
  float* hs; float* out; long ldo; const int* rev;
  float* save;           // [T*B][5H]: qs0 | r | z | n | gh_n
  const float* dhs; const float* dhs2; const float* dhs3; float* dgi; float* dgh;
  // backward link to a producer kernel that runs concurrently: step t may start once *sub_cnt >= sub_per_step * (T - t); its
  // gradient rows (dhs and sub_nparts parts, sub_stride floats apart) are then read with device-coherent loads
  const unsigned* sub_cnt; unsigned sub_per_step; const float* sub_parts; int sub_nparts; long sub_stride; unsigned* status;
  const uint32_t* rng; uint32_t site; float p;
  // lblend (model/lsthm_nsps.py:188-191): q[b,p] = ql_0 (1 - m_p) + h_s m_p with ql_0 = the state of the party NOT speaking, instead of
  // q[b,p] (1 - m_p) + h_s m_p; hli / dhli (optional): the ql_0 rows [T*B, H] as an output (the cell's h_li) and their gradient
  int lblend; float* hli; const float* dhli;
  // forward link to a consumer kernel (mser_cell_desc::ext_linked).  The consumer waits for counter >= pub_inc * (t + 1) = "EVERY
  // block of this chain has published step t".  The blocks run without a barrier among themselves, so they must not simply add
  // shares to the counter (a block two steps ahead would stand in for one that is behind): each block keeps its own progress word
  // (pub_progress[block], zeroed by the caller) and, after a step, raises every counter replica to pub_inc * min over the blocks'
  // progress words with atomicMax -- a lower bound of the true minimum at any later time, and the slowest block's own update makes
  // it exact.
  unsigned* pub_cnt; unsigned pub_inc; int pub_rep, pub_stride; unsigned* pub_progress;
};

// 16 x 32 x 128 product of one wave as two 16 x 16 tiles: A row (lane & 15) from LDS, B from registers.  v_mfma_f32_16x16x4_f32
// takes A[l&15][k = l>>4] and B[k = l>>4][l&15]; the order in which the reduction index is fed is free as long as A and B agree,
// so lane quarter h = l >> 4 takes k = 32 h + j, j = 0..31: its A values are 32 consecutive LDS words (16-byte reads).
__device__ __forceinline__ void wave_mm16(const float* arow, const float (&b0)[32], const float (&b1)[32], f32x4& acc0, f32x4& acc1) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 av = *reinterpret_cast<const float4*>(arow + 4 * c);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0[4 * c + 0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1[4 * c + 0], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0[4 * c + 1], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1[4 * c + 1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0[4 * c + 2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1[4 * c + 2], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0[4 * c + 3], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1[4 * c + 3], acc1, 0, 0, 0);
  }
}

struct GruArgs2 { GruArgs d[2]; };        // blockIdx.y selects the chain (the two directions of a bidirectional cell share a launch)

__global__ __launch_bounds__(GNT) void gru_speaker_fwd_kernel(GruArgs2 aa) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const GruArgs& a = aa.d[blockIdx.y];
  float* q = sm;                   // [RB][QS]
  float* gh = q + RB * QS;         // [RB][TS]
  int* party = (int*)(gh + RB * TS);      // [RB]
  float* qmv = (float*)(party + 32);      // [RB][2]
  const int H = GH, B = a.B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, h4 = lane >> 4;
  const int g = wave >> 2, s = wave & 3;
  const int b0 = blockIdx.x * RB;
  // W_hh slices of this wave, two 16-column tiles: B[k][n] = W_hh[g*H + s*32 + 16 i + n][k], k = 32 h4 + j
  float breg[32], breg1[32];
  {
    const float* w = a.w_hh + (long)(g * H + s * 32 + r) * H + h4 * 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) { breg[j] = w[j]; breg1[j] = w[(long)16 * H + j]; }
  }
  const float bias = a.b_hh[g * H + s * 32 + r], bias1 = a.b_hh[g * H + s * 32 + 16 + r];
  for (int e = tid; e < RB * QS; e += GNT) q[e] = 0.f;
  DropKey dk;
  if (a.rng) dk = drop_key(a.rng, a.site, a.p);
  // the step's qmask row is fetched one step ahead (threads 0..RB-1 hold it): a dependent global load would sit on every step's path
  float nm0 = 0.f, nm1 = 0.f;
  if (tid < RB && b0 + tid < B) { nm0 = a.qmask[(long)(b0 + tid) * 2]; nm1 = a.qmask[(long)(b0 + tid) * 2 + 1]; }
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    if (tid < RB) {
      const float m0 = nm0, m1 = nm1;
      qmv[tid * 2] = m0; qmv[tid * 2 + 1] = m1;
      party[tid] = m1 > m0 ? 1 : 0;          // argmax(qmask[t], 1): ties and padded (all-zero) rows -> party 0 (:175)
      if (t + 1 < a.T && b0 + tid < B) {
        nm0 = a.qmask[((long)(t + 1) * B + b0 + tid) * 2]; nm1 = a.qmask[((long)(t + 1) * B + b0 + tid) * 2 + 1];
      }
    }
    __syncthreads();
    // gh piece = qs0 W_hh^T: A[r][k] = q[r][party_r][k]; C: column = lane & 15, row = 4 (lane >> 4) + register
    {
      f32x4 acc0 = {0}, acc1 = {0};
      wave_mm16(q + r * QS + party[r] * H + h4 * 32, breg, breg1, acc0, acc1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* o = gh + (4 * h4 + i) * TS + g * H + s * 32 + r;
        o[0] = acc0[i] + bias;
        o[16] = acc1[i] + bias1;
      }
    }
    __syncthreads();
    // (the register budget of a 12-wave workgroup does not allow the loop to be unrolled NEL times with the W_hh values live)
#pragma unroll 2
    for (int k = 0; k < NEL; ++k) {
      const int e = tid + k * GNT;
      const int row = e / H, u = e - row * H;
      const int b = b0 + row;
      if (e < RB * H && b < B) {
        const float* gp = a.gi + ((long)t * B + b) * 3 * H + u;
        const float gi0 = gp[0], gi1 = gp[H], gi2 = gp[2 * H];
        const float* gr = gh + row * TS + u;
        const float hprev = q[row * QS + party[row] * H + u];
        const float rg = sigmoidf_(gi0 + gr[0]);
        const float zg = sigmoidf_(gi1 + gr[H]);
        const float ghn = gr[2 * H];
        const float ng = tanhf(gi2 + rg * ghn);
        float hv = (1.f - zg) * ng + zg * hprev;
        const long rowt = (long)t * B + b;
        if (a.rng) hv *= drop_scale(dk, (uint32_t)(rowt * H + u));         // :177 dropout(gru_s(U, qs_0)): the dropped value is the state
        // linked producer: the row goes straight to the consumer's workspace as a write-through (agent-scope) store -- it is visible
        // to the consumer's L1-bypassing loads once this wave's stores have drained; no cache write-back fence per thread and step
        if (a.pub_cnt) __hip_atomic_store(a.hs + rowt * H + u, hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.hs[rowt * H + u] = hv;
        if (a.out) {
          const int tau = a.rev ? a.rev[rowt] : t;
          if (tau >= 0) a.out[((long)tau * B + b) * a.ldo + u] = hv;
        }
        float* sv = a.save + rowt * 5 * H + u;
        sv[0] = hprev; sv[H] = rg; sv[2 * H] = zg; sv[3 * H] = ng; sv[4 * H] = ghn;
        const float m0 = qmv[row * 2], m1 = qmv[row * 2 + 1];
        float* q0 = q + row * QS + u;
        if (a.lblend) {                                                      // model/lsthm_nsps.py:188-191
          const float ql = q0[(1 - party[row]) * H];
          if (a.hli) a.hli[rowt * H + u] = ql;
          q0[0] = ql * (1.f - m0) + hv * m0;
          q0[H] = ql * (1.f - m1) + hv * m1;
        } else {
          q0[0] = q0[0] * (1.f - m0) + hv * m0;                              // model/lsthm_onlysp.py:179-181
          q0[H] = q0[H] * (1.f - m1) + hv * m1;
        }
      }
    }
    // release: every wave drains its write-through stores of step t, then the workgroup meets, then the counter moves.  (Round 2 issued
    // __threadfence() here -- an L2 write-back per thread and step -- and release / sequentially-consistent atomics below, each with
    // a fence of its own: the linked producer ran at 11.1 us per step against 5.7 alone and paced the LSTHM chain it feeds.)
    if (a.pub_cnt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.pub_cnt && tid < a.pub_rep) {
      // "Publish my progress, then read the others'" is the store-buffering pattern: with release/acquire alone two blocks that
      // finish a step together may both read the other's OLD word and both publish the smaller value -- harmless mid-sequence (the
      // next step republishes) but at the last step nothing would ever raise the counter to T * pub_inc.  Hence: (1) the progress
      // word is written with a RETURNING read-modify-write (performed at the memory side before its value comes back) and the
      // loads are sequentially consistent, so at least one of two racing blocks sees the other's new word; (2) independently of
      // that argument the host raises every replica to T * pub_inc with a stream-ordered fill right behind this kernel
      // (gru_launch), when every block has published everything.
      unsigned mine = 0;
      // (relaxed agent-scope atomics are performed at the memory side; the loads below are issued behind the exchange's RETURN -- the data
      //  dependency through `mine` -- which is all the argument above needs)
      if (tid == 0) mine = __hip_atomic_exchange(a.pub_progress + blockIdx.x, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mine = __shfl(mine, 0, 64);            // every publishing lane orders its loads behind the exchange's return
      asm volatile("" : "+v"(mine) :: "memory");      // (compiler: nothing below moves above this point; hardware: the value has arrived)
      unsigned mn = (unsigned)(t + 1) + (mine & 0u);
      for (unsigned bi = 0; bi < gridDim.x; ++bi)
        if (bi != blockIdx.x) mn = min(mn, __hip_atomic_load(a.pub_progress + bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      __hip_atomic_fetch_max(a.pub_cnt + tid * a.pub_stride, a.pub_inc * mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(GNT) void gru_speaker_bwd_kernel(GruArgs2 aa) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const GruArgs& a = aa.d[blockIdx.y];
  float* dq = sm;                  // [RB][QS]   gradient at the party states
  float* dg = dq + RB * QS;        // [RB][TS]   gradient at gh (A operand of dqs0 = dgh W_hh)
  float* part = dg + RB * TS;      // [4][RB][H + 1]: partial products per gate third, and the direct path dh' z of the step
  int* party = (int*)(part + 4 * RB * (GH + 1));
  float* qmv = (float*)(party + 32);
  const int H = GH, B = a.B, PS = GH + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, h4 = lane >> 4;
  const int g = wave >> 2, s = wave & 3;
  const int b0 = blockIdx.x * RB;
  // two 16-column tiles of slice s, gate third g of the reduction: B[k][n] = W_hh[g*H + k][s*32 + 16 i + n], k = 32 h4 + j
  float breg[32], breg1[32];
  {
    const float* w = a.w_hh + (long)(g * H + h4 * 32) * H + s * 32 + r;
#pragma unroll
    for (int j = 0; j < 32; ++j) { breg[j] = w[(long)j * H]; breg1[j] = w[(long)j * H + 16]; }
  }
  for (int e = tid; e < RB * QS; e += GNT) dq[e] = 0.f;
  for (int e = tid; e < RB * TS; e += GNT) dg[e] = 0.f;
  DropKey dk;
  if (a.rng) dk = drop_key(a.rng, a.site, a.p);
  float nm0 = 0.f, nm1 = 0.f;
  if (tid < RB && b0 + tid < B) {
    nm0 = a.qmask[((long)(a.T - 1) * B + b0 + tid) * 2]; nm1 = a.qmask[((long)(a.T - 1) * B + b0 + tid) * 2 + 1];
  }
  __syncthreads();
  for (int t = a.T - 1; t >= 0; --t) {
    if (tid < RB) {
      const float m0 = nm0, m1 = nm1;
      qmv[tid * 2] = m0; qmv[tid * 2 + 1] = m1;
      party[tid] = m1 > m0 ? 1 : 0;
      if (t > 0 && b0 + tid < B) {
        nm0 = a.qmask[((long)(t - 1) * B + b0 + tid) * 2]; nm1 = a.qmask[((long)(t - 1) * B + b0 + tid) * 2 + 1];
      }
    }
    if (a.sub_cnt && tid == 64) {      // (one lane of the second wave) wait for the producer's step t; bounded: a producer that never
      const unsigned target = a.sub_per_step * (unsigned)(a.T - t);          // comes leaves wrong numbers and a status flag, not a hang
      unsigned spins = 0;
      while (__hip_atomic_load(a.sub_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) { if (a.status) atomicOr(a.status, (unsigned)MSER_FAULT_LINK_TIMEOUT); break; }
      }
      // (no acquire fence: the rows are read with device-coherent loads below, which cannot hit a stale line; the poll's value orders
      //  them behind the producer's drained write-through stores)
    }
    __syncthreads();
    // (partially unrolled: with the 64 W_hh values live a 12-wave workgroup has ~100 registers left per lane; fully unrolled, this
    // loop spilled 55 of them to scratch in every step)
#pragma unroll 2
    for (int k = 0; k < NEL; ++k) {
      const int e = tid + k * GNT;
      const int row = e / H, u = e - row * H;
      const int b = b0 + row;
      if (e < RB * H && b < B) {
        const long rowt = (long)t * B + b;
        const float m0 = qmv[row * 2], m1 = qmv[row * 2 + 1];
        float* q0 = dq + row * QS + u;
        float dh = q0[0] * m0 + q0[H] * m1;                                 // every consumer of h_s[t]: the party states, and the cell
        if (a.sub_cnt) {       // rows a concurrently running producer has just written: device-coherent loads (no stale cache line)
          dh += __hip_atomic_load(a.dhs + rowt * H + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int pi = 0; pi < a.sub_nparts; ++pi)
            dh += __hip_atomic_load(a.sub_parts + pi * a.sub_stride + rowt * H + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          dh += a.dhs[rowt * H + u];
          if (a.dhs2) dh += a.dhs2[rowt * H + u];
          if (a.dhs3) dh += a.dhs3[rowt * H + u];
        }
        if (a.lblend) {        // both new party states took (1 - m_p) of ql_0 = q_old[1 - party]; q_old[party] is reached through qs0 only
          float dl = q0[0] * (1.f - m0) + q0[H] * (1.f - m1);
          if (a.dhli) dl += a.dhli[rowt * H + u];
          const int pr = party[row];
          q0[(1 - pr) * H] = dl;
          q0[pr * H] = 0.f;
        } else {
          q0[0] *= (1.f - m0);
          q0[H] *= (1.f - m1);
        }
        if (a.rng) dh *= drop_scale(dk, (uint32_t)(rowt * H + u));
        const float* sv = a.save + rowt * 5 * H + u;
        const float hprev = sv[0], rg = sv[H], zg = sv[2 * H], ng = sv[3 * H], ghn = sv[4 * H];
        const float dan = dh * (1.f - zg) * (1.f - ng * ng);
        const float daz = dh * (hprev - ng) * zg * (1.f - zg);
        const float dar = dan * ghn * rg * (1.f - rg);
        float* o = a.dgi + rowt * 3 * H + u;
        o[0] = dar; o[H] = daz; o[2 * H] = dan;
        float* o2 = a.dgh + rowt * 3 * H + u;
        o2[0] = dar; o2[H] = daz; o2[2 * H] = dan * rg;
        float* l = dg + row * TS + u;
        l[0] = dar; l[H] = daz; l[2 * H] = dan * rg;
        part[(3 * RB + row) * PS + u] = dh * zg;
      }
    }
    __syncthreads();
    {
      f32x4 acc0 = {0}, acc1 = {0};
      wave_mm16(dg + r * TS + g * H + h4 * 32, breg, breg1, acc0, acc1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* o = part + (g * RB + 4 * h4 + i) * PS + s * 32 + r;
        o[0] = acc0[i];
        o[16] = acc1[i];
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int k = 0; k < NEL; ++k) {
      const int e = tid + k * GNT;
      const int row = e / H, u = e - row * H;
      if (e < RB * H && b0 + row < B) {
        const float v = part[row * PS + u] + part[(RB + row) * PS + u] + part[(2 * RB + row) * PS + u] + part[(3 * RB + row) * PS + u];
        dq[row * QS + party[row] * H + u] += v;                              // gradient at qs0 = q[b, party_t[b]] (:176)
      }
    }
    __syncthreads();
  }
}

__global__ void gru_publish_all_kernel(unsigned* cnt, int replicas, int stride, unsigned value) {
  if ((int)threadIdx.x < replicas) __hip_atomic_fetch_max(cnt + threadIdx.x * stride, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

size_t gru_lds_bytes(bool bwd) {
  return ((size_t)RB * QS + RB * TS + (bwd ? 4 * RB * (GH + 1) : 0) + 32 + 64) * sizeof(float);
}

int gru_validate(const mser_gru_speaker_desc& d, bool bwd) {
  MSER_REQUIRE(d.T > 0 && d.B > 0, "mser_gru_speaker: bad sizes T=%d B=%d", d.T, d.B);
  MSER_REQUIRE(d.H == GH, "mser_gru_speaker: H=%d (built for the reference's 128)", d.H);
  MSER_REQUIRE(d.gi && d.w_hh && d.b_hh && d.qmask && d.hs && d.save, "mser_gru_speaker: null pointer");
  MSER_REQUIRE(!d.out || d.ldo >= d.H, "mser_gru_speaker: ldo=%ld < H", (long)d.ldo);
  MSER_REQUIRE(!d.rng || (d.p >= 0.f && d.p < 1.f), "mser_gru_speaker: dropout p=%f", d.p);
  MSER_REQUIRE(!d.pub_counter || (d.pub_replicas > 0 && d.pub_replicas <= 64 && d.pub_replica_stride > 0 && d.pub_per_step > 0 &&
                                  d.pub_progress),
               "mser_gru_speaker: bad publish fields");
  if (bwd) MSER_REQUIRE(d.dhs && d.dgi && d.dgh, "mser_gru_speaker_bwd: null gradient buffer");
  return 0;
}

GruArgs gru_args(const mser_gru_speaker_desc& d) {
  GruArgs a;
  a.T = d.T; a.B = d.B;
  a.gi = d.gi; a.w_hh = d.w_hh; a.b_hh = d.b_hh; a.qmask = d.qmask;
  a.hs = d.hs; a.out = d.out; a.ldo = d.ldo; a.rev = d.rev; a.save = d.save;
  a.dhs = d.dhs; a.dhs2 = d.dhs_add[0]; a.dhs3 = d.dhs_add[1]; a.dgi = d.dgi; a.dgh = d.dgh;
  a.rng = (d.rng && d.p > 0.f) ? d.rng : nullptr; a.site = d.drop_site; a.p = d.p;
  a.pub_cnt = d.pub_counter; a.pub_inc = d.pub_per_step; a.pub_rep = d.pub_replicas; a.pub_stride = d.pub_replica_stride;
  a.pub_progress = d.pub_progress;
  a.sub_cnt = d.sub_counter; a.sub_per_step = d.sub_per_step; a.sub_parts = d.sub_parts; a.sub_nparts = d.sub_nparts;
  a.sub_stride = d.sub_part_stride; a.status = d.status;
  a.lblend = d.listener_blend; a.hli = d.hli; a.dhli = d.dhli;
  return a;
}

int allow_gru(const void* kernel, size_t bytes) {
  static const void* seen[2];
  for (int i = 0; i < 2; ++i)
    if (seen[i] == kernel) return 0;
  MSER_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  for (int i = 0; i < 2; ++i)
    if (!seen[i]) { seen[i] = kernel; break; }
  return 0;
}

}  // namespace
}  // namespace mser

using namespace mser;

extern "C" {

size_t mser_gru_speaker_save_bytes(int32_t T, int32_t B, int32_t H) { return (size_t)T * B * 5 * H * sizeof(float); }

static int gru_launch(const mser_gru_speaker_desc* d, int32_t n, bool bwd, hipStream_t s) {
  const char* what = bwd ? "mser_gru_speaker_bwd" : "mser_gru_speaker_fwd";
  if (!d) { set_error("%s: null descriptor", what); return -1; }
  MSER_REQUIRE(n == 1 || n == 2, "%s: n=%d chains per launch (1 or 2)", what, n);
  GruArgs2 aa;
  for (int i = 0; i < n; ++i) {
    MSER_TRY(gru_validate(d[i], bwd));
    MSER_REQUIRE(d[i].T == d[0].T && d[i].B == d[0].B, "%s: the chains of one launch must share T and B", what);
    aa.d[i] = gru_args(d[i]);
  }
  if (n == 1) aa.d[1] = aa.d[0];
  const size_t lds = gru_lds_bytes(bwd);
  if (bwd) {
    MSER_TRY(allow_gru((const void*)gru_speaker_bwd_kernel, lds));
    hipLaunchKernelGGL(gru_speaker_bwd_kernel, dim3(cdiv(d->B, RB), n), dim3(GNT), lds, s, aa);
  } else {
    MSER_TRY(allow_gru((const void*)gru_speaker_fwd_kernel, lds));
    hipLaunchKernelGGL(gru_speaker_fwd_kernel, dim3(cdiv(d->B, RB), n), dim3(GNT), lds, s, aa);
    // stream-ordered final publication of a linked chain: behind the kernel every block has written every row, so the counter
    // replicas (consecutive words pub_replica_stride apart) are simply filled with T * per_step.  The consumer polls with
    // device-coherent loads; whatever the in-kernel min/max protocol left behind at the last step, this makes it exact.
    for (int i = 0; i < n; ++i)
      if (d[i].pub_counter) {
        hipLaunchKernelGGL(gru_publish_all_kernel, dim3(1), dim3(64), 0, s, d[i].pub_counter, d[i].pub_replicas, d[i].pub_replica_stride,
                           d[i].pub_per_step * (unsigned)d[i].T);
        MSER_TRY(check_launch("gru_publish_all"));
      }
  }
  return check_launch(what);
}

int mser_gru_speaker_fwd(const mser_gru_speaker_desc* d, int32_t n, mser_stream_t stream) { return gru_launch(d, n, false, (hipStream_t)stream); }
int mser_gru_speaker_bwd(const mser_gru_speaker_desc* d, int32_t n, mser_stream_t stream) { return gru_launch(d, n, true, (hipStream_t)stream); }

}  // extern "C"
