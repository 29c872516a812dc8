// MARN_cell on gfx950: speaker-state recurrence + LSTHM recurrence + per-step rank-1 cross-modal attention,
// forward and backward (BPTT).  Replaces model/lsthm_sps.py:132-221, :28-44, :59-72, :238-259 of the reference.
//
// Structure:
//  * Every time step is an all-to-all seam between workgroups: a workgroup owns 8 hidden units of one stream (x 4 gates = 32
//    gate columns) for a block of 32 dialogues and needs every unit of h / z / c_a for its next phase.  Measured on the box
//    (scratch/ubench*.hip): a dependent kernel boundary costs 1.56 us in a hipGraph and 2.5 us eager, a counter barrier among 32
//    co-resident workgroups with a 4 KB write-through payload 1.8 us.  With two seams per step the chains are therefore
//    PERSISTENT launches: the time loop runs inside the kernel, the workgroup's weight slice stays in registers for the whole
//    sequence, and a seam is a monotonic per-direction counter (8 replicas on separate 128-byte lines) plus `sc1` (write-through /
//    L1-bypassing) payload accesses through one buffer descriptor over the workspace.  Every spin is bounded: a workgroup that
//    waits too long sets the abort word (and the caller's sticky fault word) and every workgroup leaves the kernel.
//  * One fused launch per pass: cell_fwd_fused = LSTHM chain roles + speaker chain roles + softmax-statistics roles,
//    cell_bwd_fused = LSTHM BPTT roles + speaker BPTT roles + in-launch weight-gradient roles -- one residency guarantee for
//    every party of every hand-off, and capturable into a hipGraph.
//  * The speaker recurrence (two per-party nn.LSTMCell states indexed by compaction slot) depends on qmask only; its chain runs
//    ahead of the LSTHM chain inside the same launch and  S h_q[t]  joins the step's early product.  W x[t] for all t is hoisted
//    into one GEMM before the launch; [U | S] [h_{t-1} | h_q[t]] is formed in the shadow of the barrier that publishes z_{t-1},
//    only  V z_{t-1}  is on the critical path.
//  * A step's product is a 32(rows) x 32(gate columns) x K tile per workgroup: the 8 waves split K, each runs a short chain of
//    v_mfma_f32_32x32x2_f32 (bit-exact fp32 fmaf chains) with its B fragments in registers, the 8 partial tiles meet in LDS in a
//    fixed order (deterministic), the gate non-linearities are the epilogue of the same workgroup.
//  * The rank-1 attention never materialises the reference's [B,H,H] tensors: logits[i,j] = c_l[i] * s * Wk[j] with
//    s = <Wq, c_a>/sqrt(H); the row maximum is analytic (u * max(Wk) or u * min(Wk)), so one exp2 pass suffices.
//  * The per-step launches (spk_fwd_step, lsthm_fwd_gates, lsthm_fwd_z, lsthm_bwd_row, lsthm_bwd_mat, spk_bwd_step) remain as the
//    fallback for shapes whose workgroups cannot all be co-resident (H not in {128,256}, or more workgroups than CUs) and as the
//    cross-check of the persistent path (tests/test_gpu_model.py::test_persistent_vs_per_step_launches).
#include "common.h"
#include "../../include/mser.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace mser {

int gemm(const mser_gemm_desc& d, hipStream_t s);   // gemm.hip
int gemm_group(const mser_gemm_desc* d, int n, hipStream_t s);

struct DirP {
  // parameters
  const float *W[2], *Wb[2], *U[2], *Ub[2], *V[2], *Vb[2], *S[2], *Sb[2];
  const float *Wih[2], *Whh[2], *bih[2], *bhh[2];
  const float *attWq, *attWk;
  // wide cells (H > 512, weights streamed every step): the forward launches' B operands in MFMA fragment order, packed once per forward
  // call (cell_pack_w_kernel): [unit slab of 8][k / 8][gate * 8 + unit][8], K = [W_ih | W_hh] resp. [U | V]; nullptr otherwise
  float *WpkS[2], *WpkL[2];
  // tables (direction time order)
  int *party, *perm, *rowof, *n0;
  const int* rev;
  float* qm;
  // saved by the forward
  float *qsel, *hq_state, *cq_state, *sgates, *HQ, *tcq;   // tcq[2][T][B][H] = tanh(c_new) (saves the backward a tanhf per element)
  float *pre, *gates, *cstate, *hz;
  float* rstat;        // [T][B][H][4]: softmax statistics of the rank-1 attention row (Z, N2 = sum e*c_a*Wk, N3 = sum e*Wk, s), saved by
                       // the forward row phase so that the BPTT does not recompute its first exp2 pass
  float* out;
  const float* dout;
  // backward scratch
  float* dxc;          // [2 streams][T*B][D]: dgates_m @ W_m of THIS direction at natural time rows (pipelined persistent mode)
  float *dgates, *dc_carry, *dA, *attacc, *dHQ, *dHQp;   // dHQp[2][T][B][H]: dgates_m @ S_m per step (pipelined mode)
  float *dsg, *Xb, *dhprev, *dcprev;   // Xb[2][B][H]: grad wrt q_{t-1}[b, party_t[b]], ping-pong by step parity
  float* mnext;                        // [T][B]: qmask_t[r][party_{t+1}[r]] (0 at the last step)
  // speaker BPTT as a reduce-scatter (spk_bwd_role_ks): per-K-slice partial products, ping-pong by step parity
  float* Xp;                           // [2][H/16 slices][H/16 column groups][B][16]: partials of X_t (dialogue-row indexed)
  float* dhp;                          // [2][2 cells][H/16][H/16][B][16]: partials of dgates W_hh (slot indexed)
  // in-kernel weight-gradient roles (cell_bwd_fused): LSTHM input rows in direction time order and the gradient tensors
  const float* xw[2]; long ldxw[2];
  float *gW[2], *gU[2], *gV[2], *gS[2], *gWih[2], *gWhh[2];
  float* gbias_s[2][4];
  // dropout (CellK::rng != nullptr): sites drop_site (h_q0/h_q1, :183,:188), +1 (h_l/h_a, :211,:213), +2 (rank-1 attention, :69)
  unsigned drop_site;
  float p_state, p_attn;
  float* gbias[2][4];  // bias gradients that receive the column sums of the gate gradients: LSTHM stream m: W,U,V,S .bias; speaker
                       // cell c (wgrad_role<true>): bias_ih, bias_hh, -, -
};

struct CellK {
  int T, B, D, H, ndir, nmb;
  long ldo;
  void* wsbase;        // the cell workspace: one buffer descriptor spans it (sc1 hand-off accesses)
  unsigned wsbytes;
  unsigned* sync;      // persistent-kernel counters: [SYNC_*] words, zeroed by a memset node before every launch
  int wgrad_wgs;       // cell_bwd_fused: workgroups that accumulate the weight gradients while the BPTT chains run (0: none)
  int stats_wgs;       // cell_fwd_fused: workgroups that compute the BPTT's softmax statistics beside the chains (0: the row phase does)
  int place;           // fused launches: roles are claimed by physical XCD (claim_role); the grid then covers every CU
  short place_base[8], place_cap[8];   // XCD x hosts logical workgroups place_base[x] .. place_base[x] + place_cap[x] - 1
  int ext_spk;         // the speaker state h_q[t] comes from the caller (mser_cell_desc::ext_hq): no speaker roles in the launches
  const uint32_t* rng; // dropout generator words {seed, step} (nullptr: every dropout site of the cell is the identity)
  unsigned* fault;     // sticky fault word (mser_cell_desc::fault) or nullptr
  int fwd_sentinel;    // forward chain hand-offs through self-validating payload instead of counter barriers (MSER_OPT_FWD_SENTINEL)
  int bwd_sentinel;    // the same for the LSTHM BPTT chain (MSER_OPT_BWD_SENTINEL)
  int nodx;            // BPTT launch without the two dx = dgates W products (they run as GEMMs after the chain; frees their workgroups)
  int fwd_rowsplit, bwd_rowsplit;   // persistent chains at H = 256: a dialogue row's rank-1 attention is shared by this many workgroups
                       // (forward: 128 query units each; backward: 128 keys of the transposed pass each); 1 or 2
  int poll_delay;      // LSTHM BPTT, both seams self-validating: clocks / 64 between the arrive and the first look at the gate gradients
  int spk_ks;          // speaker BPTT roles in the K-split (reduce-scatter) form (MSER_OPT_SPK_BWD_KSPLIT)
  int ksplit;          // BPTT matvec phase: every product's K = 4H reduction is split over `ksplit` workgroups (1 or 2); the
                       // partial results live in consecutive copies of dA / dHQp / dxc and the consumers add them
  DirP d[2];
};
// Dropout inside the cell.  Element indices: h_q: ((t*2 + cell)*B + slot)*H + unit; h_l/h_a: ((t*2 + stream)*B + b)*H + unit;
// rank-1 attention: ((t*B + b)*H + i)*H + j  (t = the direction's own time index).
// (Cost of these branches on the p = 0 path, A/B of two builds on one box: step 3.354 ms with them, 3.362 ms without.)
__device__ __forceinline__ bool drop_state_on(const CellK& P, const DirP& D) { return P.rng != nullptr && D.p_state > 0.f; }
__device__ __forceinline__ bool drop_attn_on(const CellK& P, const DirP& D) { return P.rng != nullptr && D.p_attn > 0.f; }
// The three keys of the workgroup's direction are derived ONCE per kernel / role (drop_init: two global loads and four mixes) and
// kept in LDS: inside the chains a key fetched from global memory would put an L2 round trip on every step's critical path.
__shared__ DropKey s_drop[3];
__device__ __forceinline__ void drop_init(const CellK& P, const DirP& D) {
  if (P.rng == nullptr) return;          // uniform
  if (threadIdx.x < 3) s_drop[threadIdx.x] = drop_key(P.rng, D.drop_site + threadIdx.x, threadIdx.x == 2 ? D.p_attn : D.p_state);
  __syncthreads();
}
__device__ __forceinline__ float drop_hq(const CellK& P, const DirP& D, int t, int c, int slot, int u) {
  return drop_scale(s_drop[0], (uint32_t)((((long)t * 2 + c) * P.B + slot) * P.H + u));
}
__device__ __forceinline__ float drop_h(const CellK& P, const DirP& D, int t, int m, int b, int u) {
  return drop_scale(s_drop[1], (uint32_t)((((long)t * 2 + m) * P.B + b) * P.H + u));
}
__device__ __forceinline__ uint32_t drop_attn_row(const CellK& P, int t, int b) { return (uint32_t)(((long)t * P.B + b) * P.H * P.H); }

// Every counter sits on a 128-byte line of its own (SYNC_LINE words apart): arrivals (atomics) and polls of one chain never queue
// behind another chain's at the memory side.  Measured: with all eight counters on one line a second direction cost +56 % per
// forward step and +24 % per backward step although the two directions share no data.
// A counter is kept in SYNC_REP replicas (one line each): an arrival adds to every replica with ONE wave instruction (SYNC_REP
// active lanes), a waiting workgroup polls only replica (workgroup id % SYNC_REP).
#ifndef MSER_SYNC_REP
#define MSER_SYNC_REP 8
#endif
enum { SYNC_LINE = 32, SYNC_REP = MSER_SYNC_REP, SYNC_DIR = SYNC_REP * SYNC_LINE, SYNC_SPK_FWD = 0, SYNC_LSTHM_FWD = 2 * SYNC_DIR,
       SYNC_LSTHM_BWD = 4 * SYNC_DIR, SYNC_SPK_BWD = 6 * SYNC_DIR,
       // XCD placement of the fused launches (claim_role): 8 per-XCD ticket counters, "registered", "spare ticket", one line each
       SYNC_PLACE_LINES = 10, SYNC_PLACE_BWD = 8 * SYNC_DIR,          // zeroed together with the backward counters
       SYNC_ABORT = SYNC_PLACE_BWD + SYNC_PLACE_LINES * SYNC_LINE, SYNC_STAMPS = SYNC_ABORT + SYNC_LINE,
       SYNC_PLACE_FWD = SYNC_STAMPS + 2 * SYNC_LINE, SYNC_WORDS = SYNC_PLACE_FWD + SYNC_PLACE_LINES * SYNC_LINE };


// ---- cross-workgroup accessors ---------------------------------------------------------------------------------------
// PS = true inside the persistent kernels: every array that another workgroup wrote (or will read) during the same launch is
// stored write-through and loaded L1-bypassing (`sc1`: relaxed agent-scope atomics on address-space-1 pointers), which is the
// hand-off form of cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms" row 1 (sc1 payload, one counter
// add per workgroup behind s_waitcnt vmcnt(0) + barrier, sc1-load poll, workgroup barrier, sc1 loads).  PS = false in the
// per-step launches, where the kernel boundary orders everything and plain accesses are correct.
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// The hand-off payload goes through ONE buffer descriptor that spans the whole cell workspace: raw buffer loads/stores with
// aux = 16 are `buffer_load/store_dword[x4] ... sc1` (L1-bypassing / write-through), they are ordinary (non-atomic) memory
// operations for the compiler, so it can issue a phase's loads back to back and wait once, and they come 16 bytes wide.
struct WS {
  __amdgpu_buffer_rsrc_t r;
  const char* base;
};
__device__ __forceinline__ WS make_ws(void* base, unsigned bytes) {
  WS w;
  w.base = (const char*)base;
  w.r = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  return w;
}
constexpr int AUX_SC1 = 16;

// PS = 0: plain accesses; 1: write-through stores and L1 / L2-bypassing loads (the chains' hand-off form); 2: write-through stores but
// CACHED loads -- for hand-offs through arrays that are indexed by the time step (a consumer never holds a stale line of an address
// nobody has read before) and whose operands are read by many workgroups (the wide cells' persistent launches, see cell_wide_*)
template <int PS>
__device__ __forceinline__ float ldx(const WS& w, const float* p) {
  if constexpr (PS == 1) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(w.r, (int)((const char*)p - w.base), 0, AUX_SC1));
  else return *p;
}
template <int PS>
__device__ __forceinline__ void stx(const WS& w, float* p, float v) {
  if constexpr (PS != 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), w.r, (int)((const char*)p - w.base), 0, AUX_SC1);
  else *p = v;
}
__device__ __forceinline__ void load8(const float* p, float* a) {
  const float4 v0 = *reinterpret_cast<const float4*>(p);
  const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
  a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
  a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
}
template <int PS>
__device__ __forceinline__ void load8x(const WS& w, const float* p, float* a) {
  if constexpr (PS == 1) {
    const int off = (int)((const char*)p - w.base);
    const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(w.r, off, 0, AUX_SC1);
    const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(w.r, off + 16, 0, AUX_SC1);
    a[0] = __uint_as_float(v0.x); a[1] = __uint_as_float(v0.y); a[2] = __uint_as_float(v0.z); a[3] = __uint_as_float(v0.w);
    a[4] = __uint_as_float(v1.x); a[5] = __uint_as_float(v1.y); a[6] = __uint_as_float(v1.z); a[7] = __uint_as_float(v1.w);
  } else {
    load8(p, a);
  }
}
template <int PS>
__device__ __forceinline__ void store8x(const WS& w, float* p, const float* a) {
  if constexpr (PS != 0) {
    const int off = (int)((const char*)p - w.base);
    u32x4 v0, v1;
    v0.x = __float_as_uint(a[0]); v0.y = __float_as_uint(a[1]); v0.z = __float_as_uint(a[2]); v0.w = __float_as_uint(a[3]);
    v1.x = __float_as_uint(a[4]); v1.y = __float_as_uint(a[5]); v1.z = __float_as_uint(a[6]); v1.w = __float_as_uint(a[7]);
    __builtin_amdgcn_raw_buffer_store_b128(v0, w.r, off, 0, AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(v1, w.r, off + 16, 0, AUX_SC1);
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(a[4], a[5], a[6], a[7]);
  }
}
template <int PS>
__device__ __forceinline__ float4 ld4x(const WS& w, const float* p) {
  if constexpr (PS == 1) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.r, (int)((const char*)p - w.base), 0, AUX_SC1);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
template <int PS>
__device__ __forceinline__ void st4x(const WS& w, float* p, float4 v) {
  if constexpr (PS != 0) {
    u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, w.r, (int)((const char*)p - w.base), 0, AUX_SC1);
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}
__device__ __forceinline__ void zero8(float* a) {
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
}

// Barrier among the `nwg` workgroups of one direction inside a persistent launch.  One monotonic counter per direction
// (zeroed by a memset node before the launch); `target` = nwg * (index of this barrier + 1).  Every spin is bounded: on
// time-out (or when another workgroup has already given up) the abort word is set and every workgroup leaves the kernel.
constexpr unsigned SPIN_LIMIT = 1u << 22;
// Split-phase form: barrier_arrive() then (independent work, e.g. prefetching the next step's saved operands) then
// barrier_wait().  Loads issued between the two overlap the hand-off latency.
__device__ __forceinline__ void barrier_arrive(unsigned* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x < SYNC_REP) __hip_atomic_fetch_add((gu32*)cnt + threadIdx.x * SYNC_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The fused launches assign roles by physical XCD (claim_role), so a workgroup's logical id is not its blockIdx: every persistent
// kernel publishes it here first (the separate-launch kernels store blockIdx.x + y + z).
__shared__ unsigned s_logical_wg;
// Sticky fault word of the caller (mser_cell_desc::fault, may be null): unlike the abort word in the workspace, which the next
// FWD_PREP clears, it survives until the host reads it, so a timed-out chain can never pass unnoticed (the optimiser skips its
// update while the word is set, the trainer raises at its next synchronisation).
__shared__ unsigned* s_fault;
__device__ __forceinline__ void set_logical_wg(unsigned id, unsigned* fault = nullptr) {
  if (threadIdx.x == 0) { s_logical_wg = id; s_fault = fault; }
  __syncthreads();
}
__device__ __forceinline__ void raise_abort(unsigned* abortw) {
  __hip_atomic_store((gu32*)abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (s_fault) __hip_atomic_fetch_or((gu32*)s_fault, (unsigned)MSER_FAULT_CHAIN_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned sync_replica() { return (s_logical_wg % SYNC_REP) * SYNC_LINE; }
// One early look at this workgroup's replica (all lanes load the same word: uniform code, the value comes back while the
// caller keeps computing); pass it to barrier_wait, which polls only if the count was not complete yet.
__device__ __forceinline__ unsigned barrier_peek(const unsigned* cnt) {
  return __hip_atomic_load((const gu32*)(cnt + sync_replica()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool barrier_wait(const unsigned* cnt, unsigned* abortw, unsigned target, int* lds_ok, unsigned seen = 0,
                                             const unsigned* cnt2 = nullptr, unsigned target2 = 0) {
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    const unsigned rep = sync_replica();
    cnt += rep;
    if (cnt2) cnt2 += rep;
    bool a = seen >= target, b = !cnt2;
    while (true) {
      if (!a) a = __hip_atomic_load((const gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
      if (!b) b = __hip_atomic_load((const gu32*)cnt2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target2;
      if (a && b) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u) {
        if (spins > SPIN_LIMIT || __hip_atomic_load((const gu32*)abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = 0;
          break;
        }
      }
    }
    if (!ok) raise_abort(abortw);
    *lds_ok = ok;
  }
  __syncthreads();
  return *lds_ok != 0;
}

// Wait of a workgroup that is NOT on the critical chain (weight-gradient roles following a chain's step counter): long sleeps
// between polls, so that its polls do not queue in front of the chain's own arrivals and polls on the counter line.
__device__ __forceinline__ bool lazy_wait(const unsigned* cnt, unsigned* abortw, unsigned target, int* lds_ok) {
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    cnt += sync_replica();
    while (__hip_atomic_load((const gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(48);
      if ((++spins & 15u) == 0u) {
        if (spins > (SPIN_LIMIT >> 3) || __hip_atomic_load((const gu32*)abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = 0;
          break;
        }
      }
    }
    if (!ok) raise_abort(abortw);
    *lds_ok = ok;
  }
  __syncthreads();
  return *lds_ok != 0;
}

// `cnt2 / target2` (optional): additionally wait until ANOTHER kernel's counter has reached target2 -- the cross-kernel link
// of the pipelined chains (the LSTHM chain consumes h_q[t] from the concurrently running speaker chain; the speaker BPTT
// consumes dHQ[t] from the LSTHM BPTT).  `wait` = false: arrive only (last step of a producer).
__device__ __forceinline__ bool dir_barrier(unsigned* cnt, unsigned* abortw, unsigned target, int* lds_ok,
                                            const unsigned* cnt2 = nullptr, unsigned target2 = 0, bool wait = true) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
  __syncthreads();
  if (cnt && threadIdx.x < SYNC_REP) __hip_atomic_fetch_add((gu32*)cnt + threadIdx.x * SYNC_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    const unsigned rep = sync_replica();
    if (cnt) cnt += rep;
    if (cnt2) cnt2 += rep;
    while (wait) {
      const bool a = !cnt || __hip_atomic_load((const gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
      const bool b = !cnt2 || __hip_atomic_load((const gu32*)cnt2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target2;
      if (a && b) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u) {
        if (spins > SPIN_LIMIT || __hip_atomic_load((const gu32*)abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = 0;
          break;
        }
      }
    }
    if (!ok) raise_abort(abortw);
    *lds_ok = ok;
  }
  __syncthreads();
  return *lds_ok != 0;
}

// ---- self-validating hand-off (forward chain, MSER_OPT_FWD_SENTINEL) ------------------------------------------------------------
// The counter barrier above costs a store drain (s_waitcnt vmcnt(0) on write-through stores: a memory-side round trip), a workgroup
// barrier, the counter adds, a poll, a second workgroup barrier and only THEN the payload loads -- four dependent trips through the
// memory side per seam.  The chain's state arrays are indexed by the time step and written exactly once per launch, so the payload can
// validate itself: MSER_PHASE_FWD_PREP fills them with a bit pattern no finite value takes (a quiet NaN with a payload), producers
// just store (sc1), consumers load (sc1) and re-load until none of their words is the pattern.  Every 4-byte word is written by one
// store, so a word is either the pattern or final; no ordering between different words is assumed.  One store->load trip per
// seam, no drain, no atomics, and every wave proceeds as soon as ITS operands are valid (scratch/ubench_ll.hip: 1.67 us per
// all-to-all exchange among 32 workgroups against 1.97 for the counter form, before counting the barriers' own synchronisation).
// Every poll is bounded: a wave that gives up raises the abort word (and the caller's sticky fault word), sets s_poll_abort and goes
// on with garbage; all waves of the workgroup leave together at the next point where they are synchronised anyway.
constexpr unsigned SENT_BITS = 0x7fc0dead;
__device__ __forceinline__ void sleep_n(int n) {      // n x 64 clocks (s_sleep takes an immediate)
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ bool is_sent(float v) { return __float_as_uint(v) == SENT_BITS; }
__device__ __forceinline__ bool any_sent8(const float* a) {
  bool b = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) b |= is_sent(a[j]);
  return b;
}
__shared__ int s_poll_abort;
constexpr unsigned POLL_LIMIT = 1u << 20;          // re-loads (each a memory-side round trip, ~1 us): ~1 s
// call after a failed validation; returns true when the caller must stop polling
__device__ __forceinline__ bool poll_giveup(unsigned& spins, unsigned* abortw) {
  if ((++spins & 31u) == 0u) {
    if (spins > POLL_LIMIT || __hip_atomic_load((const gu32*)abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      raise_abort(abortw);
      s_poll_abort = 1;
      return true;
    }
  }
  return false;
}

// ---- XCD-aware role assignment of the fused persistent launches ---------------------------------------------------------------
// OPTIONAL (MSER_OPT_XCD_PLACEMENT, off by default).  A counter barrier + 2 KB exchange among 32 workgroups costs 1.25 us when
// they all sit on ONE XCD, 1.37 on two, 1.44 on four and 1.49 spread over all eight (scratch/ubench_xcd.hip).  The hand-off
// PROTOCOL stays placement-independent (sc1 payload, memory-side counters); only the choice of WHICH resident workgroup plays
// which role uses the physical XCD: the launch covers every CU, a workgroup takes ticket tk on its XCD x (HW_REG_XCC_ID) and
// becomes logical workgroup base[x] + tk if tk < cap[x]; the others are spares.  Spares wait until every workgroup of the launch
// has registered, then fill whatever role an XCD with too few arrivals left unclaimed (so a different dispatch pattern costs
// speed, never correctness) and otherwise exit at once, freeing their CU.
// Measured end to end: whole-XCD groups shorten the BPTT launch by 65 us but the dispatcher never migrates a workgroup to another
// XCD, so every concurrent kernel (attention branches, weight-gradient groups) stalls behind the fully occupied XCDs for the
// whole chain (+330 us per step); two XCDs per group keep half of every XCD free but then the barrier gain (0.1 us) is eaten by
// the 256-workgroup launch and registration (+100 us per step).  Hence off.
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
// returns the logical workgroup id or -1 (spare: leave).  pw: SYNC_PLACE_LINES zeroed lines; base / cap: the ids XCD x hosts.
__device__ __forceinline__ int claim_role(unsigned* pw, const short* base, const short* cap, int* lds_tmp) {
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id() & 7u;
    const unsigned tk = __hip_atomic_fetch_add((gu32*)(pw + x * SYNC_LINE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int lid = tk < (unsigned)cap[x] ? base[x] + (int)tk : -1;
    __hip_atomic_fetch_add((gu32*)(pw + 8 * SYNC_LINE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lid < 0) {
      unsigned spins = 0;
      bool all = false;
      while (!(all = __hip_atomic_load((const gu32*)(pw + 8 * SYNC_LINE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x)) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (SPIN_LIMIT >> 2)) break;
      }
      if (all) {
        const unsigned sp = __hip_atomic_fetch_add((gu32*)(pw + 9 * SYNC_LINE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned cnt = 0;
        for (int x2 = 0; x2 < 8 && lid < 0; ++x2) {
          unsigned got = __hip_atomic_load((const gu32*)(pw + x2 * SYNC_LINE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (unsigned k = got; k < (unsigned)cap[x2]; ++k) {
            if (cnt == sp) { lid = base[x2] + (int)k; break; }
            ++cnt;
          }
        }
      }
    }
    *lds_tmp = lid;
  }
  __syncthreads();
  return *lds_tmp;
}

// ---- workgroup geometry of the recurrent kernels ----------------------------------------------------------------------------
// 512 threads = 8 waves = 2 per SIMD.  The fp32 MFMA is paced per SIMD (64 cycles per 32x32x2), so the matvec phases need only
// one or two waves per SIMD; the row phases (rank-1 attention: ~16k exp2 per dialogue row) want more VALU issue slots.
constexpr int NT = 512, NW = 8;

// ---- diagnostic phase stamps (compiled only with -DMSER_STAMPS; never in the product build) -----------------------------------
#ifdef MSER_STAMPS
__shared__ unsigned long long st_acc[17];      // 16 accumulators + last stamp; 18*8 = 144 B keeps the dynamic LDS base 16-B aligned
__shared__ unsigned long long st_pad;
#define STAMP_INIT() do { if (threadIdx.x == 0) { for (int _i = 0; _i < 16; ++_i) st_acc[_i] = 0; st_acc[16] = __builtin_amdgcn_s_memrealtime(); st_pad = 0; } } while (0)
#define STAMP_ACC(k) do { if (threadIdx.x == 0) { unsigned long long _n = __builtin_amdgcn_s_memrealtime(); st_acc[k] += _n - st_acc[16]; st_acc[16] = _n; } } while (0)
#define STAMP_DUMP(P, base, sel) do { if (threadIdx.x == 0 && (sel)) for (int _i = 0; _i < 8; ++_i) (P).sync[SYNC_STAMPS - 16 + (base) + _i] = (unsigned)(st_acc[_i] / (unsigned)(P).T); } while (0)
#else
#define STAMP_INIT()
#define STAMP_ACC(k)
#define STAMP_DUMP(P, base, sel)
#endif
constexpr int RED_FLOATS = NW * 1024;       // K-split partial tiles, reused as scratch by the row phases
constexpr float LOG2E = 1.4426950408889634f;

// 32 x 32 x K product by one workgroup: the 8 waves split K, each runs a chain of v_mfma_f32_32x32x2_f32, partials are
// reduced through LDS in a fixed order.  Result (row-major [32][32]) left in tile[], all threads synced.
// aload(r, k, a[8]) returns A[row r][k..k+7]; bload(n, k, b[8]) returns B[k..k+7][col n].
// NP > 0: the B fragments of this wave's NP k-passes were loaded once into bpre[][] (persistent kernels keep the weights in
// registers across the whole time loop) and all A fragments are fetched before the first MFMA.
// UN (NP == 0 only): k-passes whose operand loads are issued together before their MFMAs run.  The streaming form used to wait for
// every pass's loads in turn and relied on 6-8 waves per SIMD to hide that; the wide persistent launches run one workgroup per CU
// (2 waves per SIMD), so they keep UN = 4 passes in flight per wave instead.
template <int NP, int UN = 1, class ALoad, class BLoad>
__device__ __forceinline__ void wg_mm32(int K, ALoad aload, BLoad bload, const float (*bpre)[8], float* red, float* tile,
                                        bool red_aliases_a = false, unsigned* poll_abortw = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int KC = ((K + NW * 16 - 1) / (NW * 16)) * 16;   // per-wave K chunk (multiple of 16)
  f32x16 acc = {0};
  const int kbeg = wave * KC;
  const int kend = min(K, kbeg + KC);
  if constexpr (NP > 0) {
    float a[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (kbeg + p * 16 < kend) aload(r, kbeg + p * 16 + half * 8, a[p]);
      else zero8(a[p]);
    }
    if (poll_abortw) {            // self-validating hand-off: the A rows come from other workgroups of this launch (see is_sent)
      unsigned spins = 0;
      while (true) {
        // only the fragments that still hold a sentinel word are requested again (a re-load of the whole slab per failed look was
        // most of the launch's memory-side read traffic)
        unsigned long long badm[NP];
        bool any = false;
#pragma unroll
        for (int p = 0; p < NP; ++p) { badm[p] = __builtin_amdgcn_ballot_w64(any_sent8(a[p])); any |= badm[p] != 0ull; }
        if (!any) break;
        if (poll_giveup(spins, poll_abortw)) break;
#pragma unroll
        for (int p = 0; p < NP; ++p)
          if (badm[p] != 0ull && kbeg + p * 16 < kend) aload(r, kbeg + p * 16 + half * 8, a[p]);
      }
      STAMP_ACC(4);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][j], bpre[p][j], acc, 0, 0, 0);
    }
    if (poll_abortw) STAMP_ACC(5);
  } else {
    for (int kb = kbeg; kb < kend; kb += 16 * UN) {
      float a[UN][8], b[UN][8];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (UN == 1 || kb + 16 * u < kend) {
          aload(r, kb + 16 * u + half * 8, a[u]);
          bload(r, kb + 16 * u + half * 8, b[u]);
        } else {
          zero8(a[u]); zero8(b[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], b[u][j], acc, 0, 0, 0);
    }
  }
  if (red_aliases_a) __syncthreads();        // `red` shares storage with the LDS-staged A operand: every wave has read its part
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
    red[wave * 1024 + row * 32 + r] = acc[i];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 1024 / NT; ++e) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * 1024 + tid + e * NT];
    tile[tid + e * NT] = s;
  }
  __syncthreads();
}
template <int NP, class BLoad>
__device__ __forceinline__ void preload_b(int K, BLoad bload, float (*bpre)[8]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int KC = ((K + NW * 16 - 1) / (NW * 16)) * 16;
  const int kend = min(K, (wave + 1) * KC);
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int kb = wave * KC + pass * 16;
    if (kb < kend) bload(r, kb + half * 8, bpre[pass]);
    else zero8(bpre[pass]);
  }
}

// block-wide helpers (NT threads)
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += sh[w];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = -INFINITY;
#pragma unroll
  for (int w = 0; w < NW; ++w) s = fmaxf(s, sh[w]);
  return s;
}

// Rank-1 attention constants of one direction in LDS: att[0..H) = Wk, att[H..2H) = Wq, att[2H] = max(Wk), att[2H+1] = min(Wk).
// (softmax_j(u * Wk[j]) has its maximum at u*max(Wk) for u >= 0 and u*min(Wk) otherwise: no running max needed.)
__device__ __forceinline__ void att_prepare(const DirP& D, int H, float* att, float* sh) {
  float wmx = -INFINITY, wmn = INFINITY;
  for (int k = threadIdx.x; k < H; k += NT) {
    const float w = D.attWk[k];
    att[k] = w;
    att[H + k] = D.attWq[k];
    wmx = fmaxf(wmx, w);
    wmn = fminf(wmn, w);
  }
  wmx = block_max(wmx, sh);
  wmn = -block_max(-wmn, sh);
  if (threadIdx.x == 0) { att[2 * H] = wmx; att[2 * H + 1] = wmn; }
  __syncthreads();
}
__device__ __forceinline__ int att_floats(int H) { return 2 * H + 16; }

// A workgroup's coordinates inside its chain's logical grid (the chains were separate launches; the fused kernels below give every
// chain a contiguous range of blockIdx.x and decode it into these coordinates).
struct Role { int x, y, z, gx, gy; };

// ================================================================================================ speaker forward
// Role: (cell c, units u0..u0+7, slot block mb) of direction D.  One nn.LSTMCell (gate order i,f,g,o) step for 8 hidden
// units of one party cell over a block of 32 compaction slots.  q_sel is rebuilt on the fly from the previous step's rows:
//   q_{t-1}[b,c] = (1-m_{t-1}[b,c]) * h0_{t-1}[b] + m_{t-1}[b,c] * hq_{t-1}[b]           (model/lsthm_sps.py:204-207)
struct SpkFwdB {
  const DirP& D; int c, u0, H;
  __device__ __forceinline__ void operator()(int n, int k, float* b) const {
    if (D.WpkS[c]) {        // fragment order: a wave's load is 1 KB contiguous (row-major it touches 32 rows x 32 bytes)
      load8(D.WpkS[c] + ((((long)(u0 >> 3) * (2 * H / 8)) + (k >> 3)) * 32 + n) * 8, b);
      return;
    }
    const long wrow = (long)(n >> 3) * H + u0 + (n & 7);
    if (k < H) load8(D.Wih[c] + wrow * H + k, b);
    else load8(D.Whh[c] + wrow * H + (k - H), b);
  }
};

// Gate non-linearities of the persistent chains on the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each): the libm
// expf / tanhf the per-step launches use cost ~60-80 instructions per value, five values per (row, unit) on the critical path of
// every step.  Absolute error ~1e-7 per value; through 128 steps the log-probs move by < 2e-6 (the parity tests run this path).
// (-DMSER_FAST_NL=0 keeps libm in the chains; measured on one box, alternating: 3.206 / 3.210 ms per step with libm, 3.182 / 3.180 with
// these forms -- while the speaker chain paced the forward at 6.5 us per step the same switch measured nothing.)
#ifndef MSER_FAST_NL
#define MSER_FAST_NL 1
#endif
constexpr bool FASTNL = MSER_FAST_NL != 0;
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * sigmoid_fast(2.0f * x) - 1.0f; }


// What a step's A rows need from the index tables (they do not depend on the chain): fetched one step ahead by the persistent kernel, so
// that the gather of q_sel costs the step ONE memory round trip (the rows themselves) instead of three dependent ones (perm -> qm -> rows).
struct SpkPre { int N0, N0p, b; float m; };
__device__ __forceinline__ SpkPre spk_fwd_prefetch(const CellK& P, const DirP& D, int t, int c, int mb) {
  SpkPre r;
  const int B = P.B;
  r.N0 = D.n0[t];
  r.N0p = t > 0 ? D.n0[t - 1] : 0;
  const int Nc = c ? B - r.N0 : r.N0, off = c ? r.N0 : 0;
  const int slot = mb * 32 + (threadIdx.x & 31);
  r.b = -1; r.m = 0.f;
  if (slot < Nc && t > 0) {
    r.b = D.perm[(long)t * B + off + slot];
    r.m = D.qm[((long)(t - 1) * B + r.b) * 2 + c];
  }
  return r;
}
// carry (persistent kernel): this thread's cell state c_q[slot][u] and its four gate biases stay in registers across the steps
struct SpkCarry { float cq, bias4[4]; };

template <int PS, int NP>
__device__ __forceinline__ void spk_fwd_body(const CellK& P, const DirP& D, const WS& ws, int t, int c, int u0, int mb, bool writer,
                                             const float (*bpre)[8], float* red, float* tile, const SpkPre* pre = nullptr,
                                             SpkCarry* carry = nullptr) {
  const int H = P.H, B = P.B, T = P.T;
  // (per-step launches: the lane's row of the index tables once per call -- inside aload they were two dependent loads in front of
  //  EVERY k-chunk's gather; at hid = 1024 the launch took 75 us for 134 MB of weights)
  SpkPre pre_local;
  if (!pre) { pre_local = spk_fwd_prefetch(P, D, t, c, mb); pre = &pre_local; }
  const int N0 = pre->N0;
  const int Nc = c ? B - N0 : N0, off = c ? N0 : 0;
  const long SB = (long)B * H;
  float* hq_new = D.hq_state + ((long)c * (T + 1) + t + 1) * SB;
  float* cq_new = D.cq_state + ((long)c * (T + 1) + t + 1) * SB;
  const float* hq_old = hq_new - SB;
  const float* cq_old = cq_new - SB;
  float* sg = D.sgates + ((long)c * T + t) * B * 4 * H;
  float* qs = D.qsel + ((long)c * T + t) * SB;
  const int tid = threadIdx.x;

  if (Nc == 0) {   // `if N0:` / `if N1:` false -> the cell is not stepped (:180,:185); carry its state
    if (tid < 256) {
      const int slot = mb * 32 + (tid >> 3), u = u0 + (tid & 7);
      if (slot < B) {
        stx<PS>(ws, hq_new + (long)slot * H + u, ldx<PS>(ws, hq_old + (long)slot * H + u));
        stx<PS>(ws, cq_new + (long)slot * H + u, ldx<PS>(ws, cq_old + (long)slot * H + u));
#pragma unroll
        for (int g = 0; g < 4; ++g) sg[(long)slot * 4 * H + g * H + u] = 0.f;
      }
    }
    if (writer)
      for (int e = tid; e < 32 * H; e += NT) {
        const int slot = mb * 32 + e / H;
        if (slot < B) stx<PS>(ws, qs + (long)slot * H + e % H, 0.f);
      }
    return;
  }

  // epilogue operands that do not depend on the matvec: fetch them first
  float cq_prev = 0.f, bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (carry) {
    cq_prev = carry->cq;
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = carry->bias4[g];
  } else if (tid < 256) {
    const int slot = mb * 32 + (tid >> 3), u = u0 + (tid & 7);
    if (slot < B) cq_prev = ldx<PS>(ws, cq_old + (long)slot * H + u);
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = D.bih[c][g * H + u] + D.bhh[c][g * H + u];
  }
  const int N0p = pre ? pre->N0p : (t > 0 ? D.n0[t - 1] : 0);
  auto aload = [&](int r, int k, float* a) {
    const int slot = mb * 32 + r;
    if (slot >= B) { zero8(a); return; }
    if (k < H) {
      if (slot < Nc && t > 0) {
        const int b = pre ? pre->b : D.perm[(long)t * B + off + slot];             // (pre: r == this lane's slot)
        const float m = pre ? pre->m : D.qm[((long)(t - 1) * B + b) * 2 + c];
        const float* h0 = (b < N0p) ? D.qsel + ((long)(0 * T + t - 1) * B + b) * H
                                    : D.qsel + ((long)(1 * T + t - 1) * B + (b - N0p)) * H;
        const float* hq = D.HQ + ((long)(t - 1) * B + b) * H;
        float x0[8], x1[8];
        load8x<PS>(ws, h0 + k, x0);
        load8x<PS>(ws, hq + k, x1);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = x0[j] * (1.f - m) + x1[j] * m;
      } else {
        zero8(a);
      }
      // q_sel is saved for the BPTT / weight gradients: every workgroup of the cell gathers the same rows, each stores the one 8-unit
      // fragment column that matches its own unit slice (one workgroup storing all of them paced the whole chain at H = 256)
      if ((k >> 3) == (u0 >> 3)) store8x<PS>(ws, qs + (long)slot * H + k, a);
    } else {
      load8x<PS>(ws, hq_old + (long)slot * H + (k - H), a);
    }
  };
  STAMP_ACC(0);
  wg_mm32<NP, (PS == 2 ? 4 : 1)>(2 * H, aload, SpkFwdB{D, c, u0, H}, bpre, red, tile);
  STAMP_ACC(1);
  if (tid < 256) {
    const int rr = tid >> 3, uu = tid & 7;
    const int slot = mb * 32 + rr, u = u0 + uu;
    if (slot < B) {
      const bool fnl = PS == 1 && FASTNL;                          // (persistent chain: hardware exp / rcp forms, see sigmoid_fast)
      const float gi = fnl ? sigmoid_fast(tile[rr * 32 + 0 + uu] + bias4[0]) : sigmoidf_(tile[rr * 32 + 0 + uu] + bias4[0]);
      const float gf = fnl ? sigmoid_fast(tile[rr * 32 + 8 + uu] + bias4[1]) : sigmoidf_(tile[rr * 32 + 8 + uu] + bias4[1]);
      const float gg = fnl ? tanh_fast(tile[rr * 32 + 16 + uu] + bias4[2]) : tanhf(tile[rr * 32 + 16 + uu] + bias4[2]);
      const float go = fnl ? sigmoid_fast(tile[rr * 32 + 24 + uu] + bias4[3]) : sigmoidf_(tile[rr * 32 + 24 + uu] + bias4[3]);
      const float cn = gf * cq_prev + gi * gg;
      const float tcn = fnl ? tanh_fast(cn) : tanhf(cn);
      float hn = go * tcn;
      if (drop_state_on(P, D)) hn *= drop_hq(P, D, t, c, slot, u);          // :183 / :188: the dropped h_q IS the carried state
      D.tcq[((long)c * T + t) * SB + (long)slot * H + u] = tcn;
      if (carry) carry->cq = cn;
      stx<PS>(ws, cq_new + (long)slot * H + u, cn);
      stx<PS>(ws, hq_new + (long)slot * H + u, hn);
      float* g = sg + (long)slot * 4 * H + u;
      g[0] = gi; g[H] = gf; g[2 * H] = gg; g[3 * H] = go;
      if (slot < Nc) {                                        // h_q = cat[h_q0[:N0], h_q1[:N1]] (:192)
        const int r = off + slot;
        stx<PS>(ws, D.HQ + ((long)t * B + r) * H + u, hn);
        const int tau = D.rev ? D.rev[(long)t * B + r] : t;   // all_hs = cat[h_l, h_a, z_l, h_q] (:218): the h_q quarter
        if (tau >= 0) D.out[((long)tau * B + r) * P.ldo + 3 * H + u] = hn;
      }
    }
  }
  STAMP_ACC(2);
}

// per-step launch: grid (H/8, 2 cells, ndir*nmb), block NT
__global__ __launch_bounds__(NT) void spk_fwd_step(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int dir = blockIdx.z / P.nmb, mb = blockIdx.z % P.nmb;
  drop_init(P, P.d[dir]);
  spk_fwd_body<false, 0>(P, P.d[dir], ws, t, blockIdx.y, blockIdx.x * 8, mb, blockIdx.x == 0, nullptr, smem, smem + RED_FLOATS);
}

// persistent launch: same grid, the whole time loop inside; weights of this workgroup's slice stay in registers.
template <int NP>
__device__ __forceinline__ void spk_fwd_role(const CellK& P, const Role R, float* smem, const WS& ws) {
  float* red = smem;
  float* tile = smem + RED_FLOATS;
  int* lds_ok = (int*)(tile + 1024);
  const int dir = R.z / P.nmb, mb = R.z % P.nmb;
  const DirP& D = P.d[dir];
  const int c = R.y, u0 = R.x * 8;
  const unsigned nwg = R.gx * R.gy * P.nmb;
  drop_init(P, D);
  float bpre[NP][8];
  preload_b<NP>(2 * P.H, SpkFwdB{D, c, u0, P.H}, bpre);
  STAMP_INIT();
  SpkCarry carry;
  carry.cq = 0.f;                                  // c_q[.][0] = 0
  {
    const int u = u0 + (threadIdx.x & 7);
#pragma unroll
    for (int g = 0; g < 4; ++g) carry.bias4[g] = threadIdx.x < 256 ? D.bih[c][g * P.H + u] + D.bhh[c][g * P.H + u] : 0.f;
  }
  SpkPre pre = spk_fwd_prefetch(P, D, 0, c, mb);
  for (int t = 0; t < P.T; ++t) {
    const SpkPre nxt = spk_fwd_prefetch(P, D, t + 1 < P.T ? t + 1 : t, c, mb);      // tables of the next step: in flight during this one
    spk_fwd_body<true, NP>(P, D, ws, t, c, u0, mb, R.x == 0, bpre, red, tile, &pre, &carry);
    pre = nxt;
    // the counter also tells the concurrently running LSTHM kernel that h_q[t] is published: arrive after the last step too
    if (!dir_barrier(P.sync + SYNC_SPK_FWD + dir * SYNC_DIR, P.sync + SYNC_ABORT, nwg * (unsigned)(t + 1), lds_ok, nullptr, 0, t + 1 < P.T)) return;
    STAMP_ACC(3);
  }
  STAMP_DUMP(P, 16, R.x == 0 && R.y == 0 && R.z == 0);
  STAMP_DUMP(P, 56, R.x == 5 && R.y == 1 && R.z == 0);
}

// ================================================================================================ LSTHM forward
// Role: (stream m, units u0..u0+7, row block mb).  gates = pre[t] + U h_{t-1} + V z_{t-1} (+ U.bias + V.bias), order f,i,o,c~.
struct LsthmFwdB {
  const DirP& D; int m, u0, H;
  __device__ __forceinline__ void operator()(int n, int k, float* b) const {
    if (D.WpkL[m] && k < 2 * H) {
      load8(D.WpkL[m] + ((((long)(u0 >> 3) * (2 * H / 8)) + (k >> 3)) * 32 + n) * 8, b);
      return;
    }
    const long wrow = (long)(n >> 3) * H + u0 + (n & 7);
    if (k < H) load8(D.U[m] + wrow * H + k, b);
    else if (k < 2 * H) load8(D.V[m] + wrow * H + (k - H), b);
    else load8(D.S[m] + wrow * H + (k - 2 * H), b);          // WITHS only (K = 3H)
  }
};

// WITHS: the speaker term S h_q[t] is part of the step (K = 3H) instead of the hoisted pre-activation GEMM, so that the chain
// can run CONCURRENTLY with the speaker chain that produces h_q (pipelined persistent kernels).
template <int PS, int NP, bool WITHS = false>
__device__ __forceinline__ void lsthm_gates_body(const CellK& P, const DirP& D, const WS& ws, int t, int m, int u0, int mb,
                                                 const float (*bpre)[8], float* red, float* tile) {
  const int H = P.H, B = P.B, T = P.T;
  const float* hz_old = D.hz + (long)t * B * 3 * H;
  float* hz_new = D.hz + (long)(t + 1) * B * 3 * H;
  const float* c_old = D.cstate + ((long)m * (T + 1) + t) * B * H;
  float* c_new = D.cstate + ((long)m * (T + 1) + t + 1) * B * H;
  const int tid = threadIdx.x;

  // epilogue operands that do not depend on the matvec: issue their loads first so their latency hides behind it
  float pre4[4] = {0.f, 0.f, 0.f, 0.f}, c_prev = 0.f;
  int tau = -1;
  if (tid < 256) {
    const int b = mb * 32 + (tid >> 3), u = u0 + (tid & 7);
    if (b < B) {
      const float* pr = D.pre + ((long)m * T * B + (long)t * B + b) * 4 * H + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pre4[g] = pr[g * H] + D.Ub[m][g * H + u] + D.Vb[m][g * H + u];
        if (WITHS) pre4[g] += D.Sb[m][g * H + u];
      }
      c_prev = ldx<PS>(ws, c_old + (long)b * H + u);
      tau = D.rev ? D.rev[(long)t * B + b] : t;
    }
  }
  auto aload = [&](int r, int k, float* a) {
    const int b = mb * 32 + r;
    if (b >= B) { zero8(a); return; }
    const float* row = hz_old + (long)b * 3 * H;
    if (k < H) load8x<PS>(ws, row + m * H + k, a);
    else if (k < 2 * H) load8x<PS>(ws, row + 2 * H + (k - H), a);
    else load8x<PS>(ws, D.HQ + ((long)t * B + b) * H + (k - 2 * H), a);
  };
  STAMP_ACC(0);
  wg_mm32<NP, (PS == 2 ? 4 : 1)>(WITHS ? 3 * H : 2 * H, aload, LsthmFwdB{D, m, u0, H}, bpre, red, tile);
  STAMP_ACC(1);
  if (tid < 256) {
    const int rr = tid >> 3, uu = tid & 7;
    const int b = mb * 32 + rr, u = u0 + uu;
    if (b < B) {
      const float gf = sigmoidf_(tile[rr * 32 + 0 + uu] + pre4[0]);
      const float gi = sigmoidf_(tile[rr * 32 + 8 + uu] + pre4[1]);
      const float go = sigmoidf_(tile[rr * 32 + 16 + uu] + pre4[2]);
      const float gc = tanhf(tile[rr * 32 + 24 + uu] + pre4[3]);
      const float cn = gf * c_prev + gi * gc;
      float hn = tanhf(cn) * go;
      if (drop_state_on(P, D)) hn *= drop_h(P, D, t, m, b, u);               // :211 / :213
      stx<PS>(ws, c_new + (long)b * H + u, cn);
      stx<PS>(ws, hz_new + (long)b * 3 * H + m * H + u, hn);
      float* g = D.gates + ((long)m * T * B + (long)t * B + b) * 4 * H + u;
      g[0] = gf; g[H] = gi; g[2 * H] = go; g[3 * H] = gc;
      if (tau >= 0) D.out[((long)tau * B + b) * P.ldo + m * H + u] = hn;
    }
  }
  STAMP_ACC(2);
}

// Row phase, one dialogue row b per call (NT threads): z[b,i] = sum_j softmax_j(c_l[i] * s_b * Wk[j]) c_a[j]  (:59-72, rank-1 form)
// thread (i = tid % H, q = tid / H) covers keys j in [q*JC, (q+1)*JC).  scr: kc[H] (float4) pZ[NT] pN[NT] sh[16] pN2[NT] pN3[NT]
// Besides z it leaves the softmax statistics of the row (Z, N2, N3, s) in rstat for the BPTT.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `after_loads` runs right behind the row's own operand loads: loads issued there are YOUNGER than the row's, so the row phase
// does not wait for them (vmcnt retires in order) -- used to fetch the next step's early-product operands under the exp2 work.
// STATS = false: the statistics are left to the stats roles of the fused launch (stats_fwd_role), off the chain.
// IWT > 0: this call covers only the IWT query units i0 .. i0 + IWT - 1 of the row (another workgroup covers the rest; every unit's
// softmax still runs over all H keys, so nothing is exchanged): thread (tid % IWT, tid / IWT), H IWT / NT keys per thread.
template <bool PS, int JCT, class Hook = NoHook, bool STATS = true, bool SV = false, int IWT = 0>       // JCT = keys per thread chunk when known at compile time (fully unrolled loops), 0 = runtime; SV: sentinel-validated loads
__device__ __forceinline__ void lsthm_z_body(const CellK& P, const DirP& D, const WS& ws, int t, int b, const float* att, float* scr,
                                             Hook after_loads = Hook(), int i0 = 0) {
  const int H = P.H, B = P.B, T = P.T;
  const int IW = IWT ? IWT : H;
  const int Q = NT / IW, JC = JCT ? JCT : H / Q;
  // per key j ONE 16-byte LDS word (Wk[j], c_a[j], c_a[j] Wk[j], -): the inner loop issues one broadcast ds_read_b128 per key
  float4* kc = reinterpret_cast<float4*>(scr);
  float* pZ = scr + 4 * H;
  float* pN = pZ + NT;
  float* sh = pN + NT;
  float* pN2 = sh + 16;
  const float* wk = att;
  const float* wq = att + H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* c_l = D.cstate + ((long)0 * (T + 1) + t + 1) * B * H + (long)b * H;
  const float* c_a = D.cstate + ((long)1 * (T + 1) + t + 1) * B * H + (long)b * H;
  const int i = i0 + (tid & (IW - 1)), q = tid / IW;
  float sp = 0.f;
  float cv = 0.f;
  if (tid < H) cv = ldx<PS>(ws, c_a + tid);
  float cli = ldx<PS>(ws, c_l + i);
  if (SV) {                          // c_l[t], c_a[t] come from the gate epilogues of all workgroups of the direction
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(is_sent(cli) || is_sent(cv)) != 0ull) {
      if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
      if (tid < H) cv = ldx<PS>(ws, c_a + tid);
      cli = ldx<PS>(ws, c_l + i);
    }
  }
  if (tid < H) {
    const float w = wk[tid];
    kc[tid] = make_float4(w, cv, cv * w, 0.f);
    sp = wq[tid] * cv;
  }
  const int tau = D.rev ? D.rev[(long)t * B + b] : t;
  if (!SV) after_loads();            // (SV: the hook runs behind the exp2 loop instead -- see there)
  sp = wave_sum(sp);
  if (lane == 0) sh[wave] = sp;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += sh[w];
  s /= sqrtf((float)H);
  const float u = cli * s;
  const float mx = (u >= 0.f) ? u * att[2 * H] : u * att[2 * H + 1];
  const float u2 = u * LOG2E, m2 = mx * LOG2E;
  float Z = 0.f, N = 0.f, N2 = 0.f, N3 = 0.f;     // N2, N3: the two extra sums the backward needs (saved below)
  if (drop_attn_on(P, D)) {      // :69 dropout(softmax): the normaliser Z (and N3) stay unmasked, the value sums take the factor
    const DropKey dk = s_drop[2];
    const uint32_t e0 = drop_attn_row(P, t, b) + (uint32_t)(i * H + q * JC);
    const float4* kcc = kc + q * JC;
    for (int jj = 0; jj < JC; jj += 2) {          // e0 and JC are even: keys jj, jj+1 share one 16+16-bit word
      const uint32_t w = drop_word16(dk, e0 + (uint32_t)jj);
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const float4 k4 = kcc[jj + o];
        const float e = __builtin_amdgcn_exp2f(fmaf(u2, k4.x, -m2));
        const float ef = e * drop_half16(dk, w, (uint32_t)o);
        Z += e;
        N = fmaf(ef, k4.y, N);
        N2 = fmaf(ef, k4.z, N2);
        N3 = fmaf(e, k4.x, N3);
      }
    }
  } else {
    const float4* kcc = kc + q * JC;
#pragma unroll
    for (int jj = 0; jj < (JCT ? JCT : JC); ++jj) {
      const float4 k4 = kcc[jj];
      const float e = __builtin_amdgcn_exp2f(fmaf(u2, k4.x, -m2));
      Z += e;
      N = fmaf(e, k4.y, N);
      if (STATS) {
        N2 = fmaf(e, k4.z, N2);
        N3 = fmaf(e, k4.x, N3);
      }
    }
  }
  float* pN3 = pN2 + NT;
  // SV: h_t of the other workgroups was stored right behind the c_t this row polled for, but nothing orders the two: requested at the
  // start of the row phase it still read as the sentinel for some producers and had to be polled again AFTER the row phase (a full
  // round trip on the critical path, 0.8 us measured).  Requested here, one exp2 loop later, it is final and lands under the reduction.
  if (SV) after_loads();
  if (q > 0) {
    pZ[tid] = Z; pN[tid] = N;
    if (STATS) { pN2[tid] = N2; pN3[tid] = N3; }
  }
  __syncthreads();
  if (q == 0) {
    for (int qq = 1; qq < Q; ++qq) {
      Z += pZ[qq * IW + tid]; N += pN[qq * IW + tid];
      if (STATS) { N2 += pN2[qq * IW + tid]; N3 += pN3[qq * IW + tid]; }
    }
    const float z = N / Z;
    stx<PS>(ws, D.hz + ((long)(t + 1) * B + b) * 3 * H + 2 * H + i, z);
    if (tau >= 0) D.out[((long)tau * B + b) * P.ldo + 2 * H + i] = z;
    if (STATS) *reinterpret_cast<float4*>(D.rstat + (((long)t * B + b) * H + i) * 4) = make_float4(Z, N2, N3, s);
  }
  __syncthreads();     // scratch is reused by the next row / phase
}

// ---- pipelined form of the forward step (persistent kernels only) --------------------------------------------------------------
// gates[t] = pre[t] + [h_{t-1} | h_q[t]] [U | S]^T  +  z_{t-1} V^T.  Only the last term needs z: the first product ("early", K = 2H)
// is known as soon as h_{t-1} is published.  Its operands are fetched at the start of the row phase and its MFMA chain runs in the
// SHADOW OF THE BARRIER that ends the row phase (arrive -> MFMAs -> wait): ~1 us in which the workgroup would only poll.  The gates
// phase then only adds the z product ("late", K = H).  The k range of both products is split over the 8 waves; each wave keeps
// ONE accumulator across the barrier (no LDS round trip in between) and the 8 partials are reduced once.
// (Measured alternatives: the early chain interleaved with the exp2 loop of the row phase in one wave, or on four dedicated
// waves next to four exp2 waves, did not overlap on this hardware: 3.2 / 4.1 us for the row phase instead of 1.6.)
// wave w, pass p < NPE: early k = w*16*NPE + 16p + half*8 (k < H: U / h, else S / h_q);  pass p >= NPE: late k = w*16*NPL + ...
template <int NP>
struct FwdSplit { static constexpr int NPE = 2 * NP / 3, NPL = NP / 3; };

template <int NP>
__device__ __forceinline__ void lsthm_preload_b_split(const DirP& D, int m, int u0, int H, float (*bpre)[8]) {
  constexpr int NPE = FwdSplit<NP>::NPE, NPL = FwdSplit<NP>::NPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  const long wrow = (long)(n >> 3) * H + u0 + (n & 7);
#pragma unroll
  for (int p = 0; p < NPE; ++p) {      // passes [0, NPE/2): U (h part), [NPE/2, NPE): S (h_q part); both cover k = 0..H-1 over the waves
    const int k = wave * 8 * NPE + 16 * (p % (NPE / 2)) + half * 8;
    load8((p < NPE / 2 ? D.U[m] : D.S[m]) + wrow * H + k, bpre[p]);
  }
#pragma unroll
  for (int p = 0; p < NPL; ++p) load8(D.V[m] + wrow * H + wave * 16 * NPL + 16 * p + half * 8, bpre[NPE + p]);
}
// A fragments of the early product for step t (rows = dialogues of block mb).  PART 0: h_m[t-1] (= hz[t], stream m) into passes
// [0, NPE/2); PART 1: h_q[t] into passes [NPE/2, NPE).
template <int NP, int PART>
__device__ __forceinline__ void lsthm_early_aload(const CellK& P, const DirP& D, const WS& ws, int t, int m, int mb, float (*a)[8]) {
  constexpr int NPE = FwdSplit<NP>::NPE;
  const int H = P.H, B = P.B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int b = mb * 32 + r;
  const int bc = b < B ? b : B - 1;           // clamped row, zeroed below (no load inside a branch)
#pragma unroll
  for (int p = 0; p < NPE / 2; ++p) {
    const int k = wave * 8 * NPE + 16 * p + half * 8;
    const float* src = PART == 0 ? D.hz + ((long)t * B + bc) * 3 * H + m * H + k : D.HQ + ((long)t * B + bc) * H + k;
    load8x<true>(ws, src, a[PART * NPE / 2 + p]);
  }
  if (b >= B) {
#pragma unroll
    for (int p = 0; p < NPE / 2; ++p) zero8(a[PART * NPE / 2 + p]);
  }
}
// The same fragments, re-loaded until none of the words is the sentinel (every wave polls for itself)
template <int NP, int PART>
__device__ __forceinline__ void lsthm_early_aload_valid(const CellK& P, const DirP& D, const WS& ws, int t, int m, int mb, float (*a)[8]) {
  constexpr int NPE = FwdSplit<NP>::NPE;
  unsigned spins = 0;
  while (true) {
    bool bad = false;
#pragma unroll
    for (int p = PART * NPE / 2; p < (PART + 1) * NPE / 2; ++p) bad |= any_sent8(a[p]);
    if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;
    if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
    lsthm_early_aload<NP, PART>(P, D, ws, t, m, mb, a);
  }
}
// PART 0 / 1: h part / h_q part of the early chain
template <int NP, int PART>
__device__ __forceinline__ f32x16 lsthm_early_mm(const float (*a)[8], const float (*bpre)[8], f32x16 acc) {
  constexpr int NPE = FwdSplit<NP>::NPE;
#pragma unroll
  for (int p = PART * NPE / 2; p < (PART + 1) * NPE / 2; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][j], bpre[p][j], acc, 0, 0, 0);
  return acc;
}
// Epilogue operands of step t that do not depend on the chain (pre-activation row + biases): fetched in the barrier shadow too.
struct GatePre { float pre4[4]; int tau; };
__device__ __forceinline__ GatePre lsthm_gate_prefetch(const CellK& P, const DirP& D, int t, int m, int u0, int mb) {
  GatePre g;
  g.pre4[0] = g.pre4[1] = g.pre4[2] = g.pre4[3] = 0.f;
  g.tau = -1;
  const int tid = threadIdx.x;
  const int H = P.H, B = P.B, T = P.T;
  if (tid < 256) {
    const int b = mb * 32 + (tid >> 3), u = u0 + (tid & 7);
    if (b < B) {
      const float* pr = D.pre + ((long)m * T * B + (long)t * B + b) * 4 * H + u;
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) g.pre4[gt] = pr[gt * H] + D.Ub[m][gt * H + u] + D.Vb[m][gt * H + u] + D.Sb[m][gt * H + u];
      g.tau = D.rev ? D.rev[(long)t * B + b] : t;
    }
  }
  return g;
}

// The pre-activation row alone (the three bias vectors do not depend on the step: the role adds their sum, fetched once), through
// buffer loads with a per-thread offset register and scalar per-step offsets (no vector address arithmetic per load).
template <int HC>
__device__ __forceinline__ GatePre lsthm_gate_prefetch_buf(const CellK& P, const DirP& D, const WS& ws, int t, int m, int u0, int mb) {
  GatePre g;
  g.pre4[0] = g.pre4[1] = g.pre4[2] = g.pre4[3] = 0.f;
  g.tau = -1;
  const int tid = threadIdx.x;
  const int B = P.B, T = P.T;
  if (tid < 256) {
    const int b = mb * 32 + (tid >> 3), u = u0 + (tid & 7);
    if (b < B) {
      const int vo = (b * 4 * HC + u) * 4;
      const int so = (int)((const char*)D.pre - ws.base) + ((m * T + t) * B) * 4 * HC * 4;
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) g.pre4[gt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ws.r, vo + gt * HC * 4, so, 0));
      g.tau = D.rev ? D.rev[(long)t * B + b] : t;
    }
  }
  return g;
}

// Gates phase of step t: acc (early product, in registers) += z_{t-1} V^T, cross-wave reduction, LSTM epilogue.
// c_state: this thread's cell state c_{t-1}[b][u] (the same thread owns the same (b, u) every step: it never leaves the register).
template <int NP, bool SV = false>
__device__ __forceinline__ void lsthm_gates_late(const CellK& P, const DirP& D, const WS& ws, int t, int m, int u0, int mb,
                                                 const float (*bpre)[8], f32x16 acc, const GatePre& gp, float& c_state, float* red,
                                                 float* tile) {
  constexpr int NPE = FwdSplit<NP>::NPE, NPL = FwdSplit<NP>::NPL;
  const int H = P.H, B = P.B, T = P.T;
  float* hz_new = D.hz + (long)(t + 1) * B * 3 * H;
  float* c_new = D.cstate + ((long)m * (T + 1) + t + 1) * B * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  // late A fragments: z_{t-1} rows (published by the row phase of the previous step)
  float a[NPL][8];
  {
    const int b = mb * 32 + r;
    const int bc = b < B ? b : B - 1;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
      load8x<true>(ws, D.hz + ((long)t * B + bc) * 3 * H + 2 * H + wave * 16 * NPL + 16 * p + half * 8, a[p]);
    if (SV && t > 0) {                 // z_{t-1} comes from the other workgroups' row phases: poll until every word is final
      unsigned spins = 0;
      while (true) {
        bool bad = false;
#pragma unroll
        for (int p = 0; p < NPL; ++p) bad |= any_sent8(a[p]);
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;
        if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          load8x<true>(ws, D.hz + ((long)t * B + bc) * 3 * H + 2 * H + wave * 16 * NPL + 16 * p + half * 8, a[p]);
      }
    }
    if (b >= B) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) zero8(a[p]);
    }
  }
  STAMP_ACC(0);
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][j], bpre[NPE + p][j], acc, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
    red[wave * 1024 + row * 32 + r] = acc[i];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 1024 / NT; ++e) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[w * 1024 + tid + e * NT];
    tile[tid + e * NT] = sum;
  }
  __syncthreads();
  STAMP_ACC(1);
  if (tid < 256) {
    const int rr = tid >> 3, uu = tid & 7;
    const int b = mb * 32 + rr, u = u0 + uu;
    if (b < B) {
      const float gf = FASTNL ? sigmoid_fast(tile[rr * 32 + 0 + uu] + gp.pre4[0]) : sigmoidf_(tile[rr * 32 + 0 + uu] + gp.pre4[0]);
      const float gi = FASTNL ? sigmoid_fast(tile[rr * 32 + 8 + uu] + gp.pre4[1]) : sigmoidf_(tile[rr * 32 + 8 + uu] + gp.pre4[1]);
      const float go = FASTNL ? sigmoid_fast(tile[rr * 32 + 16 + uu] + gp.pre4[2]) : sigmoidf_(tile[rr * 32 + 16 + uu] + gp.pre4[2]);
      const float gc = FASTNL ? tanh_fast(tile[rr * 32 + 24 + uu] + gp.pre4[3]) : tanhf(tile[rr * 32 + 24 + uu] + gp.pre4[3]);
      const float cn = gf * c_state + gi * gc;
      float hn = (FASTNL ? tanh_fast(cn) : tanhf(cn)) * go;
      if (drop_state_on(P, D)) hn *= drop_h(P, D, t, m, b, u);               // :211 / :213
      c_state = cn;
      stx<true>(ws, c_new + (long)b * H + u, cn);
      stx<true>(ws, hz_new + (long)b * 3 * H + m * H + u, hn);
      float* g = D.gates + ((long)m * T * B + (long)t * B + b) * 4 * H + u;
      g[0] = gf; g[H] = gi; g[2 * H] = go; g[3 * H] = gc;
      if (gp.tau >= 0) D.out[((long)gp.tau * B + b) * P.ldo + m * H + u] = hn;
    }
  }
  STAMP_ACC(2);
}

// per-step launches
__global__ __launch_bounds__(NT) void lsthm_fwd_gates(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int dir = blockIdx.z / P.nmb, mb = blockIdx.z % P.nmb;
  drop_init(P, P.d[dir]);
  lsthm_gates_body<false, 0>(P, P.d[dir], ws, t, blockIdx.y, blockIdx.x * 8, mb, nullptr, smem, smem + RED_FLOATS);
}
__global__ __launch_bounds__(NT) void lsthm_fwd_z(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  float* att = smem + RED_FLOATS;
  att_prepare(P.d[blockIdx.y], P.H, att, smem);
  drop_init(P, P.d[blockIdx.y]);
  lsthm_z_body<false, 0>(P, P.d[blockIdx.y], ws, t, blockIdx.x, att, smem);
}

// ---- H > 512 (per-step launches only; BASELINE configs[4] runs H = 1024) ----------------------------------------------------------
// The row phase of one dialogue row is shared by H/128 workgroups, each owning 128 query units i: the softmax of a unit runs over all
// H keys and nothing is exchanged between the workgroups.  thread (il = tid % 128, q = tid / 128) covers keys [q H/4, (q+1) H/4) of
// unit i0 + il.  LDS: att [2H+16] | kc float4[H] | partials 4 x [NT] | sh[16].  Leaves the same outputs as lsthm_z_body (z, out, rstat).
constexpr int WIDE_IW = 128;
static size_t z_wide_lds_bytes(int H) { return ((size_t)(2 * H + 16) + 4 * (size_t)H + 4 * NT + 16) * sizeof(float); }
// PS = true: inside the persistent wide launch (cell_wide_fwd_persist): c_l / c_a come from other workgroups of the same launch and z goes
// to them (write-through / L1-bypassing accesses through `ws`)
template <int PS>
__device__ __forceinline__ void lsthm_fwd_z_wide_body(const CellK& P, const DirP& D, const WS& ws, int t, int b, int zblk, float* smem) {
  const int H = P.H, B = P.B, T = P.T, i0 = zblk * WIDE_IW;
  float* att = smem;
  float4* kc = reinterpret_cast<float4*>(att + 2 * H + 16);
  float* part = reinterpret_cast<float*>(kc + H);
  float* sh = part + 4 * NT;
  att_prepare(D, H, att, sh);
  drop_init(P, D);
  constexpr int Q = NT / WIDE_IW;
  const int tid = threadIdx.x, il = tid % WIDE_IW, q = tid / WIDE_IW, i = i0 + il, JC = H / Q;
  const float* c_l = D.cstate + ((long)0 * (T + 1) + t + 1) * B * H + (long)b * H;
  const float* c_a = D.cstate + ((long)1 * (T + 1) + t + 1) * B * H + (long)b * H;
  float sp = 0.f;
  for (int k = tid; k < H; k += NT) {
    const float cv = ldx<PS>(ws, c_a + k), w = att[k];
    kc[k] = make_float4(w, cv, cv * w, 0.f);
    sp = fmaf(att[H + k], cv, sp);
  }
  const float s = block_sum(sp, sh) / sqrtf((float)H);           // (its barriers publish kc)
  const float u = ldx<PS>(ws, c_l + i) * s;
  const float mx = (u >= 0.f) ? u * att[2 * H] : u * att[2 * H + 1];
  const float u2 = u * LOG2E, m2 = mx * LOG2E;
  float Z = 0.f, N = 0.f, N2 = 0.f, N3 = 0.f;
  const float4* kcc = kc + q * JC;
  if (drop_attn_on(P, D)) {
    const DropKey dk = s_drop[2];
    const uint32_t e0 = drop_attn_row(P, t, b) + (uint32_t)(i * H + q * JC);
    for (int jj = 0; jj < JC; ++jj) {
      const float4 k4 = kcc[jj];
      const float e = __builtin_amdgcn_exp2f(fmaf(u2, k4.x, -m2));
      const float ef = e * drop_scale16(dk, e0 + (uint32_t)jj);
      Z += e;
      N = fmaf(ef, k4.y, N);
      N2 = fmaf(ef, k4.z, N2);
      N3 = fmaf(e, k4.x, N3);
    }
  } else {
#pragma unroll 8
    for (int jj = 0; jj < JC; ++jj) {
      const float4 k4 = kcc[jj];
      const float e = __builtin_amdgcn_exp2f(fmaf(u2, k4.x, -m2));
      Z += e;
      N = fmaf(e, k4.y, N);
      N2 = fmaf(e, k4.z, N2);
      N3 = fmaf(e, k4.x, N3);
    }
  }
  part[tid] = Z; part[NT + tid] = N; part[2 * NT + tid] = N2; part[3 * NT + tid] = N3;
  __syncthreads();
  if (q == 0) {
    for (int qq = 1; qq < Q; ++qq) {
      Z += part[qq * WIDE_IW + il]; N += part[NT + qq * WIDE_IW + il];
      N2 += part[2 * NT + qq * WIDE_IW + il]; N3 += part[3 * NT + qq * WIDE_IW + il];
    }
    const float z = N / Z;
    stx<PS>(ws, D.hz + ((long)(t + 1) * B + b) * 3 * H + 2 * H + i, z);
    const int tau = D.rev ? D.rev[(long)t * B + b] : t;
    if (tau >= 0) D.out[((long)tau * B + b) * P.ldo + 2 * H + i] = z;
    *reinterpret_cast<float4*>(D.rstat + (((long)t * B + b) * H + i) * 4) = make_float4(Z, N2, N3, s);
  }
}
__global__ __launch_bounds__(NT) void lsthm_fwd_z_wide(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  lsthm_fwd_z_wide_body<false>(P, P.d[blockIdx.y], ws, t, blockIdx.x, blockIdx.z, smem);
}

// persistent launch: grid (H/8, 2 streams, ndir*nmb); two barriers per step (gates -> z -> next gates).  Runs concurrently with
// spk_fwd_persist: step t starts only once the speaker counter shows h_q[t] published (checked inside the previous barrier).
// MSER_OPT_FWD_SENTINEL form of lsthm_fwd_role below: the same step without a single counter -- every operand that another
// workgroup produces (z_{t-1} for the late product, c_l / c_a for the row phase, h_t and h_q[t+1] for the early product) is loaded
// with sentinel validation.  The order of the work inside a step is unchanged (late product -> epilogue -> [h_q part of the next
// early product] -> row phase -> [h part, pre-activation fetch]).
template <int NP, bool STATS = true>
__device__ __forceinline__ void lsthm_fwd_role_sv(const CellK& P, const Role R, float* smem, const WS& ws) {
  float* red = smem;
  float* tile = smem + RED_FLOATS;
  float* att = tile + 1024;
  const int dir = R.z / P.nmb, mb = R.z % P.nmb;
  const DirP& D = P.d[dir];
  const int m = R.y, u0 = R.x * 8;
  const unsigned nwg = R.gx * R.gy * P.nmb;
  const int w = (mb * R.gy + R.y) * R.gx + R.x;
  float bpre[NP][8];
  lsthm_preload_b_split<NP>(D, m, u0, P.H, bpre);
  if (threadIdx.x == 0) s_poll_abort = 0;
  att_prepare(D, P.H, att, red);
  drop_init(P, D);
  STAMP_INIT();
  constexpr int JCT = (128 * NP / 3) * (128 * NP / 3) / NT;
  float a[FwdSplit<NP>::NPE][8];
  lsthm_early_aload<NP, 0>(P, D, ws, 0, m, mb, a);                  // h_{-1} = 0 (hz[0] is zeroed by FWD_PREP)
  lsthm_early_aload<NP, 1>(P, D, ws, 0, m, mb, a);                  // h_q[0]
  lsthm_early_aload_valid<NP, 1>(P, D, ws, 0, m, mb, a);
  f32x16 acc = lsthm_early_mm<NP, 1>(a, bpre, lsthm_early_mm<NP, 0>(a, bpre, f32x16{0}));
  GatePre gp = lsthm_gate_prefetch(P, D, 0, m, u0, mb);
  // the step-independent part of the pre-activation: U.bias + V.bias + S.bias of this thread's (gate, unit)
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < 256) {
    const int u = u0 + (threadIdx.x & 7);
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) bsum[gt] = D.Ub[m][gt * P.H + u] + D.Vb[m][gt * P.H + u] + D.Sb[m][gt * P.H + u];
  }
  float c_state = 0.f;
  for (int t = 0; t < P.T; ++t) {
    const bool more = t + 1 < P.T;
    // next step's pre-activation row: requested HERE, in front of the longest wait of the step (the poll for z_{t-1} below), because the
    // wave's loads retire in order -- requested in front of the h_t poll at the bottom of the step it was what that poll waited for
    GatePre gp_n = gp;
    if (more) {
      gp_n = lsthm_gate_prefetch_buf<128 * NP / 3>(P, D, ws, t + 1, m, u0, mb);
    }
    if (more) lsthm_early_aload<NP, 1>(P, D, ws, t + 1, m, mb, a);  // h_q[t+1] (the speaker chain normally runs far ahead): requested now,
    lsthm_gates_late<NP, true>(P, D, ws, t, m, u0, mb, bpre, acc, gp, c_state, red, tile);           // validated after the gates phase
    if (s_poll_abort) return;                                       // (read behind the phase's workgroup barriers: uniform)
    if (more) {
      lsthm_early_aload_valid<NP, 1>(P, D, ws, t + 1, m, mb, a);
      acc = lsthm_early_mm<NP, 1>(a, bpre, f32x16{0});
    }
    STAMP_ACC(3);
    bool fetched = !more;
    auto fetch = [&]() { if (!fetched) { lsthm_early_aload<NP, 0>(P, D, ws, t + 1, m, mb, a); fetched = true; } };
    if (NP == 6 && P.fwd_rowsplit == 2) {      // H = 256: 64 workgroups per direction, two per dialogue row (128 query units each)
      if (w < 2 * P.B) lsthm_z_body<true, JCT / 2, decltype(fetch), STATS, true, 128>(P, D, ws, t, w >> 1, att, red, fetch, (w & 1) * 128);
    } else {
      for (int b = w; b < P.B; b += (int)nwg) lsthm_z_body<true, JCT, decltype(fetch), STATS, true>(P, D, ws, t, b, att, red, fetch);
    }
    fetch();
    STAMP_ACC(4);
    if (!more) break;
    gp = gp_n;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) gp.pre4[gt] += bsum[gt];
    lsthm_early_aload_valid<NP, 0>(P, D, ws, t + 1, m, mb, a);      // h_t of every workgroup of the direction
    acc = lsthm_early_mm<NP, 0>(a, bpre, acc);
    STAMP_ACC(5);
  }
  STAMP_DUMP(P, 24, R.x == 3 && R.y == 1 && R.z == 0);
}

template <int NP, bool STATS = true>
__device__ __forceinline__ void lsthm_fwd_role(const CellK& P, const Role R, float* smem, const WS& ws) {
  if (P.fwd_sentinel) { lsthm_fwd_role_sv<NP, STATS>(P, R, smem, ws); return; }
  float* red = smem;
  float* tile = smem + RED_FLOATS;
  float* att = tile + 1024;
  int* lds_ok = (int*)(att + att_floats(P.H));
  const int dir = R.z / P.nmb, mb = R.z % P.nmb;
  const DirP& D = P.d[dir];
  const int m = R.y, u0 = R.x * 8;
  const unsigned nwg = R.gx * R.gy * P.nmb;
  const unsigned nwg_spk = nwg;                                          // the speaker chain uses the same logical grid
  const int w = (mb * R.gy + R.y) * R.gx + R.x;    // linear workgroup index inside the direction
  float bpre[NP][8];
  lsthm_preload_b_split<NP>(D, m, u0, P.H, bpre);
  att_prepare(D, P.H, att, red);
  drop_init(P, D);
  unsigned* cnt = P.sync + SYNC_LSTHM_FWD + dir * SYNC_DIR;
  const unsigned* spk = P.sync + SYNC_SPK_FWD + dir * SYNC_DIR;
  unsigned nbar = 0;
  // h_q[0] and h_q[1] published (the speaker chain needs only qmask and normally runs far ahead)
  if (!dir_barrier(nullptr, P.sync + SYNC_ABORT, 0, lds_ok, spk, nwg_spk * (P.T > 1 ? 2u : 1u))) return;
  STAMP_INIT();
  constexpr int JCT = (128 * NP / 3) * (128 * NP / 3) / NT;      // NP = 3H/128, keys per thread = H*H/NT
  float a[FwdSplit<NP>::NPE][8];
  lsthm_early_aload<NP, 0>(P, D, ws, 0, m, mb, a);                  // h_{-1} = 0 (hz[0] is zeroed)
  lsthm_early_aload<NP, 1>(P, D, ws, 0, m, mb, a);                  // h_q[0]
  f32x16 acc = lsthm_early_mm<NP, 1>(a, bpre, lsthm_early_mm<NP, 0>(a, bpre, f32x16{0}));
  GatePre gp = lsthm_gate_prefetch(P, D, 0, m, u0, mb);
  float c_state = 0.f;                                              // c_{-1} = 0
  for (int t = 0; t < P.T; ++t) {
    const bool more = t + 1 < P.T;
    if (more) lsthm_early_aload<NP, 1>(P, D, ws, t + 1, m, mb, a);  // h_q[t+1]: known to be published since the previous barrier
    lsthm_gates_late<NP>(P, D, ws, t, m, u0, mb, bpre, acc, gp, c_state, red, tile);   // stamps 0 (loads) 1 (mm) 2 (epilogue)
    // split-phase barriers: MFMA chains whose operands are already in registers run while the hand-off is in flight.
    // Behind this one h_t is published; in its shadow: the h_q part of the next step's early product.
    barrier_arrive(cnt);
    ++nbar;
    acc = lsthm_early_mm<NP, 1>(a, bpre, f32x16{0});
    if (!barrier_wait(cnt, P.sync + SYNC_ABORT, nwg * nbar, lds_ok)) return;
    STAMP_ACC(3);
    // h part of the next step's early product: requested behind the first row's own loads, in flight during the row phase
    bool fetched = !more;
    auto fetch = [&]() { if (!fetched) { lsthm_early_aload<NP, 0>(P, D, ws, t + 1, m, mb, a); fetched = true; } };
    if (NP == 6 && P.fwd_rowsplit == 2) {
      if (w < 2 * P.B) lsthm_z_body<true, JCT / 2, decltype(fetch), STATS, false, 128>(P, D, ws, t, w >> 1, att, red, fetch, (w & 1) * 128);
    } else {
      for (int b = w; b < P.B; b += (int)nwg) lsthm_z_body<true, JCT, decltype(fetch), STATS>(P, D, ws, t, b, att, red, fetch);
    }
    fetch();                                                        // a workgroup that owns no row
    STAMP_ACC(4);
    if (!more) break;
    // behind this one z_t is published; in its shadow: the h part of the early product and the pre-activation fetch.  The wait
    // also covers h_q[t+2] of the speaker chain (fetched at the top of the next iteration).
    barrier_arrive(cnt);
    ++nbar;
    gp = lsthm_gate_prefetch(P, D, t + 1, m, u0, mb);
    acc = lsthm_early_mm<NP, 0>(a, bpre, acc);
    if (!barrier_wait(cnt, P.sync + SYNC_ABORT, nwg * nbar, lds_ok, 0, t + 2 < P.T ? spk : nullptr, nwg_spk * (unsigned)(t + 3))) return;
    STAMP_ACC(5);
  }
  STAMP_DUMP(P, 24, R.x == 3 && R.y == 1 && R.z == 0);
}

// ================================================================================================ LSTHM backward
// Row phase (one dialogue row per call): attention backward (recomputing the softmax with exp2), then the gate backward for
// both streams.  Writes dgates[t], the dc carry, dHQ[t] (the h_q part of dout) and accumulates the attention-vector grads of
// its own row (reduced over rows once after the chain).
// scr floats: ca[H] cl[H] cw[H] coef[H][8] p[3][NT] sh[16]
// Saved-state operands of one row's backward step: everything that does NOT come from another workgroup in this launch, so it
// can be fetched while the inter-workgroup barrier of the previous phase is still in flight (threads tid < H hold unit tid).
struct RowPre {
  float cav, clv, dz_out, zi, dh_out[2], gsv[2][4], cprev[2], carry[2], dhq;
  float4 st;           // the row's softmax statistics saved by the forward (rstat: Z, N2, N3, s)
};
// known / tau_known: the natural time position of (t, b) when the caller has fetched it already (persistent roles, a step ahead: a
// table lookup here would be a DEPENDENT load, and waiting for it means waiting for every older vector-memory operation of the wave:
// loads and stores share one in-order counter on gfx9).
__device__ __forceinline__ int lsthm_tau(const DirP& D, int t, int b, int B) { return D.rev ? D.rev[(long)t * B + b] : t; }
__device__ __forceinline__ RowPre lsthm_bwd_row_prefetch(const CellK& P, const DirP& D, int t, int b, bool known = false, int tau_known = 0) {
  RowPre r;
  const int H = P.H, B = P.B, T = P.T;
  const int i = threadIdx.x;
  r.cav = r.clv = r.dz_out = r.zi = r.dhq = 0.f;
  r.st = make_float4(1.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < 2; ++m) { r.dh_out[m] = r.cprev[m] = r.carry[m] = 0.f; r.gsv[m][0] = r.gsv[m][1] = r.gsv[m][2] = r.gsv[m][3] = 0.f; }
  if (i < H) {
    const long rowt = (long)t * B + b;
    r.st = *reinterpret_cast<const float4*>(D.rstat + (rowt * H + i) * 4);
    const long SA = (long)B * H;
    const int tau = known ? tau_known : lsthm_tau(D, t, b, B);
    const float* dorow = (tau >= 0) ? D.dout + ((long)tau * B + b) * P.ldo : nullptr;
    r.clv = D.cstate[((long)0 * (T + 1) + t + 1) * B * H + (long)b * H + i];
    r.cav = D.cstate[((long)1 * (T + 1) + t + 1) * B * H + (long)b * H + i];
    r.zi = D.hz[((long)(t + 1) * B + b) * 3 * H + 2 * H + i];
    if (dorow) { r.dz_out = dorow[2 * H + i]; r.dhq = dorow[3 * H + i]; }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float* g = D.gates + ((long)m * T * B + rowt) * 4 * H + i;
      r.gsv[m][0] = g[0]; r.gsv[m][1] = g[H]; r.gsv[m][2] = g[2 * H]; r.gsv[m][3] = g[3 * H];
      if (dorow) r.dh_out[m] = dorow[m * H + i];
      r.cprev[m] = D.cstate[((long)m * (T + 1) + t) * B * H + (long)b * H + i];
      r.carry[m] = D.dc_carry[(long)m * SA + (long)b * H + i];
    }
  }
  return r;
}

// The same operands for the persistent roles, through buffer loads whose per-thread part of the address is ONE register (the unit index;
// gate / stream strides are immediate offsets) and whose per-step part is scalar arithmetic: the 64-bit vector address arithmetic of
// the plain form was ~300 instructions on the two waves that hold the row's units, issued in front of the loads the chain was waiting
// to issue (0.7 us per step).  tau: the natural time position of (t, b), uniform, already in hand (< 0: a padded step, no dout row).
// dc_carry is not fetched (the persistent roles keep it in registers).
template <int HC>
__device__ __forceinline__ RowPre lsthm_bwd_row_prefetch_buf(const CellK& P, const DirP& D, const WS& ws, __amdgpu_buffer_rsrc_t rdout, int t,
                                                             int b, int tau) {
  RowPre r;
  r.cav = r.clv = r.dz_out = r.zi = r.dhq = 0.f;
  r.st = make_float4(1.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < 2; ++m) { r.dh_out[m] = r.cprev[m] = r.carry[m] = 0.f; r.gsv[m][0] = r.gsv[m][1] = r.gsv[m][2] = r.gsv[m][3] = 0.f; }
  const int i = threadIdx.x;
  if (i < HC) {
    const int T = P.T, B = P.B;
    const int vo = i * 4;
    const int rowt = t * B + b;
    auto soff = [&](const float* base, int floats) { return (int)((const char*)base - ws.base) + floats * 4; };
    auto ld = [&](int voff, int so) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ws.r, voff, so, 0)); };
    {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.r, i * 16, soff(D.rstat, rowt * HC * 4), 0);
      r.st = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    r.clv = ld(vo, soff(D.cstate, ((0 * (T + 1) + t + 1) * B + b) * HC));
    r.cav = ld(vo, soff(D.cstate, ((1 * (T + 1) + t + 1) * B + b) * HC));
    r.zi = ld(vo, soff(D.hz, ((t + 1) * B + b) * 3 * HC + 2 * HC));
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int sg = soff(D.gates, (m * T * B + rowt) * 4 * HC);
#pragma unroll
      for (int k = 0; k < 4; ++k) r.gsv[m][k] = ld(vo + k * HC * 4, sg);
      r.cprev[m] = ld(vo, soff(D.cstate, ((m * (T + 1) + t) * B + b) * HC));
    }
    if (tau >= 0) {                // uniform
      const int sd = (tau * B + b) * (int)P.ldo * 4;
      auto ldo_ = [&](int voff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdout, voff, sd, 0)); };
      r.dh_out[0] = ldo_(vo);
      r.dh_out[1] = ldo_(vo + HC * 4);
      r.dz_out = ldo_(vo + 2 * HC * 4);
      r.dhq = ldo_(vo + 3 * HC * 4);
    }
  }
  return r;
}

// The row step in two halves.  part1 needs only saved forward state (softmax statistics of pass 1): in the persistent kernel it
// runs in the SHADOW OF THE BARRIER that publishes the carries of step t+1.  part2 needs those carries (dz, dh).
struct RowMid { float s, u, Z, N2, N3; };

template <int JCT>
__device__ __forceinline__ RowMid lsthm_bwd_row_part1(const CellK& P, const DirP& D, int t, int b, const float* att, float* scr,
                                                      const RowPre& pre) {
  const int H = P.H;
  const int Q = NT / H, JC = JCT ? JCT : H / Q;
  float* ca = scr;           float* cl = ca + H;   float* cw = cl + H;   float* coef = cw + H;
  float* p0 = coef + 8 * H;  float* p1 = p0 + NT;  float* p2 = p1 + NT;  float* sh = p2 + NT;
  const float* wk = att;
  const float* wq = att + H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float rsH = 1.0f / sqrtf((float)H);
  const int i = tid & (H - 1), q = tid / H;
  float sp = 0.f;
  if (tid < H) {
    const float cv = pre.cav, w = wk[tid];
    ca[tid] = cv; cl[tid] = pre.clv; cw[tid] = cv * w;
    sp = wq[tid] * cv;
  }
  sp = wave_sum(sp);
  if (lane == 0) sh[wave] = sp;
  __syncthreads();
  STAMP_ACC(4);
  RowMid r;
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += sh[w];
  s *= rsH;
  r.s = s;
  // ---- pass 1: per output unit i, sums over the keys j of chunk q
  const float u = cl[i] * s;
  r.u = u;
  const float mx = (u >= 0.f) ? u * att[2 * H] : u * att[2 * H + 1];
  const float u2 = u * LOG2E, m2 = mx * LOG2E;
  float Z = 0.f, N2 = 0.f, N3 = 0.f;
  if (drop_attn_on(P, D)) {      // as in the forward: only the value sum N2 takes the dropout factor
    const DropKey dk = s_drop[2];
    const uint32_t e0 = drop_attn_row(P, t, b) + (uint32_t)(i * H + q * JC);
    for (int jj = 0; jj < JC; ++jj) {
      const float wj = wk[q * JC + jj];
      const float e = __builtin_amdgcn_exp2f(fmaf(u2, wj, -m2));
      Z += e;
      N2 = fmaf(e * drop_scale16(dk, e0 + (uint32_t)jj), cw[q * JC + jj], N2);
      N3 = fmaf(e, wj, N3);
    }
  } else {
    const float* wkc = wk + q * JC;
    const float* cwc = cw + q * JC;
#pragma unroll
    for (int jj = 0; jj < (JCT ? JCT : JC); ++jj) {
      const float wj = wkc[jj];
      const float e = __builtin_amdgcn_exp2f(fmaf(u2, wj, -m2));
      Z += e;
      N2 = fmaf(e, cwc[jj], N2);
      N3 = fmaf(e, wj, N3);
    }
  }
  if (q > 0) { p0[tid] = Z; p1[tid] = N2; p2[tid] = N3; }
  __syncthreads();
  STAMP_ACC(5);
  if (q == 0) {
    for (int qq = 1; qq < Q; ++qq) { Z += p0[qq * H + i]; N2 += p1[qq * H + i]; N3 += p2[qq * H + i]; }
  }
  r.Z = Z; r.N2 = N2; r.N3 = N3;
  return r;
}

// Persistent BPTT: the same RowMid from the statistics the forward row phase saved (rstat): no exp2 pass, no reductions.  The
// first pass of the backward used to sit in the shadow of the carry barrier and was LONGER than that barrier (2.3 us against
// 1.4), so it was on the chain; as three loads it is not.
__device__ __forceinline__ RowMid lsthm_bwd_row_part1_saved(const CellK& P, const DirP& D, int t, int b, const float* att, float* scr,
                                                            const RowPre& pre) {
  const int H = P.H;
  float* ca = scr;  float* cl = ca + H;  float* cw = cl + H;
  // (the statistics and c_l come with `pre`, fetched two steps ahead: with self-validating hand-offs nothing hides a load here;
  //  only threads tid < H -- chunk q = 0 -- use the result)
  const int tid = threadIdx.x;
  const float4 st = pre.st;
  const float cli = pre.clv;
  if (tid < H) {
    const float cv = pre.cav;
    ca[tid] = cv; cl[tid] = pre.clv; cw[tid] = cv * att[tid];
  }
  RowMid r;
  r.s = st.w; r.u = cli * st.w; r.Z = st.x; r.N2 = st.y; r.N3 = st.z;
  __syncthreads();
  return r;
}

// carry[2]: the dc carry of this thread's unit, both streams (in: from step t+1, out: for step t-1).  The persistent kernel keeps it in
// registers (the same thread owns the same (row, unit) every step); it is also written to dc_carry for the per-step launches.
// RS = 2 (H = 256, persistent chain): two workgroups share the row.  Both rebuild the coefficients of all H units (element-wise), each
// runs the transposed pass for its 128 keys j0 .. j0 + 127 (thread (tid % 128, tid / 128): H / 4 units per thread instead of H / 2)
// and finishes the gate backward of those units; JCT is then the per-thread unit count of that mapping.
// attreg (persistent roles): the row's dWq / dWk contributions of this thread's unit accumulate in two registers over the whole sequence
// (the same thread owns the same (row, unit) every step) instead of a global read-modify-write per step, whose load sat on the chain.
template <bool PS, int JCT, bool SV = false, int RS = 1>
__device__ __forceinline__ void lsthm_bwd_row_part2(const CellK& P, const DirP& D, const WS& ws, int t, int b, const float* att, float* scr,
                                                    const RowPre& pre, const RowMid& mid, float* carry, int slice = 0, float* attreg = nullptr) {
  const int H = P.H, B = P.B, T = P.T;
  const int JW = H / RS;                          // keys of the transposed pass handled by this workgroup
  const int Q = NT / JW, JC = JCT ? JCT : H / Q;
  float* ca = scr;           float* cl = ca + H;   float* cw = cl + H;   float* coef = cw + H;
  float* p0 = coef + 8 * H;  float* p1 = p0 + NT;  float* p2 = p1 + NT;  float* sh = p2 + NT;
  const float* wk = att;
  const float* wq = att + H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long rowt = (long)t * B + b;
  const bool last = (t == T - 1);
  const float* dA = D.dA + (long)(t + 1 < T ? t + 1 : t) * 8 * B * H + (long)b * H;          // [2][4][B][H]: U_l, V_l, U_a, V_a products of step t+1
  const long SA = (long)B * H;
  const float rsH = 1.0f / sqrtf((float)H);
  const int i = tid & (H - 1), q = tid / H;
  const float s = mid.s, u = mid.u;
  const float mx = (u >= 0.f) ? u * att[2 * H] : u * att[2 * H + 1];
  const float u2 = u * LOG2E, m2 = mx * LOG2E;

  // the only operands that come from other workgroups of this launch: the four carry products of step t+1
  float dz_in = pre.dz_out, zi = pre.zi, dh2[2] = {pre.dh_out[0], pre.dh_out[1]}, dhq = pre.dhq;
  if (q == 0 && !last) {
    float c8[8];
    const int nc = P.ksplit == 2 ? 8 : 4;          // (second K-half of every carry product)
#pragma unroll
    for (int k = 0; k < 8; ++k) c8[k] = k < nc ? ldx<PS>(ws, dA + k * SA + i) : 0.f;
    if (SV) {                  // the carries come from the matvec roles of step t+1: poll until every word is final (see is_sent)
      unsigned spins = 0;
      while (__builtin_amdgcn_ballot_w64(any_sent8(c8)) != 0ull) {
        if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
#pragma unroll
        for (int k = 0; k < 8; ++k) c8[k] = k < nc ? ldx<PS>(ws, dA + k * SA + i) : 0.f;
      }
    }
    dz_in += c8[1] + c8[3] + c8[5] + c8[7];
    dh2[0] += c8[0] + c8[4];
    dh2[1] += c8[2] + c8[6];
  }
  float du_cl = 0.f, dcl_att = 0.f;
  if (q == 0) {
    const float Z = mid.Z, N2 = mid.N2, N3 = mid.N3;
    const float du = dz_in * (N2 - zi * N3) / Z;
    const float a = dz_in / Z;
    float* cf = coef + 8 * i;
    cf[0] = u2; cf[1] = m2; cf[2] = a; cf[3] = a * u; cf[4] = a * u * zi;
    du_cl = du * cl[i];
    dcl_att = du * s;
  }
  du_cl = wave_sum(du_cl);
  __syncthreads();               // pass-1 partials consumed; coef published
  if (lane == 0) sh[wave] = du_cl;
  STAMP_ACC(6);
  // ---- pass 2: per key index j, sums over the units ii of chunk qc (RS = 1: j = i, qc = q)
  const int j0 = slice * JW;
  const int qc = RS == 1 ? q : tid / JW;
  const int j = RS == 1 ? i : j0 + (tid & (JW - 1));
  float S1 = 0.f, S2 = 0.f, S3 = 0.f;
  if (drop_attn_on(P, D)) {      // S1, S2 run over the dropped attention (mask element (i, j), i = qc*JC + ii), S3 over the plain softmax
    const DropKey dk = s_drop[2];
    const uint32_t e0 = drop_attn_row(P, t, b) + (uint32_t)(qc * JC * H + j);
    const float wkj = wk[j];
    const float* cfc = coef + 8 * qc * JC;
    for (int ii = 0; ii < JC; ++ii) {
      const float4 c4 = *reinterpret_cast<const float4*>(cfc + 8 * ii);
      const float c5 = cfc[8 * ii + 4];
      const float e = __builtin_amdgcn_exp2f(fmaf(c4.x, wkj, -c4.y));
      const float ef = e * drop_scale16(dk, e0 + (uint32_t)(ii * H));
      S1 = fmaf(c4.z, ef, S1);
      S2 = fmaf(c4.w, ef, S2);
      S3 = fmaf(c5, e, S3);
    }
  } else {
    const float wkj = wk[j];
    const float* cfc = coef + 8 * qc * JC;
#pragma unroll
    for (int ii = 0; ii < (JCT ? JCT : JC); ++ii) {
      const float4 c4 = *reinterpret_cast<const float4*>(cfc + 8 * ii);
      const float c5 = cfc[8 * ii + 4];
      const float e = __builtin_amdgcn_exp2f(fmaf(c4.x, wkj, -c4.y));
      S1 = fmaf(c4.z, e, S1);
      S2 = fmaf(c4.w, e, S2);
      S3 = fmaf(c5, e, S3);
    }
  }
  if (RS > 1 || q > 0) { p0[tid] = S1; p1[tid] = S2; p2[tid] = S3; }
  __syncthreads();
  STAMP_ACC(7);
  // the unit whose gate backward this thread finishes: RS = 1: j (= i, threads of chunk 0); RS = 2: unit tid of this workgroup's key
  // range (the thread that holds the unit's saved state in `pre`), which collects all Q partials of key tid from LDS
  if (RS == 1 ? q == 0 : (tid >= j0 && tid < j0 + JW)) {
    float ds = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) ds += sh[w];
    if (RS == 1) {
      for (int qq = 1; qq < Q; ++qq) { S1 += p0[qq * H + j]; S2 += p1[qq * H + j]; S3 += p2[qq * H + j]; }
    } else {
      S1 = S2 = S3 = 0.f;
      for (int qq = 0; qq < Q; ++qq) { S1 += p0[qq * JW + tid - j0]; S2 += p1[qq * JW + tid - j0]; S3 += p2[qq * JW + tid - j0]; }
    }
    const int j = i;                         // (RS = 2: i == tid, the key this thread just collected)
    const float dca_att = S1 + ds * wq[j] * rsH;
    if (attreg) {
      attreg[0] += ds * ca[j] * rsH;             // dWq[j]
      attreg[1] += ca[j] * S2 - S3;              // dWk[j]
    } else {
      float* acc = D.attacc + (long)b * 2 * H;
      acc[j] += ds * ca[j] * rsH;
      acc[H + j] += ca[j] * S2 - S3;
    }
    // ---- gate backward, both streams (unit i == j)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float gf = pre.gsv[m][0], gi = pre.gsv[m][1], go = pre.gsv[m][2], gc = pre.gsv[m][3];
      float dh = dh2[m];                                   // gradient at the dropped h (:211 / :213): the same factor again
      if (drop_state_on(P, D)) dh *= drop_h(P, D, t, m, b, i);
      const float cc = m ? pre.cav : pre.clv;
      const float tc = (PS && FASTNL) ? tanh_fast(cc) : tanhf(cc);
      const float dc = carry[m] + dh * go * (1.f - tc * tc) + (m ? dca_att : dcl_att);
      float* dg = D.dgates + ((long)m * T * B + rowt) * 4 * H + i;
      stx<PS>(ws, dg, dc * pre.cprev[m] * gf * (1.f - gf));
      stx<PS>(ws, dg + H, dc * gc * gi * (1.f - gi));
      stx<PS>(ws, dg + 2 * H, dh * tc * go * (1.f - go));
      stx<PS>(ws, dg + 3 * H, dc * gi * (1.f - gc * gc));
      carry[m] = dc * gf;
      if (!attreg) D.dc_carry[(long)m * SA + (long)b * H + i] = dc * gf;       // (persistent roles: the carry lives in `carry`)
    }
    stx<PS>(ws, D.dHQ + rowt * H + i, dhq);
  }
  __syncthreads();     // scratch is reused by the next row / phase
}

template <bool PS, int JCT>
__device__ __forceinline__ void lsthm_bwd_row_body(const CellK& P, const DirP& D, const WS& ws, int t, int b, const float* att, float* scr,
                                                   const RowPre& pre) {
  const RowMid mid = lsthm_bwd_row_part1<JCT>(P, D, t, b, att, scr, pre);
  float carry[2] = {pre.carry[0], pre.carry[1]};
  lsthm_bwd_row_part2<PS, JCT>(P, D, ws, t, b, att, scr, pre, mid, carry);
}

// end of a persistent BPTT role: the attention-vector gradients this thread accumulated for (row, unit tid) join attacc (one owner per
// element: with two workgroups per row each owns the keys of its slice)
__device__ __forceinline__ void lsthm_bwd_flush_att(const CellK& P, const DirP& D, bool has_row, bool rs2, int rowb, int w, const float* attreg) {
  const int H = P.H, tid = threadIdx.x;
  if (!has_row || tid >= H) return;
  if (rs2 && (tid / (H / 2)) != (w & 1)) return;
  float* acc = D.attacc + (long)rowb * 2 * H;
  acc[tid] += attreg[0];
  acc[H + tid] += attreg[1];
}

// Matvec phase, role (product p, output slice n0..n0+31, row block mb):
// dA[p][b][n] = sum_col dgates_m[t][b][col] * Wp[col][n] with p = 0: U_l, 1: V_l, 2: U_a, 3: V_a (m = p>>1);
// p = 4, 5 (pipelined persistent mode only): dHQp[m][t][b][n] = dgates_m[t] @ S_m, the speaker-state gradient that the
// concurrently running speaker BPTT consumes (instead of a hoisted GEMM after the chain).
struct LsthmBwdB {
  const float* Wp; int n0, H;      // H = leading dimension of Wp ([K][H] row-major)
  int ncols = 1 << 30;             // valid output columns (columns >= ncols read as zero)
  __device__ __forceinline__ void operator()(int n, int k, float* bb) const {
    const int col = n0 + n;
    const int cc = col < ncols ? col : 0;       // clamped address, value zeroed afterwards (no load inside a branch)
#pragma unroll
    for (int j = 0; j < 8; ++j) bb[j] = Wp[(long)(k + j) * H + cc];
    if (col >= ncols) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bb[j] = 0.f;
    }
  }
};

// kh / ksplit: this workgroup reduces over gate columns [kh * 4H/ksplit, (kh+1) * 4H/ksplit) and writes partial copy kh.
template <int PS, int NP, bool SV = false>
__device__ __forceinline__ void lsthm_bwd_mat_body(const CellK& P, const DirP& D, const WS& ws, int t, int p, int n0, int mb,
                                                   const float (*bpre)[8], float* red, float* tile, int kh = 0, int ksplit = 1) {
  const int m = p < 4 ? p >> 1 : (p - 4) & 1;
  const int H = P.H, B = P.B, T = P.T;
  const int KH = 4 * H / ksplit, koff = kh * KH;
  const float* Wp = p >= 6 ? D.W[m] : (p >= 4 ? D.S[m] : ((p & 1) ? D.V[m] : D.U[m]));
  const float* dg = D.dgates + ((long)m * T * B + (long)t * B) * 4 * H + koff;
  auto aload = [&](int r, int k, float* a) {
    const int b = mb * 32 + r;
    if (b >= B) { zero8(a); return; }
    load8x<PS>(ws, dg + (long)b * 4 * H + k, a);
  };
  const int ldw = p >= 6 ? P.D : H;
  int taue[1024 / NT];           // dx rows: natural time positions, requested before the product (not a dependent load behind it)
#pragma unroll
  for (int e = 0; e < 1024 / NT; ++e) {
    const int b = mb * 32 + ((threadIdx.x + e * NT) >> 5);
    taue[e] = (p >= 6 && b < B) ? lsthm_tau(D, t, b, B) : -1;
  }
  wg_mm32<NP, (PS == 2 ? 4 : 1)>(KH, aload, LsthmBwdB{Wp + (long)koff * ldw, n0, ldw, ldw}, bpre, red, tile, false, SV ? P.sync + SYNC_ABORT : nullptr);
#pragma unroll
  for (int e = 0; e < 1024 / NT; ++e) {
    const int idx = threadIdx.x + e * NT;
    const int rr = idx >> 5, n = idx & 31;
    const int b = mb * 32 + rr;
    if (b < B) {
      if (p >= 6) {              // dx of this direction at the natural time position (consumed after the launch: plain store)
        const int tau = taue[e];
        if (tau >= 0 && n0 + n < P.D) D.dxc[(((long)kh * 2 + m) * T * B + (long)tau * B + b) * P.D + n0 + n] = tile[idx];
      } else {
        float* dst = p < 4 ? D.dA + ((((long)t * 2 + kh) * 4 + p) * B + b) * H : D.dHQp + ((((long)kh * 2 + m) * T + t) * B + b) * H;
        stx<PS>(ws, dst + n0 + n, tile[idx]);
      }
    }
  }
}

// per-step launches: row grid (B, ndir); mat grid (H/32, 4, ndir*nmb)
__global__ __launch_bounds__(NT) void lsthm_bwd_row(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  float* att = smem + RED_FLOATS;
  att_prepare(P.d[blockIdx.y], P.H, att, smem);
  drop_init(P, P.d[blockIdx.y]);
  lsthm_bwd_row_body<false, 0>(P, P.d[blockIdx.y], ws, t, blockIdx.x, att, smem, lsthm_bwd_row_prefetch(P, P.d[blockIdx.y], t, blockIdx.x));
}
__global__ __launch_bounds__(NT) void lsthm_bwd_mat(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int dir = blockIdx.z / P.nmb, mb = blockIdx.z % P.nmb;
  lsthm_bwd_mat_body<false, 0>(P, P.d[dir], ws, t, blockIdx.y, blockIdx.x * 32, mb, nullptr, smem, smem + RED_FLOATS);
}

// H > 512 (per-step launches only): the backward row phase of one dialogue row on H/128 workgroups, each owning 128 keys j (= the
// units whose gate gradients it writes).  Every workgroup rebuilds the per-unit coefficients of ALL units from the statistics the
// forward saved (rstat: no first exp2 pass) -- H element-wise evaluations against the H x 128 exponentials of its share of the
// transposed pass.  LDS: att [2H+16] | coef [H][8] | partials 3 x [NT] | sh[16].  Same outputs as lsthm_bwd_row_body.
static size_t bwd_row_wide_lds_bytes(int H) { return ((size_t)(2 * H + 16) + 8 * (size_t)H + 3 * NT + 16) * sizeof(float); }
template <int PS>
__device__ __forceinline__ void lsthm_bwd_row_wide_body(const CellK& P, const DirP& D, const WS& ws, int t, int b, int zblk, float* smem) {
  const int H = P.H, B = P.B, T = P.T, j0 = zblk * WIDE_IW;
  float* att = smem;
  float* coef = att + 2 * H + 16;
  float* part = coef + 8 * H;
  float* sh = part + 3 * NT;
  att_prepare(D, H, att, sh);
  drop_init(P, D);
  constexpr int Q = NT / WIDE_IW;
  const int tid = threadIdx.x, jl = tid % WIDE_IW, q = tid / WIDE_IW, j = j0 + jl, IC = H / Q;
  const long rowt = (long)t * B + b, SA = (long)B * H;
  const bool last = (t == T - 1);
  const float* dA = D.dA + (long)(last ? t : t + 1) * 8 * SA + (long)b * H;       // [2][4][B][H]: U_l, V_l, U_a, V_a products of step t+1
  const int tau = D.rev ? D.rev[rowt] : t;
  const float* dorow = (tau >= 0) ? D.dout + ((long)tau * B + b) * P.ldo : nullptr;
  const float* c_l = D.cstate + ((long)0 * (T + 1) + t + 1) * SA + (long)b * H;
  const float* c_a = D.cstate + ((long)1 * (T + 1) + t + 1) * SA + (long)b * H;
  // ---- per-unit coefficients of the whole row
  float ducl = 0.f;
  for (int k = tid; k < H; k += NT) {
    const float4 st = *reinterpret_cast<const float4*>(D.rstat + ((rowt * H) + k) * 4);     // Z, N2, N3, s
    const float cl = c_l[k];
    const float zi = D.hz[((long)(t + 1) * B + b) * 3 * H + 2 * H + k];
    float dz = dorow ? dorow[2 * H + k] : 0.f;
    if (!last) dz += ldx<PS>(ws, dA + 1 * SA + k) + ldx<PS>(ws, dA + 3 * SA + k);
    const float u = cl * st.w;
    const float mx = (u >= 0.f) ? u * att[2 * H] : u * att[2 * H + 1];
    const float du = dz * (st.y - zi * st.z) / st.x;
    const float a = dz / st.x;
    float* cf = coef + 8 * k;
    cf[0] = u * LOG2E; cf[1] = mx * LOG2E; cf[2] = a; cf[3] = a * u; cf[4] = a * u * zi; cf[5] = du;
    ducl = fmaf(du, cl, ducl);
  }
  const float ds = block_sum(ducl, sh);                            // (its barriers publish coef)
  // ---- transposed pass: key j, units of chunk q
  float S1 = 0.f, S2 = 0.f, S3 = 0.f;
  {
    const float wkj = att[j];
    const float* cfc = coef + 8 * q * IC;
    if (drop_attn_on(P, D)) {
      const DropKey dk = s_drop[2];
      const uint32_t e0 = drop_attn_row(P, t, b) + (uint32_t)(q * IC * H + j);
      for (int ii = 0; ii < IC; ++ii) {
        const float4 c4 = *reinterpret_cast<const float4*>(cfc + 8 * ii);
        const float c5 = cfc[8 * ii + 4];
        const float e = __builtin_amdgcn_exp2f(fmaf(c4.x, wkj, -c4.y));
        const float ef = e * drop_scale16(dk, e0 + (uint32_t)(ii * H));
        S1 = fmaf(c4.z, ef, S1);
        S2 = fmaf(c4.w, ef, S2);
        S3 = fmaf(c5, e, S3);
      }
    } else {
#pragma unroll 8
      for (int ii = 0; ii < IC; ++ii) {
        const float4 c4 = *reinterpret_cast<const float4*>(cfc + 8 * ii);
        const float c5 = cfc[8 * ii + 4];
        const float e = __builtin_amdgcn_exp2f(fmaf(c4.x, wkj, -c4.y));
        S1 = fmaf(c4.z, e, S1);
        S2 = fmaf(c4.w, e, S2);
        S3 = fmaf(c5, e, S3);
      }
    }
  }
  part[tid] = S1; part[NT + tid] = S2; part[2 * NT + tid] = S3;
  __syncthreads();
  if (q == 0) {
  for (int qq = 1; qq < Q; ++qq) { S1 += part[qq * WIDE_IW + jl]; S2 += part[NT + qq * WIDE_IW + jl]; S3 += part[2 * NT + qq * WIDE_IW + jl]; }
  const float rsH = 1.0f / sqrtf((float)H);
  const float cav = c_a[j], clv = c_l[j];
  const float s = D.rstat[(rowt * H + j) * 4 + 3];
  const float dca_att = S1 + ds * att[H + j] * rsH;
  const float dcl_att = coef[8 * j + 5] * s;
  float* acc = D.attacc + (long)b * 2 * H;
  acc[j] += ds * cav * rsH;                    // dWq[j]
  acc[H + j] += cav * S2 - S3;                 // dWk[j]
  // ---- gate backward, both streams, unit j
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float* g = D.gates + ((long)m * T * B + rowt) * 4 * H + j;
    const float gf = g[0], gi = g[H], go = g[2 * H], gc = g[3 * H];
    float dh = (dorow ? dorow[m * H + j] : 0.f) + (last ? 0.f : ldx<PS>(ws, dA + (2 * m) * SA + j));
    if (drop_state_on(P, D)) dh *= drop_h(P, D, t, m, b, j);
    const float cc = m ? cav : clv;
    const float tc = tanhf(cc);
    const float cprev = D.cstate[((long)m * (T + 1) + t) * SA + (long)b * H + j];
    float* carry = D.dc_carry + (long)m * SA + (long)b * H + j;
    const float dc = *carry + dh * go * (1.f - tc * tc) + (m ? dca_att : dcl_att);
    float* dg = D.dgates + ((long)m * T * B + rowt) * 4 * H + j;
    stx<PS>(ws, dg, dc * cprev * gf * (1.f - gf));
    stx<PS>(ws, dg + H, dc * gc * gi * (1.f - gi));
    stx<PS>(ws, dg + 2 * H, dh * tc * go * (1.f - go));
    stx<PS>(ws, dg + 3 * H, dc * gi * (1.f - gc * gc));
    *carry = dc * gf;
  }
  D.dHQ[rowt * H + j] = dorow ? dorow[3 * H + j] : 0.f;
  }
}
__global__ __launch_bounds__(NT) void lsthm_bwd_row_wide(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  lsthm_bwd_row_wide_body<false>(P, P.d[blockIdx.y], ws, t, blockIdx.x, blockIdx.z, smem);
}

// persistent launch: grid (nwg, 1, ndir), nwg >= (H/32)*6*nmb.  Row phase: rows round-robin over all nwg workgroups;
// matvec phase: the first (H/32)*6*nmb workgroups (4 carry products + 2 speaker-gradient products).  Two barriers per step;
// the counter doubles as the "dHQ[t] is complete" signal for the concurrently running speaker BPTT (value 2*(T-t)*nwg).
// MSER_OPT_BWD_SENTINEL form of lsthm_bwd_role below: the second seam of the step (carry products dA[t] of the matvec roles -> row
// phase of step t-1) and the hand-off to the speaker BPTT roles (dHQ[t] / dHQp[t]) are self-validating -- written once per launch
// into step-indexed arrays that MSER_PHASE_BWD_PREP filled with the sentinel, polled by their consumers -- while the first seam
// (gate gradients -> matvec roles, 32 KB per consumer) keeps its counter barrier, which the weight-gradient roles also follow.
template <int NP, int KSPLIT = 1>
__device__ __forceinline__ void lsthm_bwd_role_sv(const CellK& P, const Role R, float* smem, const WS& ws) {
  constexpr int NPM = NP / KSPLIT;
  float* red = smem;
  float* tile = smem + RED_FLOATS;
  float* att = tile + 1024;
  const int dir = R.z;
  const DirP& D = P.d[dir];
  const int H = P.H;
  const unsigned nwg = R.gx;
  const int w = R.x;
  if (threadIdx.x == 0) s_poll_abort = 0;
  drop_init(P, D);
  const int nsl = H / 32, nslx = P.nodx ? 0 : (P.D + 31) / 32;
  const int per_kh = 6 * nsl + 2 * nslx;
  const int per_mb = per_kh * KSPLIT;
  const bool has_mat = w < per_mb * P.nmb;
  const int mb = w / per_mb, kh = (w % per_mb) / per_kh, wr = w % per_kh;
  const int p = wr < 6 * nsl ? wr / nsl : 6 + (wr - 6 * nsl) / nslx;
  const int n0 = (wr < 6 * nsl ? wr % nsl : (wr - 6 * nsl) % nslx) * 32;
  float bpre[NPM][8];
  if (has_mat) {
    const int m = p < 4 ? p >> 1 : (p - 4) & 1;
    const float* Wp = p >= 6 ? D.W[m] : (p >= 4 ? D.S[m] : ((p & 1) ? D.V[m] : D.U[m]));
    const int ldw = p >= 6 ? P.D : H;
    preload_b<NPM>(4 * H / KSPLIT, LsthmBwdB{Wp + (long)kh * (4 * H / KSPLIT) * ldw, n0, ldw, ldw}, bpre);
  }
  att_prepare(D, H, att, red);
  int* lds_ok = (int*)(att + att_floats(P.H));
  unsigned nbar = 0;
  unsigned* cnt = P.sync + SYNC_LSTHM_BWD + dir * SYNC_DIR;
  STAMP_INIT();
  constexpr int JCB = 2 * NP * NP;
  const bool rs2 = P.bwd_rowsplit == 2;                       // two workgroups per dialogue row (needs 2 B <= nwg)
  const bool has_row = rs2 ? w < 2 * P.B : w < P.B;
  const int rowb = has_row ? (rs2 ? w >> 1 : w) : 0;
  RowPre pre = lsthm_bwd_row_prefetch(P, D, P.T - 1, rowb);
  RowPre pre_n = lsthm_bwd_row_prefetch(P, D, P.T > 1 ? P.T - 2 : 0, rowb);
  int tau_pp = lsthm_tau(D, P.T > 3 ? P.T - 3 : 0, rowb, P.B);      // natural time position of step t - 2, fetched a step before its use
  const __amdgpu_buffer_rsrc_t rdout = __builtin_amdgcn_make_buffer_rsrc((void*)D.dout, 0, 0x7ffffffc, 0x00020000);
  RowMid mid = lsthm_bwd_row_part1_saved(P, D, P.T - 1, rowb, att, red, pre);
  float carry[2] = {0.f, 0.f};
  float attreg[2] = {0.f, 0.f};
  for (int t = P.T - 1; t >= 0; --t) {
    if (has_row) {
      if (rs2) lsthm_bwd_row_part2<true, JCB / 2, true, 2>(P, D, ws, t, rowb, att, red, pre, mid, carry, w & 1, attreg);
      else lsthm_bwd_row_part2<true, JCB, true>(P, D, ws, t, w, att, red, pre, mid, carry, 0, attreg);
    }
    for (int b = w + (int)nwg; b < (rs2 ? 0 : P.B); b += (int)nwg) {
      const RowPre pr = lsthm_bwd_row_prefetch(P, D, t, b);
      const RowMid md = lsthm_bwd_row_part1<JCB>(P, D, t, b, att, red, pr);
      float cr[2] = {pr.carry[0], pr.carry[1]};
      lsthm_bwd_row_part2<true, JCB, true>(P, D, ws, t, b, att, red, pr, md, cr);
    }
    if (s_poll_abort) return;                    // (read behind part2's closing workgroup barrier: uniform)
    STAMP_ACC(0);
    // Where the saved state of step t-2 is requested matters: the wave's vector-memory operations retire in order, so whatever waits
    // next also waits for these loads (HBM latency).  Right behind the arrive nothing is outstanding (its drain), and the next wait is
    // the matvec phase's for the OTHER workgroups' gate gradients, which is longer than the loads take: hidden.  Requested behind the
    // matvec phase instead they stood in front of the carry poll of the next row phase (measured: 1.2 us per step).
    auto rotate = [&]() {
      if (t == 0) return;
      asm volatile("" : "+v"(tau_pp));           // the table word fetched a step ago is consumed here, where waiting for it is free
      pre = pre_n;
      if (t > 1) pre_n = lsthm_bwd_row_prefetch_buf<32 * NP>(P, D, ws, rdout, t - 2, rowb, __builtin_amdgcn_readfirstlane(tau_pp));
      tau_pp = lsthm_tau(D, t > 3 ? t - 3 : 0, rowb, P.B);
    };
    // seam 1 (gate gradients -> matvec roles) keeps its counter barrier: every matvec workgroup reads a 32 KB slab of dgates[t];
    // polling a payload of that size from 128 workgroups at once costs more fabric traffic than the barrier's bookkeeping saves
    // (measured: 1379 us per launch with both seams self-validating against 1341 with both on the counter)
    nbar += P.ext_spk ? 2u : 1u;
    if (P.bwd_sentinel == 2) {
      // both seams self-validating: the counter only advances (the weight-gradient roles and a linked consumer follow it); the
      // drain of this workgroup's own stores overlaps the wait for everybody else's inside the matvec phase's validated A loads
      barrier_arrive(cnt);
      STAMP_ACC(1);
      rotate();
      // the other workgroups' gate gradients cannot be there yet (they finish their row phases when this one does, and a store takes
      // as long again to become visible): the first look is delayed by about one round trip instead of being issued in vain
      if (P.poll_delay > 0) sleep_n(P.poll_delay);
      if (has_mat && (t > 0 || p >= 4)) lsthm_bwd_mat_body<true, NPM, true>(P, D, ws, t, p, n0, mb, bpre, red, tile, kh, KSPLIT);
      if (s_poll_abort) return;                  // (read behind the product's workgroup barriers: uniform)
    } else {
      if (!dir_barrier(cnt, P.sync + SYNC_ABORT, nwg * (nbar - (P.ext_spk ? 1u : 0u)), lds_ok)) return;
      STAMP_ACC(1);
      rotate();
      if (has_mat && (t > 0 || p >= 4)) lsthm_bwd_mat_body<true, NPM, false>(P, D, ws, t, p, n0, mb, bpre, red, tile, kh, KSPLIT);
    }
    STAMP_ACC(2);
    // seam 2 (carry products -> row phase of step t-1: 8 floats per unit) is self-validating: no arrive, no wait.
    // With an external speaker state a linked consumer of the caller follows the COUNTER ("2 nwg (T - t) arrivals = dHQ[t] and its
    // parts are published").  That reading is only sound if no workgroup adds an arrival of step t-1 before every workgroup has added
    // both of step t -- and nothing else orders the workgroups that form dgates_m S_m (nobody in the chain consumes their tiles):
    // so this seam stays a full counter barrier there (the next step's preparation runs in its shadow).
    if (P.ext_spk) {
      barrier_arrive(cnt);
      if (t > 0) mid = lsthm_bwd_row_part1_saved(P, D, t - 1, rowb, att, red, pre);
      if (!barrier_wait(cnt, P.sync + SYNC_ABORT, nwg * nbar, lds_ok)) return;
    } else if (t > 0) {
      mid = lsthm_bwd_row_part1_saved(P, D, t - 1, rowb, att, red, pre);
    }
    STAMP_ACC(3);
  }
  lsthm_bwd_flush_att(P, D, has_row, rs2, rowb, w, attreg);
  STAMP_DUMP(P, 32, R.x == 1 && R.z == 0);
}

template <int NP, int KSPLIT = 1>
__device__ __forceinline__ void lsthm_bwd_role(const CellK& P, const Role R, float* smem, const WS& ws) {
  if (P.bwd_sentinel) { lsthm_bwd_role_sv<NP, KSPLIT>(P, R, smem, ws); return; }
  constexpr int NPM = NP / KSPLIT;                 // k-passes per wave of the (K-split) matvec products
  float* red = smem;
  float* tile = smem + RED_FLOATS;
  float* att = tile + 1024;
  int* lds_ok = (int*)(att + att_floats(P.H));
  const int dir = R.z;
  const DirP& D = P.d[dir];
  const int H = P.H;
  const unsigned nwg = R.gx;
  const int w = R.x;
  drop_init(P, D);
  // matvec roles: 6 products x H/32 slices (carries + speaker gradient), then 2 products x ceil(D/32) slices (dx = dgates W)
  const int nsl = H / 32, nslx = P.nodx ? 0 : (P.D + 31) / 32;
  const int per_kh = 6 * nsl + 2 * nslx;
  const int per_mb = per_kh * KSPLIT;              // K-halves of one product sit per_kh workgroups apart
  const bool has_mat = w < per_mb * P.nmb;
  const int mb = w / per_mb, kh = (w % per_mb) / per_kh, wr = w % per_kh;
  const int p = wr < 6 * nsl ? wr / nsl : 6 + (wr - 6 * nsl) / nslx;
  const int n0 = (wr < 6 * nsl ? wr % nsl : (wr - 6 * nsl) % nslx) * 32;
  float bpre[NPM][8];
  if (has_mat) {
    const int m = p < 4 ? p >> 1 : (p - 4) & 1;
    const float* Wp = p >= 6 ? D.W[m] : (p >= 4 ? D.S[m] : ((p & 1) ? D.V[m] : D.U[m]));
    const int ldw = p >= 6 ? P.D : H;
    preload_b<NPM>(4 * H / KSPLIT, LsthmBwdB{Wp + (long)kh * (4 * H / KSPLIT) * ldw, n0, ldw, ldw}, bpre);
  }
  att_prepare(D, H, att, red);
  unsigned nbar = 0;
  unsigned* cnt = P.sync + SYNC_LSTHM_BWD + dir * SYNC_DIR;
  STAMP_INIT();
  constexpr int JCB = 2 * NP * NP;                         // NP = H/32, keys per thread = H*H/NT
  const bool rs2 = P.bwd_rowsplit == 2;                       // two workgroups per dialogue row (needs 2 B <= nwg)
  const bool has_row = rs2 ? w < 2 * P.B : w < P.B;
  // saved-state operands are fetched TWO steps ahead (a whole step for the loads to land); the dc carry stays in registers
  const int rowb = has_row ? (rs2 ? w >> 1 : w) : 0;
  RowPre pre = lsthm_bwd_row_prefetch(P, D, P.T - 1, rowb);
  RowPre pre_n = lsthm_bwd_row_prefetch(P, D, P.T > 1 ? P.T - 2 : 0, rowb);
  int tau_pp = lsthm_tau(D, P.T > 3 ? P.T - 3 : 0, rowb, P.B);      // natural time position of step t - 2, fetched a step before its use
  const __amdgpu_buffer_rsrc_t rdout = __builtin_amdgcn_make_buffer_rsrc((void*)D.dout, 0, 0x7ffffffc, 0x00020000);
  RowMid mid = lsthm_bwd_row_part1_saved(P, D, P.T - 1, rowb, att, red, pre);
  float carry[2] = {0.f, 0.f};
  float attreg[2] = {0.f, 0.f};
  for (int t = P.T - 1; t >= 0; --t) {
    if (has_row) {
      if (rs2) lsthm_bwd_row_part2<true, JCB / 2, false, 2>(P, D, ws, t, rowb, att, red, pre, mid, carry, w & 1, attreg);
      else lsthm_bwd_row_part2<true, JCB>(P, D, ws, t, w, att, red, pre, mid, carry, 0, attreg);
    }
    for (int b = w + (int)nwg; b < (rs2 ? 0 : P.B); b += (int)nwg)
      lsthm_bwd_row_body<true, JCB>(P, D, ws, t, b, att, red, lsthm_bwd_row_prefetch(P, D, t, b));
    STAMP_ACC(0);
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, nwg * ++nbar, lds_ok)) return;
    STAMP_ACC(1);
    if (has_mat && (t > 0 || p >= 4)) lsthm_bwd_mat_body<true, NPM>(P, D, ws, t, p, n0, mb, bpre, red, tile, kh, KSPLIT);   // t == 0: no carries needed
    STAMP_ACC(2);
    // split-phase barrier: while the carries are handed over, the half of step t-1's row work that needs only saved forward
    // state (softmax statistics, pass 1) is computed and the saved state of step t-2 is requested
    barrier_arrive(cnt);
    ++nbar;
    if (t > 0) {
      pre = pre_n;
      if (t > 1) pre_n = lsthm_bwd_row_prefetch_buf<32 * NP>(P, D, ws, rdout, t - 2, rowb, __builtin_amdgcn_readfirstlane(tau_pp));
      tau_pp = lsthm_tau(D, t > 3 ? t - 3 : 0, rowb, P.B);
      mid = lsthm_bwd_row_part1_saved(P, D, t - 1, rowb, att, red, pre);
      if (!barrier_wait(cnt, P.sync + SYNC_ABORT, nwg * nbar, lds_ok)) return;
    }
    STAMP_ACC(3);
  }
  lsthm_bwd_flush_att(P, D, has_row, rs2, rowb, w, attreg);
  STAMP_DUMP(P, 32, R.x == 1 && R.z == 0);
}

// ================================================================================================ speaker backward
// Role (product p, output slice n0, slot block mb), one phase per step (descending t).
// product p: cell c = p>>1, (p&1) ? W_hh : W_ih.
//   X_t[b]  := grad wrt q_{t-1}[b, party_t[b]]                         (dialogue-row indexed, ping-pong buffer Xb)
//   dh_q_t[r] = dHQ[t][r] + mnext[t][r] * X_{t+1}[r]                   (blend, :204-207, h_q branch)
//   dh_0_t[r] =            (1 - mnext[t][r]) * X_{t+1}[r]              (blend, h_0 branch: flows straight into q_sel_t)
//   X_t[perm_t[off+k]] = (dgates_t[c][k] @ W_ih)[.] + dh_0_t[off+k]    (gather of q_sel, :242-257)
// Prologue: every workgroup of the cell rebuilds the LSTMCell gate gradients for its 32 slots (element-wise, all loads issued
// up front: one memory round trip), then raw = dsg @ W on the MFMA.  The redundant element-wise results are written to
// global memory by exactly one workgroup per iteration slice (wsel), so no workgroup carries all the stores.
// LDS of one speaker-BPTT workgroup: K-split partials `red` [NW][1024], reduced tile [1024], gate-gradient tile dsg_s [32][4H+4].
// At H = 128 the three lie side by side (103 KB).  At H = 256 that would be 168 KB (> 160 KB): the partials then reuse the storage of
// the gate-gradient tile, whose last read (A operand of the product) is separated from their first write by one barrier.
struct SpkBwdLds { float* red; float* tile; float* dsg_s; int* lds_ok; bool alias; };
__host__ __device__ __forceinline__ size_t spk_bwd_lds_floats(int H) {
  return (H > 128 ? 0 : (size_t)RED_FLOATS) + 1024 + 32 * (4 * (size_t)H + 4);
}
__device__ __forceinline__ SpkBwdLds spk_bwd_lds(float* smem, int H) {
  SpkBwdLds l;
  l.alias = H > 128;
  l.tile = smem + (l.alias ? 0 : RED_FLOATS);
  l.dsg_s = l.tile + 1024;
  l.red = l.alias ? l.dsg_s : smem;
  l.lds_ok = (int*)(l.dsg_s + 32 * (4 * H + 4));
  return l;
}
// MODE 0: the whole step.  H >= 512 (per-step launches only; the gate-gradient tile would not fit the LDS) runs it as two launches:
// MODE 1 = the element-wise prologue alone (publishes dsg[t] and the carried cell gradient), MODE 2 = the products alone, with
// the A operand read back from dsg[t].
template <int PS, int NP, bool WITHP = false, int MODE = 0>
__device__ __forceinline__ void spk_bwd_body(const CellK& P, const DirP& D, const WS& ws, int t, int p, int n0, int mb, int wsel,
                                             const float (*bpre)[8], float* red, float* tile, float* dsg_s) {
  const int c = p >> 1;
  const int H = P.H, B = P.B, T = P.T;
  const int LDS_LD = 4 * H + 4;
  const long SB = (long)B * H;
  const int N0 = D.n0[t];
  const int Nc = c ? B - N0 : N0, off = c ? N0 : 0;
  const bool last = (t == T - 1);
  const int cur = t & 1, nxt = cur ^ 1;       // buffers written at step t / step t+1
  const float* X_n = D.Xb + (long)nxt * SB;
  const float* dhprev_n = D.dhprev + (long)nxt * 2 * SB + (long)c * SB;
  const float* dcprev_n = D.dcprev + (long)nxt * 2 * SB + (long)c * SB;
  float* X_c = D.Xb + (long)cur * SB;
  float* dhprev_c = D.dhprev + (long)cur * 2 * SB + (long)c * SB;
  float* dcprev_c = D.dcprev + (long)cur * 2 * SB + (long)c * SB;
  const int tid = threadIdx.x;
  const float* sg = D.sgates + ((long)c * T + t) * B * 4 * H;
  const float* cq_old = D.cq_state + ((long)c * (T + 1) + t) * SB;
  const float* tcq_t = D.tcq + ((long)c * T + t) * SB;
  float* dsg_g = D.dsg + ((long)c * T + t) * B * 4 * H;
  const float* mn = D.mnext + (long)t * B;

  // operands of the matvec epilogue (W_ih products only): dh_0 of this slice, fetched before anything else
  float dh0e[1024 / NT];
  int be[1024 / NT];
#pragma unroll
  for (int e = 0; e < 1024 / NT; ++e) {
    const int idx = tid + e * NT;
    const int rr = idx >> 5, n = idx & 31;
    const int slot = mb * 32 + rr;
    dh0e[e] = 0.f;
    be[e] = -1;
    if (MODE != 1 && (p & 1) == 0 && slot < Nc) {
      const int r = off + slot;
      be[e] = D.perm[(long)t * B + r];
      if (!last) dh0e[e] = (1.f - mn[r]) * ldx<(PS ? 1 : 0)>(ws, X_n + (long)r * H + n0 + n);
    }
  }

  STAMP_ACC(4);
  // element-wise prologue over 32 slots x H units, 4 consecutive units per thread-iteration (16-byte accesses: the phase is
  // bound by vector-memory instruction throughput, not by arithmetic).  All loads of an iteration batch are issued first.
  constexpr int MAXIT = NP > 0 ? 2 : 4;         // groups in flight per batch (persistent kernels: H = 128 has two per thread in all; H = 256 four, and with its 64 weight registers per lane four in flight would spill)
  const int G4 = H / 4;                          // float4 groups per slot row
  const int npub = 2 * (H / 32);                 // workgroups of this cell; group g is published by workgroup (g % G4) % npub
  // MODE 1 (the prologue as its own launch / phase: nothing stays in LDS) visits ONLY the groups this workgroup publishes -- 32 slots x
  // G4 / npub = 4 column groups -- instead of rebuilding all 32 x G4 of them (at hid = 1024 every one of the cell's 64 workgroups loaded the
  // whole 32 x 4096 tile's operands to store 1/64 of it: 45 of the launch's 50 us)
  const int cgn = G4 / npub;
  const int ngrp = MODE == 1 ? 32 * cgn : 32 * G4;
  const int nit = (ngrp + NT - 1) / NT;          // MODE 0 / 2: = H/64
  auto grp = [&](int e) { return MODE == 1 ? (e / cgn) * G4 + wsel + npub * (e % cgn) : e; };
  auto f4 = [](float4 v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };
  for (int it0 = 0; it0 < (MODE == 2 ? 0 : nit); it0 += MAXIT) {
    float4 v_dh[MAXIT], v_dc[MAXIT], v_x[MAXIT], v_hq[MAXIT], v_g[MAXIT][4], v_tc[MAXIT], v_co[MAXIT];
    float v_m[MAXIT];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ii = 0; ii < MAXIT; ++ii) {
      const int ge = tid + (it0 + ii) * NT;
      const int g = grp(ge < ngrp ? ge : 0);
      const int rr = g / G4, u = (g % G4) * 4;
      const int slot = mb * 32 + rr;
      v_dh[ii] = v_dc[ii] = v_x[ii] = v_hq[ii] = v_tc[ii] = v_co[ii] = z4;
      v_g[ii][0] = v_g[ii][1] = v_g[ii][2] = v_g[ii][3] = z4;
      v_m[ii] = 0.f;
      if (ge < ngrp && slot < B) {
        if (!last) {
          v_dh[ii] = ld4x<(PS ? 1 : 0)>(ws, dhprev_n + (long)slot * H + u);      // (ping-pong buffers: addresses are reused every second
          v_dc[ii] = ld4x<(PS ? 1 : 0)>(ws, dcprev_n + (long)slot * H + u);      //  step, so their loads bypass the caches in mode 2 as well)
        }
        if (slot < Nc) {
          const int r = off + slot;
          if (!last) { v_x[ii] = ld4x<(PS ? 1 : 0)>(ws, X_n + (long)r * H + u); v_m[ii] = mn[r]; }
          if (WITHP) {       // pipelined: dHQ[t] = dout_hq (row phase) + dgates_l S_l + dgates_a S_a (matvec phase), all just published
            const float4 q0 = ld4x<PS>(ws, D.dHQ + ((long)t * B + r) * H + u);
            const float4 q1 = ld4x<PS>(ws, D.dHQp + (((long)0 * T + t) * B + r) * H + u);
            const float4 q2 = ld4x<PS>(ws, D.dHQp + (((long)1 * T + t) * B + r) * H + u);
            v_hq[ii] = make_float4(q0.x + q1.x + q2.x, q0.y + q1.y + q2.y, q0.z + q1.z + q2.z, q0.w + q1.w + q2.w);
            if (P.ksplit == 2) {     // second K-half of the two speaker-gradient products
              const float4 q3 = ld4x<PS>(ws, D.dHQp + (((long)2 * T + t) * B + r) * H + u);
              const float4 q4 = ld4x<PS>(ws, D.dHQp + (((long)3 * T + t) * B + r) * H + u);
              v_hq[ii].x += q3.x + q4.x; v_hq[ii].y += q3.y + q4.y; v_hq[ii].z += q3.z + q4.z; v_hq[ii].w += q3.w + q4.w;
            }
          } else {
            v_hq[ii] = *reinterpret_cast<const float4*>(D.dHQ + ((long)t * B + r) * H + u);
          }
        }
        if (Nc != 0) {
          const float* g0 = sg + (long)slot * 4 * H + u;
#pragma unroll
          for (int k = 0; k < 4; ++k) v_g[ii][k] = *reinterpret_cast<const float4*>(g0 + k * H);
          v_tc[ii] = *reinterpret_cast<const float4*>(tcq_t + (long)slot * H + u);
          v_co[ii] = *reinterpret_cast<const float4*>(cq_old + (long)slot * H + u);
        }
      }
    }
    STAMP_ACC(5);
    if (WITHP && P.bwd_sentinel) {
      // self-validating hand-off from the LSTHM BPTT (no counter in this mode): dHQ[t] and its per-product parts were filled with the
      // sentinel (a NaN) by BWD_PREP, so a sum that is NaN still holds a word that has not been written: re-load until it is not
#pragma unroll
      for (int ii = 0; ii < MAXIT; ++ii) {
        const int ge = tid + (it0 + ii) * NT;
      const int g = grp(ge < ngrp ? ge : 0);
        const int rr = g / G4, u = (g % G4) * 4;
        const int slot = mb * 32 + rr;
        const bool mine = ge < ngrp && slot < B && slot < Nc;
        unsigned spins = 0;
        while (__builtin_amdgcn_ballot_w64(mine && (v_hq[ii].x != v_hq[ii].x || v_hq[ii].y != v_hq[ii].y || v_hq[ii].z != v_hq[ii].z ||
                                                    v_hq[ii].w != v_hq[ii].w)) != 0ull) {
          if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
          if (mine) {
            const int r = off + slot;
            const int np_ = P.ksplit == 2 ? 4 : 2;
            float4 acc4 = ld4x<PS>(ws, D.dHQ + ((long)t * B + r) * H + u);
            for (int k = 0; k < np_; ++k) {
              const float4 qk = ld4x<PS>(ws, D.dHQp + (((long)k * T + t) * B + r) * H + u);
              acc4.x += qk.x; acc4.y += qk.y; acc4.z += qk.z; acc4.w += qk.w;
            }
            v_hq[ii] = acc4;
          }
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < MAXIT; ++ii) {
      if (it0 + ii >= nit) break;
      const int ge = tid + (it0 + ii) * NT;
      if (ge >= ngrp) continue;
      const int g = grp(ge);
      const int rr = g / G4, u = (g % G4) * 4;
      const int slot = mb * 32 + rr;
      const bool wr = ((g % G4) % npub) == wsel;          // this workgroup publishes this group's results
      float d4[4][4];                                     // [gate][j]
      float dcp[4], dhp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float dh = f4(v_dh[ii], j) + f4(v_hq[ii], j) + v_m[ii] * f4(v_x[ii], j);
        const float dc_in = f4(v_dc[ii], j);
        dhp[j] = dh;                                      // skipped cell: no dropout was drawn (:180 / :185)
        if (Nc != 0 && slot < B && drop_state_on(P, D)) dh *= drop_hq(P, D, t, c, slot, u + j);
        if (Nc == 0) {                                    // skipped cell: identity on (h, c)
          dcp[j] = dc_in;
          d4[0][j] = d4[1][j] = d4[2][j] = d4[3][j] = 0.f;
        } else {
          const float gi = f4(v_g[ii][0], j), gf = f4(v_g[ii][1], j), gg = f4(v_g[ii][2], j), go = f4(v_g[ii][3], j);
          const float tc = f4(v_tc[ii], j);               // tanh(c_new), saved by the forward
          const float dcn = dc_in + dh * go * (1.f - tc * tc);
          d4[0][j] = dcn * gg * gi * (1.f - gi);
          d4[1][j] = dcn * f4(v_co[ii], j) * gf * (1.f - gf);
          d4[2][j] = dcn * gi * (1.f - gg * gg);
          d4[3][j] = dh * tc * go * (1.f - go);
          dcp[j] = dcn * gf;
        }
      }
      if (slot < B && wr) {
        st4x<PS>(ws, dcprev_c + (long)slot * H + u, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
        if (Nc == 0) st4x<PS>(ws, dhprev_c + (long)slot * H + u, make_float4(dhp[0], dhp[1], dhp[2], dhp[3]));
        float* o = dsg_g + (long)slot * 4 * H + u;
#pragma unroll
        for (int k = 0; k < 4; ++k) st4x<PS>(ws, o + k * H, make_float4(d4[k][0], d4[k][1], d4[k][2], d4[k][3]));   // read by the wgrad roles
      }
      if (MODE == 0) {
        float* l = dsg_s + rr * LDS_LD + u;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(l + k * H) = make_float4(d4[k][0], d4[k][1], d4[k][2], d4[k][3]);
      }
    }
  }
  STAMP_ACC(6);
  if (Nc == 0 || MODE == 1) return;     // uniform per workgroup
  __syncthreads();
  STAMP_ACC(0);

  const float* Wp = (p & 1) ? D.Whh[c] : D.Wih[c];
  auto aload = [&](int r, int k, float* a) {
    if (MODE == 2) {
      const int slot = mb * 32 + r;
      if (slot < B) load8x<PS>(ws, dsg_g + (long)slot * 4 * H + k, a);
      else zero8(a);
    } else {
      load8(dsg_s + r * LDS_LD + k, a);
    }
  };
  wg_mm32<NP, (PS == 2 ? 4 : 1)>(4 * H, aload, LsthmBwdB{Wp, n0, H}, bpre, red, tile, MODE == 0 && red == dsg_s);
  STAMP_ACC(1);
#pragma unroll
  for (int e = 0; e < 1024 / NT; ++e) {
    const int idx = tid + e * NT;
    const int rr = idx >> 5, n = idx & 31;
    const int slot = mb * 32 + rr;
    if (p & 1) {
      if (slot < B) stx<PS>(ws, dhprev_c + (long)slot * H + n0 + n, tile[idx]);
    } else if (be[e] >= 0) {
      stx<PS>(ws, X_c + (long)be[e] * H + n0 + n, tile[idx] + dh0e[e]);
    }
  }
}

// per-step launch: grid (H/32, 4 products, ndir*nmb), block NT
__global__ __launch_bounds__(NT) void spk_bwd_step(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int dir = blockIdx.z / P.nmb, mb = blockIdx.z % P.nmb;
  const SpkBwdLds l = spk_bwd_lds(smem, P.H);
  drop_init(P, P.d[dir]);
  spk_bwd_body<false, 0>(P, P.d[dir], ws, t, blockIdx.y, blockIdx.x * 32, mb, (int)((blockIdx.y & 1) * gridDim.x + blockIdx.x), nullptr, l.red, l.tile,
                         l.dsg_s);
}

// H >= 512: the step as two launches (see spk_bwd_body), same grid each
__global__ __launch_bounds__(NT) void spk_bwd_pre_wide(CellK P, int t) {
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int dir = blockIdx.z / P.nmb, mb = blockIdx.z % P.nmb;
  drop_init(P, P.d[dir]);
  spk_bwd_body<false, 0, false, 1>(P, P.d[dir], ws, t, blockIdx.y, blockIdx.x * 32, mb, (int)((blockIdx.y & 1) * gridDim.x + blockIdx.x), nullptr, nullptr,
                                   nullptr, nullptr);
}
__global__ __launch_bounds__(NT) void spk_bwd_mat_wide(CellK P, int t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int dir = blockIdx.z / P.nmb, mb = blockIdx.z % P.nmb;
  spk_bwd_body<false, 0, false, 2>(P, P.d[dir], ws, t, blockIdx.y, blockIdx.x * 32, mb, 0, nullptr, smem, smem + RED_FLOATS, nullptr);
}

// persistent launch: same grid, one barrier per step.  Runs concurrently with lsthm_bwd_persist: step t starts once that kernel's
// counter shows dHQ[t] complete (value 2*(T-t)*nwg_l; checked inside the previous step's barrier).
template <int NP>
__device__ __forceinline__ void spk_bwd_role(const CellK& P, const Role R, float* smem, const WS& ws, unsigned nwg_l) {
  const SpkBwdLds l = spk_bwd_lds(smem, P.H);
  float* red = l.red;
  float* tile = l.tile;
  float* dsg_s = l.dsg_s;
  int* lds_ok = l.lds_ok;
  const int dir = R.z / P.nmb, mb = R.z % P.nmb;
  const DirP& D = P.d[dir];
  const int p = R.y, n0 = R.x * 32;
  const unsigned nwg = R.gx * R.gy * P.nmb;
  drop_init(P, D);
  float bpre[NP][8];
  preload_b<NP>(4 * P.H, LsthmBwdB{(p & 1) ? D.Whh[p >> 1] : D.Wih[p >> 1], n0, P.H}, bpre);
  unsigned* cnt = P.sync + SYNC_SPK_BWD + dir * SYNC_DIR;
  const unsigned* lcnt = P.sync + SYNC_LSTHM_BWD + dir * SYNC_DIR;
  unsigned nbar = 0;
  const bool sv = P.bwd_sentinel != 0;          // the LSTHM BPTT publishes no counter: dHQ[t] / dHQp[t] validate themselves (spk_bwd_body)
  if (threadIdx.x == 0) s_poll_abort = 0;
  if (!sv && !dir_barrier(nullptr, P.sync + SYNC_ABORT, 0, lds_ok, lcnt, 2u * nwg_l)) return;          // dHQ[T-1] complete
  STAMP_INIT();
  for (int t = P.T - 1; t >= 0; --t) {
    spk_bwd_body<true, NP, true>(P, D, ws, t, p, n0, mb, (int)((p & 1) * R.gx + R.x), bpre, red, tile, dsg_s);
    STAMP_ACC(2);
    if (t == 0) {              // no consumer inside the chain, but the weight-gradient roles wait for dsg[0]: arrive only
      barrier_arrive(cnt);
      break;
    }
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, nwg * ++nbar, lds_ok, sv ? nullptr : lcnt, 2u * nwg_l * (unsigned)(P.T - t + 1))) return;
    if (s_poll_abort) return;
    STAMP_ACC(3);
  }
  STAMP_DUMP(P, 40, R.x == 0 && R.y == 0 && R.z == 0);
  STAMP_DUMP(P, 48, R.x == 2 && R.y == 1 && R.z == 0);
}

// ---- speaker BPTT as a reduce-scatter (MSER_OPT_SPK_BWD_KSPLIT, persistent launch only) -------------------------------------------
// spk_bwd_role above splits the step's product over its OUTPUT columns: every one of a cell's 2 H/32 workgroups needs the cell's
// whole 32 x 4H gate-gradient tile as its A operand and therefore rebuilds it from 245 KB (H = 128) of write-through loads per step --
// an all-gather, ~1 GB per launch of redundant traffic and the pacing phase of the launch (5.4 of 9.8 us per step).
// Here a workgroup owns SPK_UW = 16 hidden units of one cell for all 32 slots of its block, i.e. a K-slice of 64 gate columns:
//   * everything element-wise about those units is local: the dc carry never leaves its register (thread = (slot, unit) in every
//     step), gates / tanh c / c_prev are this thread's own six saved values (fetched a step ahead);
//   * the product is [32 x 64] . [64 x 2H] = the K-slice's CONTRIBUTION to every column of dgates W_ih and dgates W_hh: one 32 x 32
//     output tile per wave (two at H = 256, sharing the A fragments), K = 64 entirely inside the wave -- no cross-wave reduction;
//   * the H/16 partial tiles of a column meet at the NEXT step's owner of that column, which adds them in a fixed order
//     (deterministic): 16 KB + 16 KB of partials and 6-10 KB of the LSTHM chain's dHQ parts per workgroup and step instead of 245 KB.
//     Layout of a partial array: [producer slice w][column group n/16][row][16], so a consumer reads 2 KB contiguous per producer.
// Still ONE seam per step (the counter barrier of the speaker group, which the weight-gradient roles follow as before).  The terms
// that bypass the product -- dh_0 = (1 - mnext) X_{t+1} into X_t (:204-207, h_0 branch) and the identity of a skipped cell (:180 /
// :185) -- are added by the owner of the column to its own partial (it is the one workgroup that holds them).
constexpr int SPK_UW = 16;
struct SpkKsPre { int N0, b; float mn, g4[4], tc, co; };
// thread (slot = tid / 16, unit = u0 + tid % 16) of step t: index tables and the unit's saved forward state
// (N0 = n0[t] is passed in: the role fetches it one step earlier still, so that no address here waits for a load)
__device__ __forceinline__ SpkKsPre spk_ks_prefetch(const CellK& P, const DirP& D, int t, int N0, int c, int mb, int u0) {
  SpkKsPre r;
  const int H = P.H, B = P.B, T = P.T;
  const long SB = (long)B * H;
  const int slot = mb * 32 + (threadIdx.x >> 4), u = u0 + (threadIdx.x & 15);
  r.N0 = N0;
  const int Nc = c ? B - r.N0 : r.N0, off = c ? r.N0 : 0;
  r.b = -1; r.mn = 0.f; r.tc = 0.f; r.co = 0.f;
  r.g4[0] = r.g4[1] = r.g4[2] = r.g4[3] = 0.f;
  if (slot < Nc) {
    r.b = D.perm[(long)t * B + off + slot];
    r.mn = D.mnext[(long)t * B + off + slot];          // (0 at the last step)
  }
  if (slot < B && Nc != 0) {
    const float* g0 = D.sgates + ((long)c * T + t) * B * 4 * H + (long)slot * 4 * H + u;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.g4[k] = g0[k * H];
    r.tc = D.tcq[((long)c * T + t) * SB + (long)slot * H + u];
    r.co = D.cq_state[((long)c * (T + 1) + t) * SB + (long)slot * H + u];
  }
  return r;
}

// LDS (floats): pa[4][32][16] | px[4][32][16] | dsg_s[32][68] | xs[32][16] | hs[32][16] | rowb[32] (int) | lds_ok
__host__ __device__ __forceinline__ size_t spk_ks_lds_floats() { return 2 * 2048 + 32 * 68 + 2 * 512 + 32 + 16; }

template <int NPS>       // NPS = H / 32
__device__ __forceinline__ void spk_bwd_role_ks(const CellK& P, const Role R, float* smem, const WS& ws, unsigned nwg_l) {
  constexpr int H = 32 * NPS, NSL = H / SPK_UW, NG = H / 16, PW = NSL / 4, NTW = 2 * NPS / NW, LDA = 68;
  static_assert(NTW == 1 || NTW == 2, "speaker BPTT K-split: H = 128 or 256");
  float* pa = smem;
  float* px = pa + 2048;
  float* dsg_s = px + 2048;
  float* xs = dsg_s + 32 * LDA;
  float* hs = xs + 512;
  int* rowb = (int*)(hs + 512);
  int* lds_ok = rowb + 32;
  const int B = P.B, T = P.T;
  const long SB = (long)B * H;
  const int dir = R.z / P.nmb, mb = R.z % P.nmb;
  const DirP& D = P.d[dir];
  const int lin = R.y * R.gx + R.x;                    // 0 .. H/8 - 1: (cell, K-slice)
  const int c = lin / NSL, w = lin % NSL, u0 = w * SPK_UW;
  const unsigned nwg = R.gx * R.gy * P.nmb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  drop_init(P, D);
  // ---- this wave's output tiles: NTW == 1: waves 0..NPS-1 -> W_ih column tiles, NPS..2NPS-1 -> W_hh; NTW == 2: tile 0 = W_ih, 1 = W_hh
  int tkind[NTW], tnt[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    tkind[i] = NTW == 1 ? wave / NPS : i;
    tnt[i] = NTW == 1 ? wave % NPS : wave;
  }
  // B fragments: k = gate p (0..3) x unit (half * 8 + j) of the slice; element W[(p H + u0 + half 8 + j)][32 nt + r]
  float bpre[NTW][4][8];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const float* Wp = tkind[i] ? D.Whh[c] : D.Wih[c];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < 8; ++j) bpre[i][p][j] = Wp[(long)(p * H + u0 + half * 8 + j) * H + 32 * tnt[i] + r];
  }
  unsigned* cnt = P.sync + SYNC_SPK_BWD + dir * SYNC_DIR;
  const unsigned* lcnt = P.sync + SYNC_LSTHM_BWD + dir * SYNC_DIR;
  unsigned nbar = 0;
  const bool sv = P.bwd_sentinel != 0;          // the LSTHM BPTT publishes no counter: dHQ[t] / dHQp[t] validate themselves (see is_sent)
  if (threadIdx.x == 0) s_poll_abort = 0;
  if (!sv && !dir_barrier(nullptr, P.sync + SYNC_ABORT, 0, lds_ok, lcnt, 2u * nwg_l)) return;          // dHQ[T-1] complete
  STAMP_INIT();
  // phase-1 coordinates: thread (g = tid / 128, slot = (tid / 4) % 32, unit quad): g takes the producers w' = g, g + 4, ...
  const int g1 = tid >> 7, sl1 = (tid >> 2) & 31, q4 = (tid & 3) * 4;
  const int slot1 = mb * 32 + sl1;
  // phase-2 coordinates: thread (slot = tid / 16, unit)
  const int sl2 = tid >> 4, uu = tid & 15;
  const int slot2 = mb * 32 + sl2;
  const int nparts = 1 + 2 * P.ksplit;                 // dHQ (output quarter) + the per-product parts dgates_m S_m
  float dc_reg = 0.f;                                  // dc carry of (slot2, u0 + uu): in a register for the whole sequence
  SpkKsPre pre = spk_ks_prefetch(P, D, T - 1, D.n0[T - 1], c, mb, u0);
  int n0_next = D.n0[T > 1 ? T - 2 : 0];               // n0[t - 1], fetched two steps ahead of its use as an address
  for (int t = T - 1; t >= 0; --t) {
    const bool last = (t == T - 1);
    const int cur = t & 1, nxt = cur ^ 1;
    const int N0 = pre.N0;
    const int Nc = c ? B - N0 : N0, off = c ? N0 : 0;
    const float* dhp_n = D.dhp + ((long)nxt * 2 + c) * NSL * SB;
    const float* Xp_n = D.Xp + (long)nxt * NSL * SB;
    float* dhp_c = D.dhp + ((long)cur * 2 + c) * NSL * SB;
    float* Xp_c = D.Xp + (long)cur * NSL * SB;
    // ---- phase 1: this workgroup's 16 columns of the previous step's partial products and of dHQ[t], 16-byte loads, all issued first
    float4 va[PW], vx[PW], vh[2];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < PW; ++k) va[k] = vx[k] = z4;
    vh[0] = vh[1] = z4;
    if (slot1 < B) {
      if (!last) {
#pragma unroll
        for (int k = 0; k < PW; ++k) va[k] = ld4x<true>(ws, dhp_n + (((long)(g1 + 4 * k) * NG + w) * B + slot1) * 16 + q4);
      }
      if (slot1 < Nc) {
        const int r1 = off + slot1;
        if (!last) {
#pragma unroll
          for (int k = 0; k < PW; ++k) vx[k] = ld4x<true>(ws, Xp_n + (((long)(g1 + 4 * k) * NG + w) * B + r1) * 16 + q4);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int a = g1 + 4 * k;
          if (a < nparts) {
            const float* src = a == 0 ? D.dHQ + ((long)t * B + r1) * H : D.dHQp + (((long)(a - 1) * T + t) * B + r1) * H;
            vh[k] = ld4x<true>(ws, src + u0 + q4);
          }
        }
      }
    }
    // the next step's tables and saved state: younger than the loads above, consumed a step later
    const SpkKsPre nxtp = spk_ks_prefetch(P, D, t > 0 ? t - 1 : 0, n0_next, c, mb, u0);
    n0_next = D.n0[t > 1 ? t - 2 : 0];
    if (sv) {                // self-validating hand-off from the LSTHM BPTT: BWD_PREP filled dHQ / dHQp with the sentinel; re-load until final
      const bool mine = slot1 < B && slot1 < Nc;
      auto bad4 = [](float4 v) { return is_sent(v.x) || is_sent(v.y) || is_sent(v.z) || is_sent(v.w); };
      unsigned spins = 0;
      while (__builtin_amdgcn_ballot_w64(mine && (bad4(vh[0]) || bad4(vh[1]))) != 0ull) {
        if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
        if (mine) {
          const int r1 = off + slot1;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int a = g1 + 4 * k;
            if (a < nparts) {
              const float* src = a == 0 ? D.dHQ + ((long)t * B + r1) * H : D.dHQp + (((long)(a - 1) * T + t) * B + r1) * H;
              vh[k] = ld4x<true>(ws, src + u0 + q4);
            }
          }
        }
      }
    }
    STAMP_ACC(4);
    {
      float4 sa = vh[0], sx = z4;
      sa.x += vh[1].x; sa.y += vh[1].y; sa.z += vh[1].z; sa.w += vh[1].w;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        sa.x += va[k].x; sa.y += va[k].y; sa.z += va[k].z; sa.w += va[k].w;
        sx.x += vx[k].x; sx.y += vx[k].y; sx.z += vx[k].z; sx.w += vx[k].w;
      }
      *reinterpret_cast<float4*>(pa + (g1 * 32 + sl1) * 16 + q4) = sa;
      *reinterpret_cast<float4*>(px + (g1 * 32 + sl1) * 16 + q4) = sx;
    }
    __syncthreads();
    STAMP_ACC(5);
    // ---- phase 2: element-wise LSTMCell backward of (slot2, u0 + uu); fixed summation order over the four thread groups
    {
      const int o = sl2 * 16 + uu;
      const float a = ((pa[o] + pa[512 + o]) + pa[1024 + o]) + pa[1536 + o];
      const float x = ((px[o] + px[512 + o]) + px[1024 + o]) + px[1536 + o];
      float dh = a + pre.mn * x;                          // dh_q = dhprev + dHQ + mnext X_{t+1}   (x = 0 beyond Nc and at the last step)
      float d4[4] = {0.f, 0.f, 0.f, 0.f};
      float hfold = 0.f;
      if (Nc == 0) {                                      // skipped cell: identity on (h, c); no dropout was drawn (:180 / :185)
        hfold = dh;
      } else {
        if (slot2 < B && drop_state_on(P, D)) dh *= drop_hq(P, D, t, c, slot2, u0 + uu);
        const float gi = pre.g4[0], gf = pre.g4[1], gg = pre.g4[2], go = pre.g4[3], tc = pre.tc;
        const float dcn = dc_reg + dh * go * (1.f - tc * tc);
        d4[0] = dcn * gg * gi * (1.f - gi);
        d4[1] = dcn * pre.co * gf * (1.f - gf);
        d4[2] = dcn * gi * (1.f - gg * gg);
        d4[3] = dh * tc * go * (1.f - go);
        dc_reg = dcn * gf;
      }
      if (slot2 >= B) { d4[0] = d4[1] = d4[2] = d4[3] = 0.f; hfold = 0.f; }
#pragma unroll
      for (int k = 0; k < 4; ++k) dsg_s[sl2 * LDA + k * 16 + uu] = d4[k];
      xs[o] = (1.f - pre.mn) * x;                         // dh_0 branch of the blend: straight into X_t (rows beyond Nc: x = 0)
      hs[o] = hfold;
      if (uu == 0) rowb[sl2] = pre.b;
      if (slot2 < B) {                                    // dsg[t]: read by the weight-gradient roles behind this step's barrier
        float* og = D.dsg + ((long)c * T + t) * B * 4 * H + (long)slot2 * 4 * H + u0 + uu;
#pragma unroll
        for (int k = 0; k < 4; ++k) stx<true>(ws, og + k * H, d4[k]);
      }
    }
    STAMP_ACC(6);
    __syncthreads();
    STAMP_ACC(0);
    // ---- phase 3: the K-slice's contribution to all columns of dgates W_ih | dgates W_hh
    {
      float a[4][8];
      if (Nc != 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) load8(dsg_s + r * LDA + p * 16 + half * 8, a[p]);
      }
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        f32x16 acc = {0};
        if (Nc != 0) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][j], bpre[i][p][j], acc, 0, 0, 0);
        }
        const int n = 32 * tnt[i] + r;                    // output column of this lane
        const bool mine = (n >> 4) == w;                  // a column of this workgroup's own units: takes the bypass terms
        float* dst = (tkind[i] ? dhp_c : Xp_c) + ((long)w * NG + (n >> 4)) * B * 16 + (n & 15);
        if (tkind[i] == 0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int b = rowb[row];
            float v = acc[e];
            if (mine) v += xs[row * 16 + (n & 15)];
            if (b >= 0) stx<true>(ws, dst + (long)b * 16, v);          // X_t[perm_t[off + slot]] (gather of q_sel, :242-257)
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int slot = mb * 32 + row;
            float v = acc[e];
            if (mine) v += hs[row * 16 + (n & 15)];
            if (slot < B) stx<true>(ws, dst + (long)slot * 16, v);
          }
        }
      }
    }
    STAMP_ACC(1);
    pre = nxtp;
    STAMP_ACC(2);
    if (t == 0) {              // no consumer inside the chain, but the weight-gradient roles wait for dsg[0]: arrive only
      barrier_arrive(cnt);
      break;
    }
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, nwg * ++nbar, lds_ok, sv ? nullptr : lcnt, 2u * nwg_l * (unsigned)(T - t + 1))) return;
    if (s_poll_abort) return;
    STAMP_ACC(3);
  }
  STAMP_DUMP(P, 40, lin == 0 && R.z == 0);
  STAMP_DUMP(P, 48, lin == NSL + 2 && R.z == 0);
}

// ================================================================================================ fused persistent launches
// Both chains of a pass live in ONE launch: blockIdx.x < n_l are the LSTHM-chain workgroups (the critical chain first), the rest
// the speaker-chain workgroups.  They run concurrently as independent groups linked only by the producer's step counter
// (forward: h_q[t]; backward: dHQ[t]).  One launch = one residency guarantee (grid <= CUs, one workgroup per CU through the LDS
// request), independent of how a stream or hipGraph executor would have ordered two separate kernels.
// Softmax statistics of the rank-1 attention rows for the BPTT (rstat: Z, N2 = sum e c_a Wk, N3 = sum e Wk, s), computed by
// extra workgroups of the fused forward launch that follow the LSTHM chain through its step counter: in the row phase of the chain
// the two extra sums cost 0.5-1 us per step of pure critical path; here they cost nothing (one workgroup handles rows
// b = id, id + nsw, ... of a direction, ~1.2 us per row and step against a 7 us step).
template <int JCT>
__device__ __forceinline__ void stats_fwd_role(const CellK& P, int id, int nsw, float* smem, const WS& ws, unsigned nwg_l) {
  const int H = P.H, B = P.B, T = P.T;
  const int dir = id / nsw, w = id % nsw;
  const DirP& D = P.d[dir];
  float* scr = smem;
  float* att = smem + RED_FLOATS;
  int* lds_ok = (int*)(att + att_floats(H));
  if (threadIdx.x == 0) s_poll_abort = 0;
  att_prepare(D, H, att, scr);
  float4* kc = reinterpret_cast<float4*>(scr);
  float* pZ = scr + 4 * H;
  float* pN2 = pZ + NT;
  float* pN3 = pN2 + NT;
  float* sh = pN3 + NT;
  const int Q = NT / H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = tid & (H - 1), q = tid / H;
  const unsigned* cnt = P.sync + SYNC_LSTHM_FWD + dir * SYNC_DIR;
  for (int t = 0; t < T; ++t) {
    // c_l[t], c_a[t] are published behind the barrier that follows the gates phase of step t: barrier 2t + 1 of the chain
    if (!P.fwd_sentinel && !lazy_wait(cnt, P.sync + SYNC_ABORT, nwg_l * (2u * (unsigned)t + 1u), lds_ok)) return;
    for (int b = w; b < B; b += nsw) {
      const float* c_l = D.cstate + ((long)0 * (T + 1) + t + 1) * B * H + (long)b * H;
      const float* c_a = D.cstate + ((long)1 * (T + 1) + t + 1) * B * H + (long)b * H;
      float sp = 0.f;
      float cv = 0.f;
      if (tid < H) cv = ldx<true>(ws, c_a + tid);
      float cli = ldx<true>(ws, c_l + i);
      if (P.fwd_sentinel) {            // the chain publishes no counter in this mode: the rows validate themselves (long sleeps between
        unsigned spins = 0;            // polls: this role is off the chain and must not crowd its requests)
        while (__builtin_amdgcn_ballot_w64(is_sent(cli) || is_sent(cv)) != 0ull) {
          __builtin_amdgcn_s_sleep(32);
          if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
          if (tid < H) cv = ldx<true>(ws, c_a + tid);
          cli = ldx<true>(ws, c_l + i);
        }
      }
      if (tid < H) {
        const float wk = att[tid];
        kc[tid] = make_float4(wk, cv, cv * wk, 0.f);
        sp = att[H + tid] * cv;
      }
      sp = wave_sum(sp);
      if (lane == 0) sh[wave] = sp;
      __syncthreads();
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) s += sh[k];
      s /= sqrtf((float)H);
      const float u = cli * s;
      const float mx = (u >= 0.f) ? u * att[2 * H] : u * att[2 * H + 1];
      const float u2 = u * LOG2E, m2 = mx * LOG2E;
      float Z = 0.f, N2 = 0.f, N3 = 0.f;
      const float4* kcc = kc + q * JCT;
#pragma unroll
      for (int jj = 0; jj < JCT; ++jj) {
        const float4 k4 = kcc[jj];
        const float e = __builtin_amdgcn_exp2f(fmaf(u2, k4.x, -m2));
        Z += e;
        N2 = fmaf(e, k4.z, N2);
        N3 = fmaf(e, k4.x, N3);
      }
      if (q > 0) { pZ[tid] = Z; pN2[tid] = N2; pN3[tid] = N3; }
      __syncthreads();
      if (q == 0) {
        for (int qq = 1; qq < Q; ++qq) { Z += pZ[qq * H + i]; N2 += pN2[qq * H + i]; N3 += pN3[qq * H + i]; }
        *reinterpret_cast<float4*>(D.rstat + (((long)t * B + b) * H + i) * 4) = make_float4(Z, N2, N3, s);
      }
      __syncthreads();
      if (s_poll_abort) return;
    }
  }
}

template <int NPS, int NPL>
__global__ __launch_bounds__(NT) void cell_fwd_fused(CellK P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int gx = P.H / 8, gy = 2;
  const int n_l = gx * gy * P.ndir * P.nmb;
  int id = blockIdx.x;
  if (P.place) {          // every 32-workgroup chain group on TWO XCDs (16 + 16): cheaper barriers, half of every XCD stays free
    id = claim_role(P.sync + SYNC_PLACE_FWD, P.place_base, P.place_cap, (int*)smem);
    if (id < 0) return;
  }
  set_logical_wg((unsigned)id, P.fault);
  if (id >= 2 * n_l) {        // statistics roles (P.stats_wgs > 0): off both chains
    constexpr int JCT = (128 * NPL / 3) * (128 * NPL / 3) / NT;      // = H*H/NT, as in lsthm_fwd_role
    stats_fwd_role<JCT>(P, id - 2 * n_l, P.stats_wgs / P.ndir, smem, ws, (unsigned)(gx * gy * P.nmb));
    return;
  }
  const bool lsthm = id < n_l;
  if (!lsthm) id -= n_l;
  const Role R{id % gx, (id / gx) % gy, id / (gx * gy), gx, gy};
  if (lsthm) {
    if (P.stats_wgs > 0) lsthm_fwd_role<NPL, false>(P, R, smem, ws);
    else lsthm_fwd_role<NPL, true>(P, R, smem, ws);
  } else {
    spk_fwd_role<NPS>(P, R, smem, ws);
  }
}

// The forward chains as two separate launches (MSER_PHASE_SEPARATE_SPEAKER): the speaker chain needs only qmask, so an eager
// caller can start it on a side stream long before the encoders have produced the LSTHM inputs; the LSTHM launch follows it
// through the same step counter.  Only valid when the speaker launch is already executing or queued ahead on another REAL
// stream (not inside a captured graph, whose executor may serialise branches in any order).
template <int NP>
__global__ __launch_bounds__(NT) void spk_fwd_persist(CellK P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  set_logical_wg(blockIdx.x + blockIdx.y + blockIdx.z, P.fault);
  spk_fwd_role<NP>(P, Role{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y}, smem, ws);
}
template <int NP>
__global__ __launch_bounds__(NT) void lsthm_fwd_persist(CellK P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  set_logical_wg(blockIdx.x + blockIdx.y + blockIdx.z, P.fault);
  lsthm_fwd_role<NP>(P, Role{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y}, smem, ws);
}

// ================================================================================================ weight gradients inside the BPTT launch
// dW_m += dgates_m^T x, dS_m += dgates_m^T h_q, dU_m += dgates_m^T h_prev, dV_m += dgates_m^T z_prev and, for the two speaker cells,
// dW_ih += dsg^T q_sel, dW_hh += dsg^T h_q_prev are reductions over all T*B rows: 12.4 GFLOP of exact-fp32 MFMA work that, run
// after the chains, is MFMA-bound (~100-270 us) on a chip that sat 60 % idle for the 1.5 ms of the BPTT.  These roles run beside
// the chains in the same launch (one residency guarantee): every wave owns FOUR 32x32 output tiles for the whole launch
// (accumulators in registers, no split-K, no atomics: one deterministic read-modify-write per element at the end) and follows the
// producing chain through its step counter, two time steps per poll.  The gate gradients are read with the same write-through /
// L1-bypassing hand-off form as inside the chains.
//   LSTHM tiles : (dir, stream m, row tile mt of the 4H gate rows, 16 column tiles = W | S | U | V segments)  -> 4 waves per mt
//   speaker     : (dir, cell c,   row tile mt,                     8 column tiles = W_ih | W_hh)            -> 2 waves per mt
constexpr int WG_CH = 2;        // time steps per poll
struct WgSeg { const float* base; long ld; int width; float* g; long ldg; };

template <bool SPK>
__device__ __forceinline__ void wgrad_role(const CellK& P, int id, const WS& ws, unsigned nwg_src, float* smem) {
  const int H = P.H, B = P.B, T = P.T;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the segment (and its buffer descriptor) is per wave
  int* lds_ok = (int*)smem;
  const int MT = 4 * H / 32;
  // ---- decode (workgroup, wave) -> (dir, m or c, row tile, first of 4 column tiles)
  const int wpm = SPK ? 2 : 4;                    // waves per row tile
  const int mpw = NW / wpm;                       // row tiles per workgroup
  const int unit = id / (MT / mpw);               // (dir, m/c)
  const int dir = unit >> 1, mc = unit & 1;
  const int mt = (id % (MT / mpw)) * mpw + wave / wpm;
  const int nt0 = (wave % wpm) * 4;
  const DirP& D = P.d[dir];
  const long TB = (long)T * B, SB = (long)B * H;
  const float* A = (SPK ? D.dsg : D.dgates) + (long)mc * TB * 4 * H + mt * 32 + r;
  // column segment of tile nt (compile-time unrolled selects: no runtime-indexed local arrays, they would live in scratch)
  const int tps = H / 32;                         // column tiles per segment (the W segment is padded to the same count)
  auto segment = [&](int nt) -> WgSeg {
    const int sg = nt / tps;
    if (SPK) {
      if (sg == 0) return WgSeg{D.qsel + (long)mc * TB * H, H, H, D.gWih[mc], H};
      return WgSeg{D.hq_state + (long)mc * (T + 1) * SB, H, H, D.gWhh[mc], H};
    }
    if (sg == 0) return WgSeg{D.xw[mc], D.ldxw[mc], P.D, D.gW[mc], P.D};
    if (sg == 1) return WgSeg{D.HQ, H, H, D.gS[mc], H};
    if (sg == 2) return WgSeg{D.hz + mc * H, 3L * H, H, D.gU[mc], H};
    return WgSeg{D.hz + 2 * H, 3L * H, H, D.gV[mc], H};
  };
  // the four tiles of a wave lie in ONE segment (tps == 4 and nt0 is a multiple of 4)
  const WgSeg sg0 = segment(nt0);
  // B rows through one buffer descriptor per wave with 32-bit byte offsets (64-bit per-load addresses would push the role past
  // the register budget the chain roles are compiled for)
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)sg0.base, 0, 0x7ffffffc, 0x00020000);
  bool ok[4];
  int coff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = ((nt0 + j) % tps) * 32 + r;
    ok[j] = col < sg0.width;
    coff[j] = (ok[j] ? col : 0) * 4;
  }
  const int ldb4 = (int)sg0.ld * 4;
  const int aoff = (int)((const char*)A - ws.base);
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x16{0};
  // the first wave of every row tile also sums its 32 gate columns over all rows: the bias gradients (a colsum launch less per
  // stream / cell after the chains)
  const bool do_bias = nt0 == 0;
  float csum = 0.f;
  const unsigned* cnt = P.sync + (SPK ? SYNC_SPK_BWD : SYNC_LSTHM_BWD) + dir * SYNC_DIR;
  for (int t_hi = T - 1; t_hi >= 0; t_hi -= WG_CH) {
    const int t_lo = t_hi - WG_CH + 1 > 0 ? t_hi - WG_CH + 1 : 0;
    // dgates[t] is complete behind the row-phase barrier of step t (barrier 2(T-1-t)+1 of the LSTHM chain); dsg[t] behind
    // barrier T-t of the speaker chain
    const unsigned target = SPK ? nwg_src * (unsigned)(T - t_lo) : nwg_src * (2u * (unsigned)(T - 1 - t_lo) + 1u);
    // (MSER_OPT_BWD_SENTINEL: the LSTHM BPTT passes ONE counter barrier per step -- two with an external speaker state -- and
    // dgates[t] is complete behind the first of step t)
    const unsigned per = P.ext_spk ? 2u : 1u;
    const unsigned target_sv = nwg_src * (per * (unsigned)(T - 1 - t_lo) + 1u);
    // MSER_OPT_BWD_SENTINEL = 2: nothing waits at the chain's first seam any more, so the counter is only a HINT here -- a workgroup
    // that owns no dialogue row adds its arrival of a step at once, and one whose product needs only half of the gate columns may
    // even have added the NEXT arrival before a row workgroup has drained its last stores: "count >= target" no longer implies
    // "dgates[t] complete" (found with an external speaker state, two arrivals per step: every weight gradient NaN).  The gate
    // gradients validate themselves instead, as for the chain's own consumers.
    const bool sv = !SPK && P.bwd_sentinel == 2;
    if (!lazy_wait(cnt, P.sync + SYNC_ABORT, (!SPK && P.bwd_sentinel) ? target_sv : target, lds_ok)) return;
#ifdef MSER_WGRAD_EXPERIMENT_SKIP       // diagnostic build only: follow the counters, do no work
    continue;
#endif
    for (int t = t_hi; t >= t_lo; --t) {
      for (int k0 = 0; k0 < B; k0 += 8) {
        // MFMA j of this group consumes rows k0 + 2j + half; rows >= B read row B-1 and are zeroed (no load inside a branch).
        // Groups of 4 MFMAs per tile keep the role under the 128 VGPRs the chain roles are compiled for.
        float a[4], b[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + 2 * j + half;
          const int row = t * B + (k < B ? k : B - 1);
          a[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ws.r, aoff + row * 16 * H, 0, AUX_SC1));
          const int roff = row * ldb4;
#pragma unroll
          for (int q = 0; q < 4; ++q) b[q][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, roff + coff[q], 0, 0));
        }
        if (sv) {                // long sleeps between polls: this role is off the chain and must not crowd its requests
          unsigned spins = 0;
          while (__builtin_amdgcn_ballot_w64(is_sent(a[0]) || is_sent(a[1]) || is_sent(a[2]) || is_sent(a[3])) != 0ull) {
            __builtin_amdgcn_s_sleep(48);
            if (poll_giveup(spins, P.sync + SYNC_ABORT)) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int k = k0 + 2 * j + half;
              const int row = t * B + (k < B ? k : B - 1);
              a[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ws.r, aoff + row * 16 * H, 0, AUX_SC1));
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kok = k0 + 2 * j + half < B;
          const float av = kok ? a[j] : 0.f;
          csum += av;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ok[q] ? b[q][j] : 0.f, acc[q], 0, 0, 0);
        }
      }
    }
  }
  // ---- accumulate into the gradient tensors (each element has exactly one owner in the launch)
  csum += __shfl_xor(csum, 32, 64);              // the two k-halves of the wave hold the same gate column
  if (do_bias && half == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* gb = SPK ? D.gbias_s[mc][q] : D.gbias[mc][q];
      if (gb) gb[mt * 32 + r] += csum;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = ((nt0 + q) % tps) * 32 + r;
    if (!ok[q]) continue;
    float* g = sg0.g + col;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      g[(long)row * sg0.ldg] += acc[q][i];
    }
  }
}

template <int NPL, int NPS, int KSPLIT = 1>
__global__ __launch_bounds__(NT) void cell_bwd_fused(CellK P, unsigned bwd_nwg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WS ws = make_ws(P.wsbase, P.wsbytes);
  const int n_l = (int)bwd_nwg * P.ndir;
  const int gx = P.H / 32, gy = 4;
  const int n_s = P.ext_spk ? 0 : gx * gy * P.nmb * P.ndir;
  int id = blockIdx.x;
  if (P.place) {          // LSTHM BPTT groups on two XCDs each, each speaker group on one, the wgrad roles on the rest
    id = claim_role(P.sync + SYNC_PLACE_BWD, P.place_base, P.place_cap, (int*)smem);
    if (id < 0) return;
  }
  set_logical_wg((unsigned)id, P.fault);
  if (id < n_l) {
    const Role R{id % (int)bwd_nwg, 0, id / (int)bwd_nwg, (int)bwd_nwg, 1};
    lsthm_bwd_role<NPL, KSPLIT>(P, R, smem, ws);
  } else if (id < n_l + n_s) {
    id -= n_l;
    const Role R{id % gx, (id / gx) % gy, id / (gx * gy), gx, gy};
    if (P.spk_ks) spk_bwd_role_ks<NPS>(P, R, smem, ws, bwd_nwg);
    else spk_bwd_role<NPS>(P, R, smem, ws, bwd_nwg);
  } else {
    id -= n_l + n_s;
    // LSTHM weight gradients first ((4H/32)/2 workgroups per (dir, stream)), then the speaker cells ((4H/32)/4 per (dir, cell))
    const int n_wl = P.ndir * 2 * (4 * P.H / 32) / 2;
    if (id < n_wl) wgrad_role<false>(P, id, ws, bwd_nwg, smem);
    else wgrad_role<true>(P, id - n_wl, ws, (unsigned)(gx * gy * P.nmb), smem);
  }
}

// ================================================================================================ wide cells (H > 512): persistent form
// BASELINE configs[4] runs hid = 1024: the recurrent weights (134 MB per phase) cannot be register-resident, so every step streams them
// from the infinity cache / HBM -- round 2 did that with SEVEN launches per time step (speaker step, gates, row phase; BPTT row phase,
// matvec, speaker prologue, speaker products), 1792 launches per training step whose dispatch gaps (~10 us each) were a fifth of the
// step.  These three kernels run the same per-step bodies inside ONE launch per pass: the grid is one workgroup per CU, a phase's
// (virtual) blocks are dealt round-robin to the workgroups, and a phase boundary is a counter barrier among all of them (2 us instead
// of a launch).
// Coherence (accessor mode 2): what a phase hands to the next is STORED write-through and LOADED through the caches.  A step's A rows
// (393 KB at hid = 1024) are read by 512 virtual blocks and must hit the L2: with the narrow chains' L1 / L2-bypassing loads every one
// of those reads went to the memory side (122 ms per step against 95 for the per-step launches), and agent-scope release / acquire
// fences around a plain-access phase cost ~60 us each (they walk the whole L2: 216 ms per step).  Cached loads are safe here because
// every hand-off array is indexed by the time step -- no cache can hold a stale line of an address that nobody has read since the
// launch began -- except the speaker BPTT's three ping-pong buffers, whose loads bypass the caches.  The workspace exceeds the 2 GiB one
// buffer descriptor addresses at this width (5 GB): each direction gets a descriptor over just its hand-off arrays (forward
// qsel .. hz: 1.3 GB at B = 32, L = 256; backward dgates .. dcprev).
struct WideWS { WS f[2], b[2]; };
__device__ __forceinline__ WideWS make_wide_ws(const CellK& P) {
  WideWS w;
  for (int i = 0; i < P.ndir; ++i) {
    const DirP& D = P.d[i];
    w.f[i] = make_ws(D.qsel, (unsigned)((const char*)(D.hz + (long)(P.T + 1) * P.B * 3 * P.H) - (const char*)D.qsel));
    w.b[i] = make_ws(D.dgates, (unsigned)((const char*)(D.dcprev + 2L * 2 * P.B * P.H) - (const char*)D.dgates));
  }
  if (P.ndir == 1) { w.f[1] = w.f[0]; w.b[1] = w.b[0]; }
  return w;
}
static size_t wide_lds_bytes(int H) {
  size_t m = (RED_FLOATS + 1024) * sizeof(float);
  m = std::max(m, z_wide_lds_bytes(H));
  m = std::max(m, bwd_row_wide_lds_bytes(H));
  return m + 64;                  // + the barrier's ok word
}

// The speaker chain and the LSTHM chain stay TWO launches (as in the per-step form): fused into one time loop their weights -- 134 MB
// of speaker-cell matrices and 134 MB of U / V per step -- no longer fit the 256 MB infinity cache together and every step streamed
// from HBM (49.5 ms for the fused forward launch against 19.3 + 18.4 ms for the two loops; scratch/wide_prof.py).
__global__ __launch_bounds__(NT) void cell_wide_spkfwd_persist(CellK P, unsigned lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WideWS W = make_wide_ws(P);
  const int G = gridDim.x, w = blockIdx.x;
  set_logical_wg((unsigned)w, P.fault);
  int* lds_ok = (int*)(smem + lds_floats);
  unsigned* cnt = P.sync + SYNC_SPK_FWD;
  unsigned nbar = 0;
  const int gx = P.H / 8;
  const int Vs = gx * 2 * P.ndir * P.nmb;
  for (int t = 0; t < P.T; ++t) {
    for (int vb = w; vb < Vs; vb += G) {
      const int x = vb % gx, y = (vb / gx) & 1, z = vb / (gx * 2);
      const int dir = z / P.nmb, mb = z % P.nmb;
      drop_init(P, P.d[dir]);
      spk_fwd_body<2, 0>(P, P.d[dir], W.f[dir], t, y, x * 8, mb, x == 0, nullptr, smem, smem + RED_FLOATS);
      __syncthreads();
    }
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, (unsigned)G * ++nbar, lds_ok, nullptr, 0, t + 1 < P.T)) return;
  }
}

// LSTHM forward: gates(t) -> row phase(t) (two barriers per step); S h_q[t] arrives in the hoisted pre-activations, as in the per-step form
__global__ __launch_bounds__(NT) void cell_wide_fwd_persist(CellK P, unsigned lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WideWS W = make_wide_ws(P);
  const int G = gridDim.x, w = blockIdx.x;
  set_logical_wg((unsigned)w, P.fault);
  int* lds_ok = (int*)(smem + lds_floats);
  unsigned* cnt = P.sync + SYNC_LSTHM_FWD;
  unsigned nbar = 0;
  const int gx = P.H / 8, nz = P.H / WIDE_IW;
  const int Vs = gx * 2 * P.ndir * P.nmb, Vz = P.B * P.ndir * nz;
  for (int t = 0; t < P.T; ++t) {
    for (int vb = w; vb < Vs; vb += G) {
      const int x = vb % gx, y = (vb / gx) & 1, z = vb / (gx * 2);
      const int dir = z / P.nmb, mb = z % P.nmb;
      drop_init(P, P.d[dir]);
      lsthm_gates_body<2, 0>(P, P.d[dir], W.f[dir], t, y, x * 8, mb, nullptr, smem, smem + RED_FLOATS);
      __syncthreads();
    }
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, (unsigned)G * ++nbar, lds_ok)) return;
    for (int vb = w; vb < Vz; vb += G) {
      const int b = vb % P.B, rest = vb / P.B;
      const int dir = rest % P.ndir, zb = rest / P.ndir;
      lsthm_fwd_z_wide_body<2>(P, P.d[dir], W.f[dir], t, b, zb, smem);
      __syncthreads();
    }
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, (unsigned)G * ++nbar, lds_ok, nullptr, 0, t + 1 < P.T)) return;
  }
}

// BPTT of the LSTHM streams: row phase(t) -> the four carry products of step t (two barriers per step); dHQ += dgates S and the
// speaker BPTT follow after the launch, as in the per-step form
__global__ __launch_bounds__(NT) void cell_wide_bwd_persist(CellK P, unsigned lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WideWS W = make_wide_ws(P);
  const int G = gridDim.x, w = blockIdx.x;
  set_logical_wg((unsigned)w, P.fault);
  int* lds_ok = (int*)(smem + lds_floats);
  unsigned* cnt = P.sync + SYNC_LSTHM_BWD;
  unsigned nbar = 0;
  const int nz = P.H / WIDE_IW, gm = P.H / 32;
  const int Vr = P.B * P.ndir * nz, Vm = gm * 4 * P.ndir * P.nmb;
  for (int t = P.T - 1; t >= 0; --t) {
    for (int vb = w; vb < Vr; vb += G) {
      const int b = vb % P.B, rest = vb / P.B;
      const int dir = rest % P.ndir, zb = rest / P.ndir;
      lsthm_bwd_row_wide_body<2>(P, P.d[dir], W.b[dir], t, b, zb, smem);
      __syncthreads();
    }
    if (t == 0) break;                       // no carries needed behind the first step
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, (unsigned)G * ++nbar, lds_ok)) return;
    for (int vb = w; vb < Vm; vb += G) {
      const int x = vb % gm, y = (vb / gm) & 3, z = vb / (gm * 4);
      const int dir = z / P.nmb, mb = z % P.nmb;
      lsthm_bwd_mat_body<2, 0>(P, P.d[dir], W.b[dir], t, y, x * 32, mb, nullptr, smem, smem + RED_FLOATS);
      __syncthreads();
    }
    if (!dir_barrier(cnt, P.sync + SYNC_ABORT, (unsigned)G * ++nbar, lds_ok)) return;
  }
}

// speaker BPTT: prologue(t) (publishes the gate-gradient tile and the carried cell gradient) -> products(t) (two barriers per step)
__global__ __launch_bounds__(NT) void cell_wide_spkbwd_persist(CellK P, unsigned lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WideWS W = make_wide_ws(P);
  const int G = gridDim.x, w = blockIdx.x;
  set_logical_wg((unsigned)w, P.fault);
  int* lds_ok = (int*)(smem + lds_floats);
  unsigned* cnt = P.sync + SYNC_SPK_BWD;
  unsigned nbar = 0;
  const int gm = P.H / 32;
  const int Vm = gm * 4 * P.ndir * P.nmb;
  for (int t = P.T - 1; t >= 0; --t) {
    for (int pass = 0; pass < 2; ++pass) {
      for (int vb = w; vb < Vm; vb += G) {
        const int x = vb % gm, y = (vb / gm) & 3, z = vb / (gm * 4);
        const int dir = z / P.nmb, mb = z % P.nmb;
        drop_init(P, P.d[dir]);
        if (pass == 0) spk_bwd_body<2, 0, false, 1>(P, P.d[dir], W.b[dir], t, y, x * 32, mb, (y & 1) * gm + x, nullptr, nullptr, nullptr, nullptr);
        else spk_bwd_body<2, 0, false, 2>(P, P.d[dir], W.b[dir], t, y, x * 32, mb, 0, nullptr, smem, smem + RED_FLOATS, nullptr);
        __syncthreads();
      }
      if (t == 0 && pass == 1) break;
      if (!dir_barrier(cnt, P.sync + SYNC_ABORT, (unsigned)G * ++nbar, lds_ok)) return;
    }
  }
}

// ================================================================================================ small helpers
// mnext[t][r] = qm[t][r][party[t+1][r]]: the blend weight with which row r's h_q feeds the state that dialogue r reads at t+1
// out[((slab * K8 + k8) * 32 + n) * 8 + j] = [Wa | Wb][(n >> 3) * H + slab * 8 + (n & 7)][k8 * 8 + j]   (K = 2 H; both halves [4H, H] row-major)
__global__ void cell_pack_w_kernel(const float* Wa, const float* Wb, int H, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)4 * H * 2 * H;
  if (i >= total) return;
  const int j = (int)(i & 7), n = (int)((i >> 3) & 31);
  const long q = i >> 8;
  const int K8 = 2 * H / 8;
  const int k = (int)(q % K8) * 8 + j, slab = (int)(q / K8);
  const long row = (long)(n >> 3) * H + slab * 8 + (n & 7);
  out[i] = k < H ? Wa[row * H + k] : Wb[row * H + (k - H)];
}

__global__ void mnext_kernel(const float* qm, const int* party, float* mnext, int T, int B) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)T * B) return;
  const long t = i / B;
  mnext[i] = (t + 1 < T) ? qm[i * 2 + party[i + B]] : 0.f;
}

// Every fill of MSER_PHASE_FWD_PREP as ONE launch (a hipGraph replay dispatches about one node per 5 us: ten memset nodes in front of
// the forward delayed everything behind them).  Segments of 32-bit words with a value each (16-byte aligned starts, as Carver gives),
// plus strided row blocks to zero (the reversed direction's output rows).
struct FillArgs {
  unsigned* p[32]; long n[32]; unsigned v[32]; int nseg;
  float* q[2]; long rows[2], ld[2]; int width[2]; int nq;
};
__global__ __launch_bounds__(256) void cell_fill_kernel(FillArgs a) {
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const unsigned v = a.v[sgi];
    const long n4 = a.n[sgi] >> 2;
    u32x4* p4 = reinterpret_cast<u32x4*>(a.p[sgi]);
    const u32x4 v4 = {v, v, v, v};
    for (long i = t0; i < n4; i += stride) p4[i] = v4;
    for (long i = (n4 << 2) + t0; i < a.n[sgi]; i += stride) a.p[sgi][i] = v;
  }
  for (int qi = 0; qi < a.nq; ++qi) {
    const long n = a.rows[qi] * a.width[qi];
    for (long i = t0; i < n; i += stride) a.q[qi][(i / a.width[qi]) * a.ld[qi] + i % a.width[qi]] = 0.f;
  }
}

// The index tables of both directions in ONE launch (grid (T, ndir), one thread per dialogue): party, the stable partition perm and
// its inverse rowof, n0, the qmask rows in direction time order and mnext[t][b] = qmask_t[b][party_{t+1}[b]] -- what
// mser_build_slot_tables + cell_prep_kernel's table half did in two launches per direction.
struct TabArgs { const float* qmask[2]; const int* rev[2]; int *party[2], *perm[2], *rowof[2], *n0[2]; float *qm[2], *mnext[2]; int T, B; };
__global__ __launch_bounds__(1024) void cell_tables_kernel(TabArgs a) {
  __shared__ int cnt0[17];
  const int t = blockIdx.x, dir = blockIdx.y, b = threadIdx.x, T = a.T, B = a.B;
  const int lane = b & 63, wave = b >> 6;
  const float* qmask = a.qmask[dir];
  const int* rev = a.rev[dir];
  int p = 1;          // inactive lanes count as party 1 so they never enter the party-0 ballot
  if (b < B) {
    float m0 = 0.f, m1 = 0.f, n0v = 0.f, n1v = 0.f;
    const int src_t = rev ? rev[(long)t * B + b] : t;
    if (src_t >= 0) { m0 = qmask[((long)src_t * B + b) * 2]; m1 = qmask[((long)src_t * B + b) * 2 + 1]; }
    p = (m1 > m0) ? 1 : 0;
    float mn = 0.f;
    if (t + 1 < T) {
      const int src_n = rev ? rev[(long)(t + 1) * B + b] : t + 1;
      if (src_n >= 0) { n0v = qmask[((long)src_n * B + b) * 2]; n1v = qmask[((long)src_n * B + b) * 2 + 1]; }
      mn = (n1v > n0v) ? m1 : m0;          // this step's mask value at the NEXT step's party
    }
    a.party[dir][(long)t * B + b] = p;
    a.qm[dir][((long)t * B + b) * 2] = m0;
    a.qm[dir][((long)t * B + b) * 2 + 1] = m1;
    a.mnext[dir][(long)t * B + b] = mn;
  }
  const unsigned long long bal = __ballot(p == 0);
  const int before0 = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) cnt0[wave] = __popcll(bal);
  __syncthreads();
  int base0 = 0, total0 = 0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) {
    if (w < wave) base0 += cnt0[w];
    total0 += cnt0[w];
  }
  if (b < B) {
    const int idx0 = base0 + before0;
    const int row = (p == 0) ? idx0 : total0 + (b - idx0);
    a.perm[dir][(long)t * B + row] = b;
    a.rowof[dir][(long)t * B + b] = row;
  }
  if (b == 0) a.n0[dir][t] = total0;
}

__global__ void rowof_kernel(const int* perm, int* rowof, long TB, int B) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TB) return;
  const long t = i / B;
  rowof[t * B + perm[i]] = (int)(i - t * B);
}

// out[tau,b,:D] += X[t,b,:D] with tau = rev[t,b] >= 0   (adjoint of reverse_by_length)
__global__ void reverse_acc_kernel(const float* X, const int* rev, float* out, long ldo, int L, int B, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)L * B * D) return;
  const long r = i / D;
  const int j = (int)(i - r * D);
  const int b = (int)(r % B);
  const int tau = rev[r];
  if (tau >= 0) out[((long)tau * B + b) * ldo + j] += X[i];
}

// out_k[n] += sum_m X[m,n] for up to 4 destinations (the four LSTHM biases share one gradient)
__global__ __launch_bounds__(256) void colsum4_kernel(const float* X, long rows, int n, long ld, float* o0, float* o1, float* o2,
                                                      float* o3) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * 64;
  float s = 0.f;
  if (c < n) {
#pragma unroll 8
    for (long r = r0 + rl; r < min(rows, r0 + 64); r += 4) s += X[r * ld + c];
  }
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < n) {
    const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    if (o0) atomicAdd(&o0[c], v);
    if (o1) atomicAdd(&o1[c], v);
    if (o2) atomicAdd(&o2[c], v);
    if (o3) atomicAdd(&o3[c], v);
  }
}

static int colsum4(const float* X, long rows, int n, long ld, float* o0, float* o1, float* o2, float* o3, hipStream_t s) {
  hipLaunchKernelGGL(colsum4_kernel, dim3(cdiv(n, 64), cdiv(rows, 64)), dim3(256), 0, s, X, rows, n, ld, o0, o1, o2, o3);
  return check_launch("colsum4");
}

// dx_l / dx_a (contiguous [T*B, D]) += up to four contiguous addends each, both modalities in one launch (blockIdx.y):
// the BPTT launch's dx products of the two directions and the caller's partial sums (sequence-level attention branches).
// ext_dhq = dHQ + its per-product partials: the total gradient at an externally supplied speaker state (mser_cell_desc::ext_hq)
__global__ void dhq_total_kernel(float* dst, const float* a, const float* p0, const float* p1, const float* p2, const float* p3, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = a[i];
  if (p0) v += p0[i] + p1[i];
  if (p2) v += p2[i] + p3[i];
  dst[i] = v;
}

struct SumArgs { float* out[2]; const float* src[2][6]; long n; };
template <int VEC>
__global__ void sum_into_kernel(SumArgs a) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (i >= a.n) return;
  const int m = blockIdx.y;
  if (VEC == 4) {
    float4 v = *reinterpret_cast<const float4*>(a.out[m] + i);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (a.src[m][k]) {
        const float4 w = *reinterpret_cast<const float4*>(a.src[m][k] + i);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
    }
    *reinterpret_cast<float4*>(a.out[m] + i) = v;
  } else {
    float v = a.out[m][i];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (a.src[m][k]) v += a.src[m][k][i];
    a.out[m][i] = v;
  }
}

// ================================================================================================ host side
struct Carver {
  char* base; size_t off;
  template <class T> T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

static void carve_dir(Carver& cv, DirP& d, int T, int B, int D, int H) {
  const size_t TB = (size_t)T * B, SB = (size_t)B * H;
  d.party = cv.take<int>(TB); d.perm = cv.take<int>(TB); d.rowof = cv.take<int>(TB); d.n0 = cv.take<int>(T);
  d.qm = cv.take<float>(TB * 2);
  d.qsel = cv.take<float>(2 * TB * H);
  d.hq_state = cv.take<float>(2 * (T + 1) * SB);
  d.cq_state = cv.take<float>(2 * (T + 1) * SB);
  d.sgates = cv.take<float>(2 * TB * 4 * H);
  d.tcq = cv.take<float>(2 * TB * H);
  d.pre = cv.take<float>(2 * TB * 4 * H);
  d.gates = cv.take<float>(2 * TB * 4 * H);
  // HQ | cstate | hz back to back: the arrays the forward chains hand from workgroup to workgroup (ONE sentinel fill per direction)
  d.HQ = cv.take<float>(TB * H);
  d.cstate = cv.take<float>(2 * (T + 1) * SB);
  d.hz = cv.take<float>((T + 1) * (size_t)B * 3 * H);
  d.rstat = cv.take<float>(TB * H * 4);
  d.dgates = cv.take<float>(2 * TB * 4 * H);
  d.dA = cv.take<float>((size_t)T * 2 * 4 * SB);   // [T][2 K-split copies][4 products][B][H]: indexed by the PRODUCING step (written once per
                                                   // launch: the backward chain's self-validating hand-off needs that; 17 MB at c2)
  d.dHQ = cv.take<float>(TB * H);
  d.dHQp = cv.take<float>(2 * 2 * TB * H);
  // zeroed before every backward in ONE memset: [dc_carry | attacc | dxc] back to back (sizes rounded to 64 floats)
  {
    auto r64 = [](size_t n) { return (n + 63) & ~size_t(63); };
    float* z = cv.take<float>(r64(2 * SB) + r64((size_t)B * 2 * H) + 2 * 2 * TB * D);
    d.dc_carry = z;
    d.attacc = z ? z + r64(2 * SB) : nullptr;
    d.dxc = z ? z + r64(2 * SB) + r64((size_t)B * 2 * H) : nullptr;
  }
  d.dsg = cv.take<float>(2 * TB * 4 * H);
  d.Xb = cv.take<float>(2 * SB);
  d.dhprev = cv.take<float>(2 * 2 * SB);
  d.dcprev = cv.take<float>(2 * 2 * SB);
  d.mnext = cv.take<float>(TB);
  // partial products of the K-split speaker BPTT (persistent widths only)
  const size_t nsl = (H == 128 || H == 256) ? (size_t)H / 16 : 0;
  d.Xp = cv.take<float>(2 * nsl * SB);
  d.dhp = cv.take<float>(2 * 2 * nsl * SB);
  // fragment-ordered forward weights of the wide cells
  for (int i = 0; i < 2; ++i) {
    d.WpkS[i] = H > 512 ? cv.take<float>((size_t)4 * H * 2 * H) : nullptr;
    d.WpkL[i] = H > 512 ? cv.take<float>((size_t)4 * H * 2 * H) : nullptr;
  }
}

struct CellHost {
  CellK k;
  unsigned* sync;
  float* xrev[2];     // reversed x_l / x_a for the backward direction [T*B, D]
  float* dxtmp;       // [T*B, D]
};

static size_t carve_all(char* base, const mser_cell_desc& d, CellHost* out) {
  Carver cv{base, 0};
  CellHost h;
  h.k.T = d.T; h.k.B = d.B; h.k.D = d.D; h.k.H = d.H; h.k.ndir = d.ndir; h.k.nmb = cdiv(d.B, 32); h.k.ldo = d.ldo;
  h.k.wgrad_wgs = 0; h.k.spk_ks = 0; h.k.poll_delay = 0; h.k.ksplit = 1; h.k.place = 0; h.k.stats_wgs = 0; h.k.nodx = 0; h.k.fwd_rowsplit = h.k.bwd_rowsplit = 1;
  h.sync = cv.take<unsigned>(SYNC_WORDS);
  h.k.sync = h.sync;
  h.k.wsbase = base;
  h.k.wsbytes = (unsigned)(d.workspace_bytes < 0x7fffffffULL ? d.workspace_bytes : 0x7fffffffULL);
  for (int i = 0; i < d.ndir; ++i) carve_dir(cv, h.k.d[i], d.T, d.B, d.D, d.H);
  const size_t TBD = (size_t)d.T * d.B * d.D;
  h.xrev[0] = cv.take<float>(TBD);
  h.xrev[1] = cv.take<float>(TBD);
  h.dxtmp = cv.take<float>(TBD);
  if (out) *out = h;
  return (cv.off + 255) & ~size_t(255);
}

static int validate(const mser_cell_desc& d, bool bwd) {
  MSER_REQUIRE(d.T > 0 && d.B > 0 && d.D > 0, "marn_cell: bad sizes T=%d B=%d D=%d", d.T, d.B, d.D);
  MSER_REQUIRE(d.H >= 32 && d.H <= 2048 && (d.H & (d.H - 1)) == 0, "marn_cell: H=%d must be a power of two in [32,2048]", d.H);
  MSER_REQUIRE(d.ndir == 1 || d.ndir == 2, "marn_cell: ndir=%d", d.ndir);
  MSER_REQUIRE(d.x_l && d.x_a && d.workspace, "marn_cell: null input/workspace");
  MSER_REQUIRE(((uintptr_t)d.workspace & 255) == 0, "marn_cell: workspace must be 256-byte aligned");
  MSER_REQUIRE(d.workspace_bytes >= mser_marn_cell_workspace_bytes(d.T, d.B, d.D, d.H, d.ndir),
               "marn_cell: workspace too small (%zu < %zu)", d.workspace_bytes,
               mser_marn_cell_workspace_bytes(d.T, d.B, d.D, d.H, d.ndir));
  MSER_REQUIRE(d.ldo >= 4L * d.H, "marn_cell: ldo=%ld < 4H", (long)d.ldo);
  if (d.rng)
    for (int i = 0; i < d.ndir; ++i)
      MSER_REQUIRE(d.p_state[i] >= 0.f && d.p_state[i] < 1.f && d.p_attn[i] >= 0.f && d.p_attn[i] < 1.f,
                   "marn_cell: dropout p out of [0,1) (direction %d: %f, %f)", i, d.p_state[i], d.p_attn[i]);
  // the persistent chains (H = 128, 256) address everything they hand over through ONE buffer descriptor over the workspace
  if (d.H == 128 || d.H == 256)
    MSER_REQUIRE(d.workspace_bytes < 0x7fffffffULL, "marn_cell: workspace of %zu bytes exceeds the 2 GiB addressable through one buffer descriptor", d.workspace_bytes);
  for (int i = 0; i < d.ndir; ++i) {
    const mser_cell_dir& r = d.dir[i];
    MSER_REQUIRE(r.qmask && r.out, "marn_cell: dir %d null qmask/out", i);
    const mser_cell_params& p = r.p;
    for (int m = 0; m < 2; ++m)
      MSER_REQUIRE(p.lsthm_W[m] && p.lsthm_Wb[m] && p.lsthm_U[m] && p.lsthm_Ub[m] && p.lsthm_V[m] && p.lsthm_Vb[m] &&
                       p.lsthm_S[m] && p.lsthm_Sb[m] && (d.ext_hq[0] || (p.q_Wih[m] && p.q_Whh[m] && p.q_bih[m] && p.q_bhh[m])),
                   "marn_cell: dir %d null parameter", i);
    MSER_REQUIRE(p.att_Wq && p.att_Wk, "marn_cell: dir %d null attention vector", i);
    if (bwd) MSER_REQUIRE(r.dout, "marn_cell_bwd: dir %d null dout", i);
  }
  if (bwd) MSER_REQUIRE(d.dx_l && d.dx_a, "marn_cell_bwd: null dx");
  return 0;
}

static void fill_params(DirP& k, const mser_cell_dir& r) {
  for (int m = 0; m < 2; ++m) {
    k.W[m] = r.p.lsthm_W[m]; k.Wb[m] = r.p.lsthm_Wb[m]; k.U[m] = r.p.lsthm_U[m]; k.Ub[m] = r.p.lsthm_Ub[m];
    k.V[m] = r.p.lsthm_V[m]; k.Vb[m] = r.p.lsthm_Vb[m]; k.S[m] = r.p.lsthm_S[m]; k.Sb[m] = r.p.lsthm_Sb[m];
    k.Wih[m] = r.p.q_Wih[m]; k.Whh[m] = r.p.q_Whh[m]; k.bih[m] = r.p.q_bih[m]; k.bhh[m] = r.p.q_bhh[m];
  }
  k.attWq = r.p.att_Wq; k.attWk = r.p.att_Wk;
  k.rev = r.rev; k.out = r.out; k.dout = r.dout;
}
static void fill_dropout(CellK& K, const mser_cell_desc& d) {
  K.rng = d.rng;
  K.fault = d.fault;
  for (int i = 0; i < d.ndir; ++i) {
    K.d[i].drop_site = d.drop_site[i];
    K.d[i].p_state = d.rng ? d.p_state[i] : 0.f;
    K.d[i].p_attn = d.rng ? d.p_attn[i] : 0.f;
  }
}

static mser_gemm_desc gd(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc, int M,
                         int N, int K) {
  mser_gemm_desc g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K;
  g.sAm = sAm; g.sAk = sAk; g.sBk = sBk; g.sBn = sBn; g.ldc = ldc;
  g.batch1 = g.batch2 = 1; g.alpha = 1.f; g.splitk = 1;
  return g;
}

// Raise the dynamic-LDS limit of a kernel (> 64 KiB needs the attribute); remembered per kernel address so that
// steady-state calls (and hipGraph captures) issue no runtime call.
static int allow_lds(const void* kernel, size_t bytes) {
  static const void* seen[16];
  static size_t granted[16];
  static int nseen = 0;
  if (bytes <= 64 * 1024) return 0;
  int i = 0;
  for (; i < nseen; ++i)
    if (seen[i] == kernel) break;
  if (i < nseen && granted[i] >= bytes) return 0;
  MSER_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (i == nseen && nseen < 16) { seen[nseen] = kernel; ++nseen; }
  if (i < 16) granted[i] = bytes;
  return 0;
}

static size_t row_lds_bytes(int H) { return ((size_t)RED_FLOATS + 2 * (size_t)H + 16) * sizeof(float); }

// ---- launch mode ---------------------------------------------------------------------------------------------------------
static int g_opt_persistent = 1;      // MSER_OPT_PERSISTENT
static int g_opt_wgrad_inkernel = 1;  // MSER_OPT_WGRAD_INKERNEL
static int g_opt_ksplit = 1;          // MSER_OPT_BPTT_KSPLIT
static int g_opt_stats_roles = 1;     // MSER_OPT_FWD_STATS_ROLES
static int g_opt_xcd_place = 0;       // MSER_OPT_XCD_PLACEMENT (off: measured slower end to end)
static int g_opt_fwd_sentinel = 1;    // MSER_OPT_FWD_SENTINEL
static int g_opt_rowsplit = 1;        // MSER_OPT_H256_SPLIT: H = 256 persistent chains share a row phase between two workgroups, BPTT products K-split
static int g_opt_spk_ks = 1;          // MSER_OPT_SPK_BWD_KSPLIT
static int g_opt_poll_delay = 0;      // MSER_OPT_BWD_POLL_DELAY
extern int g_opt_drnn_persist;         // dialogue.hip
static int g_opt_wide_persist = 0;    // MSER_OPT_WIDE_PERSISTENT (off: the four launches take 75 ms against 81 ms of per-step kernel time at the configs[4]
                                      // shard, but they hold every CU, so the weight-gradient GEMMs and attention branches that the per-step
                                      // launches overlap are serialised behind them: 102-135 ms per step against 90.6)
static int g_opt_bwd_sentinel = 2;    // MSER_OPT_BWD_SENTINEL (2: both seams of the LSTHM BPTT self-validating; round 2 measured no gain -- the speaker roles
                                      // paced the launch then; with them out of the way: 1321 -> 1225 (seam 2) -> 1166 us (both) per launch)
static int g_num_cus = 0;
constexpr size_t PERSIST_MIN_LDS = 84 * 1024;     // > half of the 160 KiB LDS: at most ONE persistent workgroup per CU

static int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) g_num_cus = n;
    if (g_num_cus <= 0) g_num_cus = 1;
  }
  return g_num_cus;
}
// A persistent launch needs every workgroup co-resident (one per CU) and its weight slice in registers (H = 128 or 256).
static bool persist_ok(int H, long total_wgs) {
  return g_opt_persistent && (H == 128 || H == 256) && total_wgs <= num_cus();
}
static size_t persist_lds(size_t need) { return need > PERSIST_MIN_LDS ? need : PERSIST_MIN_LDS; }
// H > 512: the per-step bodies inside one launch per pass (cell_wide_*_persist): one workgroup per CU, no external speaker state
static bool wide_persist_ok(int H, bool ext) { return g_opt_persistent && g_opt_wide_persist && H > 512 && !ext && num_cus() >= 64; }

// phases: MSER_PHASE_SPEAKER_FWD builds the tables, zeroes the initial states and runs the speaker chain (needs only qmask / rev:
// it can overlap the encoders on another stream); MSER_PHASE_LSTHM_FWD runs the pre-activation GEMMs and the LSTHM chain.
int marn_cell_fwd(const mser_cell_desc& d, hipStream_t s, int phases) {
  MSER_TRY(validate(d, false));
  CellHost h;
  carve_all((char*)d.workspace, d, &h);
  CellK& K = h.k;
  const int T = d.T, B = d.B, D = d.D, H = d.H;
  const long TB = (long)T * B, SB = (long)B * H;
  for (int i = 0; i < d.ndir; ++i) fill_params(K.d[i], d.dir[i]);
  fill_dropout(K, d);
  // External speaker state (the GRU-speaker variants): h_q[t] = ext_hq rows, complete before this call.  The LSTHM
  // chain then runs as its own persistent launch (the form MSER_PHASE_SEPARATE_SPEAKER uses) with the speaker's step counter preset.
  // (The rows are COPIED into the workspace's HQ array: the chains address everything they share through one buffer descriptor
  // over the workspace, an outside pointer would read as zeros, and the backward wants them saved there anyway.)
  const bool ext = d.ext_hq[0] != nullptr;
  K.ext_spk = ext ? 1 : 0;
  const size_t mm_lds = (RED_FLOATS + 1024) * sizeof(float);
  const long fwd_wgs = (long)(H / 8) * 2 * d.ndir * K.nmb;
  const bool persist = persist_ok(H, ext ? fwd_wgs : 2 * fwd_wgs);        // speaker and LSTHM chains share one launch: all workgroups co-resident
  const size_t p_lds = persist_lds(mm_lds + (2 * (size_t)H + 16) * sizeof(float) + 64);
  // (an external speaker state keeps the counters: its producer publishes through the step counter anyway, and polling the rows of a
  // linked producer measured slower -- MARN1_onlysp 4.16 / 4.25 ms per step self-validating against 4.06 / 4.16 on the counters)
  K.fwd_sentinel = (persist && g_opt_fwd_sentinel && !ext) ? 1 : 0;
  K.fwd_rowsplit = (persist && g_opt_rowsplit && H == 256 && 2 * B <= (H / 8) * 2 * K.nmb) ? 2 : 1;
  if (phases & MSER_PHASE_FWD_PREP) {
  if (H > 512) {      // the wide cells stream their weights every step: fragment order for the forward launches (once per call)
    const long total = (long)4 * H * 2 * H;
    for (int i = 0; i < d.ndir; ++i)
      for (int c = 0; c < 2; ++c) {
        hipLaunchKernelGGL(cell_pack_w_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, K.d[i].Wih[c], K.d[i].Whh[c], H, K.d[i].WpkS[c]);
        hipLaunchKernelGGL(cell_pack_w_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, K.d[i].U[c], K.d[i].V[c], H, K.d[i].WpkL[c]);
      }
    MSER_TRY(check_launch("cell_pack_w"));
  }
  {
    // counters zeroed; every word the forward chains hand from workgroup to workgroup starts as the sentinel (HQ | cstate | hz, carved
    // back to back; cell_prep_kernel, below, then zeroes the index-0 states; an external speaker state is copied over HQ by
    // MSER_PHASE_LSTHM_FWD, a linked producer writes its rows while the chain polls them); MSER_PHASE_PREP_BOTH: what
    // MSER_PHASE_BWD_PREP would do, now (off the critical path between the head's backward and the BPTT launch) -- the carries and
    // accumulators over their whole allocated extent, the BPTT counters being part of the sync words; rows at and beyond len_b stay
    // zero in the reversed direction's output (pad_sequence, :410)
    FillArgs fa;
    memset(&fa, 0, sizeof fa);
    auto seg = [&](void* p, size_t words, unsigned v) { fa.p[fa.nseg] = (unsigned*)p; fa.n[fa.nseg] = (long)words; fa.v[fa.nseg] = v; ++fa.nseg; };
    seg(h.sync, SYNC_WORDS, 0u);
    for (int i = 0; i < d.ndir; ++i) {
      DirP& k = K.d[i];
      // index 0 of the (T+1)-long state arrays = the zero initial states; everything else of HQ | cstate | hz the sentinel
      const size_t SBw = (size_t)SB, TSB = (size_t)(T + 1) * SB;
      for (int c2 = 0; c2 < 2; ++c2) { seg(k.hq_state + c2 * TSB, SBw, 0u); seg(k.cq_state + c2 * TSB, SBw, 0u); seg(k.cstate + c2 * TSB, SBw, 0u); }
      seg(k.hz, (size_t)B * 3 * H, 0u);
      if (K.fwd_sentinel) {
        seg(k.HQ, (size_t)((char*)k.cstate - (char*)k.HQ) / 4, SENT_BITS);
        seg(k.cstate + SBw, TSB - SBw, SENT_BITS);
        seg(k.cstate + TSB + SBw, (size_t)((char*)k.hz - (char*)(k.cstate + TSB + SBw)) / 4, SENT_BITS);
        seg(k.hz + (size_t)B * 3 * H, (size_t)T * B * 3 * H, SENT_BITS);
      }
      if (phases & MSER_PHASE_PREP_BOTH) {
        seg(k.dc_carry, (size_t)((char*)(k.dxc + (size_t)2 * 2 * TB * D) - (char*)k.dc_carry) / 4, 0u);
        if (g_opt_bwd_sentinel) seg(k.dgates, (size_t)((char*)(k.dHQp + 2 * 2 * TB * H) - (char*)k.dgates) / 4, SENT_BITS);
      }
      if (k.rev) { fa.q[fa.nq] = k.out; fa.rows[fa.nq] = TB; fa.ld[fa.nq] = d.ldo; fa.width[fa.nq] = 4 * H; ++fa.nq; }
    }
    hipLaunchKernelGGL(cell_fill_kernel, dim3(2048), dim3(256), 0, s, fa);
    MSER_TRY(check_launch("cell_fill"));
  }
  if (ext && !d.ext_linked)      // "every h_q[t] is published": the LSTHM chain's waits on the speaker counter fall through
    MSER_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(h.sync + SYNC_SPK_FWD), 0x3fffffff, 2 * SYNC_DIR, s));
  {
    MSER_REQUIRE(B <= 1024, "marn_cell: B=%d > 1024 dialogues per call", B);
    TabArgs ta;
    memset(&ta, 0, sizeof ta);
    ta.T = T; ta.B = B;
    for (int i = 0; i < d.ndir; ++i) {
      DirP& k = K.d[i];
      ta.qmask[i] = d.dir[i].qmask; ta.rev[i] = k.rev; ta.party[i] = k.party; ta.perm[i] = k.perm; ta.rowof[i] = k.rowof;
      ta.n0[i] = k.n0; ta.qm[i] = k.qm; ta.mnext[i] = k.mnext;
    }
    hipLaunchKernelGGL(cell_tables_kernel, dim3(T, d.ndir), dim3(cdiv(B, 64) * 64), 0, s, ta);
    MSER_TRY(check_launch("cell_tables"));
  }
  }
  const bool separate = persist && ((phases & MSER_PHASE_SEPARATE_SPEAKER) || ext);
  // XCD placement of the fused launch: needs 32-workgroup chain groups (H = 128, B <= 32) and a launch over every CU
  K.place = (persist && !separate && g_opt_xcd_place && (H / 8) * 2 * K.nmb == 32 && num_cus() == 256) ? 1 : 0;
  if (K.place) {          // groups (LSTHM d0, [LSTHM d1,] speaker d0 [, speaker d1]) = logical ids 32 g .. 32 g + 31 -> XCDs 2g, 2g + 1
    for (int x = 0; x < 8; ++x) {
      const int g = x / 2;
      K.place_cap[x] = (short)(g < 2 * d.ndir ? 16 : 0);
      K.place_base[x] = (short)(g * 32 + (x & 1) * 16);
    }
  }
  if ((phases & MSER_PHASE_SPEAKER_FWD) && separate && !ext) {
    ProfScope ps(MSER_PROF_SPK_FWD, s);
    if (H == 128) {
      MSER_TRY(allow_lds((const void*)spk_fwd_persist<2>, p_lds));
      hipLaunchKernelGGL(spk_fwd_persist<2>, dim3(H / 8, 2, d.ndir * K.nmb), dim3(NT), p_lds, s, K);
    } else {
      MSER_TRY(allow_lds((const void*)spk_fwd_persist<4>, p_lds));
      hipLaunchKernelGGL(spk_fwd_persist<4>, dim3(H / 8, 2, d.ndir * K.nmb), dim3(NT), p_lds, s, K);
    }
    MSER_TRY(check_launch("spk_fwd_persist"));
  }
  const bool wide = wide_persist_ok(H, ext);
  if ((phases & MSER_PHASE_SPEAKER_FWD) && wide) {      // H > 512: the whole speaker chain as one launch (one barrier per step)
    const size_t wl = persist_lds(wide_lds_bytes(H));
    MSER_TRY(allow_lds((const void*)cell_wide_spkfwd_persist, wl));
    ProfScope ps(MSER_PROF_SPK_FWD, s);
    hipLaunchKernelGGL(cell_wide_spkfwd_persist, dim3(num_cus()), dim3(NT), wl, s, K, (unsigned)((wl - 64) / sizeof(float)));
    MSER_TRY(check_launch("cell_wide_spkfwd_persist"));
  }
  if ((phases & MSER_PHASE_SPEAKER_FWD) && !persist && !ext && !wide) {
    // ---- speaker chain as per-step launches (the persistent mode runs it inside the fused launch of the LSTHM phase)
    MSER_TRY(allow_lds((const void*)spk_fwd_step, mm_lds));
    for (int t = 0; t < T; ++t) {
      ProfScope ps(MSER_PROF_SPK_FWD, s);
      hipLaunchKernelGGL(spk_fwd_step, dim3(H / 8, 2, d.ndir * K.nmb), dim3(NT), mm_lds, s, K, t);
    }
    MSER_TRY(check_launch("spk_fwd"));
  }
  // ---- hoisted pre-activations: pre_m = xdir W_m^T + W.bias (+ HQ S_m^T + S.bias, below, when the chain does not add it itself).  The
  // x W^T products of both streams and directions are independent: ONE grouped launch (they sit on the critical path right in front
  // of the chain).  MSER_PHASE_LSTHM_PRE runs them alone (they need only x_l / x_a: the caller can overlap MSER_PHASE_FWD_PREP on
  // another stream), MSER_PHASE_LSTHM_FWD | MSER_PHASE_PRE_DONE then skips them.
  if ((phases & (MSER_PHASE_LSTHM_PRE | MSER_PHASE_LSTHM_PRE_L | MSER_PHASE_LSTHM_PRE_A)) ||
      ((phases & MSER_PHASE_LSTHM_FWD) && !(phases & MSER_PHASE_PRE_DONE))) {
    // which streams: PRE_L / PRE_A = the products over x_l / x_a alone (each needs only its own encoder branch), otherwise both
    const bool only = (phases & (MSER_PHASE_LSTHM_PRE_L | MSER_PHASE_LSTHM_PRE_A)) && !(phases & (MSER_PHASE_LSTHM_PRE | MSER_PHASE_LSTHM_FWD));
    const bool want[2] = {!only || (phases & MSER_PHASE_LSTHM_PRE_L) != 0, !only || (phases & MSER_PHASE_LSTHM_PRE_A) != 0};
    std::vector<mser_gemm_desc> pg;
    for (int i = 0; i < d.ndir; ++i) {
      DirP& k = K.d[i];
      const float* xs[2] = {d.x_l, d.x_a};
      long lds[2] = {d.ldxl, d.ldxa};
      if (k.rev) {
        if (want[0]) MSER_TRY(mser_reverse_by_length(d.x_l, d.ldxl, k.rev, h.xrev[0], D, T, B, D, s));
        if (want[1]) MSER_TRY(mser_reverse_by_length(d.x_a, d.ldxa, k.rev, h.xrev[1], D, T, B, D, s));
        xs[0] = h.xrev[0]; xs[1] = h.xrev[1]; lds[0] = lds[1] = D;
      }
      for (int m = 0; m < 2; ++m) {
        if (!want[m]) continue;
        float* pre = k.pre + (long)m * TB * 4 * H;
        mser_gemm_desc g = gd(xs[m], lds[m], 1, k.W[m], 1, D, pre, 4 * H, (int)TB, 4 * H, D);
        g.bias = k.Wb[m];
        pg.push_back(g);
      }
    }
    MSER_TRY(gemm_group(pg.data(), (int)pg.size(), s));
  }
  if (!(phases & MSER_PHASE_LSTHM_FWD)) return 0;
  if (ext && d.ext_linked) {
    // the producer (a concurrently running kernel of the caller, on another stream) writes the rows straight into the workspace and
    // bumps the speaker counter per step (mser_marn_cell_ext_link): the LSTHM chain follows it through its ordinary per-step waits
    MSER_REQUIRE(persist, "marn_cell_fwd: ext_linked needs the persistent launch (all LSTHM workgroups co-resident)");
  } else if (ext) {
    for (int i = 0; i < d.ndir; ++i)
      MSER_CHECK_HIP(hipMemcpyAsync(K.d[i].HQ, d.ext_hq[i], (size_t)TB * H * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  if (!persist) {      // the pipelined persistent kernel adds S h_q[t] (+ S.bias) inside the step instead
    for (int i = 0; i < d.ndir; ++i) {
      DirP& k = K.d[i];
      for (int m = 0; m < 2; ++m) {
        float* pre = k.pre + (long)m * TB * 4 * H;
        mser_gemm_desc g = gd(k.HQ, H, 1, k.S[m], 1, H, pre, 4 * H, (int)TB, 4 * H, H);
        g.bias = k.Sb[m];
        g.flags = MSER_GEMM_ACCUM;
        MSER_TRY(gemm(g, s));
      }
    }
  }
  // ---- LSTHM chain
  const size_t z_lds = row_lds_bytes(H);
  if (separate) {
    ProfScope ps(MSER_PROF_LSTHM_FWD_GATES, s);
    if (H == 128) {
      MSER_TRY(allow_lds((const void*)lsthm_fwd_persist<3>, p_lds));
      hipLaunchKernelGGL(lsthm_fwd_persist<3>, dim3(H / 8, 2, d.ndir * K.nmb), dim3(NT), p_lds, s, K);
    } else {
      MSER_TRY(allow_lds((const void*)lsthm_fwd_persist<6>, p_lds));
      hipLaunchKernelGGL(lsthm_fwd_persist<6>, dim3(H / 8, 2, d.ndir * K.nmb), dim3(NT), p_lds, s, K);
    }
  } else if (persist) {
    // ONE launch for both chains: 2 x fwd_wgs workgroups (LSTHM roles first), linked by the speaker's step counter, plus (H = 128)
    // 16 workgroups per direction that compute the BPTT's softmax statistics beside the chains
    ProfScope ps(MSER_PROF_LSTHM_FWD_GATES, s);
    K.stats_wgs = (H == 128 && !K.place && g_opt_stats_roles && !d.rng && 2 * fwd_wgs + 16 * d.ndir <= num_cus()) ? 16 * d.ndir : 0;
    if (H == 128) {
      MSER_TRY(allow_lds((const void*)cell_fwd_fused<2, 3>, p_lds));
      hipLaunchKernelGGL((cell_fwd_fused<2, 3>), dim3(K.place ? num_cus() : 2 * fwd_wgs + K.stats_wgs), dim3(NT), p_lds, s, K);
    } else {
      MSER_TRY(allow_lds((const void*)cell_fwd_fused<4, 6>, p_lds));
      hipLaunchKernelGGL((cell_fwd_fused<4, 6>), dim3(2 * fwd_wgs), dim3(NT), p_lds, s, K);
    }
  } else if (wide) {
    // H > 512: gates and row phase of every time step inside ONE launch (phase boundaries = counter barriers)
    // (one workgroup per CU; the memory-level parallelism the per-step launches get from 6-8 waves per SIMD comes from four k-passes
    // of operand loads in flight per wave here: wg_mm32's UN.  Two workgroups per CU were tried: 450 ms per step, the 512 workgroups
    // were not all resident and the barriers ran into their bounds.)
    const size_t wl = persist_lds(wide_lds_bytes(H));
    MSER_TRY(allow_lds((const void*)cell_wide_fwd_persist, wl));
    ProfScope ps(MSER_PROF_LSTHM_FWD_GATES, s);
    hipLaunchKernelGGL(cell_wide_fwd_persist, dim3(num_cus()), dim3(NT), wl, s, K, (unsigned)((wl - 64) / sizeof(float)));
  } else {
    MSER_TRY(allow_lds((const void*)lsthm_fwd_gates, mm_lds));
    if (H > 512) MSER_TRY(allow_lds((const void*)lsthm_fwd_z_wide, z_wide_lds_bytes(H)));
    for (int t = 0; t < T; ++t) {
      {
        ProfScope ps(MSER_PROF_LSTHM_FWD_GATES, s);
        hipLaunchKernelGGL(lsthm_fwd_gates, dim3(H / 8, 2, d.ndir * K.nmb), dim3(NT), mm_lds, s, K, t);
      }
      {
        ProfScope ps(MSER_PROF_LSTHM_FWD_Z, s);
        if (H > 512) hipLaunchKernelGGL(lsthm_fwd_z_wide, dim3(B, d.ndir, H / WIDE_IW), dim3(NT), z_wide_lds_bytes(H), s, K, t);
        else hipLaunchKernelGGL(lsthm_fwd_z, dim3(B, d.ndir), dim3(NT), z_lds, s, K, t);
      }
    }
  }
  return check_launch("lsthm_fwd");
}

// phases: MSER_PHASE_LSTHM_BWD = BPTT of the LSTHM chain + its deferred gradient GEMMs (produces dx_l / dx_a and dHQ);
// MSER_PHASE_SPEAKER_BWD = BPTT of the speaker chain + its deferred GEMMs (touches only speaker-cell gradients: it can overlap
// the encoder backward on another stream once the LSTHM phase has finished).
int marn_cell_bwd(const mser_cell_desc& d, hipStream_t s, int phases) {
  MSER_TRY(validate(d, true));
  CellHost h;
  carve_all((char*)d.workspace, d, &h);
  CellK& K = h.k;
  const int T = d.T, B = d.B, D = d.D, H = d.H;
  const long TB = (long)T * B, SB = (long)B * H;
  for (int i = 0; i < d.ndir; ++i) fill_params(K.d[i], d.dir[i]);
  fill_dropout(K, d);
  const bool ext = d.ext_hq[0] != nullptr;                   // external speaker state: no speaker chain, ext_dhq = the gradient at it
  K.ext_spk = ext ? 1 : 0;
  if (ext) MSER_REQUIRE(d.ext_dhq[0] && (d.ndir == 1 || d.ext_dhq[1]), "marn_cell_bwd: ext_hq needs ext_dhq");
  const size_t mm_lds = (RED_FLOATS + 1024) * sizeof(float);
  const size_t row_lds = row_lds_bytes(H);
  const int spk_wgs = ext ? 0 : (H / 32) * 4 * K.nmb;        // speaker BPTT: 4 products
  // H = 256: the dx products leave the chain (GEMMs after it, as in the per-step mode) when that frees enough workgroups for the K-split
  const bool nodx = g_opt_persistent && g_opt_rowsplit && g_opt_ksplit && H == 256 && !ext &&
                    ((long)2 * (H / 32) * 6 * K.nmb + spk_wgs) * d.ndir <= num_cus();
  K.nodx = nodx ? 1 : 0;
  const int mat_wgs1 = ((H / 32) * 6 + (nodx ? 0 : ((D + 31) / 32) * 2)) * K.nmb;  // LSTHM BPTT matvec roles: 4 carries + 2 speaker-gradient + 2 dx products
  // K-split of the matvec phase (the longest phase of a BPTT step, MFMA-paced: 8 waves x 32 MFMAs on 4 SIMDs): two workgroups per
  // product halve it when the doubled grid still fits beside the speaker chain
  // XCD placement (claim_role) wants one 32-workgroup LSTHM group per XCD, which excludes the K-split (64 per direction); the
  // placement is worth more (-0.45 us per hand-off against -12 us per launch)
  const bool place_ok = g_opt_persistent && g_opt_xcd_place && H == 128 && mat_wgs1 <= 32 && num_cus() == 256 && !ext;
  const int ksplit = (!place_ok && g_opt_persistent && g_opt_ksplit && (H == 128 || nodx) && ((long)2 * mat_wgs1 + spk_wgs) * d.ndir <= num_cus()) ? 2 : 1;
  K.ksplit = ksplit;
  const int mat_wgs = mat_wgs1 * ksplit;
  const int bwd_nwg = mat_wgs > 32 ? mat_wgs : 32;           // row phase spreads the B rows over all of them
  // both BPTT kernels run concurrently (pipelined): all their workgroups must be co-resident
  const bool persist = persist_ok(H, ((long)bwd_nwg + spk_wgs) * d.ndir);
  if (!persist) K.nodx = 0;
  K.bwd_rowsplit = (persist && g_opt_rowsplit && 2 * B <= bwd_nwg) ? 2 : 1;      // (H = 128: 64 workgroups per direction with the K-split)
  const size_t p_lds = persist_lds(mm_lds + (2 * (size_t)H + 16) * sizeof(float) + 64);
  const int SPLITK = 16;
  // Weight gradients inside the fused BPTT launch (wgrad_role): needs the persistent launch, the H = 128 tiling (16 / 8 column
  // tiles per row tile), D <= H, every gradient tensor present and room for its workgroups beside the chains.
  const int wgrad_wgs = d.ndir * 2 * ((4 * H / 32) / 2 + (ext ? 0 : (4 * H / 32) / 4));
  bool wgrad_in = persist && H == 128 && D <= H && ((long)bwd_nwg + spk_wgs) * d.ndir + wgrad_wgs <= num_cus() && g_opt_wgrad_inkernel;
  for (int i = 0; i < d.ndir && wgrad_in; ++i) {
    const mser_cell_params& G = d.dir[i].g;
    for (int m = 0; m < 2; ++m)
      wgrad_in = wgrad_in && G.lsthm_W[m] && G.lsthm_U[m] && G.lsthm_V[m] && G.lsthm_S[m] && (ext || (G.q_Wih[m] && G.q_Whh[m]));
  }
  K.wgrad_wgs = wgrad_in ? wgrad_wgs : 0;
  if (wgrad_in) {
    for (int i = 0; i < d.ndir; ++i) {
      DirP& k = K.d[i];
      const mser_cell_params& G = d.dir[i].g;
      k.xw[0] = k.rev ? h.xrev[0] : d.x_l; k.xw[1] = k.rev ? h.xrev[1] : d.x_a;
      k.ldxw[0] = k.rev ? D : d.ldxl; k.ldxw[1] = k.rev ? D : d.ldxa;
      for (int m = 0; m < 2; ++m) {
        k.gW[m] = G.lsthm_W[m]; k.gU[m] = G.lsthm_U[m]; k.gV[m] = G.lsthm_V[m]; k.gS[m] = G.lsthm_S[m];
        k.gWih[m] = G.q_Wih[m]; k.gWhh[m] = G.q_Whh[m];
      }
      // NOTE: gbias is indexed [stream m] by the LSTHM roles and [cell c] by the speaker roles; both live in the same launch, so
      // the two sets are kept in separate descriptors: LSTHM in gbias, speaker in gbias_s
      for (int m = 0; m < 2; ++m) {
        k.gbias[m][0] = G.lsthm_Wb[m]; k.gbias[m][1] = G.lsthm_Ub[m]; k.gbias[m][2] = G.lsthm_Vb[m]; k.gbias[m][3] = G.lsthm_Sb[m];
        k.gbias_s[m][0] = G.q_bih[m]; k.gbias_s[m][1] = G.q_bhh[m]; k.gbias_s[m][2] = nullptr; k.gbias_s[m][3] = nullptr;
      }
    }
  }
  K.bwd_sentinel = persist ? g_opt_bwd_sentinel : 0;
  const bool wide = !persist && wide_persist_ok(H, ext);
  K.spk_ks = (persist && !ext && g_opt_spk_ks) ? 1 : 0;
  K.poll_delay = g_opt_poll_delay;
  if (phases & MSER_PHASE_BWD_PREP) {
    if (K.bwd_sentinel) {
      // every word the BPTT chains hand from workgroup to workgroup starts as the sentinel (dgates | dA | dHQ | dHQp are carved back to
      // back: one fill per direction)
      for (int i = 0; i < d.ndir; ++i) {
        DirP& k = K.d[i];
        MSER_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)k.dgates, SENT_BITS, (size_t)((char*)(k.dHQp + 2 * 2 * TB * H) - (char*)k.dgates) / 4, s));
      }
    }
    for (int i = 0; i < d.ndir; ++i) {
      DirP& k = K.d[i];
      // dc_carry | attacc | dxc are carved back to back: one memset (dxc only where it is read without having been fully written:
      // rows at and beyond len_b receive no gradient from the reversed direction)
      const size_t zbytes = (persist && !K.nodx && k.rev) ? (size_t)((char*)(k.dxc + (size_t)ksplit * 2 * TB * D) - (char*)k.dc_carry)
                                               : (size_t)((char*)k.dxc - (char*)k.dc_carry);
      MSER_CHECK_HIP(hipMemsetAsync(k.dc_carry, 0, zbytes, s));
    }
    MSER_CHECK_HIP(hipMemsetAsync(h.sync + SYNC_LSTHM_BWD, 0, (4 * SYNC_DIR + SYNC_PLACE_LINES * SYNC_LINE) * sizeof(unsigned), s));
  }
  if (phases & MSER_PHASE_LSTHM_BWD) {
  // ---- LSTHM chain, reverse time
  if (persist) {
    // ONE launch for both BPTT chains: bwd_nwg*ndir LSTHM workgroups + spk_wgs*ndir speaker workgroups
    const size_t f_lds = persist_lds(std::max(p_lds, (K.spk_ks ? spk_ks_lds_floats() : spk_bwd_lds_floats(H)) * sizeof(float) + 64));
    const unsigned roles = (unsigned)(((long)bwd_nwg + spk_wgs) * d.ndir + K.wgrad_wgs);
    K.place = (place_ok && bwd_nwg == 32 && spk_wgs == 16 && num_cus() == 256) ? 1 : 0;
    if (K.place) {
      // logical ids: LSTHM d0 [0,32) [, d1 [32,64)], speaker groups of 16, then the wgrad roles.  LSTHM groups on two XCDs each,
      // every speaker group on one XCD, the wgrad roles spread over what is left (at most 24 per XCD: every XCD keeps free CUs)
      int x = 0, id = 0;
      for (int i = 0; i < d.ndir; ++i)
        for (int hx = 0; hx < 2; ++hx) { K.place_base[x] = (short)id; K.place_cap[x] = 16; id += 16; ++x; }
      for (int i = 0; i < d.ndir; ++i) { K.place_base[x] = (short)id; K.place_cap[x] = 16; id += 16; ++x; }
      const int left = 8 - x;
      int rem = K.wgrad_wgs;
      for (int j = 0; j < left; ++j) {
        const int take = (rem + (left - j) - 1) / (left - j);
        K.place_base[x] = (short)id; K.place_cap[x] = (short)take; id += take; rem -= take; ++x;
      }
      if (id != (int)roles) K.place = 0;
    }
    const unsigned grid = K.place ? (unsigned)num_cus() : roles;
    ProfScope ps(MSER_PROF_LSTHM_BWD_ROW, s);
    if (H == 128 && ksplit == 2) {
      MSER_TRY(allow_lds((const void*)cell_bwd_fused<4, 4, 2>, f_lds));
      hipLaunchKernelGGL((cell_bwd_fused<4, 4, 2>), dim3(grid), dim3(NT), f_lds, s, K, (unsigned)bwd_nwg);
    } else if (H == 128) {
      MSER_TRY(allow_lds((const void*)cell_bwd_fused<4, 4>, f_lds));
      hipLaunchKernelGGL((cell_bwd_fused<4, 4>), dim3(grid), dim3(NT), f_lds, s, K, (unsigned)bwd_nwg);
    } else if (ksplit == 2) {
      MSER_TRY(allow_lds((const void*)cell_bwd_fused<8, 8, 2>, f_lds));
      hipLaunchKernelGGL((cell_bwd_fused<8, 8, 2>), dim3(grid), dim3(NT), f_lds, s, K, (unsigned)bwd_nwg);
    } else {
      MSER_TRY(allow_lds((const void*)cell_bwd_fused<8, 8>, f_lds));
      hipLaunchKernelGGL((cell_bwd_fused<8, 8>), dim3(grid), dim3(NT), f_lds, s, K, (unsigned)bwd_nwg);
    }
  } else if (wide) {
    const size_t wl = persist_lds(wide_lds_bytes(H));
    MSER_TRY(allow_lds((const void*)cell_wide_bwd_persist, wl));
    ProfScope ps(MSER_PROF_LSTHM_BWD_ROW, s);
    hipLaunchKernelGGL(cell_wide_bwd_persist, dim3(num_cus()), dim3(NT), wl, s, K, (unsigned)((wl - 64) / sizeof(float)));
  } else {
    MSER_TRY(allow_lds((const void*)lsthm_bwd_mat, mm_lds));
    if (H > 512) MSER_TRY(allow_lds((const void*)lsthm_bwd_row_wide, bwd_row_wide_lds_bytes(H)));
    for (int t = T - 1; t >= 0; --t) {
      {
        ProfScope ps(MSER_PROF_LSTHM_BWD_ROW, s);
        if (H > 512) hipLaunchKernelGGL(lsthm_bwd_row_wide, dim3(B, d.ndir, H / WIDE_IW), dim3(NT), bwd_row_wide_lds_bytes(H), s, K, t);
        else hipLaunchKernelGGL(lsthm_bwd_row, dim3(B, d.ndir), dim3(NT), row_lds, s, K, t);
      }
      if (t > 0) {
        ProfScope ps(MSER_PROF_LSTHM_BWD_MAT, s);
        hipLaunchKernelGGL(lsthm_bwd_mat, dim3(H / 32, 4, d.ndir * K.nmb), dim3(NT), mm_lds, s, K, t);
      }
    }
  }
  MSER_TRY(check_launch("lsthm_bwd"));
  }
  // ---- deferred (non-recurrent) GEMMs of the LSTHM streams.  DX: what the rest of the backward waits for (dHQ for the speaker
  // chain, dx_l / dx_a for the encoders).  WGRAD: parameter gradients only -- nothing downstream reads them in this step; all
  // of them (4 per stream and direction) go out as ONE grouped launch.
  std::vector<mser_gemm_desc> wg;
  for (int i = 0; i < d.ndir; ++i) {
    DirP& k = K.d[i];
    const mser_cell_params& G = d.dir[i].g;
    const float* xs[2] = {d.x_l, d.x_a};
    long lds[2] = {d.ldxl, d.ldxa};
    float* dxs[2] = {d.dx_l, d.dx_a};
    if (k.rev) { xs[0] = h.xrev[0]; xs[1] = h.xrev[1]; lds[0] = lds[1] = D; }
    for (int m = 0; m < 2; ++m) {
      const float* dg = k.dgates + (long)m * TB * 4 * H;
      mser_gemm_desc g;
      if (phases & MSER_PHASE_LSTHM_BWD_DX) {
        if (!persist) {    // dHQ += dg S_m (the pipelined persistent kernels exchange this per step instead)
          g = gd(dg, 4 * H, 1, k.S[m], H, 1, k.dHQ, H, (int)TB, H, 4 * H);
          g.flags = MSER_GEMM_ACCUM;
          MSER_TRY(gemm(g, s));
        }
        // dx (direction order) = dg W_m
        if (persist && !K.nodx) {       // produced inside the BPTT kernel at natural time rows: summed below, both directions in one launch
        } else if (!k.rev) {
          g = gd(dg, 4 * H, 1, k.W[m], D, 1, dxs[m], D, (int)TB, D, 4 * H);
          g.flags = MSER_GEMM_ACCUM;
          MSER_TRY(gemm(g, s));
        } else {
          g = gd(dg, 4 * H, 1, k.W[m], D, 1, h.dxtmp, D, (int)TB, D, 4 * H);
          MSER_TRY(gemm(g, s));
          hipLaunchKernelGGL(reverse_acc_kernel, dim3(cdiv(TB * D, 256)), dim3(256), 0, s, h.dxtmp, k.rev, dxs[m], (long)D, T, B, D);
          MSER_TRY(check_launch("reverse_acc"));
        }
      }
      if ((phases & MSER_PHASE_LSTHM_WGRAD) && G.lsthm_W[m]) {
        // dW_m += dg^T xdir ; dS_m += dg^T HQ ; dU_m += dg^T h_prev ; dV_m += dg^T z_prev  (unless the BPTT launch did them)
        if (wgrad_in) continue;        // weights AND biases were accumulated by the BPTT launch's wgrad roles
        g = gd(dg, 1, 4 * H, xs[m], lds[m], 1, G.lsthm_W[m], D, 4 * H, D, (int)TB);
        g.splitk = SPLITK;
        wg.push_back(g);
        g = gd(dg, 1, 4 * H, k.HQ, H, 1, G.lsthm_S[m], H, 4 * H, H, (int)TB);
        g.splitk = SPLITK;
        wg.push_back(g);
        g = gd(dg, 1, 4 * H, k.hz + m * H, 3 * H, 1, G.lsthm_U[m], H, 4 * H, H, (int)TB);
        g.splitk = SPLITK;
        wg.push_back(g);
        g = gd(dg, 1, 4 * H, k.hz + 2 * H, 3 * H, 1, G.lsthm_V[m], H, 4 * H, H, (int)TB);
        g.splitk = SPLITK;
        wg.push_back(g);
        MSER_TRY(colsum4(dg, TB, 4 * H, 4 * H, G.lsthm_Wb[m], G.lsthm_Ub[m], G.lsthm_Vb[m], G.lsthm_Sb[m], s));
      }
    }
    if ((phases & MSER_PHASE_LSTHM_WGRAD) && G.att_Wq) {
      MSER_TRY(colsum4(k.attacc, B, H, 2 * H, G.att_Wq, nullptr, nullptr, nullptr, s));
      MSER_TRY(colsum4(k.attacc + H, B, H, 2 * H, G.att_Wk, nullptr, nullptr, nullptr, s));
    }
  }
  if ((phases & MSER_PHASE_LSTHM_BWD_DX) && ((persist && !K.nodx) || d.dx_l_add[0] || d.dx_l_add[1] || d.dx_a_add[0] || d.dx_a_add[1])) {
    SumArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.out[0] = d.dx_l; sa.out[1] = d.dx_a;
    sa.n = TB * D;
    int n[2] = {0, 0};
    for (int i = 0; i < d.ndir && persist && !K.nodx; ++i)
      for (int kh = 0; kh < ksplit; ++kh)
        for (int m = 0; m < 2; ++m) sa.src[m][n[m]++] = K.d[i].dxc + ((long)kh * 2 + m) * TB * D;
    for (int j = 0; j < 2; ++j) {
      if (d.dx_l_add[j]) sa.src[0][n[0]++] = d.dx_l_add[j];
      if (d.dx_a_add[j]) sa.src[1][n[1]++] = d.dx_a_add[j];
    }
    bool vec = (sa.n & 3) == 0;
    for (int m = 0; m < 2; ++m) {
      vec = vec && ((uintptr_t)sa.out[m] & 15) == 0;
      for (int j = 0; j < 6; ++j) vec = vec && ((uintptr_t)sa.src[m][j] & 15) == 0;
    }
    if (vec) hipLaunchKernelGGL(sum_into_kernel<4>, dim3(cdiv(sa.n / 4, 256), 2), dim3(256), 0, s, sa);
    else hipLaunchKernelGGL(sum_into_kernel<1>, dim3(cdiv(sa.n, 256), 2), dim3(256), 0, s, sa);
    MSER_TRY(check_launch("sum_into"));
  }
  MSER_TRY(gemm_group(wg.data(), (int)wg.size(), s));
  wg.clear();
  if (!(phases & MSER_PHASE_SPEAKER_BWD)) return 0;
  if (ext) {
    // no speaker chain of our own: hand the total gradient at the external speaker state to the caller.  Persistent mode left the
    // two (K-split: four) per-product parts dgates_m S_m in dHQp; the per-step mode accumulated them into dHQ already (DX phase).
    MSER_TRY(gemm_group(wg.data(), (int)wg.size(), s));
    for (int i = 0; i < d.ndir; ++i) {
      DirP& k = K.d[i];
      const long n = TB * H;
      const float* p0 = persist ? k.dHQp : nullptr;
      const float* p2 = (persist && ksplit == 2) ? k.dHQp + 2 * n : nullptr;
      hipLaunchKernelGGL(dhq_total_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, d.ext_dhq[i], k.dHQ, p0, p0 ? p0 + n : nullptr, p2,
                         p2 ? p2 + n : nullptr, n);
    }
    return check_launch("dhq_total");
  }
  // ---- speaker chain, reverse time
  const size_t spk_lds = spk_bwd_lds_floats(H) * sizeof(float) + 64;
  if (!persist && spk_lds <= 160 * 1024) {      // persistent mode: the speaker BPTT chain already ran inside the fused launch of the LSTHM_BWD phase
    MSER_TRY(allow_lds((const void*)spk_bwd_step, spk_lds));
    for (int t = T - 1; t >= 0; --t) {
      ProfScope ps(MSER_PROF_SPK_BWD, s);
      hipLaunchKernelGGL(spk_bwd_step, dim3(H / 32, 4, d.ndir * K.nmb), dim3(NT), spk_lds, s, K, t);
    }
  } else if (!persist && wide) {                // H > 512: prologue and products of every step inside one launch
    const size_t wl = persist_lds(wide_lds_bytes(H));
    MSER_TRY(allow_lds((const void*)cell_wide_spkbwd_persist, wl));
    ProfScope ps(MSER_PROF_SPK_BWD, s);
    hipLaunchKernelGGL(cell_wide_spkbwd_persist, dim3(num_cus()), dim3(NT), wl, s, K, (unsigned)((wl - 64) / sizeof(float)));
  } else if (!persist) {                        // H >= 512: the gate-gradient tile goes through global memory (two launches per step)
    MSER_TRY(allow_lds((const void*)spk_bwd_mat_wide, mm_lds));
    for (int t = T - 1; t >= 0; --t) {
      ProfScope ps(MSER_PROF_SPK_BWD, s);
      hipLaunchKernelGGL(spk_bwd_pre_wide, dim3(H / 32, 4, d.ndir * K.nmb), dim3(NT), 0, s, K, t);
      hipLaunchKernelGGL(spk_bwd_mat_wide, dim3(H / 32, 4, d.ndir * K.nmb), dim3(NT), mm_lds, s, K, t);
    }
  }
  MSER_TRY(check_launch("spk_bwd"));
  for (int i = 0; i < d.ndir; ++i) {
    DirP& k = K.d[i];
    const mser_cell_params& G = d.dir[i].g;
    if (!G.q_Wih[0]) continue;
    for (int c = 0; c < 2; ++c) {
      const float* dsg = k.dsg + (long)c * TB * 4 * H;
      if (wgrad_in) continue;
      mser_gemm_desc g = gd(dsg, 1, 4 * H, k.qsel + (long)c * TB * H, H, 1, G.q_Wih[c], H, 4 * H, H, (int)TB);
      g.splitk = SPLITK;
      wg.push_back(g);
      g = gd(dsg, 1, 4 * H, k.hq_state + (long)c * (T + 1) * SB, H, 1, G.q_Whh[c], H, 4 * H, H, (int)TB);
      g.splitk = SPLITK;
      wg.push_back(g);
      MSER_TRY(colsum4(dsg, TB, 4 * H, 4 * H, G.q_bih[c], G.q_bhh[c], nullptr, nullptr, s));
    }
  }
  return gemm_group(wg.data(), (int)wg.size(), s);
}

// ================================================================================================ module-level single steps
__global__ void lsthm_step_kernel(const float* x, const float* c, const float* h, const float* z, const float* sp, const float* W,
                                  const float* Wb, const float* U, const float* Ub, const float* V, const float* Vb, const float* S,
                                  const float* Sb, float* c_out, float* h_out, float* gates, int B, int D, int H, int Hz, int Hs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H) return;
  const int b = (int)(i / H), u = (int)(i % H);
  float g[4];
  for (int gt = 0; gt < 4; ++gt) {
    const long r = (long)gt * H + u;
    float a = Wb[r] + Ub[r] + Vb[r] + Sb[r];
    for (int k = 0; k < D; ++k) a = fmaf(x[(long)b * D + k], W[r * D + k], a);
    for (int k = 0; k < H; ++k) a = fmaf(h[(long)b * H + k], U[r * H + k], a);
    for (int k = 0; k < Hz; ++k) a = fmaf(z[(long)b * Hz + k], V[r * Hz + k], a);
    for (int k = 0; k < Hs; ++k) a = fmaf(sp[(long)b * Hs + k], S[r * Hs + k], a);
    g[gt] = a;
  }
  const float gf = sigmoidf_(g[0]), gi = sigmoidf_(g[1]), go = sigmoidf_(g[2]), gc = tanhf(g[3]);
  const float cn = gf * c[i] + gi * gc;
  c_out[i] = cn;
  h_out[i] = tanhf(cn) * go;
  if (gates) {
    float* o = gates + (long)b * 4 * H + u;
    o[0] = gf; o[H] = gi; o[2 * H] = go; o[3 * H] = gc;
  }
}

__global__ __launch_bounds__(NT) void rank1_attention_kernel(const float* x1, const float* x2, const float* Wq, const float* Wk,
                                                             float* out, int B, int H, const uint32_t* rng, uint32_t site, float p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Q = NT / H, JC = H / Q;
  float* ca = smem; float* wk = ca + H; float* pZ = wk + H; float* pN = pZ + NT; float* sh = pN + NT;
  const int b = blockIdx.x, tid = threadIdx.x;
  float sp = 0.f, wmx = -INFINITY, wmn = INFINITY;
  for (int k = tid; k < H; k += NT) {
    const float cv = x2[(long)b * H + k], w = Wk[k];
    ca[k] = cv; wk[k] = w;
    sp += Wq[k] * cv;
    wmx = fmaxf(wmx, w);
    wmn = fminf(wmn, w);
  }
  const float s = block_sum(sp, sh) / sqrtf((float)H);
  wmx = block_max(wmx, sh);
  wmn = -block_max(-wmn, sh);
  const int i = tid & (H - 1), q = tid / H;
  const float u = x1[(long)b * H + i] * s;
  const float mx = (u >= 0.f) ? u * wmx : u * wmn;
  const float u2 = u * LOG2E, m2 = mx * LOG2E;
  float Z = 0.f, N = 0.f;
  if (rng) {               // :69 dropout on the normalised attention: element (b, i, j)
    const DropKey dk = drop_key(rng, site, p);
    const uint32_t e0 = (uint32_t)(((long)b * H + i) * H);
    for (int j = q * JC; j < (q + 1) * JC; ++j) {
      const float e = __builtin_amdgcn_exp2f(fmaf(u2, wk[j], -m2));
      Z += e;
      N = fmaf(e * drop_scale16(dk, e0 + (uint32_t)j), ca[j], N);
    }
  } else {
    for (int j = q * JC; j < (q + 1) * JC; ++j) {
      const float e = __builtin_amdgcn_exp2f(fmaf(u2, wk[j], -m2));
      Z += e;
      N = fmaf(e, ca[j], N);
    }
  }
  pZ[tid] = Z; pN[tid] = N;
  __syncthreads();
  if (q == 0) {
    for (int qq = 1; qq < Q; ++qq) { Z += pZ[qq * H + i]; N += pN[qq * H + i]; }
    out[(long)b * H + i] = N / Z;
  }
}


// Gate backward of ONE LSTHM1 step (module-level API; training goes through the fused BPTT): from d(c_t), d(h_t) to the
// pre-activation gradients [B, 4H] in gate order f, i, o, c~ and d(c_{t-1}).  The products with W, U, V, S are GEMMs.
__global__ void lsthm_step_bwd_kernel(const float* gates, const float* c_prev, const float* c_new, const float* dc_new,
                                      const float* dh_new, float* dgates, float* dc_prev, int B, int H) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H) return;
  const int b = (int)(i / H), u = (int)(i % H);
  const float* g = gates + (long)b * 4 * H + u;
  const float gf = g[0], gi = g[H], go = g[2 * H], gc = g[3 * H];
  const float tc = tanhf(c_new[i]);
  const float dh = dh_new ? dh_new[i] : 0.f;
  const float dc = (dc_new ? dc_new[i] : 0.f) + dh * go * (1.f - tc * tc);
  float* o = dgates + (long)b * 4 * H + u;
  o[0] = dc * c_prev[i] * gf * (1.f - gf);
  o[H] = dc * gc * gi * (1.f - gi);
  o[2 * H] = dh * tc * go * (1.f - go);
  o[3 * H] = dc * gi * (1.f - gc * gc);
  dc_prev[i] = dc * gf;
}

// Backward of the rank-1 CrossAttention (module-level API): out[i] = sum_j a_ij x2[j], a_ij = softmax_j(u_i Wk[j]),
// u_i = x1[i] * s, s = <Wq, x2> / sqrt(H).  One workgroup of H threads per row; pass A (thread = unit i) builds the softmax
// statistics, pass B (thread = key j) the sums over i.  dWq / dWk are accumulated over the rows with float atomics.
__global__ void rank1_attention_bwd_kernel(const float* x1, const float* x2, const float* Wq, const float* Wk, const float* dout,
                                           float* dx1, float* dx2, float* gWq, float* gWk, int B, int H, const uint32_t* rng,
                                           uint32_t site, float p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ca = smem;            // x2
  float* wk = ca + H;
  float* cu = wk + H;          // u_i
  float* cm = cu + H;          // row maximum
  float* cz = cm + H;          // z_i
  float* cd = cz + H;          // dz_i / Z_i
  float* red = cd + H;         // [H] reduction scratch
  const int b = blockIdx.x, t = threadIdx.x;
  const float rsH = 1.0f / sqrtf((float)H);
  const float x2v = x2[(long)b * H + t], wq = Wq[t], wkv = Wk[t];
  ca[t] = x2v; wk[t] = wkv; red[t] = wq * x2v;
  __syncthreads();
  for (int o = H / 2; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  const float s = red[0] * rsH;
  __syncthreads();
  // ---- pass A: unit i = t
  const float x1v = x1[(long)b * H + t];
  const float u = x1v * s;
  float mx = -INFINITY;
  for (int j = 0; j < H; ++j) mx = fmaxf(mx, u * wk[j]);
  float Z = 0.f, N = 0.f, M = 0.f, Wn = 0.f;
  DropKey dk;
  if (rng) dk = drop_key(rng, site, p);
  const uint32_t e0 = (uint32_t)((long)b * H * H);
  for (int j = 0; j < H; ++j) {
    const float e = expf(u * wk[j] - mx);
    const float ef = rng ? e * drop_scale16(dk, e0 + (uint32_t)(t * H + j)) : e;   // the value sums see the dropped attention
    Z += e; N = fmaf(ef, ca[j], N); M = fmaf(ef, ca[j] * wk[j], M); Wn = fmaf(e, wk[j], Wn);
  }
  const float z = N / Z, dz = dout[(long)b * H + t];
  const float du = dz * (M - z * Wn) / Z;
  cu[t] = u; cm[t] = mx; cz[t] = z; cd[t] = dz / Z;
  dx1[(long)b * H + t] = du * s;
  red[t] = du * x1v;
  __syncthreads();
  for (int o = H / 2; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  const float ds = red[0];
  // ---- pass B: key j = t
  float dxj = 0.f, dwj = 0.f;
  for (int i = 0; i < H; ++i) {
    const float a = expf(cu[i] * wkv - cm[i]) * cd[i];
    const float f = rng ? drop_scale16(dk, e0 + (uint32_t)(i * H + t)) : 1.f;
    dxj = fmaf(a, f, dxj);
    dwj = fmaf(a * (f * x2v - cz[i]), cu[i], dwj);
  }
  dx2[(long)b * H + t] = dxj + ds * wq * rsH;
  atomicAdd(gWq + t, ds * x2v * rsH);
  atomicAdd(gWk + t, dwj);
}

}  // namespace mser

using namespace mser;

extern "C" {

int mser_set_option(int32_t key, int32_t value) {
  switch (key) {
    case MSER_OPT_PERSISTENT: g_opt_persistent = value ? 1 : 0; return 0;
    case MSER_OPT_WGRAD_INKERNEL: g_opt_wgrad_inkernel = value ? 1 : 0; return 0;
    case MSER_OPT_BPTT_KSPLIT: g_opt_ksplit = value ? 1 : 0; return 0;
    case MSER_OPT_XCD_PLACEMENT: g_opt_xcd_place = value ? 1 : 0; return 0;
    case MSER_OPT_FWD_STATS_ROLES: g_opt_stats_roles = value ? 1 : 0; return 0;
    case MSER_OPT_FWD_SENTINEL: g_opt_fwd_sentinel = value ? 1 : 0; return 0;
    case MSER_OPT_BWD_SENTINEL: g_opt_bwd_sentinel = value == 2 ? 2 : (value ? 1 : 0); return 0;
    case MSER_OPT_H256_SPLIT: g_opt_rowsplit = value ? 1 : 0; return 0;
    case MSER_OPT_SPK_BWD_KSPLIT: g_opt_spk_ks = value ? 1 : 0; return 0;
    case MSER_OPT_WIDE_PERSISTENT: g_opt_wide_persist = value ? 1 : 0; return 0;
    case MSER_OPT_DRNN_PERSISTENT: mser::g_opt_drnn_persist = value ? 1 : 0; return 0;
    case MSER_OPT_BWD_POLL_DELAY: g_opt_poll_delay = value < 0 ? 0 : (value > 256 ? 256 : value); return 0;
    default: set_error("mser_set_option: unknown key %d", key); return -1;
  }
}

int mser_marn_cell_status(const mser_cell_desc* d, mser_stream_t stream) {
  if (!d || !d->workspace) { set_error("mser_marn_cell_status: null descriptor/workspace"); return -1; }
  CellHost h;
  carve_all((char*)d->workspace, *d, &h);
  unsigned words[SYNC_WORDS];
  MSER_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  MSER_CHECK_HIP(hipMemcpy(words, h.sync, sizeof(words), hipMemcpyDeviceToHost));
#ifdef MSER_STAMPS
  {
    const char* names[6] = {"spk_fwd wg(0,0)", "lsthm_fwd wg(3,1)", "lsthm_bwd wg 1", "spk_bwd wg(0,0)", "spk_bwd wg(2,1)", "spk_fwd wg(5,1)"};
    for (int k = 0; k < 6; ++k) {
      fprintf(stderr, "[stamps %-18s 10ns ticks/step]", names[k]);
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %u", words[SYNC_STAMPS + 8 * k + i]);
      fprintf(stderr, "\n");
    }
  }
#endif
  if (words[SYNC_ABORT] != 0) {
    set_error("marn_cell: a persistent kernel gave up waiting at an inter-workgroup barrier (counters: spk_fwd %u/%u lsthm_fwd %u/%u "
              "lsthm_bwd %u/%u spk_bwd %u/%u)", words[SYNC_SPK_FWD], words[SYNC_SPK_FWD + SYNC_DIR], words[SYNC_LSTHM_FWD], words[SYNC_LSTHM_FWD + SYNC_DIR], words[SYNC_LSTHM_BWD],
              words[SYNC_LSTHM_BWD + SYNC_DIR], words[SYNC_SPK_BWD], words[SYNC_SPK_BWD + SYNC_DIR]);
    return -2;
  }
  return 0;
}

size_t mser_marn_cell_workspace_bytes(int32_t T, int32_t B, int32_t D, int32_t H, int32_t ndir) {
  mser_cell_desc d;
  memset(&d, 0, sizeof(d));
  d.T = T; d.B = B; d.D = D; d.H = H; d.ndir = ndir;
  return carve_all(nullptr, d, nullptr);
}

int mser_marn_cell_fwd(const mser_cell_desc* d, mser_stream_t stream) {
  if (!d) { set_error("mser_marn_cell_fwd: null descriptor"); return -1; }
  return marn_cell_fwd(*d, (hipStream_t)stream, MSER_PHASE_FWD_PREP | MSER_PHASE_SPEAKER_FWD | MSER_PHASE_LSTHM_FWD);
}

int mser_marn_cell_bwd(const mser_cell_desc* d, mser_stream_t stream) {
  if (!d) { set_error("mser_marn_cell_bwd: null descriptor"); return -1; }
  return marn_cell_bwd(*d, (hipStream_t)stream, MSER_PHASE_BWD_PREP | MSER_PHASE_LSTHM_BWD | MSER_PHASE_LSTHM_BWD_DX | MSER_PHASE_LSTHM_WGRAD | MSER_PHASE_SPEAKER_BWD);
}

int mser_marn_cell_ext_link(const mser_cell_desc* d, int32_t dir, int32_t partner_wgs, float** hq_rows, uint32_t** counter,
                            int32_t* replicas, int32_t* replica_stride, uint32_t* per_step) {
  if (!d || dir < 0 || dir >= d->ndir || !hq_rows || !counter || !replicas || !replica_stride || !per_step) {
    set_error("mser_marn_cell_ext_link: bad arguments");
    return -1;
  }
  CellHost h;
  carve_all((char*)d->workspace, *d, &h);
  const long fwd_wgs = (long)(d->H / 8) * 2 * d->ndir * h.k.nmb;
  *hq_rows = h.k.d[dir].HQ;
  *counter = h.sync + SYNC_SPK_FWD + dir * SYNC_DIR;
  *replicas = SYNC_REP;
  *replica_stride = SYNC_LINE;
  *per_step = (unsigned)((d->H / 8) * 2 * h.k.nmb);          // what the LSTHM chain expects per published step (its nwg_spk)
  // 1: the persistent LSTHM launch will run AND the partner's workgroups fit beside it (a consumer that filled every CU before the
  // producer was dispatched would spin to its bound): one CU per workgroup of either kernel
  return (partner_wgs >= 0 && persist_ok(d->H, fwd_wgs) && fwd_wgs + partner_wgs <= num_cus()) ? 1 : 0;
}

int mser_marn_cell_ext_link_bwd(const mser_cell_desc* d, int32_t dir, int32_t partner_wgs, const float** dhq, const float** dhq_parts,
                                int32_t* n_parts, int64_t* part_stride, uint32_t** counter, int32_t* replicas, int32_t* replica_stride,
                                uint32_t* per_step) {
  if (!d || dir < 0 || dir >= d->ndir || !dhq || !dhq_parts || !n_parts || !part_stride || !counter || !replicas || !replica_stride ||
      !per_step) {
    set_error("mser_marn_cell_ext_link_bwd: bad arguments");
    return -1;
  }
  CellHost h;
  carve_all((char*)d->workspace, *d, &h);
  // the same plan as marn_cell_bwd makes for an external speaker state (no speaker workgroups)
  const int H = d->H, D = d->D;
  const int mat_wgs1 = ((H / 32) * 6 + ((D + 31) / 32) * 2) * h.k.nmb;
  const int ksplit = (g_opt_persistent && g_opt_ksplit && H == 128 && ((long)2 * mat_wgs1) * d->ndir <= num_cus()) ? 2 : 1;
  const int mat_wgs = mat_wgs1 * ksplit;
  const int bwd_nwg = mat_wgs > 32 ? mat_wgs : 32;
  *dhq = h.k.d[dir].dHQ;
  *dhq_parts = h.k.d[dir].dHQp;
  *n_parts = 2 * ksplit;
  *part_stride = (int64_t)d->T * d->B * H;
  *counter = h.sync + SYNC_LSTHM_BWD + dir * SYNC_DIR;
  *replicas = SYNC_REP;
  *replica_stride = SYNC_LINE;
  *per_step = 2u * (unsigned)bwd_nwg;           // the BPTT chain passes two barriers per step: dHQ[t] and its parts are complete at
                                               // counter >= per_step * (T - t)
  // the BPTT launch also carries the in-launch weight-gradient roles when they fit (marn_cell_bwd's plan): count them
  const long wgrad_wgs = (long)d->ndir * 2 * ((4 * H / 32) / 2);
  const long cell_wgs = (long)bwd_nwg * d->ndir + ((g_opt_wgrad_inkernel && H == 128 && D <= H) ? wgrad_wgs : 0);
  return (partner_wgs >= 0 && persist_ok(H, (long)bwd_nwg * d->ndir) && cell_wgs + partner_wgs <= num_cus()) ? 1 : 0;
}

int mser_marn_cell_pipelined(int32_t B, int32_t H, int32_t ndir) {
  // Since the chains of a pass share ONE fused launch, the caller never has to overlap phases itself: always 0 (kept for ABI
  // stability; the phases may simply be issued in their listed order).
  (void)B; (void)H; (void)ndir;
  return 0;
}

int mser_marn_cell_run(const mser_cell_desc* d, int32_t phases, mser_stream_t stream) {
  if (!d) { set_error("mser_marn_cell_run: null descriptor"); return -1; }
  if (phases & (MSER_PHASE_FWD_PREP | MSER_PHASE_SPEAKER_FWD | MSER_PHASE_LSTHM_FWD | MSER_PHASE_LSTHM_PRE | MSER_PHASE_LSTHM_PRE_L | MSER_PHASE_LSTHM_PRE_A))
    MSER_TRY(marn_cell_fwd(*d, (hipStream_t)stream, phases));
  if (phases & (MSER_PHASE_BWD_PREP | MSER_PHASE_LSTHM_BWD | MSER_PHASE_LSTHM_BWD_DX | MSER_PHASE_LSTHM_WGRAD | MSER_PHASE_SPEAKER_BWD))
    MSER_TRY(marn_cell_bwd(*d, (hipStream_t)stream, phases));
  return 0;
}

int mser_lsthm_step_fwd(const float* x, const float* c, const float* h, const float* z, const float* s, const float* W,
                        const float* Wb, const float* U, const float* Ub, const float* V, const float* Vb, const float* S,
                        const float* Sb, float* c_out, float* h_out, float* gates, int32_t B, int32_t D, int32_t H, int32_t Hz,
                        int32_t Hs, mser_stream_t stream) {
  MSER_REQUIRE(x && c && h && z && s && W && Wb && U && Ub && V && Vb && S && Sb && c_out && h_out, "mser_lsthm_step_fwd: null pointer");
  if (B <= 0) return 0;
  hipLaunchKernelGGL(lsthm_step_kernel, dim3(cdiv((long)B * H, 128)), dim3(128), 0, (hipStream_t)stream, x, c, h, z, s, W, Wb, U,
                     Ub, V, Vb, S, Sb, c_out, h_out, gates, B, D, H, Hz, Hs);
  return check_launch("mser_lsthm_step_fwd");
}

int mser_lsthm_step_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dc_new, const float* dh_new,
                        float* dgates, float* dc_prev, int32_t B, int32_t H, mser_stream_t stream) {
  MSER_REQUIRE(gates && c_prev && c_new && dgates && dc_prev, "mser_lsthm_step_bwd: null pointer");
  if (B <= 0 || H <= 0) return 0;
  hipLaunchKernelGGL(lsthm_step_bwd_kernel, dim3(cdiv((long)B * H, 128)), dim3(128), 0, (hipStream_t)stream, gates, c_prev, c_new,
                     dc_new, dh_new, dgates, dc_prev, B, H);
  return check_launch("mser_lsthm_step_bwd");
}

int mser_rank1_attention_bwd(const float* x1, const float* x2, const float* Wq, const float* Wk, const float* dout, float* dx1,
                             float* dx2, float* gWq, float* gWk, int32_t B, int32_t H, const uint32_t* rng, uint32_t site, float p,
                             mser_stream_t stream) {
  MSER_REQUIRE(!rng || (p >= 0.f && p < 1.f), "mser_rank1_attention_bwd: dropout p=%f", p);
  MSER_REQUIRE(x1 && x2 && Wq && Wk && dout && dx1 && dx2 && gWq && gWk, "mser_rank1_attention_bwd: null pointer");
  MSER_REQUIRE(H >= 32 && H <= 1024 && (H & (H - 1)) == 0, "mser_rank1_attention_bwd: H=%d must be a power of two in [32,1024]", H);
  if (B <= 0) return 0;
  hipLaunchKernelGGL(rank1_attention_bwd_kernel, dim3(B), dim3(H), 7 * (size_t)H * sizeof(float), (hipStream_t)stream, x1, x2, Wq,
                     Wk, dout, dx1, dx2, gWq, gWk, B, H, rng, site, p);
  return check_launch("mser_rank1_attention_bwd");
}

int mser_rank1_attention_fwd(const float* x1, const float* x2, const float* Wq, const float* Wk, float* out, int32_t B, int32_t H,
                             const uint32_t* rng, uint32_t site, float p, mser_stream_t stream) {
  MSER_REQUIRE(!rng || (p >= 0.f && p < 1.f), "mser_rank1_attention_fwd: dropout p=%f", p);
  MSER_REQUIRE(x1 && x2 && Wq && Wk && out, "mser_rank1_attention_fwd: null pointer");
  MSER_REQUIRE(H >= 32 && H <= 512 && (H & (H - 1)) == 0, "mser_rank1_attention_fwd: H=%d must be a power of two in [32,512]", H);
  if (B <= 0) return 0;
  hipLaunchKernelGGL(rank1_attention_kernel, dim3(B), dim3(NT), (2 * (size_t)H + 2 * NT + 16) * sizeof(float), (hipStream_t)stream, x1, x2,
                     Wq, Wk, out, B, H, rng, site, p);
  return check_launch("mser_rank1_attention_fwd");
}

}  // extern "C"
