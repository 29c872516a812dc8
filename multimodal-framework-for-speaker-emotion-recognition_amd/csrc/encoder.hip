// One post-LN Transformer encoder layer (reference model/encoder.py:116-133 = MultiHeadAttention :7-60 +
// ScaledDotProductAttention :63-86 + PositionwiseFeedForward :89-113) as THREE launches forward and THREE on the
// activation-gradient chain backward, instead of 9 / 25 generic launches:
//
//   forward : qkv = e0 Wqkv^T (gemm.hip)  ->  attn_fwd_kernel   (one workgroup per (dialogue, head): S = q k^T / sqrt(dk),
//             softmax, O = P v; q/k/v/S live in LDS, P is written once for the backward)
//                                         ->  post_fwd_kernel   (32 rows per workgroup: fc + residual + LayerNorm + FFN +
//             residual + LayerNorm; every intermediate stays in LDS, only the tensors the backward needs are written)
//   backward: post_bwd_kernel (LayerNorm, FFN, LayerNorm, fc backward for 32 rows; bias / LayerNorm parameter gradients by
//             one float atomic per column per workgroup)  ->  attn_bwd_kernel (per (dialogue, head): dV, dP, dS, dQ, dK with P/dS
//             and the head's q, k, v, dO in LDS)  ->  de0 = dqkv Wqkv + dy1 (gemm.hip);
//             weight gradients (four split-K GEMMs) are a separate phase so the caller can park them on a side stream.
//
// Why this shape: at the reference sizes (D = 100, 8 heads x 40, d_inner = 40, L <= 128) every product of the layer is far too
// small to fill the chip as a launch of its own (8-30 us each, mostly fixed cost); what bounds the layer is the NUMBER of
// dependent launches on the critical path, not HBM bytes or MFMA rate.  All arithmetic is fp32 on v_mfma_f32_32x32x2_f32
// (bit-exact fmaf chains), as everywhere on this path (parity gate 1e-4 on the log-probs).
//
// Operand conventions of the 32x32x2 MFMA used below: lane l supplies A[row = l & 31][k = l >> 5] and B[k = l >> 5][col = l & 31];
// a "chunk" is 8 consecutive k: lane (r, half) holds k = 8c + 4 half + {0..3} as one float4 and issues 4 MFMAs.
// Accumulator register i of lane l is C[row = (i & 3) + 8 (i >> 2) + 4 (l >> 5)][col = l & 31].
#include "common.h"
#include "../../include/mser.h"
#include <cmath>
#include <cstring>

namespace mser {

int gemm(const mser_gemm_desc& d, hipStream_t s);   // gemm.hip
int gemm_group(const mser_gemm_desc* d, int n, hipStream_t s);

namespace {

// 1024 threads = 16 waves = 4 per SIMD: these kernels are phase-structured (stage, product, softmax / LayerNorm, product ...)
// with a workgroup barrier between phases, and PMC showed their waves parked at s_waitcnt / s_barrier for half of their life with
// 8 waves (MFMA pipe 20-27 % busy); twice the waves per SIMD hide the LDS / L2 latencies of one another.
constexpr int ET = 1024, EW = 16;         // threads / waves per workgroup
constexpr int RT = 32;                    // rows per workgroup of the row-tiled kernels

__device__ __forceinline__ f32x16 mfma4(const float4& a, const float4& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }

// ================================================================================================ attention core
struct AttnArgs {
  const float* q; const float* k; const float* v;     // row views [rows, ld]; head h occupies columns h*dk .. h*dk+dk-1
  long ldq, ldk, ldv;
  float* o; long ldo;                                  // forward output / saved O (backward: delta = <dO, O>)
  float* P;                                            // [nb, nh, L, L]
  const unsigned char* mask;                           // [nb, nh, L, L], 0 = masked (logit := fill) or null
  const float* dO; long lddo;
  float* dq; float* dk_; float* dv; long lddq, lddk, lddv;
  int nb, nh, L, dk;
  long sb, sl;                                         // row of (dialogue b, position l) = b*sb + l*sl
  float scale, fill;
  const uint32_t* rng; uint32_t site; float p;      // attention Dropout (encoder.py:83); rng == nullptr: identity
};

__device__ __forceinline__ void stage_head(const float* src, long ld, int col0, int b, const AttnArgs& a, int LP, int SD, float* dst) {
  const int vpr = a.dk >> 2;
  for (int e = threadIdx.x; e < LP * vpr; e += ET) {
    const int l = e / vpr, c4 = (e - l * vpr) << 2;
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l < a.L) v4 = *reinterpret_cast<const float4*>(src + ((long)b * a.sb + (long)l * a.sl) * ld + col0 + c4);
    *reinterpret_cast<float4*>(dst + l * SD + c4) = v4;
  }
}

// C[32 x 32] (+)= A-rows (float4 along k) x B-rows (float4 along k): both operands row-major in LDS with k contiguous
__device__ __forceinline__ f32x16 tile_rr(const float* Arow, const float* Brow, int K, f32x16 acc) {
  for (int kc = 0; kc < K; kc += 8)
    acc = mfma4(*reinterpret_cast<const float4*>(Arow + kc), *reinterpret_cast<const float4*>(Brow + kc), acc);
  return acc;
}
// A-rows (float4 along k) x B stored [k][n] (n = this lane's column, stride ldb along k); ok = column valid
__device__ __forceinline__ f32x16 tile_rc(const float* Arow, const float* Bcol, int ldb, int K, bool ok, f32x16 acc) {
  for (int kc = 0; kc < K; kc += 8) {
    float4 b;
    b.x = Bcol[(kc + 0) * ldb]; b.y = Bcol[(kc + 1) * ldb]; b.z = Bcol[(kc + 2) * ldb]; b.w = Bcol[(kc + 3) * ldb];
    if (!ok) b = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = mfma4(*reinterpret_cast<const float4*>(Arow + kc), b, acc);
  }
  return acc;
}
// A stored [k][m] (m = this lane's row, stride lda along k) x B stored [k][n]
__device__ __forceinline__ f32x16 tile_cc(const float* Acol, int lda, const float* Bcol, int ldb, int K, bool ok, f32x16 acc) {
  for (int kc = 0; kc < K; kc += 8) {
    float4 a, b;
    a.x = Acol[(kc + 0) * lda]; a.y = Acol[(kc + 1) * lda]; a.z = Acol[(kc + 2) * lda]; a.w = Acol[(kc + 3) * lda];
    b.x = Bcol[(kc + 0) * ldb]; b.y = Bcol[(kc + 1) * ldb]; b.z = Bcol[(kc + 2) * ldb]; b.w = Bcol[(kc + 3) * ldb];
    if (!ok) b = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = mfma4(a, b, acc);
  }
  return acc;
}

__global__ __launch_bounds__(ET) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int h = blockIdx.x, b = blockIdx.y;
  const int L = a.L, LP = (L + 31) & ~31, dk = a.dk, SD = dk + 4, SS = LP + 4;
  float* qs = sm;
  float* ks = qs + LP * SD;
  float* vs = ks + LP * SD;
  float* S = vs + LP * SD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
  stage_head(a.q, a.ldq, h * dk, b, a, LP, SD, qs);
  stage_head(a.k, a.ldk, h * dk, b, a, LP, SD, ks);
  stage_head(a.v, a.ldv, h * dk, b, a, LP, SD, vs);
  __syncthreads();
  const int nt = LP >> 5;
  const long pbase = ((long)b * a.nh + h) * L * L;
  // ---- S = scale * q k^T (+ mask), key padding -> -inf
  for (int t = wave; t < nt * nt; t += EW) {
    const int ti = t / nt, tj = t - ti * nt;
    f32x16 acc = {0};
    acc = tile_rr(qs + (ti * 32 + r) * SD + half * 4, ks + (tj * 32 + r) * SD + half * 4, dk, acc);
    const int col = tj * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = ti * 32 + acc_row(i, half);
      float s = acc[i] * a.scale;
      if (a.mask && row < L && col < L && a.mask[pbase + (long)row * L + col] == 0) s = a.fill;
      if (col >= L) s = -INFINITY;
      S[row * SS + col] = s;
    }
  }
  __syncthreads();
  // ---- row softmax (one wave per row, two columns per lane); padded query rows become zero rows.  With dropout the global copy
  //      keeps the plain softmax (its backward needs it), the LDS copy that feeds O = P v takes the factor of element pbase + row*L + col
  DropKey dk_;
  if (a.rng) dk_ = drop_key(a.rng, a.site, a.p);
  for (int row = wave; row < LP; row += EW) {
    float* Sr = S + row * SS;
    const bool has1 = lane + 64 < LP;
    if (row >= L) {
      if (lane < LP) Sr[lane] = 0.f;
      if (has1) Sr[lane + 64] = 0.f;
      continue;
    }
    const float v0 = lane < LP ? Sr[lane] : -INFINITY, v1 = has1 ? Sr[lane + 64] : -INFINITY;
    const float m = wave_max(fmaxf(v0, v1));
    const float e0 = expf(v0 - m), e1 = expf(v1 - m);
    const float inv = 1.0f / wave_sum(e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    float q0 = p0, q1 = p1;
    if (a.rng) {
      const uint32_t e0i = (uint32_t)(pbase + (long)row * L);
      q0 *= drop_scale(dk_, e0i + (uint32_t)lane);
      q1 *= drop_scale(dk_, e0i + (uint32_t)(lane + 64));
    }
    if (lane < LP) Sr[lane] = q0;
    if (has1) Sr[lane + 64] = q1;
    if (a.P) {
      if (lane < L) a.P[pbase + (long)row * L + lane] = p0;
      if (lane + 64 < L) a.P[pbase + (long)row * L + lane + 64] = p1;
    }
  }
  __syncthreads();
  // ---- O = P v
  const int ntn = (dk + 31) >> 5;
  for (int t = wave; t < nt * ntn; t += EW) {
    const int ti = t / ntn, tn = t - ti * ntn;
    const int col = tn * 32 + r;
    const bool cok = col < dk;
    f32x16 acc = {0};
    acc = tile_rc(S + (ti * 32 + r) * SS + half * 4, vs + (half * 4) * SD + (cok ? col : 0), SD, LP, cok, acc);
    if (cok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = ti * 32 + acc_row(i, half);
        if (row < L) a.o[((long)b * a.sb + (long)row * a.sl) * a.ldo + h * dk + col] = acc[i];
      }
    }
  }
}

// Backward of the core for one (dialogue, head):  dV = P^T dO,  dP = dO V^T,  dS = scale * P o (dP - delta),
// delta_i = <dO_i, O_i> (= sum_j P_ij dP_ij),  dQ = dS K,  dK = dS^T Q.
__global__ __launch_bounds__(ET) void attn_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int h = blockIdx.x, b = blockIdx.y;
  const int L = a.L, LP = (L + 31) & ~31, dk = a.dk, SD = dk + 4, SS = LP + 4;
  float* Pb = sm;                       // P, then dS in place
  float* s0 = Pb + LP * SS;             // dO, then Q
  float* s1 = s0 + LP * SD;             // V, then K
  float* delta = s1 + LP * SD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
  const long pbase = ((long)b * a.nh + h) * L * L;
  DropKey dk_;
  if (a.rng) dk_ = drop_key(a.rng, a.site, a.p);
  for (int e = tid; e < LP * LP; e += ET) {        // dropout: the dropped attention (what multiplied V in the forward) for dV
    const int row = e / LP, col = e - row * LP;
    float pv = (row < L && col < L) ? a.P[pbase + (long)row * L + col] : 0.f;
    if (a.rng) pv *= drop_scale(dk_, (uint32_t)(pbase + (long)row * L + col));
    Pb[row * SS + col] = pv;
  }
  stage_head(a.dO, a.lddo, h * dk, b, a, LP, SD, s0);
  stage_head(a.v, a.ldv, h * dk, b, a, LP, SD, s1);
  for (int row = wave; row < LP; row += EW) {
    float d = 0.f;
    if (row < L && lane < dk) {
      const long g = (long)b * a.sb + (long)row * a.sl;
      d = a.dO[g * a.lddo + h * dk + lane] * a.o[g * a.ldo + h * dk + lane];
    }
    d = wave_sum(d);
    if (lane == 0) delta[row] = d;
  }
  __syncthreads();
  const int nt = LP >> 5, ntn = (dk + 31) >> 5;
  // ---- dV = P^T dO on the lower half of the waves, dP = dO V^T on the upper half (its tiles stay in registers: at most two
  //      per wave since LP <= 128)
  constexpr int HW = EW / 2;
  f32x16 dp[2];
  dp[0] = f32x16{0}; dp[1] = f32x16{0};
  if (wave < HW) {
    for (int t = wave; t < nt * ntn; t += HW) {
      const int ti = t / ntn, tn = t - ti * ntn;
      const int col = tn * 32 + r;
      const bool cok = col < dk;
      f32x16 acc = {0};
      acc = tile_cc(Pb + (half * 4) * SS + ti * 32 + r, SS, s0 + (half * 4) * SD + (cok ? col : 0), SD, LP, cok, acc);
      if (cok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = ti * 32 + acc_row(i, half);
          if (row < L) a.dv[((long)b * a.sb + (long)row * a.sl) * a.lddv + h * dk + col] = acc[i];
        }
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = (wave - HW) + u * HW;
      if (t < nt * nt) {
        const int ti = t / nt, tj = t - ti * nt;
        dp[u] = tile_rr(s0 + (ti * 32 + r) * SD + half * 4, s1 + (tj * 32 + r) * SD + half * 4, dk, dp[u]);
      }
    }
  }
  __syncthreads();                      // every wave is done with P (as P^T), dO and V
  if (wave >= HW) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = (wave - HW) + u * HW;
      if (t < nt * nt) {
        const int ti = t / nt, tj = t - ti * nt;
        const int col = tj * 32 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = ti * 32 + acc_row(i, half);
          float* p = Pb + row * SS + col;
          if (a.rng) {       // dS = scale * P o (mask o dP - delta) with the plain softmax P (re-read: the tile holds the dropped one)
            const bool in = row < L && col < L;
            const long g = pbase + (long)row * L + col;
            const float pv = in ? a.P[g] : 0.f;
            *p = a.scale * pv * (dp[u][i] * drop_scale(dk_, (uint32_t)g) - delta[row]);
          } else {
            *p = a.scale * *p * (dp[u][i] - delta[row]);
          }
        }
      }
    }
  }
  stage_head(a.q, a.ldq, h * dk, b, a, LP, SD, s0);
  stage_head(a.k, a.ldk, h * dk, b, a, LP, SD, s1);
  __syncthreads();
  // ---- dQ = dS K (work items [0, nt*ntn)) ;  dK = dS^T Q (work items [nt*ntn, 2 nt*ntn)), spread over all waves
  for (int it = wave; it < 2 * nt * ntn; it += EW) {
    const bool isk = it >= nt * ntn;
    const int t = isk ? it - nt * ntn : it;
    const int ti = t / ntn, tn = t - ti * ntn;
    const int col = tn * 32 + r;
    const bool cok = col < dk;
    f32x16 acc = {0};
    if (!isk) acc = tile_rc(Pb + (ti * 32 + r) * SS + half * 4, s1 + (half * 4) * SD + (cok ? col : 0), SD, LP, cok, acc);
    else acc = tile_cc(Pb + (half * 4) * SS + ti * 32 + r, SS, s0 + (half * 4) * SD + (cok ? col : 0), SD, LP, cok, acc);
    if (cok) {
      float* dst = isk ? a.dk_ : a.dq;
      const long ldd = isk ? a.lddk : a.lddq;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = ti * 32 + acc_row(i, half);
        if (row < L) dst[((long)b * a.sb + (long)row * a.sl) * ldd + h * dk + col] = acc[i];
      }
    }
  }
}

// ================================================================================================ row-tiled blocks
// C[32 x N] = A[32 x K] B for one workgroup.  A lives in LDS (row stride lda, zero-padded to a multiple of 8 along k).
//   BT = 0: B[n][k], k contiguous (y = x W^T with W = [N, K] row-major): one float4 per chunk and lane
//   BT = 1: B[k][n], n contiguous (dx = dy W with W = [K, N] row-major): four coalesced dwords per chunk and lane
// The column tiles (and, when there are fewer than 8 of them, k-slices of each) are spread over the 8 waves; the partial
// accumulators meet in `red` (8 x 32 x 33 floats) and emit(row, col, value) receives every finished element.
// The B operand comes straight from global memory (the weights are a few hundred KB, L2-resident and shared by all
// workgroups): loads run one pass (PASS chunks) ahead of the MFMAs, issued unconditionally from clamped addresses.
constexpr int PASS = 4;
constexpr int RED_LD = 33;

template <int BT>
__device__ __forceinline__ void load_pass(const float* Bg, long ldb, int nc, int K, int c, int c1, int half, float4* bv) {
#pragma unroll
  for (int j = 0; j < PASS; ++j) {
    const int cc = min(c + j, c1 - 1);                  // clamped chunk (values of chunks >= c1 are never used)
    const int k = cc * 8 + half * 4;
    if (BT == 0) {
      const int kk = min(k, K - 4);
      bv[j] = *reinterpret_cast<const float4*>(Bg + (long)nc * ldb + kk);
    } else {
      bv[j].x = Bg[(long)min(k + 0, K - 1) * ldb + nc];
      bv[j].y = Bg[(long)min(k + 1, K - 1) * ldb + nc];
      bv[j].z = Bg[(long)min(k + 2, K - 1) * ldb + nc];
      bv[j].w = Bg[(long)min(k + 3, K - 1) * ldb + nc];
    }
  }
}
__device__ __forceinline__ float4 mask_k(float4 v, int k, int K, bool nok) {
  // zero the k positions past the end (the LDS A operand is zero there too, but never multiply by stray bits) and invalid columns
  if (!nok || k >= K) v.x = 0.f;
  if (!nok || k + 1 >= K) v.y = 0.f;
  if (!nok || k + 2 >= K) v.z = 0.f;
  if (!nok || k + 3 >= K) v.w = 0.f;
  return v;
}

template <int BT, class Emit>
__device__ __forceinline__ void wg_gemm32(const float* A, int lda, int K, const float* Bg, long ldb, int N, float* red, Emit emit) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
  const int NT = (N + 31) >> 5, nch = (K + 7) >> 3;
  for (int t0 = 0; t0 < NT; t0 += EW) {
    const int ntg = min(EW, NT - t0);
    const int KS = EW / ntg;                            // k-slices per column tile
    const int tile = wave % ntg, slice = wave / ntg;
    const int cps = (nch + KS - 1) / KS;
    const int c0 = slice * cps, c1 = min(nch, c0 + cps);
    f32x16 acc = {0};
    if (slice < KS && c0 < c1) {
      const int n = (t0 + tile) * 32 + r;
      const bool nok = n < N;
      const int nc = nok ? n : N - 1;
      const float* Arow = A + r * lda + half * 4;
      float4 bcur[PASS], bnext[PASS];
      load_pass<BT>(Bg, ldb, nc, K, c0, c1, half, bcur);
      for (int c = c0; c < c1; c += PASS) {
        if (c + PASS < c1) load_pass<BT>(Bg, ldb, nc, K, c + PASS, c1, half, bnext);
#pragma unroll
        for (int j = 0; j < PASS; ++j) {
          if (c + j < c1) {
            const int k = (c + j) * 8 + half * 4;
            acc = mfma4(*reinterpret_cast<const float4*>(Arow + (c + j) * 8), mask_k(bcur[j], k, K, nok), acc);
          }
        }
#pragma unroll
        for (int j = 0; j < PASS; ++j) bcur[j] = bnext[j];
      }
    }
    float* my = red + wave * (32 * RED_LD);
#pragma unroll
    for (int i = 0; i < 16; ++i) my[acc_row(i, half) * RED_LD + r] = acc[i];
    __syncthreads();
    for (int e = tid; e < ntg * 1024; e += ET) {
      const int tl = e >> 10, rc = e & 1023, row = rc >> 5, col = rc & 31;
      float s = 0.f;
      for (int sl = 0; sl < KS; ++sl) s += red[(sl * ntg + tl) * (32 * RED_LD) + row * RED_LD + col];
      const int n = (t0 + tl) * 32 + col;
      if (n < N) emit(row, n, s);
    }
    __syncthreads();
  }
}

struct PostArgs {
  int rows, D, NO, F;
  float eps;
  // forward inputs / parameters
  const float* O; const float* x;
  const float* fc; const float* g1; const float* be1; const float* W1; const float* bb1; const float* W2; const float* bb2;
  const float* g2; const float* be2;
  // saved by the forward
  float* y1; float* mean1; float* rstd1; float* e1; float* hdn; float* y2; float* mean2; float* rstd2; float* out;
  // backward
  const float* dout; float* dy2; float* dh; float* dy1; float* dO;
  float* gg1; float* gbe1; float* gbb1; float* gbb2; float* gg2; float* gbe2;
  // Dropout after fc (encoder.py:54, site_fc) and after w_2 (:106, site_fc + 1), element row*D + col; rng == nullptr: identity.
  // dt1: the gradient at the fc output (masked dy1) for the fc weight gradient; dy1 itself stays the residual's gradient.
  const uint32_t* rng; uint32_t site_fc; float p_fc, p_ffn; float* dt1;
};

__device__ __forceinline__ int pad8(int n) { return ((n + 7) & ~7) + 4; }   // LDS row stride: k padded to 8, +4 (= 4 * odd mod 32 for D=100,F=40,NO=320)

// LayerNorm of the wave's rows of `ys` (+ residual / bias) -> normalised rows into `dst` (LDS, zero-padded) and global.
// v(row, j) supplies the pre-norm value.
template <class V>
__device__ __forceinline__ void ln_rows(const PostArgs& p, long r0, int D, V v, const float* gamma, const float* beta, float* ysum,
                                        float* mean, float* rstd, float* yout, float* dst, int ldd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = wave * (RT / EW); rr < (wave + 1) * (RT / EW); ++rr) {
    const long row = r0 + rr;
    const bool rok = row < p.rows;
    float x0 = 0.f, x1 = 0.f;
    if (rok && lane < D) x0 = v(rr, lane);
    if (rok && lane + 64 < D) x1 = v(rr, lane + 64);
    const float mu = wave_sum(x0 + x1) / D;
    const float d0 = lane < D ? x0 - mu : 0.f, d1 = lane + 64 < D ? x1 - mu : 0.f;
    const float rs = 1.0f / sqrtf(wave_sum(d0 * d0 + d1 * d1) / D + p.eps);
    float o0 = 0.f, o1 = 0.f;
    if (lane < D) o0 = d0 * rs * gamma[lane] + beta[lane];
    if (lane + 64 < D) o1 = d1 * rs * gamma[lane + 64] + beta[lane + 64];
    if (!rok) { o0 = 0.f; o1 = 0.f; }
    if (dst) {
      if (lane < ldd) dst[rr * ldd + lane] = lane < D ? o0 : 0.f;
      if (lane + 64 < ldd) dst[rr * ldd + lane + 64] = lane + 64 < D ? o1 : 0.f;
    }
    if (rok) {
      if (lane < D) { ysum[row * D + lane] = x0; yout[row * D + lane] = o0; }
      if (lane + 64 < D) { ysum[row * D + lane + 64] = x1; yout[row * D + lane + 64] = o1; }
      if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
  }
}

__global__ __launch_bounds__(ET) void post_fwd_kernel(PostArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = p.D, NO = p.NO, F = p.F;
  const int NOP = pad8(NO), DP = pad8(D), FP = pad8(F);
  float* red = sm;                          // 8 * 32 * 33
  float* As = red + EW * 32 * RED_LD;       // [32][NOP]  attention output tile
  float* ys = As + RT * NOP;                // [32][DP]   pre-norm sums
  float* e1s = ys + RT * DP;                // [32][DP]
  float* hs = e1s + RT * DP;                // [32][FP]
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RT;
  {
    const int vpr = NO >> 2, vpp = NOP >> 2;
    for (int e = tid; e < RT * vpp; e += ET) {
      const int rr = e / vpp, c4 = (e - rr * vpp) << 2;
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + rr < p.rows && c4 < vpr * 4) v4 = *reinterpret_cast<const float4*>(p.O + (r0 + rr) * NO + c4);
      *reinterpret_cast<float4*>(As + rr * NOP + c4) = v4;
    }
  }
  __syncthreads();
  // t = dropout(O fc^T)
  DropKey dk1, dk2;
  if (p.rng) { dk1 = drop_key(p.rng, p.site_fc, p.p_fc); dk2 = drop_key(p.rng, p.site_fc + 1u, p.p_ffn); }
  wg_gemm32<0>(As, NOP, NO, p.fc, NO, D, red, [&](int row, int n, float s) {
    ys[row * DP + n] = p.rng ? s * drop_scale(dk1, (uint32_t)((r0 + row) * D + n)) : s;
  });
  // y1 = t + e0 ; e1 = LayerNorm(y1)
  ln_rows(p, r0, D, [&](int rr, int j) { return ys[rr * DP + j] + p.x[(r0 + rr) * D + j]; }, p.g1, p.be1, p.y1, p.mean1, p.rstd1,
          p.e1, e1s, DP);
  __syncthreads();
  // hdn = relu(e1 W1^T + b1)
  for (int e = tid; e < RT * (FP - F); e += ET) hs[(e / (FP - F)) * FP + F + e % (FP - F)] = 0.f;
  wg_gemm32<0>(e1s, DP, D, p.W1, D, F, red, [&](int row, int n, float s) {
    const float hv = fmaxf(s + p.bb1[n], 0.f);
    hs[row * FP + n] = hv;
    if (r0 + row < p.rows) p.hdn[(r0 + row) * F + n] = hv;
  });
  // t2 = hdn W2^T + b2 ; y2 = t2 + e1 ; out = LayerNorm(y2)
  wg_gemm32<0>(hs, FP, F, p.W2, F, D, red, [&](int row, int n, float s) {
    float t2 = s + p.bb2[n];
    if (p.rng) t2 *= drop_scale(dk2, (uint32_t)((r0 + row) * D + n));
    ys[row * DP + n] = t2 + e1s[row * DP + n];
  });
  ln_rows(p, r0, D, [&](int rr, int j) { return ys[rr * DP + j]; }, p.g2, p.be2, p.y2, p.mean2, p.rstd2, p.out, nullptr, 0);
}

// LayerNorm backward for the wave's rows: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma.
// dyv(rr, j) supplies dy; results go to dst (LDS, zero-padded) and global dxg; per-column sums of dy*xhat / dy are left in
// cs[wave][0/1][128] for the caller to reduce.
template <class V>
__device__ __forceinline__ void ln_bwd_rows(const PostArgs& p, long r0, int D, V dyv, const float* ysum, const float* mean,
                                            const float* rstd, const float* gamma, float* dst, int ldd, float* dxg, float* cs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag0 = 0.f, ag1 = 0.f, ab0 = 0.f, ab1 = 0.f;
  const float gm0 = lane < D ? gamma[lane] : 0.f, gm1 = lane + 64 < D ? gamma[lane + 64] : 0.f;
  for (int rr = wave * (RT / EW); rr < (wave + 1) * (RT / EW); ++rr) {
    const long row = r0 + rr;
    const bool rok = row < p.rows;
    float d0 = 0.f, d1 = 0.f, xh0 = 0.f, xh1 = 0.f, rs = 0.f;
    if (rok) {
      const float mu = mean[row];
      rs = rstd[row];
      if (lane < D) { d0 = dyv(rr, lane); xh0 = (ysum[row * D + lane] - mu) * rs; }
      if (lane + 64 < D) { d1 = dyv(rr, lane + 64); xh1 = (ysum[row * D + lane + 64] - mu) * rs; }
    }
    const float g0 = d0 * gm0, g1 = d1 * gm1;
    ag0 += d0 * xh0; ag1 += d1 * xh1; ab0 += d0; ab1 += d1;
    const float s1 = wave_sum(g0 + g1) / D, s2 = wave_sum(g0 * xh0 + g1 * xh1) / D;
    const float o0 = rs * (g0 - s1 - xh0 * s2), o1 = rs * (g1 - s1 - xh1 * s2);
    if (lane < ldd) dst[rr * ldd + lane] = lane < D ? o0 : 0.f;
    if (lane + 64 < ldd) dst[rr * ldd + lane + 64] = lane + 64 < D ? o1 : 0.f;
    if (rok) {
      if (lane < D) dxg[row * D + lane] = o0;
      if (lane + 64 < D) dxg[row * D + lane + 64] = o1;
    }
  }
  cs[(wave * 2 + 0) * 128 + lane] = ag0; cs[(wave * 2 + 0) * 128 + lane + 64] = ag1;
  cs[(wave * 2 + 1) * 128 + lane] = ab0; cs[(wave * 2 + 1) * 128 + lane + 64] = ab1;
}
__device__ __forceinline__ void flush_cs(const float* cs, int D, float* gg, float* gb) {
  const int tid = threadIdx.x;
  if (tid < 2 * 128) {
    const int which = tid >> 7, j = tid & 127;
    if (j < D) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < EW; ++w) s += cs[(w * 2 + which) * 128 + j];
      atomicAdd((which ? gb : gg) + j, s);
    }
  }
}
// out[j] += sum over the 32 tile rows of T[row][j]
__device__ __forceinline__ void colsum_tile(const float* T, int ld, int n, float* out) {
  for (int j = threadIdx.x; j < n; j += ET) {
    float s = 0.f;
#pragma unroll 8
    for (int rr = 0; rr < RT; ++rr) s += T[rr * ld + j];
    atomicAdd(out + j, s);
  }
}

__global__ __launch_bounds__(ET) void post_bwd_kernel(PostArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = p.D, NO = p.NO, F = p.F;
  const int DP = pad8(D), FP = pad8(F);
  float* red = sm;                          // 8 * 32 * 33
  float* dys = red + EW * 32 * RED_LD;      // [32][DP]  dy2, later dy1
  float* des = dys + RT * DP;               // [32][DP]  de1
  float* dhs = des + RT * DP;               // [32][FP]
  float* cs = dhs + RT * FP;                // [8][2][128]
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RT;
  // ---- LayerNorm 2 backward -> dy2
  ln_bwd_rows(p, r0, D, [&](int rr, int j) { return p.dout[(r0 + rr) * D + j]; }, p.y2, p.mean2, p.rstd2, p.g2, dys, DP, p.dy2, cs);
  for (int e = tid; e < RT * (FP - F); e += ET) dhs[(e / (FP - F)) * FP + F + e % (FP - F)] = 0.f;
  __syncthreads();
  flush_cs(cs, D, p.gg2, p.gbe2);
  // dropout after w_2: the FFN path (bias sum, dh, the deferred W2 gradient through p.dy2) sees mask o dy2, the residual (de1
  // below) the plain dy2.  The masked tile borrows `des`, which is only written by the W1 product further down.
  float* dms = dys;
  if (p.rng) {
    const DropKey dk2 = drop_key(p.rng, p.site_fc + 1u, p.p_ffn);
    dms = des;
    for (int e = tid; e < RT * DP; e += ET) {
      const int rr = e / DP, j = e - rr * DP;
      float v = 0.f;
      if (j < D && r0 + rr < p.rows) {
        v = dys[e] * drop_scale(dk2, (uint32_t)((r0 + rr) * D + j));
        p.dy2[(r0 + rr) * D + j] = v;
      }
      des[e] = v;
    }
    __syncthreads();
  }
  colsum_tile(dms, DP, D, p.gbb2);
  // ---- dh = (dy2 W2) o (hdn > 0)
  wg_gemm32<1>(dms, DP, D, p.W2, F, F, red, [&](int row, int n, float s) {
    const bool rok = r0 + row < p.rows;
    const float v = (rok && p.hdn[(r0 + row) * F + n] > 0.f) ? s : 0.f;
    dhs[row * FP + n] = v;
    if (rok) p.dh[(r0 + row) * F + n] = v;
  });
  colsum_tile(dhs, FP, F, p.gbb1);
  // ---- de1 = dh W1 + dy2
  wg_gemm32<1>(dhs, FP, F, p.W1, D, D, red, [&](int row, int n, float s) { des[row * DP + n] = s + dys[row * DP + n]; });
  // ---- LayerNorm 1 backward -> dy1 (overwrites the dy2 tile)
  ln_bwd_rows(p, r0, D, [&](int rr, int j) { return des[rr * DP + j]; }, p.y1, p.mean1, p.rstd1, p.g1, dys, DP, p.dy1, cs);
  __syncthreads();
  flush_cs(cs, D, p.gg1, p.gbe1);
  if (p.rng) {      // dropout after fc: dO and the fc weight gradient (through p.dt1) see mask o dy1; p.dy1 stays the residual's gradient
    const DropKey dk1 = drop_key(p.rng, p.site_fc, p.p_fc);
    for (int e = tid; e < RT * DP; e += ET) {
      const int rr = e / DP, j = e - rr * DP;
      if (j < D && r0 + rr < p.rows) {
        const float v = dys[e] * drop_scale(dk1, (uint32_t)((r0 + rr) * D + j));
        dys[e] = v;
        p.dt1[(r0 + rr) * D + j] = v;
      }
    }
    __syncthreads();
  }
  // ---- dO = dy1 fc
  wg_gemm32<1>(dys, DP, D, p.fc, NO, NO, red, [&](int row, int n, float s) {
    if (r0 + row < p.rows) p.dO[(r0 + row) * NO + n] = s;
  });
}

// ================================================================================================ classifier tail of the fusion head
// model/lsthm_sps.py:390-393 after the fc GEMM: y1r = relu(fc(h)) + x_l + x_a ; y2 = relu(nn_out.0(y1r)) ; y3 = nn_out.3(y2) ;
// log_probs[b*L + t] = log_softmax(y3[t*B + b]).  One row-tiled launch forward, one backward (instead of 5 / 9 small ones that
// sat between the recurrent chains on the critical path).  The two products are K = D and K = hidden(32) wide: tiny.
struct TailArgs {
  int L, B, D, F, C;                      // F = nn_out hidden width (32), C = classes
  const float* y1; const float* x_l; const float* x_a;
  const float* W0; const float* b0; const float* W3; const float* b3;
  float* y1r; float* y2; float* lp;
  // backward
  const float* dlp; const float* dxl_in; const float* dxa_in;
  float* dy3; float* dy2; float* dy1; float* dx_l; float* dx_a;
  float* g_b0; float* g_b3; float* g_bfc;
  // dropout (rng == nullptr: identity): nn_out's site is evaluated here, fc's is applied by the caller on y1 before the forward
  const uint32_t* rng; uint32_t site_out; float p_out, scale_fc;
};

__global__ __launch_bounds__(ET) void tail_fwd_kernel(TailArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = p.D, F = p.F, C = p.C, rows = p.L * p.B;
  const int DP = pad8(D), FP = pad8(F), CP = C + 1;
  float* red = sm;
  float* ys = red + EW * 32 * RED_LD;       // [32][DP] y1r
  float* hs = ys + RT * DP;                 // [32][FP] y2
  float* zs = hs + RT * FP;                 // [32][CP] y3
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RT;
  for (int e = tid; e < RT * DP; e += ET) {
    const int rr = e / DP, j = e - rr * DP;
    float v = 0.f;
    if (r0 + rr < rows && j < D) {
      const long g = (r0 + rr) * D + j;
      v = p.y1[g] + p.x_l[g] + p.x_a[g];
      p.y1r[g] = v;
    }
    ys[e] = v;
  }
  for (int e = tid; e < RT * (FP - F); e += ET) hs[(e / (FP - F)) * FP + F + e % (FP - F)] = 0.f;
  __syncthreads();
  DropKey dk;
  if (p.rng) dk = drop_key(p.rng, p.site_out, p.p_out);
  wg_gemm32<0>(ys, DP, D, p.W0, D, F, red, [&](int row, int n, float s) {
    float hv = fmaxf(s + p.b0[n], 0.f);
    if (p.rng) hv *= drop_scale(dk, (uint32_t)((r0 + row) * F + n));       // nn_out Dropout (:323)
    hs[row * FP + n] = hv;
    if (r0 + row < rows) p.y2[(r0 + row) * F + n] = hv;
  });
  wg_gemm32<0>(hs, FP, F, p.W3, F, C, red, [&](int row, int n, float s) { zs[row * CP + n] = s + p.b3[n]; });
  if (tid < RT && r0 + tid < rows) {
    const long r = r0 + tid;
    const int t = (int)(r / p.B), b = (int)(r - (long)t * p.B);
    const float* z = zs + tid * CP;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
    float ssum = 0.f;
    for (int c = 0; c < C; ++c) ssum += expf(z[c] - mx);
    const float lse = mx + logf(ssum);
    float* dst = p.lp + ((long)b * p.L + t) * C;
    for (int c = 0; c < C; ++c) dst[c] = z[c] - lse;
  }
}

__global__ __launch_bounds__(ET) void tail_bwd_kernel(TailArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = p.D, F = p.F, C = p.C, rows = p.L * p.B;
  const int DP = pad8(D), FP = pad8(F), CP8 = pad8(C);
  float* red = sm;
  float* zs = red + EW * 32 * RED_LD;       // [32][CP8] dy3
  float* hs = zs + RT * CP8;                // [32][FP]  dy2
  float* ds = hs + RT * FP;                 // [32][DP]  dy1 (masked)
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RT;
  for (int e = tid; e < RT * CP8; e += ET) zs[e] = 0.f;
  for (int e = tid; e < RT * (FP - F); e += ET) hs[(e / (FP - F)) * FP + F + e % (FP - F)] = 0.f;
  __syncthreads();
  if (tid < RT && r0 + tid < rows) {
    const long r = r0 + tid;
    const int t = (int)(r / p.B), b = (int)(r - (long)t * p.B);
    const long q = ((long)b * p.L + t) * C;
    float ssum = 0.f;
    for (int c = 0; c < C; ++c) ssum += p.dlp[q + c];
    for (int c = 0; c < C; ++c) {
      const float v = p.dlp[q + c] - expf(p.lp[q + c]) * ssum;
      zs[tid * CP8 + c] = v;
      p.dy3[r * C + c] = v;
    }
  }
  __syncthreads();
  colsum_tile(zs, CP8, C, p.g_b3);
  // dy2 = (dy3 W3) o (y2 > 0); with dropout the saved y2 is the dropped one: a dropped unit reads 0 here, a kept one needs 1/(1-p)
  const float sc_out = p.rng ? 1.0f / (1.0f - p.p_out) : 1.0f;
  const float sc_fc = p.rng ? p.scale_fc : 1.0f;
  wg_gemm32<1>(zs, CP8, C, p.W3, F, F, red, [&](int row, int n, float s) {
    const bool rok = r0 + row < rows;
    const float v = (rok && p.y2[(r0 + row) * F + n] > 0.f) ? s * sc_out : 0.f;
    hs[row * FP + n] = v;
    if (rok) p.dy2[(r0 + row) * F + n] = v;
  });
  colsum_tile(hs, FP, F, p.g_b0);
  // d(y1r) = dy2 W0 -> dx_l, dx_a (+ the gradients of the returned x_l / x_a) ; dy1 = d(y1r) o (y1 > 0)
  wg_gemm32<1>(hs, FP, F, p.W0, D, D, red, [&](int row, int n, float s) {
    const bool rok = r0 + row < rows;
    float m = 0.f;
    if (rok) {
      const long g = (r0 + row) * D + n;
      p.dx_l[g] = s + (p.dxl_in ? p.dxl_in[g] : 0.f);
      p.dx_a[g] = s + (p.dxa_in ? p.dxa_in[g] : 0.f);
      m = p.y1[g] > 0.f ? s * sc_fc : 0.f;
      p.dy1[g] = m;
    }
    ds[row * DP + n] = m;
  });
  colsum_tile(ds, DP, D, p.g_bfc);
}

size_t tail_lds_bytes(int D, int F, int C, bool bwd) {
  const size_t DP = ((D + 7) & ~7) + 4, FP = ((F + 7) & ~7) + 4, red = EW * 32 * RED_LD;
  return (red + RT * DP + RT * FP + RT * (bwd ? ((C + 7) & ~7) + 4 : C + 1)) * sizeof(float);
}

size_t attn_lds_bytes(int L, int dk, bool bwd) {
  const size_t LP = (size_t)((L + 31) & ~31), SD = dk + 4, SS = LP + 4;
  return (bwd ? LP * SS + 2 * LP * SD + LP : 3 * LP * SD + LP * SS) * sizeof(float);
}
size_t post_lds_bytes(int D, int NO, int F, bool bwd) {
  const size_t DP = ((D + 7) & ~7) + 4, FP = ((F + 7) & ~7) + 4, NOP = ((NO + 7) & ~7) + 4;
  const size_t red = EW * 32 * RED_LD;
  return (bwd ? red + 2 * RT * DP + RT * FP + EW * 2 * 128 : red + RT * NOP + 2 * RT * DP + RT * FP) * sizeof(float);
}
constexpr size_t LDS_MAX = 160 * 1024;

int allow(const void* kernel, size_t bytes) {
  if (bytes > 64 * 1024) MSER_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}
bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

static const char* enc_unsupported(const mser_encoder_desc& d) {
  if (d.nb <= 0 || d.nl <= 0 || d.D <= 0) return "empty shape";
  if (d.nl > 128) return "sequence longer than 128 (the S tile of one head must fit LDS)";
  if (d.dk != d.dv) return "d_k != d_v";
  if (d.dk % 8 || d.dk > 64) return "head width must be a multiple of 8 and <= 64";
  if (d.D % 4 || d.D > 128) return "model width must be a multiple of 4 and <= 128";
  if (d.dff % 4 || d.dff > 128) return "FFN width must be a multiple of 4 and <= 128";
  if (d.nh * d.dk > 1024) return "n_head * d_k > 1024";
  if (attn_lds_bytes(d.nl, d.dk, false) > LDS_MAX || attn_lds_bytes(d.nl, d.dk, true) > LDS_MAX) return "attention tile exceeds LDS";
  if (post_lds_bytes(d.D, d.nh * d.dv, d.dff, false) > LDS_MAX) return "row tile exceeds LDS";
  return nullptr;
}

static int enc_validate(const mser_encoder_desc& d, bool bwd) {
  const char* why = enc_unsupported(d);
  MSER_REQUIRE(!why, "mser_encoder_layer: unsupported configuration: %s", why ? why : "");
  MSER_REQUIRE(d.x && d.w_qs && d.w_ks && d.w_vs && d.fc && d.ln1_g && d.ln1_b && d.w1 && d.b1 && d.w2 && d.b2 && d.ln2_g && d.ln2_b,
               "mser_encoder_layer: null parameter / input");
  MSER_REQUIRE(d.qkv && d.P && d.O && d.y1 && d.mean1 && d.rstd1 && d.e1 && d.hdn && d.y2 && d.mean2 && d.rstd2 && d.out,
               "mser_encoder_layer: null saved-tensor / output buffer");
  MSER_REQUIRE(al16(d.x) && al16(d.qkv) && al16(d.O) && al16(d.fc) && al16(d.w1) && al16(d.w2) && al16(d.w_qs) && al16(d.w_ks) && al16(d.w_vs),
               "mser_encoder_layer: buffers must be 16-byte aligned");
  if (bwd) {
    MSER_REQUIRE(d.dout && d.dy2 && d.dh && d.dy1 && d.dO && d.dqkv && d.dx, "mser_encoder_layer_bwd: null gradient buffer");
    MSER_REQUIRE(al16(d.dO) && al16(d.dqkv), "mser_encoder_layer_bwd: buffers must be 16-byte aligned");
    MSER_REQUIRE(!(d.rng && (d.p_fc > 0.f || d.p_ffn > 0.f)) || d.dt1, "mser_encoder_layer_bwd: dropout needs the dt1 buffer");
  }
  return 0;
}

static AttnArgs attn_args(const mser_encoder_desc& d) {
  AttnArgs a;
  const int nq = d.nh * d.dk;
  a.q = d.qkv; a.k = d.qkv + nq; a.v = d.qkv + 2 * nq;
  a.ldq = a.ldk = a.ldv = 3L * nq;
  a.o = d.O; a.ldo = nq;
  a.P = d.P; a.mask = d.mask;
  a.dO = d.dO; a.lddo = nq;
  a.dq = d.dqkv; a.dk_ = d.dqkv ? d.dqkv + nq : nullptr; a.dv = d.dqkv ? d.dqkv + 2 * nq : nullptr;
  a.lddq = a.lddk = a.lddv = 3L * nq;
  a.nb = d.nb; a.nh = d.nh; a.L = d.nl; a.dk = d.dk;
  a.sb = d.sb; a.sl = d.sl;
  a.scale = 1.0f / sqrtf((float)d.dk); a.fill = -1e9f;
  a.rng = (d.rng && d.p_attn > 0.f) ? d.rng : nullptr; a.site = d.drop_site; a.p = d.p_attn;
  return a;
}
static PostArgs post_args(const mser_encoder_desc& d) {
  PostArgs p;
  p.rows = d.nb * d.nl; p.D = d.D; p.NO = d.nh * d.dv; p.F = d.dff; p.eps = d.eps;
  p.O = d.O; p.x = d.x; p.fc = d.fc; p.g1 = d.ln1_g; p.be1 = d.ln1_b; p.W1 = d.w1; p.bb1 = d.b1; p.W2 = d.w2; p.bb2 = d.b2;
  p.g2 = d.ln2_g; p.be2 = d.ln2_b;
  p.y1 = d.y1; p.mean1 = d.mean1; p.rstd1 = d.rstd1; p.e1 = d.e1; p.hdn = d.hdn; p.y2 = d.y2; p.mean2 = d.mean2; p.rstd2 = d.rstd2;
  p.out = d.out;
  p.dout = d.dout; p.dy2 = d.dy2; p.dh = d.dh; p.dy1 = d.dy1; p.dO = d.dO;
  p.gg1 = d.g_ln1_g; p.gbe1 = d.g_ln1_b; p.gbb1 = d.g_b1; p.gbb2 = d.g_b2; p.gg2 = d.g_ln2_g; p.gbe2 = d.g_ln2_b;
  p.rng = (d.rng && (d.p_fc > 0.f || d.p_ffn > 0.f)) ? d.rng : nullptr;
  p.site_fc = d.drop_site + 1u; p.p_fc = d.p_fc; p.p_ffn = d.p_ffn; p.dt1 = d.dt1;
  return p;
}
// w_qs, w_ks, w_vs back to back (the flat parameter buffer lays them out that way) -> one N = 3*nh*dk projection
static bool qkv_adjacent(const float* a, const float* b, const float* c, long n) { return b == a + n && c == b + n; }

static mser_gemm_desc gd0() {
  mser_gemm_desc g;
  memset(&g, 0, sizeof g);
  g.batch1 = g.batch2 = 1; g.alpha = 1.0f; g.splitk = 1;
  return g;
}

int encoder_layer_fwd(const mser_encoder_desc& d, hipStream_t s) {
  MSER_TRY(enc_validate(d, false));
  const int rows = d.nb * d.nl, nq = d.nh * d.dk, D = d.D;
  // ---- q | k | v = e0 [Wq; Wk; Wv]^T
  mser_gemm_desc g = gd0();
  g.A = d.x; g.sAm = D; g.sAk = 1; g.M = rows; g.K = D; g.sBk = 1; g.sBn = D; g.ldc = 3L * nq;
  if (qkv_adjacent(d.w_qs, d.w_ks, d.w_vs, (long)nq * D)) {
    g.B = d.w_qs; g.C = d.qkv; g.N = 3 * nq;
    MSER_TRY(gemm(g, s));
  } else {
    const float* w[3] = {d.w_qs, d.w_ks, d.w_vs};
    for (int i = 0; i < 3; ++i) {
      g.B = w[i]; g.C = d.qkv + i * nq; g.N = nq;
      MSER_TRY(gemm(g, s));
    }
  }
  const AttnArgs a = attn_args(d);
  const size_t lds_a = attn_lds_bytes(d.nl, d.dk, false);
  MSER_TRY(allow((const void*)attn_fwd_kernel, lds_a));
  {
    ProfScope ps(MSER_PROF_ENC_ATTN_FWD, s);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(d.nh, d.nb), dim3(ET), lds_a, s, a);
  }
  MSER_TRY(check_launch("attn_fwd_kernel"));
  const PostArgs p = post_args(d);
  const size_t lds_p = post_lds_bytes(d.D, nq, d.dff, false);
  MSER_TRY(allow((const void*)post_fwd_kernel, lds_p));
  hipLaunchKernelGGL(post_fwd_kernel, dim3(cdiv(rows, RT)), dim3(ET), lds_p, s, p);
  return check_launch("post_fwd_kernel");
}

// The layer's weight-gradient products dW[N, K] += dY^T X (A = dY^T, m contiguous; B = X, n contiguous; reduction over the rows,
// split-K float atomics) as descriptors: launched here as one group, or handed to the caller who batches them with the rest of
// the model's weight gradients (mser_gemm_grouped).  Valid once the MSER_ENC_BWD_ACT phase has been enqueued.
int encoder_layer_wgrad_descs(const mser_encoder_desc& d, mser_gemm_desc* out, int cap) {
  MSER_TRY(enc_validate(d, true));
  MSER_REQUIRE(d.g_w_qs && d.g_w_ks && d.g_w_vs && d.g_fc && d.g_w1 && d.g_w2, "mser_encoder_layer_bwd: null weight gradient");
  MSER_REQUIRE(out && cap >= 6, "mser_encoder_layer_wgrad_descs: need room for 6 descriptors");
  const int rows = d.nb * d.nl, nq = d.nh * d.dk, D = d.D, F = d.dff;
  int n = 0;
  auto wgrad = [&](const float* dY, long lddy, int N, const float* X, long ldx, int K, float* dW) {
    mser_gemm_desc g = gd0();
    g.A = dY; g.sAm = 1; g.sAk = lddy; g.B = X; g.sBk = ldx; g.sBn = 1; g.C = dW; g.ldc = K; g.M = N; g.N = K; g.K = rows;
    g.splitk = 16;
    out[n++] = g;
  };
  if (qkv_adjacent(d.w_qs, d.w_ks, d.w_vs, (long)nq * D) && qkv_adjacent(d.g_w_qs, d.g_w_ks, d.g_w_vs, (long)nq * D)) {
    wgrad(d.dqkv, 3L * nq, 3 * nq, d.x, D, D, d.g_w_qs);
  } else {
    float* gw[3] = {d.g_w_qs, d.g_w_ks, d.g_w_vs};
    for (int i = 0; i < 3; ++i) wgrad(d.dqkv + i * nq, 3L * nq, nq, d.x, D, D, gw[i]);
  }
  wgrad((d.rng && (d.p_fc > 0.f || d.p_ffn > 0.f)) ? d.dt1 : d.dy1, D, D, d.O, nq, nq, d.g_fc);
  wgrad(d.dh, F, F, d.e1, D, D, d.g_w1);
  wgrad(d.dy2, D, D, d.hdn, F, F, d.g_w2);
  return n;
}

int encoder_layer_bwd(const mser_encoder_desc& d, int phases, hipStream_t s) {
  MSER_TRY(enc_validate(d, true));
  const int rows = d.nb * d.nl, nq = d.nh * d.dk, D = d.D, F = d.dff;
  const bool adj = qkv_adjacent(d.w_qs, d.w_ks, d.w_vs, (long)nq * D);
  if (phases & MSER_ENC_BWD_ACT) {
    MSER_REQUIRE(d.g_ln1_g && d.g_ln1_b && d.g_b1 && d.g_b2 && d.g_ln2_g && d.g_ln2_b, "mser_encoder_layer_bwd: null bias / LayerNorm gradient");
    const PostArgs p = post_args(d);
    const size_t lds_p = post_lds_bytes(d.D, nq, d.dff, true);
    MSER_TRY(allow((const void*)post_bwd_kernel, lds_p));
    hipLaunchKernelGGL(post_bwd_kernel, dim3(cdiv(rows, RT)), dim3(ET), lds_p, s, p);
    MSER_TRY(check_launch("post_bwd_kernel"));
    const AttnArgs a = attn_args(d);
    const size_t lds_a = attn_lds_bytes(d.nl, d.dk, true);
    MSER_TRY(allow((const void*)attn_bwd_kernel, lds_a));
    {
      ProfScope ps(MSER_PROF_ENC_ATTN_BWD, s);
      hipLaunchKernelGGL(attn_bwd_kernel, dim3(d.nh, d.nb), dim3(ET), lds_a, s, a);
    }
    MSER_TRY(check_launch("attn_bwd_kernel"));
    // de0 = dy1 (residual) + dq Wq + dk Wk + dv Wv
    mser_gemm_desc g = gd0();
    g.C = d.dx; g.ldc = D; g.M = rows; g.N = D; g.sAk = 1; g.sBn = 1; g.sBk = D;
    g.R1 = d.dy1; g.ldr1 = D;
    if (adj) {
      g.A = d.dqkv; g.sAm = 3L * nq; g.K = 3 * nq; g.B = d.w_qs;
      MSER_TRY(gemm(g, s));
    } else {
      const float* w[3] = {d.w_qs, d.w_ks, d.w_vs};
      for (int i = 0; i < 3; ++i) {
        g.A = d.dqkv + i * nq; g.sAm = 3L * nq; g.K = nq; g.B = w[i];
        if (i > 0) { g.R1 = nullptr; g.flags = MSER_GEMM_ACCUM; }
        MSER_TRY(gemm(g, s));
      }
    }
  }
  if (phases & MSER_ENC_BWD_WGRAD) {
    mser_gemm_desc wg[6];
    const int nwg = encoder_layer_wgrad_descs(d, wg, 6);
    if (nwg < 0) return nwg;
    MSER_TRY(gemm_group(wg, nwg, s));        // one grouped launch for the layer's weight gradients
  }
  return 0;
}

static int tail_validate(const mser_head_tail_desc& d, bool bwd) {
  MSER_REQUIRE(d.L > 0 && d.B > 0 && d.D > 0 && d.F > 0 && d.C > 0, "mser_head_tail: bad sizes");
  MSER_REQUIRE(d.D % 4 == 0 && d.F % 4 == 0 && d.D <= 1024 && d.F <= 256 && d.C <= 32, "mser_head_tail: unsupported widths D=%d F=%d C=%d", d.D, d.F, d.C);
  MSER_REQUIRE(d.y1 && d.x_l && d.x_a && d.w0 && d.b0 && d.w3 && d.b3 && d.y1r && d.y2 && d.lp, "mser_head_tail: null pointer");
  MSER_REQUIRE(al16(d.w0) && al16(d.w3), "mser_head_tail: weights must be 16-byte aligned");
  MSER_REQUIRE(tail_lds_bytes(d.D, d.F, d.C, bwd) <= LDS_MAX, "mser_head_tail: row tile exceeds LDS");
  MSER_REQUIRE(!d.rng || (d.p_out >= 0.f && d.p_out < 1.f && d.p_fc >= 0.f && d.p_fc < 1.f), "mser_head_tail: dropout p out of [0,1)");
  if (bwd)
    MSER_REQUIRE(d.dlp && d.dy3 && d.dy2 && d.dy1 && d.dx_l && d.dx_a && d.g_b0 && d.g_b3 && d.g_bfc, "mser_head_tail_bwd: null gradient buffer");
  return 0;
}
static TailArgs tail_args(const mser_head_tail_desc& d) {
  TailArgs p;
  p.L = d.L; p.B = d.B; p.D = d.D; p.F = d.F; p.C = d.C;
  p.y1 = d.y1; p.x_l = d.x_l; p.x_a = d.x_a; p.W0 = d.w0; p.b0 = d.b0; p.W3 = d.w3; p.b3 = d.b3;
  p.y1r = d.y1r; p.y2 = d.y2; p.lp = d.lp;
  p.dlp = d.dlp; p.dxl_in = d.dx_l_in; p.dxa_in = d.dx_a_in;
  p.dy3 = d.dy3; p.dy2 = d.dy2; p.dy1 = d.dy1; p.dx_l = d.dx_l; p.dx_a = d.dx_a;
  p.g_b0 = d.g_b0; p.g_b3 = d.g_b3; p.g_bfc = d.g_bfc;
  p.rng = d.rng; p.site_out = d.site_out; p.p_out = d.p_out; p.scale_fc = 1.0f / (1.0f - d.p_fc);
  return p;
}
int head_tail(const mser_head_tail_desc& d, bool bwd, hipStream_t s) {
  MSER_TRY(tail_validate(d, bwd));
  const TailArgs p = tail_args(d);
  const size_t lds = tail_lds_bytes(d.D, d.F, d.C, bwd);
  const int grid = cdiv((long)d.L * d.B, RT);
  if (!bwd) {
    MSER_TRY(allow((const void*)tail_fwd_kernel, lds));
    hipLaunchKernelGGL(tail_fwd_kernel, dim3(grid), dim3(ET), lds, s, p);
    return check_launch("tail_fwd_kernel");
  }
  MSER_TRY(allow((const void*)tail_bwd_kernel, lds));
  hipLaunchKernelGGL(tail_bwd_kernel, dim3(grid), dim3(ET), lds, s, p);
  return check_launch("tail_bwd_kernel");
}

}  // namespace mser


extern "C" {

int mser_head_tail_fwd(const mser_head_tail_desc* d, mser_stream_t stream) {
  if (!d) { mser::set_error("mser_head_tail_fwd: null descriptor"); return -1; }
  return mser::head_tail(*d, false, (hipStream_t)stream);
}
int mser_head_tail_bwd(const mser_head_tail_desc* d, mser_stream_t stream) {
  if (!d) { mser::set_error("mser_head_tail_bwd: null descriptor"); return -1; }
  return mser::head_tail(*d, true, (hipStream_t)stream);
}
int mser_encoder_layer_supported(const mser_encoder_desc* d) {
  if (!d) return 0;
  return mser::enc_unsupported(*d) == nullptr ? 1 : 0;
}
int mser_encoder_layer_fwd(const mser_encoder_desc* d, mser_stream_t stream) {
  if (!d) { mser::set_error("mser_encoder_layer_fwd: null descriptor"); return -1; }
  return mser::encoder_layer_fwd(*d, (hipStream_t)stream);
}
int mser_encoder_layer_wgrad_descs(const mser_encoder_desc* d, mser_gemm_desc* out, int32_t cap) {
  if (!d) { mser::set_error("mser_encoder_layer_wgrad_descs: null descriptor"); return -1; }
  return mser::encoder_layer_wgrad_descs(*d, out, cap);
}
int mser_encoder_layer_bwd(const mser_encoder_desc* d, int32_t phases, mser_stream_t stream) {
  if (!d) { mser::set_error("mser_encoder_layer_bwd: null descriptor"); return -1; }
  return mser::encoder_layer_bwd(*d, phases, (hipStream_t)stream);
}

}  // extern "C"
