// Sequence-level cross-modal attention (reference model/lsthm_sps.py:88-101 CrossAttention2 and :116-129 CrossAttention3:
// softmax(Q K^T / sqrt(dk)) V across the utterance axis between the two modality streams) as ONE fused launch per direction of the
// autograd graph, after the projection GEMMs:
//
//   forward : one workgroup per (32-query tile, head, dialogue): the Q tile and K of the head in LDS -> S = scale Q K^T on the fp32
//             MFMA -> row softmax by wavefront butterflies (two columns per lane) -> dropout -> V over K's LDS buffer -> O = P V.
//             Nothing of size [B, L, L] goes to HBM: only the row statistics (max, 1 / sum) are saved.
//   backward: the same tiling; P is recomputed from Q, K and the statistics; dQ = dS K is owned by the tile.  The query tiles of a
//             dialogue meet in dV = sum Pd^T dO and dK = sum dS^T Q: with mser_xattn_desc::part_stride != 0 every tile STORES its
//             contribution into its own slab and a second launch adds the (<= 4) slabs in a fixed order -- deterministic, and plain
//             coalesced stores instead of 32 k float atomics per workgroup; part_stride == 0 keeps the atomic accumulation (the
//             caller zeroes dk / dv).  (An owner-computes form -- key-tile workgroups walking the query tiles, recomputing the
//             transposed blocks -- was built and measured: same 30 us alone, but twice the workgroups: 175 instead of 118 us beside
//             the BPTT launch, which leaves ~48 CUs free.  Not kept.)
//
// A (dialogue, head) has L <= 128 keys of width dk <= 128: K (and then V) take 66 KB of LDS, the query tile 17 KB, the score tile
// 17 KB -- 101 KB forward, 135 KB backward -- where the encoder's per-head kernel (csrc/encoder.hip), which keeps q, k, v and the
// whole L x L tile of a 40-wide head resident, would need 270 KB at dk = 128.  Tiling the queries also gives 4x the workgroups
// (128 at B = 32, L = 128).  All products are v_mfma_f32_32x32x2_f32 chains (exact fp32: the parity gate is 1e-4 on log-probs).
// MFMA operand conventions as in csrc/encoder.hip.
#include "common.h"
#include "../../include/mser.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace mser {
namespace {

constexpr int XT = 1024, XW = 16;       // threads / waves per workgroup
constexpr int QT = 32;                  // query rows per workgroup

__device__ __forceinline__ f32x16 mfma4(const float4& a, const float4& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }
// A rows (float4 along k) x B rows (float4 along k)
__device__ __forceinline__ f32x16 tile_rr(const float* Arow, const float* Brow, int K, f32x16 acc) {
  for (int kc = 0; kc < K; kc += 8)
    acc = mfma4(*reinterpret_cast<const float4*>(Arow + kc), *reinterpret_cast<const float4*>(Brow + kc), acc);
  return acc;
}
// A rows (float4 along k) x B stored [k][n]
__device__ __forceinline__ f32x16 tile_rc(const float* Arow, const float* Bcol, int ldb, int K, bool ok, f32x16 acc) {
  for (int kc = 0; kc < K; kc += 8) {
    float4 b;
    b.x = Bcol[(kc + 0) * ldb]; b.y = Bcol[(kc + 1) * ldb]; b.z = Bcol[(kc + 2) * ldb]; b.w = Bcol[(kc + 3) * ldb];
    if (!ok) b = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = mfma4(*reinterpret_cast<const float4*>(Arow + kc), b, acc);
  }
  return acc;
}
// A stored [k][m] x B stored [k][n]
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

struct XArgs {
  const float* q; const float* k; const float* v; long ldq, ldk, ldv;
  long sbq, slq, sbk, slk;
  float* o; long ldo;
  float* stats;                        // [nb, nh, Lq, 2]
  const float* dO; long lddo;
  float* dq; float* dk_; float* dv; long lddq, lddk, lddv;
  long part_stride;                    // floats between the per-query-tile slabs of dk_ / dv (0: accumulate atomically)
  int nb, nh, Lq, Lk, dk;
  float scale;
  const uint32_t* rng; uint32_t site; float p;
};

// rows [r0, r0 + nrows) of a [*, ld] view, columns col0 .. col0 + dk, into dst [nrows_padded][SD]; rows >= lim read as zero
__device__ __forceinline__ void stage_rows(const float* src, long ld, long sb, long sl, int b, int r0, int nrows, int lim, int col0, int dk,
                                           int SD, float* dst) {
  const int vpr = dk >> 2;
  for (int e = threadIdx.x; e < nrows * vpr; e += XT) {
    const int l = e / vpr, c4 = (e - l * vpr) << 2;
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + l < lim) v4 = *reinterpret_cast<const float4*>(src + ((long)b * sb + (long)(r0 + l) * sl) * ld + col0 + c4);
    *reinterpret_cast<float4*>(dst + l * SD + c4) = v4;
  }
}

// S tile [32][LPk] = Qt K^T (unscaled), one wave per 32-column tile
__device__ __forceinline__ void score_tile(const float* Qt, const float* Ks, int SD, int dk, int ntk, float* S, int SS) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
  if (wave < ntk) {
    f32x16 acc = {0};
    acc = tile_rr(Qt + r * SD + half * 4, Ks + (wave * 32 + r) * SD + half * 4, dk, acc);
#pragma unroll
    for (int i = 0; i < 16; ++i) S[acc_row(i, half) * SS + wave * 32 + r] = acc[i];
  }
}

__global__ __launch_bounds__(XT) void xattn_fwd_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int Lq = a.Lq, Lk = a.Lk, LPk = (Lk + 31) & ~31, dk = a.dk, SD = dk + 4, SS = LPk + 4;
  const int SP = SD > SS ? SD : SS;            // the query tile's buffer later holds P
  float* Qt = sm;                              // [32][SP]
  float* KV = Qt + QT * SP;                    // [LPk][SD]
  float* S = KV + LPk * SD;                    // [32][SS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
  const int q0 = qt * QT;
  stage_rows(a.q, a.ldq, a.sbq, a.slq, b, q0, QT, Lq, h * dk, dk, SD, Qt);
  stage_rows(a.k, a.ldk, a.sbk, a.slk, b, 0, LPk, Lk, h * dk, dk, SD, KV);
  __syncthreads();
  const int ntk = LPk >> 5;
  score_tile(Qt, KV, SD, dk, ntk, S, SS);
  __syncthreads();
  // ---- V over K's buffer while the softmax runs on S (disjoint LDS regions; both need only the barrier above)
  stage_rows(a.v, a.ldv, a.sbk, a.slk, b, 0, LPk, Lk, h * dk, dk, SD, KV);
  // ---- row softmax: one wave per row, two columns per lane; P (dropped) goes to the query tile's buffer with row stride SS
  float* Pt = Qt;
  DropKey dkey;
  if (a.rng) dkey = drop_key(a.rng, a.site, a.p);
  for (int row = wave; row < QT; row += XW) {
    const int gq = q0 + row;
    const bool has1 = lane + 64 < LPk;
    float p0 = 0.f, p1 = 0.f;
    if (gq < Lq) {
      const float v0 = lane < Lk ? S[row * SS + lane] * a.scale : -INFINITY;
      const float v1 = lane + 64 < Lk ? S[row * SS + lane + 64] * a.scale : -INFINITY;
      const float m = wave_max(fmaxf(v0, v1));
      const float e0 = expf(v0 - m), e1 = expf(v1 - m);
      const float inv = 1.0f / wave_sum(e0 + e1);
      p0 = e0 * inv; p1 = e1 * inv;
      if (lane == 0) {
        float* st = a.stats + (((long)b * a.nh + h) * Lq + gq) * 2;
        st[0] = m; st[1] = inv;
      }
      if (a.rng) {
        const uint32_t e0i = (uint32_t)((((long)b * a.nh + h) * Lq + gq) * Lk);
        p0 *= drop_scale(dkey, e0i + (uint32_t)lane);
        p1 *= drop_scale(dkey, e0i + (uint32_t)(lane + 64));
      }
    }
    // (Pt aliases the query tile: every wave finished reading Qt behind the barrier that follows score_tile)
    if (lane < LPk) Pt[row * SS + lane] = p0;
    if (has1) Pt[row * SS + lane + 64] = p1;
  }
  __syncthreads();
  // ---- O tile = P V: one wave per 32-column tile of the head
  const int ntn = (dk + 31) >> 5;
  if (wave < ntn) {
    const int col = wave * 32 + r;
    const bool cok = col < dk;
    f32x16 acc = {0};
    acc = tile_rc(Pt + r * SS + half * 4, KV + (half * 4) * SD + (cok ? col : 0), SD, LPk, cok, acc);
    if (cok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int gq = q0 + acc_row(i, half);
        if (gq < Lq) a.o[((long)b * a.sbq + (long)gq * a.slq) * a.ldo + h * dk + col] = acc[i];
      }
    }
  }
}

// dV += Pd^T dO, dP = dO V^T, dS = scale P o (mask o dP - delta), delta_i = <dO_i, O_i>, dQ = dS K, dK += dS^T Q
__global__ __launch_bounds__(XT) void xattn_bwd_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int Lq = a.Lq, Lk = a.Lk, LPk = (Lk + 31) & ~31, dk = a.dk, SD = dk + 4, SS = LPk + 4;
  float* Qt = sm;                              // [32][SD]
  float* dOt = Qt + QT * SD;                   // [32][SD]
  float* KV = dOt + QT * SD;                   // [LPk][SD]
  float* Pt = KV + LPk * SD;                   // [32][SS]   plain softmax
  float* Xt = Pt + QT * SS;                    // [32][SS]   dropped P, then dP, then dS
  float* delta = Xt + QT * SS;                 // [32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
  const int q0 = qt * QT;
  stage_rows(a.q, a.ldq, a.sbq, a.slq, b, q0, QT, Lq, h * dk, dk, SD, Qt);
  stage_rows(a.dO, a.lddo, a.sbq, a.slq, b, q0, QT, Lq, h * dk, dk, SD, dOt);
  stage_rows(a.k, a.ldk, a.sbk, a.slk, b, 0, LPk, Lk, h * dk, dk, SD, KV);
  for (int row = wave; row < QT; row += XW) {         // delta_i = <dO_i, O_i> (= sum_j Pd_ij dPd_ij)
    const int gq = q0 + row;
    float d = 0.f;
    if (gq < Lq) {
      const long g = (long)b * a.sbq + (long)gq * a.slq;
      for (int c = lane; c < dk; c += 64) d = fmaf(a.dO[g * a.lddo + h * dk + c], a.o[g * a.ldo + h * dk + c], d);
    }
    d = wave_sum(d);
    if (lane == 0) delta[row] = d;
  }
  __syncthreads();
  const int ntk = LPk >> 5, ntn = (dk + 31) >> 5;
  score_tile(Qt, KV, SD, dk, ntk, Pt, SS);
  __syncthreads();
  // ---- V over K's buffer; P from the saved statistics (plain -> Pt, dropped -> Xt)
  stage_rows(a.v, a.ldv, a.sbk, a.slk, b, 0, LPk, Lk, h * dk, dk, SD, KV);
  DropKey dkey;
  if (a.rng) dkey = drop_key(a.rng, a.site, a.p);
  for (int row = wave; row < QT; row += XW) {
    const int gq = q0 + row;
    float m = 0.f, inv = 0.f;
    if (gq < Lq) { const float* st = a.stats + (((long)b * a.nh + h) * Lq + gq) * 2; m = st[0]; inv = st[1]; }
    const uint32_t e0i = (uint32_t)((((long)b * a.nh + h) * Lq + gq) * Lk);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = lane + 64 * u;
      if (col < LPk) {
        float pv = 0.f;
        if (gq < Lq && col < Lk) pv = expf(Pt[row * SS + col] * a.scale - m) * inv;
        Pt[row * SS + col] = pv;
        Xt[row * SS + col] = (a.rng && gq < Lq && col < Lk) ? pv * drop_scale(dkey, e0i + (uint32_t)col) : pv;
      }
    }
  }
  __syncthreads();
  // ---- dV += Pd^T dO (tiles [key tile][column tile], K = the 32 query rows) on waves 0 .. ntk*ntn-1; dP = dO V^T on the next ntk waves
  //      (its tile stays in registers until the dropped P has been consumed)
  f32x16 dp = {0};
  const int ndv = ntk * ntn;
  if (wave < ndv) {
    const int ti = wave / ntn, tn = wave - ti * ntn;
    const int col = tn * 32 + r;
    const bool cok = col < dk;
    f32x16 acc = {0};
    acc = tile_cc(Xt + (half * 4) * SS + ti * 32 + r, SS, dOt + (half * 4) * SD + (cok ? col : 0), SD, QT, cok, acc);
    if (cok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = ti * 32 + acc_row(i, half);
        if (key < Lk) {
          float* dst = a.dv + ((long)b * a.sbk + (long)key * a.slk) * a.lddv + h * dk + col;
          if (a.part_stride) dst[(long)qt * a.part_stride] = acc[i];
          else atomicAdd(dst, acc[i]);
        }
      }
    }
  }
  // (ndv + ntk <= 16 does not hold at dk = 128, L = 128: the dP tiles then follow in a second round on waves 0 .. ntk-1)
  const bool dp_same_round = ndv + ntk <= XW;
  if (dp_same_round && wave >= ndv && wave < ndv + ntk) {
    const int tj = wave - ndv;
    dp = tile_rr(dOt + r * SD + half * 4, KV + (tj * 32 + r) * SD + half * 4, dk, dp);
  }
  if (!dp_same_round && wave < ntk) dp = tile_rr(dOt + r * SD + half * 4, KV + (wave * 32 + r) * SD + half * 4, dk, dp);
  __syncthreads();                      // the dropped P (Xt) and V have been consumed
  {
    const int tj = dp_same_round ? wave - ndv : wave;
    if (tj >= 0 && tj < ntk) {
      const int col = tj * 32 + r;
      const uint32_t* rng = a.rng;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = acc_row(i, half);
        const int gq = q0 + row;
        float g = dp[i];
        if (rng && gq < Lq && col < Lk) g *= drop_scale(dkey, (uint32_t)((((long)b * a.nh + h) * Lq + gq) * Lk + col));
        Xt[row * SS + col] = a.scale * Pt[row * SS + col] * (g - delta[row]);
      }
    }
  }
  stage_rows(a.k, a.ldk, a.sbk, a.slk, b, 0, LPk, Lk, h * dk, dk, SD, KV);
  __syncthreads();
  // ---- dQ = dS K on waves 0 .. ntn-1 (tile owned by this workgroup); dK += dS^T Q on the following ntk*ntn waves (second round if needed)
  for (int it = wave; it < ntn + ntk * ntn; it += XW) {
    if (it < ntn) {
      const int col = it * 32 + r;
      const bool cok = col < dk;
      f32x16 acc = {0};
      acc = tile_rc(Xt + r * SS + half * 4, KV + (half * 4) * SD + (cok ? col : 0), SD, LPk, cok, acc);
      if (cok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int gq = q0 + acc_row(i, half);
          if (gq < Lq) a.dq[((long)b * a.sbq + (long)gq * a.slq) * a.lddq + h * dk + col] = acc[i];
        }
      }
    } else {
      const int t = it - ntn;
      const int ti = t / ntn, tn = t - ti * ntn;
      const int col = tn * 32 + r;
      const bool cok = col < dk;
      f32x16 acc = {0};
      acc = tile_cc(Xt + (half * 4) * SS + ti * 32 + r, SS, Qt + (half * 4) * SD + (cok ? col : 0), SD, QT, cok, acc);
      if (cok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = ti * 32 + acc_row(i, half);
          if (key < Lk) {
            float* dst = a.dk_ + ((long)b * a.sbk + (long)key * a.slk) * a.lddk + h * dk + col;
            if (a.part_stride) dst[(long)qt * a.part_stride] = acc[i];
            else atomicAdd(dst, acc[i]);
          }
        }
      }
    }
  }
}

// slab 0 += slab 1 + slab 2 + ... (fixed order) for the dK and dV views: rows x width each
__global__ void xattn_reduce_kernel(float* dk_, long lddk, float* dv, long lddv, long rows, int width, long part_stride, int nslab) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w4 = width >> 2;
  if (i >= rows * w4) return;
  const long r = i / w4;
  const int c = (int)(i - r * w4) << 2;
  float* base = (blockIdx.y ? dv + r * lddv : dk_ + r * lddk) + c;
  float4 s4 = *reinterpret_cast<const float4*>(base);
  for (int k = 1; k < nslab; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(base + (long)k * part_stride);
    s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
  }
  *reinterpret_cast<float4*>(base) = s4;
}

size_t xattn_lds_bytes(int Lk, int dk, bool bwd) {
  const size_t LPk = (size_t)((Lk + 31) & ~31), SD = dk + 4, SS = LPk + 4, SP = SD > SS ? SD : SS;
  return (bwd ? 2 * QT * SD + LPk * SD + 2 * QT * SS + QT : QT * SP + LPk * SD + QT * SS) * sizeof(float);
}

const char* xattn_unsupported(const mser_xattn_desc& d) {
  if (d.nb <= 0 || d.Lq <= 0 || d.Lk <= 0 || d.nh <= 0) return "empty shape";
  if (d.Lk > 128) return "more than 128 keys (K / V of one head must fit LDS)";
  if (d.dk % 8 || d.dk <= 0 || d.dk > 128) return "head width must be a multiple of 8 and <= 128";
  if (((d.Lk + 31) / 32) * ((d.dk + 31) / 32) > 16) return "too many dV / dK tiles for one round";
  if (xattn_lds_bytes(d.Lk, d.dk, true) > 160 * 1024) return "tiles exceed LDS";
  return nullptr;
}
bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

XArgs xargs(const mser_xattn_desc& d) {
  XArgs a;
  memset(&a, 0, sizeof(a));
  a.q = d.q; a.k = d.k; a.v = d.v; a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv;
  a.sbq = d.sbq; a.slq = d.slq; a.sbk = d.sbk; a.slk = d.slk;
  a.o = d.o; a.ldo = d.ldo; a.stats = d.stats;
  a.dO = d.dO; a.lddo = d.lddo; a.dq = d.dq; a.dk_ = d.dk_; a.dv = d.dv; a.lddq = d.lddq; a.lddk = d.lddk; a.lddv = d.lddv;
  a.part_stride = d.part_stride;
  a.nb = d.nb; a.nh = d.nh; a.Lq = d.Lq; a.Lk = d.Lk; a.dk = d.dk; a.scale = d.scale;
  a.rng = (d.rng && d.p > 0.f) ? d.rng : nullptr; a.site = d.site; a.p = d.p;
  return a;
}

int allow(const void* kernel, size_t bytes) {
  if (bytes > 64 * 1024) MSER_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

int validate(const mser_xattn_desc& d, bool bwd) {
  const char* why = xattn_unsupported(d);
  MSER_REQUIRE(!why, "mser_xattn_seq: unsupported configuration: %s", why ? why : "");
  MSER_REQUIRE(d.q && d.k && d.v && d.o && d.stats, "mser_xattn_seq: null pointer");
  MSER_REQUIRE(al16(d.q) && al16(d.k) && al16(d.v) && d.ldq % 4 == 0 && d.ldk % 4 == 0 && d.ldv % 4 == 0,
               "mser_xattn_seq: q / k / v must be 16-byte aligned with leading dimensions that are multiples of 4");
  MSER_REQUIRE(!d.rng || (d.p >= 0.f && d.p < 1.f), "mser_xattn_seq: dropout p=%f", d.p);
  if (bwd) {
    MSER_REQUIRE(d.dO && d.dq && d.dk_ && d.dv, "mser_xattn_seq_bwd: null gradient buffer");
    MSER_REQUIRE(al16(d.dO) && d.lddo % 4 == 0, "mser_xattn_seq_bwd: dO must be 16-byte aligned, lddo a multiple of 4");
    if (d.part_stride != 0)
      MSER_REQUIRE(d.part_stride > 0 && d.part_stride % 4 == 0 && al16(d.dk_) && al16(d.dv) && d.lddk % 4 == 0 && d.lddv % 4 == 0 && (d.nh * d.dk) % 4 == 0,
                   "mser_xattn_seq_bwd: slabs need 16-byte aligned dk_ / dv, leading dimensions and part_stride that are multiples of 4");
  }
  return 0;
}

}  // namespace
}  // namespace mser

using namespace mser;

extern "C" {

int mser_xattn_seq_supported(const mser_xattn_desc* d) { return d && !xattn_unsupported(*d) ? 1 : 0; }

int mser_xattn_seq_fwd(const mser_xattn_desc* d, mser_stream_t stream) {
  if (!d) { set_error("mser_xattn_seq_fwd: null descriptor"); return -1; }
  MSER_TRY(validate(*d, false));
  const size_t lds = xattn_lds_bytes(d->Lk, d->dk, false);
  MSER_TRY(allow((const void*)xattn_fwd_kernel, lds));
  ProfScope ps(MSER_PROF_XATTN_FWD, (hipStream_t)stream);
  hipLaunchKernelGGL(xattn_fwd_kernel, dim3((d->Lq + QT - 1) / QT, d->nh, d->nb), dim3(XT), lds, (hipStream_t)stream, xargs(*d));
  return check_launch("mser_xattn_seq_fwd");
}

int mser_xattn_seq_bwd(const mser_xattn_desc* d, mser_stream_t stream) {
  if (!d) { set_error("mser_xattn_seq_bwd: null descriptor"); return -1; }
  MSER_TRY(validate(*d, true));
  const size_t lds = xattn_lds_bytes(d->Lk, d->dk, true);
  MSER_TRY(allow((const void*)xattn_bwd_kernel, lds));
  ProfScope ps(MSER_PROF_XATTN_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL(xattn_bwd_kernel, dim3((d->Lq + QT - 1) / QT, d->nh, d->nb), dim3(XT), lds, (hipStream_t)stream, xargs(*d));
  MSER_TRY(check_launch("mser_xattn_seq_bwd"));
  const int nslab = (d->Lq + QT - 1) / QT;
  if (d->part_stride != 0 && nslab > 1) {
    // the rows the launch wrote: every key row of every dialogue, i.e. the whole [rows, nh*dk] views (rows = the views' row count)
    long rows = 0;
    for (int bb = 0; bb < 2; ++bb) rows = std::max(rows, (long)(d->nb - 1) * d->sbk + (long)(d->Lk - 1) * d->slk + 1);
    const int width = d->nh * d->dk;
    hipLaunchKernelGGL(xattn_reduce_kernel, dim3((unsigned)((rows * (width / 4) + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, d->dk_,
                       (long)d->lddk, d->dv, (long)d->lddv, rows, width, (long)d->part_stride, nslab);
    return check_launch("mser_xattn_seq_bwd (slab sum)");
  }
  return 0;
}

}  // extern "C"
