// Generic strided batched fp32 GEMM for gfx950 on v_mfma_f32_32x32x2_f32.
//
// Why fp32 MFMA: the parity gate is 1e-4 on log-probs after a 128-step recurrence, so the path computes in exact
// fp32; on CDNA4 the f32 MFMA is a k-ordered fmaf chain (bit-identical to a scalar fp32 loop) at the f32 vector peak
// but needs one VGPR per operand and leaves the VALU free for the epilogue (guide: "FP32-input MFMA").
//
// The matrices on this path are small (M = B*L = 4096 rows, N <= 1280, K <= 1280, or a 4096-long split reduction into a
// small weight-gradient) and the fp32 MFMA is paced per SIMD (64 cycles per 32x32x2), so a call is bound by (a) how many of
// the chip's 1024 SIMDs it keeps busy and (b) the fixed cost per k-tile around the MFMA chain.  Hence:
//  * two tile configurations, both 4 waves / 256 threads with ONE 32x32 accumulator per wave:
//      C0  64x64 tile, waves 2x2,            BK = 32  -- grids that already fill the chip
//      C1  64x32 tile, waves 2x1 x 2 k-halves, BK = 64  -- twice the workgroups for the many M=4096, N<=128 products
//    (the k-halves of C1 are summed through LDS before the epilogue);  split-K over workgroups (float atomics) for the
//    K = 4096 weight-gradient reductions, with the split chosen here so that the grid fills the chip;
//  * k-major LDS tiles [k][rows + 2]: operand reads are 32 consecutive words (conflict-free ds_read_b32, all 32 fetched
//    before the MFMA chain starts), and the staging stores of all three load modes are at worst 2-way conflicted
//    (the previous [parity][row][k/2] layout made the row-contiguous mode 8-way conflicted: 2.7 us per k-tile);
//  * a branch-free software pipeline: global loads run TWO tiles ahead of the MFMAs (register staging), issued
//    unconditionally with clamped addresses -- a load under a condition gets copies (and a wait) right behind it.
// Generality (any strides, two batch levels, fused bias/ReLU/residual epilogue) is kept because every transposed / strided /
// head-interleaved product of the forward and backward pass goes through this one kernel.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "../../include/mser.h"

namespace mser {

constexpr int GEMM_ATOMIC = 1 << 8;   // internal flag: accumulate with float atomics even without a split (grouped launches)

struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;
  long sAm, sAk, sBk, sBn, ldc;
  int batch2, splitk, kchunk;
  long sA1, sA2, sB1, sB2, sC1, sC2;
  const float* bias; const float* alpha_dev; float alpha; int flags;
  const float* R1; const float* R2; long ldr1, ldr2, sR1, sR2;
};

// One operand tile (TR rows/cols x BKT k) staged global -> registers -> LDS (k-major [BKT][TR + 2]).
// MODE 0: k contiguous, float4 along k (host guarantees 16-B alignment, stride % 4 == 0, K-chunk % 4 == 0)
// MODE 1: m/n contiguous, float4 along m/n (alignment, stride % 4 == 0, extent % 4 == 0)
// MODE 2: any strides, scalar loads.
// Loads are UNCONDITIONAL: addresses are clamped into the matrix and the value is zeroed by a select at the LDS store -- a
// load inside a per-element branch makes hipcc wait vmcnt(0) per element (serial memory round trips).
template <int MODE, int TR, int BKT>
struct TileLoader {
  static constexpr int NF = TR * BKT / 256;            // floats per thread
  static constexpr int NV = (MODE == 2) ? NF : NF / 4; // register slots: scalars or float4
  static constexpr int LD = TR + 2;
  const float* p[NV];
  int mm[NV], kk[NV];
  bool ok[NV];
  long sk;                                             // element stride along k

  __device__ __forceinline__ void init(const float* base, long sm, long sk_, int m0, int M) {
    sk = sk_;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * 256;
      if (MODE == 0) { mm[i] = e / (BKT / 4); kk[i] = (e % (BKT / 4)) * 4; }
      else if (MODE == 1) { mm[i] = (e % (TR / 4)) * 4; kk[i] = e / (TR / 4); }
      else { mm[i] = e / BKT; kk[i] = e % BKT; }
      ok[i] = (m0 + mm[i]) < M;
      p[i] = base + (long)(ok[i] ? m0 + mm[i] : 0) * sm + (long)kk[i] * sk;
    }
  }
  // load() ONLY issues the loads (no use of the data: any use would make the compiler wait for them right here and defeat the
  // two-tile prefetch); invalid rows / k are zeroed in store().
  __device__ __forceinline__ void load(int k0, int klen, float* r) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k = k0 + kk[i];
      // clamped so that the (possibly 4-wide) access stays inside [0, klen): vector modes have klen % 4 == 0
      const int kc = (k < klen ? k : klen - (MODE == 0 ? 4 : 1)) - kk[i];
      const float* q = p[i] + (long)kc * sk;
      if (MODE == 2) {
        r[i] = *q;
      } else {
        const float4 v = *reinterpret_cast<const float4*>(q);
        r[4 * i + 0] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
      }
    }
  }
  // T: [BKT][LD]; element (k, m) lives at T[k * LD + m]
  __device__ __forceinline__ void store(float* T, const float* r, int k0, int klen) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool good = ok[i] && (k0 + kk[i]) < klen;     // vectors never straddle the valid range (host checks % 4)
      if (MODE == 0) {            // k .. k+3 of one row
#pragma unroll
        for (int j = 0; j < 4; ++j) T[(kk[i] + j) * LD + mm[i]] = good ? r[4 * i + j] : 0.f;
      } else if (MODE == 1) {     // rows m .. m+3 at one k (8-byte aligned: LD even, m % 4 == 0)
        float2* d = reinterpret_cast<float2*>(&T[kk[i] * LD + mm[i]]);
        d[0] = make_float2(good ? r[4 * i + 0] : 0.f, good ? r[4 * i + 1] : 0.f);
        d[1] = make_float2(good ? r[4 * i + 2] : 0.f, good ? r[4 * i + 3] : 0.f);
      } else {
        T[kk[i] * LD + mm[i]] = good ? r[i] : 0.f;
      }
    }
  }
};

// WM x WN waves side by side, KS = 4 / (WM * WN) k-slices of every BK tile (one per remaining wave).
template <int AMODE, int BMODE, int WM, int WN>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz) {
  constexpr int KS = 4 / (WM * WN), TM = 32 * WM, TN = 32 * WN, BKT = 32 * KS;
  using LA = TileLoader<AMODE, TM, BKT>;
  using LB = TileLoader<BMODE, TN, BKT>;
  constexpr int ASZ = BKT * LA::LD, BSZ = BKT * LB::LD;
  __shared__ __attribute__((aligned(16))) float As[2][ASZ];
  __shared__ __attribute__((aligned(16))) float Bs[2][BSZ];
  static_assert(KS == 1 || 2 * ASZ >= (KS - 1) * WM * WN * 1024, "k-slice reduction reuses the A stage buffers");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kh = wave / (WM * WN), wt = wave % (WM * WN);
  const int wr = wt / WN, wc = wt % WN;
  const int m0 = by * TM, n0 = bx * TN;
  int z = bz;
  const int ks = z % g.splitk; z /= g.splitk;
  const int z2 = z % g.batch2, z1 = z / g.batch2;
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int klen = kend - kbeg;

  LA la;
  LB lb;
  la.init(g.A + z1 * g.sA1 + z2 * g.sA2 + (long)kbeg * g.sAk, g.sAm, g.sAk, m0, g.M);
  lb.init(g.B + z1 * g.sB1 + z2 * g.sB2 + (long)kbeg * g.sBk, g.sBn, g.sBk, n0, g.N);

  f32x16 acc = {0};
  const int nt = (klen + BKT - 1) / BKT;
  const int half = lane >> 5, l31 = lane & 31;
  // all 16 A and 16 B operands of this wave's k-slice are fetched BEFORE the MFMA chain (just-in-time LDS reads expose the
  // LDS latency once per MFMA)
  auto compute = [&](int buf) {
    float a[16], b[16];
    const float* pa = &As[buf][(kh * 32 + half) * LA::LD + wr * 32 + l31];
    const float* pb = &Bs[buf][(kh * 32 + half) * LB::LD + wc * 32 + l31];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = pa[2 * i * LA::LD]; b[i] = pb[2 * i * LB::LD]; }
    // straight-line chain of 16 MFMAs (a short last tile multiplies stored zeros: cheaper than a branch per MFMA)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
  };

  // Branch-free pipeline: tile t lives in LDS buffer t&1; register set (t+1)&1 holds tile t+1; tile t+2 is requested into the
  // freed set.  Out-of-range tiles are loaded from clamped addresses, stored as zeros and multiply as zeros.
  float ra0[LA::NF], rb0[LB::NF], ra1[LA::NF], rb1[LB::NF];
  la.load(0, klen, ra0); lb.load(0, klen, rb0);
  la.load(BKT, klen, ra1); lb.load(BKT, klen, rb1);
  la.store(As[0], ra0, 0, klen); lb.store(Bs[0], rb0, 0, klen);
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    la.load((t + 2) * BKT, klen, ra0); lb.load((t + 2) * BKT, klen, rb0);
    compute(0);
    la.store(As[1], ra1, (t + 1) * BKT, klen); lb.store(Bs[1], rb1, (t + 1) * BKT, klen);
    __syncthreads();
    la.load((t + 3) * BKT, klen, ra1); lb.load((t + 3) * BKT, klen, rb1);
    compute(1);                       // odd tile count: the last pass multiplies a zero tile
    la.store(As[0], ra0, (t + 2) * BKT, klen); lb.store(Bs[0], rb0, (t + 2) * BKT, klen);
    __syncthreads();
  }

  if (KS > 1) {       // sum the k-slices: slices 1.. park their accumulators in LDS (stage buffers are dead by now)
    float* red = &As[0][0];
    if (kh > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((kh - 1) * WM * WN + wt) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh > 0) return;
#pragma unroll
    for (int q = 0; q < KS - 1; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[((q * WM * WN + wt) * 16 + r) * 64 + lane];
  }

  // epilogue. C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float alpha = g.alpha;
  if (g.alpha_dev) alpha *= *g.alpha_dev;
  float* C = g.C + z1 * g.sC1 + z2 * g.sC2;
  const float* R1 = g.R1 ? g.R1 + z1 * g.sR1 + z2 * g.sR2 : nullptr;
  const float* R2 = g.R2 ? g.R2 + z1 * g.sR1 + z2 * g.sR2 : nullptr;
  const int n = n0 + wc * 32 + (lane & 31);
  if (n >= g.N) return;
  const float bias = (g.bias && ks == 0) ? g.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= g.M) continue;
    float v = alpha * acc[r] + bias;
    if (g.flags & MSER_GEMM_RELU) v = fmaxf(v, 0.f);
    if (ks == 0) {
      if (R1) v += R1[(long)m * g.ldr1 + n];
      if (R2) v += R2[(long)m * g.ldr2 + n];
    }
    float* dst = C + (long)m * g.ldc + n;
    if (g.splitk > 1 || (g.flags & GEMM_ATOMIC)) atomicAdd(dst, v);
    else if (g.flags & MSER_GEMM_ACCUM) *dst += v;
    else *dst = v;
  }
}

template <int AMODE, int BMODE, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  gemm_body<AMODE, BMODE, WM, WN>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped launch: up to MAXG independent products (same load modes and tile configuration) share ONE grid.  The many small
// weight-gradient products of a training step (M, N <= 1280, K = B*L rows) are each far too small to fill the chip and cost a
// launch apiece; as one grid they fill it together, which also lets every member use a SMALLER split-K (fewer float atomics).
// Workgroup -> (member, tile) through a prefix table in the kernel arguments (uniform scalar search).
constexpr int MAXG = 16;
struct GroupArgs {
  int n;
  int start[MAXG + 1];
  int gx[MAXG], gy[MAXG];
  GemmArgs a[MAXG];
};
static_assert(sizeof(GroupArgs) <= 4096, "kernel arguments are limited to 4 KB");

template <int AMODE, int BMODE, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_group_kernel(GroupArgs G) {
  const int w = blockIdx.x;
  int p = 0;
#pragma unroll 1
  while (p + 1 < G.n && w >= G.start[p + 1]) ++p;
  const int local = w - G.start[p];
  const int gx = G.gx[p], gy = G.gy[p];
  const int bx = local % gx, t = local / gx;
  gemm_body<AMODE, BMODE, WM, WN>(G.a[p], bx, t % gy, t / gy);
}

template <int AM, int BM_, int WM, int WN>
static void launch_one(dim3 grid, hipStream_t s, const GemmArgs& a) {
  hipLaunchKernelGGL((gemm_kernel<AM, BM_, WM, WN>), grid, dim3(256), 0, s, a);
}
template <int WM, int WN>
static void launch_modes(int amode, int bmode, dim3 grid, hipStream_t s, const GemmArgs& a) {
  switch (amode * 3 + bmode) {
    case 0: launch_one<0, 0, WM, WN>(grid, s, a); break;
    case 1: launch_one<0, 1, WM, WN>(grid, s, a); break;
    case 2: launch_one<0, 2, WM, WN>(grid, s, a); break;
    case 3: launch_one<1, 0, WM, WN>(grid, s, a); break;
    case 4: launch_one<1, 1, WM, WN>(grid, s, a); break;
    case 5: launch_one<1, 2, WM, WN>(grid, s, a); break;
    case 6: launch_one<2, 0, WM, WN>(grid, s, a); break;
    case 7: launch_one<2, 1, WM, WN>(grid, s, a); break;
    default: launch_one<2, 2, WM, WN>(grid, s, a); break;
  }
}

struct Plan {
  GemmArgs a;
  dim3 grid;
  int amode, bmode;
  bool c1;
};

// group_tiles = 64x64 tiles of all the products that will share the launch (0: this product alone)
static int plan(const mser_gemm_desc& d, long group_tiles, Plan& P) {
  MSER_REQUIRE(d.A && d.B && d.C, "mser_gemm: null operand");
  MSER_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, "mser_gemm: negative size");
  const int b1 = d.batch1 > 0 ? d.batch1 : 1, b2 = d.batch2 > 0 ? d.batch2 : 1;
  MSER_REQUIRE(!(d.splitk > 1 && (d.flags & MSER_GEMM_RELU)), "mser_gemm: split-K cannot fuse ReLU");
  // ---- tile configuration: C0 (64x64) when it fills the chip on its own, otherwise C1 (64x32, two k-slices per workgroup)
  constexpr long FILL = 256;                         // one workgroup per CU
  const long tiles0 = (long)cdiv(d.M, 64) * cdiv(d.N, 64) * b1 * b2;
  const long fill0 = group_tiles > 0 ? group_tiles : tiles0;
  const bool c1 = fill0 * (d.splitk > 1 ? 4 : 1) < FILL + FILL / 2 || d.N <= 32;
  const int TM = 64, TN = c1 ? 32 : 64, BKT = c1 ? 64 : 32;
  const long tiles = (long)cdiv(d.M, TM) * cdiv(d.N, TN) * b1 * b2;
  // ---- split-K: `splitk > 1` means "C is initialised, accumulate atomically"; the split itself is chosen here so that the grid
  // (of the whole group) is about two workgroups per CU (four for a product on its own) while every workgroup still runs at least two k-tiles
  int splitk = 1;
  if (d.splitk > 1 && d.K > 0) {
    const long all = group_tiles > 0 ? group_tiles * (c1 ? 2 : 1) : tiles;
    // a product on its own: four workgroups per CU (two left 288-tile weight gradients at 2.25 workgroups per CU -- three rounds for
    // 2.25 rounds of work: DialogueRNN's weight-gradient tail 7.5 -> 6.5 ms; MSER_GEMM_WGS_PER_CU overrides for measurements)
    static const long per_cu = getenv("MSER_GEMM_WGS_PER_CU") ? atol(getenv("MSER_GEMM_WGS_PER_CU")) : 4;
    long want = ((group_tiles > 0 ? 2 : per_cu) * FILL + all - 1) / all;
    if (group_tiles > 0 && tiles < 64) {
      // inside a group no SMALL member may become the straggler: at most 16 k-tiles per workgroup (a tiny member that the group-wide
      // count left unsplit ran its whole K = B*L reduction in ONE workgroup: 90 us at the end of the step), and many short
      // workgroups also even out the last round of the grid.  A member of 64 tiles or more spreads over the chip by itself: splitting
      // it only multiplies its output atomics (hid = 256 / 1024 and DialogueRNN weight gradients: 8 float atomics per output element)
      const long ktiles = cdiv(d.K, BKT);
      const long cap = (ktiles + 15) / 16;
      if (want < cap) want = cap;
    }
    const long most = cdiv(d.K, 2 * BKT);
    if (want > most) want = most;
    splitk = (int)(want < 1 ? 1 : want);
  }
  GemmArgs& a = P.a;
  a.A = d.A; a.B = d.B; a.C = d.C; a.M = d.M; a.N = d.N; a.K = d.K;
  a.sAm = d.sAm; a.sAk = d.sAk; a.sBk = d.sBk; a.sBn = d.sBn; a.ldc = d.ldc;
  a.batch2 = b2;
  int kchunk = cdiv(d.K > 0 ? d.K : 1, splitk);
  kchunk = cdiv(kchunk, BKT) * BKT;
  a.kchunk = kchunk;
  a.splitk = splitk = (d.K > 0) ? cdiv(d.K, kchunk) : 1;
  if (splitk == 1 && d.splitk > 1) {
    // degenerate split: a read-modify-write accumulate (caller promised C is initialised).  Inside a group two members may
    // accumulate into the SAME tensor (a layer applied twice shares its weight gradient): those must stay atomic.
    a.flags = d.flags | (group_tiles > 0 ? GEMM_ATOMIC : MSER_GEMM_ACCUM);
  } else {
    a.flags = d.flags;
  }
  a.sA1 = d.sA1; a.sA2 = d.sA2; a.sB1 = d.sB1; a.sB2 = d.sB2; a.sC1 = d.sC1; a.sC2 = d.sC2;
  a.bias = d.bias; a.alpha_dev = d.alpha_dev; a.alpha = d.alpha;
  a.R1 = d.R1; a.R2 = d.R2; a.ldr1 = d.ldr1; a.ldr2 = d.ldr2; a.sR1 = d.sR1_1; a.sR2 = d.sR1_2;
  P.grid = dim3(cdiv(d.N, TN), cdiv(d.M, TM), b1 * b2 * splitk);
  MSER_REQUIRE(P.grid.y <= 65535 && P.grid.z <= 65535, "mser_gemm: grid too large (M=%d batch=%d)", d.M, b1 * b2);
  // vector (float4) staging needs: unit stride along the vector, every other stride % 4 == 0, a 16-byte aligned base, and
  // extents such that no vector straddles the valid range
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  auto m4 = [](long v) { return (v & 3) == 0; };
  const bool kvec_ok = m4(d.K) && m4(a.kchunk);
  int amode = 2, bmode = 2;
  if (d.sAk == 1 && m4(d.sAm) && m4(d.sA1) && m4(d.sA2) && al16(d.A) && kvec_ok) amode = 0;
  else if (d.sAm == 1 && m4(d.sAk) && m4(d.sA1) && m4(d.sA2) && al16(d.A) && m4(d.M)) amode = 1;
  if (d.sBk == 1 && m4(d.sBn) && m4(d.sB1) && m4(d.sB2) && al16(d.B) && kvec_ok) bmode = 0;
  else if (d.sBn == 1 && m4(d.sBk) && m4(d.sB1) && m4(d.sB2) && al16(d.B) && m4(d.N)) bmode = 1;
  P.amode = amode; P.bmode = bmode; P.c1 = c1;
  static const bool log_calls = getenv("MSER_GEMM_LOG") != nullptr;      // diagnostic: one line per call on stderr
  if (log_calls)
    fprintf(stderr, "[gemm] M %d N %d K %d b %d splitk %d modes %d%d flags %d sAm %ld sAk %ld sBk %ld sBn %ld ldc %ld bias %d R %d grid %u cfg %d grp %ld\n",
            d.M, d.N, d.K, b1 * b2, d.splitk, amode, bmode, d.flags, d.sAm, d.sAk, d.sBk, d.sBn, d.ldc, d.bias != nullptr,
            (d.R1 != nullptr) + (d.R2 != nullptr), P.grid.x * P.grid.y * P.grid.z, c1 ? 1 : 0, group_tiles);
  return 0;
}

int gemm(const mser_gemm_desc& d, hipStream_t s) {
  if (d.M == 0 || d.N == 0) {
    MSER_REQUIRE(d.M >= 0 && d.N >= 0, "mser_gemm: negative size");
    return 0;
  }
  Plan P;
  MSER_TRY(plan(d, 0, P));
  if (P.c1) launch_modes<2, 1>(P.amode, P.bmode, P.grid, s, P.a);
  else launch_modes<2, 2>(P.amode, P.bmode, P.grid, s, P.a);
  return check_launch("mser_gemm");
}

template <int WM, int WN>
static void launch_group_modes(int amode, int bmode, unsigned grid, hipStream_t s, const GroupArgs& G) {
#define MSER_GRP(AM, BM_) hipLaunchKernelGGL((gemm_group_kernel<AM, BM_, WM, WN>), dim3(grid), dim3(256), 0, s, G)
  switch (amode * 3 + bmode) {
    case 0: MSER_GRP(0, 0); break;
    case 1: MSER_GRP(0, 1); break;
    case 2: MSER_GRP(0, 2); break;
    case 3: MSER_GRP(1, 0); break;
    case 4: MSER_GRP(1, 1); break;
    case 5: MSER_GRP(1, 2); break;
    case 6: MSER_GRP(2, 0); break;
    case 7: MSER_GRP(2, 1); break;
    default: MSER_GRP(2, 2); break;
  }
#undef MSER_GRP
}

// n independent products in as few launches as their (load mode, tile configuration) classes allow.
int gemm_group(const mser_gemm_desc* d, int n, hipStream_t s) {
  MSER_REQUIRE(n >= 0 && (d || n == 0), "mser_gemm_grouped: null descriptor array");
  if (n == 0) return 0;
  if (n == 1) return gemm(d[0], s);
  long group_tiles = 0;
  for (int i = 0; i < n; ++i) {
    MSER_REQUIRE(d[i].M >= 0 && d[i].N >= 0, "mser_gemm_grouped: negative size");
    group_tiles += (long)cdiv(d[i].M, 64) * cdiv(d[i].N, 64) * (d[i].batch1 > 0 ? d[i].batch1 : 1) * (d[i].batch2 > 0 ? d[i].batch2 : 1);
  }
  std::vector<Plan> plans(n);
  std::vector<char> done(n, 0);
  for (int i = 0; i < n; ++i) {
    if (d[i].M == 0 || d[i].N == 0) { done[i] = 1; continue; }
    MSER_TRY(plan(d[i], group_tiles, plans[i]));
  }
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    GroupArgs G;
    G.n = 0;
    G.start[0] = 0;
    const Plan& lead = plans[i];
    for (int j = i; j < n && G.n < MAXG; ++j) {
      if (done[j]) continue;
      const Plan& q = plans[j];
      if (q.amode != lead.amode || q.bmode != lead.bmode || q.c1 != lead.c1) continue;
      G.a[G.n] = q.a;
      G.gx[G.n] = q.grid.x; G.gy[G.n] = q.grid.y;
      G.start[G.n + 1] = G.start[G.n] + (int)(q.grid.x * q.grid.y * q.grid.z);
      ++G.n;
      done[j] = 1;
    }
    if (lead.c1) launch_group_modes<2, 1>(lead.amode, lead.bmode, (unsigned)G.start[G.n], s, G);
    else launch_group_modes<2, 2>(lead.amode, lead.bmode, (unsigned)G.start[G.n], s, G);
    MSER_TRY(check_launch("mser_gemm_grouped"));
  }
  return 0;
}

}  // namespace mser

extern "C" int mser_gemm_grouped(const mser_gemm_desc* d, int32_t n, mser_stream_t stream) {
  return mser::gemm_group(d, n, (hipStream_t)stream);
}

extern "C" int mser_gemm(const mser_gemm_desc* d, mser_stream_t stream) {
  if (!d) { mser::set_error("mser_gemm: null descriptor"); return -1; }
  return mser::gemm(*d, (hipStream_t)stream);
}
