// Row-wise / element-wise kernels of the hot path (softmax, add+LayerNorm, reductions, bookkeeping, loss, Adam).
// All of these are HBM/L2-bandwidth kernels: one 64-lane wave per row with butterfly (DPP shuffle) reductions,
// rows short enough (<= 1280 floats) to live in registers between the passes.
#include "common.h"
#include <algorithm>
#include "../../include/mser.h"

namespace mser {

constexpr int WPB = 4;  // waves per block for the one-wave-per-row kernels

// ---------------------------------------------------------------------------------------------- softmax
template <int NV>  // NV = ceil(n / 64) values per lane
__global__ __launch_bounds__(64 * WPB) void softmax_rows_kernel(float* S, long rows, int n, long ld, const float* mul,
                                                                const uint8_t* mask, int mask_on, float fill) {
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* p = S + row * ld;
  float v[NV];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + i * 64;
    float x = -INFINITY;
    if (j < n) {
      x = p[j];
      if (mul) x *= mul[row * ld + j];
      if (mask && mask[row * ld + j] == mask_on) x = fill;
    }
    v[i] = x;
    mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + i * 64;
    v[i] = (j < n) ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + i * 64;
    if (j < n) p[j] = v[i] * inv;
  }
}

template <int NV>
__global__ __launch_bounds__(64 * WPB) void softmax_bwd_rows_kernel(const float* P, float* dP, long rows, int n, long ld,
                                                                    const float* mul) {
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = P + row * ld;
  float* d = dP + row * ld;
  float pv[NV], dv[NV];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + i * 64;
    pv[i] = (j < n) ? p[j] : 0.f;
    dv[i] = (j < n) ? d[j] : 0.f;
    dot += pv[i] * dv[i];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + i * 64;
    if (j < n) {
      float g = pv[i] * (dv[i] - dot);
      if (mul) g *= mul[row * ld + j];
      d[j] = g;
    }
  }
}

// ---------------------------------------------------------------------------------------------- LayerNorm
// D <= 256 (the path uses D = 100); one wave per row, two-pass mean/variance in registers (matches
// torch's layer_norm to ~1e-7: biased variance, rstd = 1/sqrt(var + eps)).
__global__ __launch_bounds__(64 * WPB) void add_layernorm_fwd_kernel(const float* x, long ldx, const float* res, long ldres,
                                                                     const float* gamma, const float* beta, float* y,
                                                                     float* sum_out, float* mean, float* rstd, long rows,
                                                                     int D, float eps) {
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + i * 64;
    float t = 0.f;
    if (j < D) {
      t = x[row * ldx + j];
      if (res) t += res[row * ldres + j];
      if (sum_out) sum_out[row * D + j] = t;
    }
    v[i] = t;
    s += t;
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + i * 64;
    const float dlt = (j < D) ? v[i] - mu : 0.f;
    q += dlt * dlt;
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + i * 64;
    if (j < D) y[row * D + j] = (v[i] - mu) * rs * gamma[j] + beta[j];
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  dgamma / dbeta: per-block partial sums in
// LDS, then one float atomic per column per block.
constexpr int LN_ROWS_PER_BLOCK = 16;
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* dy, const float* xsum, const float* mean,
                                                            const float* rstd, const float* gamma, float* dx,
                                                            float* dgamma, float* dbeta, long rows, int D) {
  __shared__ float sg[256], sb[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 256) { sg[threadIdx.x] = 0.f; sb[threadIdx.x] = 0.f; }
  __syncthreads();
  float ag[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
  const long r0 = (long)blockIdx.x * LN_ROWS_PER_BLOCK;
  for (int rr = wave; rr < LN_ROWS_PER_BLOCK; rr += 4) {
    const long row = r0 + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float g[4], xh[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = lane + i * 64;
      g[i] = 0.f; xh[i] = 0.f;
      if (j < D) {
        const float d = dy[row * D + j];
        xh[i] = (xsum[row * D + j] - mu) * rs;
        g[i] = d * gamma[j];
        ag[i] += d * xh[i];
        ab[i] += d;
      }
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = lane + i * 64;
      if (j < D) dx[row * D + j] = rs * (g[i] - s1 - xh[i] * s2);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + i * 64;
    if (j < D) { atomicAdd(&sg[j], ag[i]); atomicAdd(&sb[j], ab[i]); }
  }
  __syncthreads();
  if (threadIdx.x < D) {
    atomicAdd(&dgamma[threadIdx.x], sg[threadIdx.x]);
    atomicAdd(&dbeta[threadIdx.x], sb[threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------------------- batch ingest / metrics
// x[r, 0:dr] = (((r1 + r2) + r3) + r4) / 4 ;  x[r, dr:dr+da] = acouf[r, :]   (model_trainer.py:104-105: textf average + cat).
// Same association order as the reference expression, so the result is bit-identical to the torch CPU evaluation.
// HBM-bound: (4 dr + da) floats read and (dr + da) written per utterance; 16-byte accesses when every width is a multiple of 4.
template <int VEC>
__global__ __launch_bounds__(256) void ingest_kernel(const float* r1, const float* r2, const float* r3, const float* r4,
                                                     const float* ac, float* x, long rows, int dr, int da) {
  const int W = (dr + da) / VEC;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * W) return;
  const long r = i / W;
  const int j = (int)(i - r * W) * VEC;
  float* dst = x + r * (dr + da) + j;
  if (VEC == 4) {
    float4 v;
    if (j < dr) {
      const long o = r * dr + j;
      const float4 a = *reinterpret_cast<const float4*>(r1 + o), b = *reinterpret_cast<const float4*>(r2 + o);
      const float4 c = *reinterpret_cast<const float4*>(r3 + o), d = *reinterpret_cast<const float4*>(r4 + o);
      v.x = (((a.x + b.x) + c.x) + d.x) / 4.f; v.y = (((a.y + b.y) + c.y) + d.y) / 4.f;
      v.z = (((a.z + b.z) + c.z) + d.z) / 4.f; v.w = (((a.w + b.w) + c.w) + d.w) / 4.f;
    } else {
      v = *reinterpret_cast<const float4*>(ac + r * da + (j - dr));
    }
    *reinterpret_cast<float4*>(dst) = v;
  } else {
    if (j < dr) {
      const long o = r * dr + j;
      *dst = (((r1[o] + r2[o]) + r3[o]) + r4[o]) / 4.f;
    } else {
      *dst = ac[r * da + (j - dr)];
    }
  }
}

// Confusion matrix of one evaluation batch (model_trainer.py:142-156): pred = argmax_c lp[r, c] (first maximum),
// conf[label[r]][pred] += mask[r] in float64 (sklearn's sample_weight sums), pred_out[r] = pred (the res.csv column).
// One thread per utterance row; C*C float64 partial sums per workgroup in LDS, one atomic per non-zero cell per workgroup.
constexpr int CONF_MAXC = 16;
__global__ __launch_bounds__(256) void confusion_kernel(const float* lp, const long* label, const float* mask, long rows, int C,
                                                        double* conf, long* pred_out) {
  __shared__ double part[CONF_MAXC * CONF_MAXC];
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) part[i] = 0.0;
  __syncthreads();
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) {
    const float* p = lp + r * C;
    int best = 0;
    float bv = p[0];
    for (int c = 1; c < C; ++c) {
      const float v = p[c];
      if (v > bv) { bv = v; best = c; }
    }
    if (pred_out) pred_out[r] = best;
    const long y = label[r];
    const float w = mask[r];
    if (w != 0.f && y >= 0 && y < C) atomicAdd(&part[(int)y * C + best], (double)w);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += blockDim.x)
    if (part[i] != 0.0) atomicAdd(&conf[i], part[i]);
}

// ---------------------------------------------------------------------------------------------- reductions
// out[n] += sum_m X[m,n]; block = 256 threads covering 64 columns x 4 row-lanes, 64 rows per block.
__global__ __launch_bounds__(256) void colsum_kernel(const float* X, long rows, int n, long ld, float* out) {
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
  if (rl == 0 && c < n) atomicAdd(&out[c], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ void relu_bwd_kernel(float* dY, const float* Y, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count && !(Y[i] > 0.f)) dY[i] = 0.f;
}

__global__ void add_rows_kernel(float* out, long ldo, const float* a, long lda, const float* b, long ldb, long rows, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * D) return;
  const long r = i / D;
  const int j = (int)(i - r * D);
  float v = a[r * lda + j];
  if (b) v += b[r * ldb + j];
  out[r * ldo + j] = v;
}

__global__ __launch_bounds__(256) void scale_acc_dot_kernel(float* acc, long ldacc, const float* t, long ldt, const float* x,
                                                            long ldx, const float* s_dev, float* ds, long rows, int D) {
  __shared__ float red[4];
  const float s = s_dev ? *s_dev : 1.f;
  float dot = 0.f;
  const long total = rows * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int j = (int)(i - r * D);
    const float tv = t[r * ldt + j];
    acc[r * ldacc + j] += s * tv;
    if (ds) dot += tv * x[r * ldx + j];
  }
  if (ds) {
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(ds, red[0] + red[1] + red[2] + red[3]);
  }
}

// ---------------------------------------------------------------------------------------------- bookkeeping
__global__ void reverse_index_kernel(const float* umask, int B, int L, int* lens, int* rev) {
  const int b = blockIdx.x;
  __shared__ int len_s;
  float s = 0.f;
  for (int t = threadIdx.x; t < L; t += 64) s += umask[(long)b * L + t];
  s = wave_sum(s);
  if (threadIdx.x == 0) { len_s = (int)s; lens[b] = (int)s; }   // torch: sum(mask,1).int() truncates
  __syncthreads();
  const int len = len_s;
  for (int t = threadIdx.x; t < L; t += 64) rev[(long)t * B + b] = (t < len) ? (len - 1 - t) : -1;
}

__global__ void reverse_rows_kernel(const float* X, long ldx, const int* rev, float* out, long ldo, int L, int B, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)L * B * D) return;
  const long r = i / D;
  const int j = (int)(i - r * D);
  const int b = (int)(r % B);
  const int src = rev[r];
  out[r * ldo + j] = (src >= 0) ? X[((long)src * B + b) * ldx + j] : 0.f;
}

// One block per time step: party = argmax over the 2 mask values (ties -> 0, like torch.argmax returning the first
// maximum), then a stable partition of the dialogues by party (ballot-based prefix counts, B <= 1024).
__global__ __launch_bounds__(1024) void slot_tables_kernel(const float* qmask, const int* rev, int T, int B, int* party,
                                                           int* perm, int* n0, float* qm_out) {
  __shared__ int cnt0[17];   // per-wave counts of party-0 rows
  const int t = blockIdx.x, b = threadIdx.x;
  const int lane = b & 63, wave = b >> 6;
  int p = 1;   // inactive lanes count as party 1 so they never enter the party-0 ballot
  float m0 = 0.f, m1 = 0.f;
  if (b < B) {
    int src_t = t;
    if (rev) src_t = rev[(long)t * B + b];
    if (src_t >= 0) {
      m0 = qmask[((long)src_t * B + b) * 2 + 0];
      m1 = qmask[((long)src_t * B + b) * 2 + 1];
    }
    p = (m1 > m0) ? 1 : 0;
    party[(long)t * B + b] = p;
    qm_out[((long)t * B + b) * 2 + 0] = m0;
    qm_out[((long)t * B + b) * 2 + 1] = m1;
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
    const int idx0 = base0 + before0;             // rank among party-0 rows
    const int idx1 = b - idx0;                    // rank among party-1 rows (rows before b that are not party 0)
    const int row = (p == 0) ? idx0 : total0 + idx1;
    perm[(long)t * B + row] = b;
  }
  if (b == 0) n0[t] = total0;
}

// ---------------------------------------------------------------------------------------------- head / loss
__global__ void logsoftmax_tb_fwd_kernel(const float* y, float* lp, int L, int B, int C) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;   // r = b*L + t (output row)
  if (r >= (long)L * B) return;
  const int b = (int)(r / L), t = (int)(r % L);
  const float* src = y + ((long)t * B + b) * C;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, src[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(src[c] - mx);
  const float lse = mx + logf(s);
  for (int c = 0; c < C; ++c) lp[r * C + c] = src[c] - lse;
}

__global__ void logsoftmax_tb_bwd_kernel(const float* dlp, const float* lp, float* dy, int L, int B, int C) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (long)L * B) return;
  const int b = (int)(r / L), t = (int)(r % L);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += dlp[r * C + c];
  float* dst = dy + ((long)t * B + b) * C;
  for (int c = 0; c < C; ++c) dst[c] = dlp[r * C + c] - expf(lp[r * C + c]) * s;
}

// single block: deterministic two-level reduction (the loss is a scalar; rows <= ~1e6)
__global__ __launch_bounds__(1024) void masked_nll_fwd_kernel(const float* pred, const long* target, const float* mask,
                                                              long rows, int C, float* loss_out) {
  __shared__ float s_num[16], s_den[16];
  float num = 0.f, den = 0.f;
  for (long r = threadIdx.x; r < rows; r += 1024) {
    const float m = mask[r];
    num -= pred[r * C + target[r]] * m;
    den += m;
  }
  num = wave_sum(num);
  den = wave_sum(den);
  if ((threadIdx.x & 63) == 0) { s_num[threadIdx.x >> 6] = num; s_den[threadIdx.x >> 6] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < 16; ++i) { a += s_num[i]; b += s_den[i]; }
    loss_out[0] = a / b;
    loss_out[1] = b;
  }
}

// MaskedLoss in full (loss.py:13-25): losser in {NLLLoss, CrossEntropyLoss}(weight=w, reduction='sum') applied to pred * mask,
// divided by sum(mask) (w == null) or sum(w[target] * mask).  CrossEntropyLoss re-applies log_softmax to pred * mask: on the
// log-probabilities of a valid row that is the identity, but a masked row (pred * 0) contributes w[y] * log(C) to the numerator --
// the reference's reported loss on padded batches with its default --loss CrossEntropy (train.py:117) includes that term.
// Labels: torch's lossers skip rows whose target is ignore_index (-100: no term in the sum; the reference's denominator
// sum(mask) still counts the row) and raise on any other target outside [0, C).  A launch cannot raise: such a row is skipped and
// MSER_FAULT_BAD_LABEL is ORed into *fault (the trainer raises at its next synchronisation); with class weights an ignored row
// would make the reference index weight[-100], so it counts as a bad label there.
__device__ __forceinline__ bool label_ok(long y, int C, bool weighted, unsigned* fault) {
  if (y >= 0 && y < C) return true;
  if (!(y == -100 && !weighted) && fault) atomicOr(fault, (unsigned)MSER_FAULT_BAD_LABEL);
  return false;
}

__global__ __launch_bounds__(1024) void masked_loss_fwd_kernel(const float* pred, const long* target, const float* mask,
                                                               const float* weight, int is_ce, long rows, int C, float* loss_out,
                                                               unsigned* fault) {
  __shared__ float s_num[16], s_den[16];
  float num = 0.f, den = 0.f;
  for (long r = threadIdx.x; r < rows; r += 1024) {
    const float m = mask[r];
    const long y = target[r];
    if (!label_ok(y, C, weight != nullptr, fault)) {
      if (!weight) den += m;
      continue;
    }
    const float w = weight ? weight[y] : 1.f;
    const float* p = pred + r * C;
    if (is_ce) {
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, m * p[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(m * p[c] - mx);
      num += w * (mx + logf(se) - m * p[y]);
    } else {
      num -= w * m * p[y];
    }
    den += weight ? w * m : m;
  }
  num = wave_sum(num);
  den = wave_sum(den);
  if ((threadIdx.x & 63) == 0) { s_num[threadIdx.x >> 6] = num; s_den[threadIdx.x >> 6] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < 16; ++i) { a += s_num[i]; b += s_den[i]; }
    loss_out[0] = a / b;
    loss_out[1] = b;
  }
}

__global__ void masked_loss_bwd_kernel(const float* pred, const long* target, const float* mask, const float* weight, int is_ce,
                                       const float* loss_out, const float* gscale_dev, float* dpred, long rows, int C) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float m = mask[r];
  const long y = target[r];
  float* d = dpred + r * C;
  // an empty shard (sum(mask) == 0: the forward reports 0/0 like the reference) contributes a ZERO gradient, so that the
  // data-parallel sum n_r * g_r over the ranks stays finite; ignored / out-of-range labels contribute nothing either
  if (!(y >= 0 && y < C) || loss_out[1] == 0.f) {
    for (int c = 0; c < C; ++c) d[c] = 0.f;
    return;
  }
  const float k = (gscale_dev ? *gscale_dev : 1.f) * (weight ? weight[y] : 1.f) / loss_out[1];
  const float* p = pred + r * C;
  if (is_ce) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, m * p[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(m * p[c] - mx);
    for (int c = 0; c < C; ++c) d[c] = k * m * (expf(m * p[c] - mx) / se - (c == (int)y ? 1.f : 0.f));
  } else {
    for (int c = 0; c < C; ++c) d[c] = (c == (int)y) ? -k * m : 0.f;
  }
}

__global__ void masked_nll_bwd_kernel(const long* target, const float* mask, const float* loss_out, const float* gscale_dev,
                                      float* dpred, long rows, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  const float gs = gscale_dev ? *gscale_dev : 1.f;
  dpred[i] = (c == (int)target[r]) ? -gs * mask[r] / loss_out[1] : 0.f;
}

// ---------------------------------------------------------------------------------------------- Adam
__global__ void adam_flat_kernel(float* p, const float* g, float* m, float* v, const uint8_t* live, long n, float lr_bc1,
                                 float inv_sqrt_bc2, float b1, float b2, float eps, float wd, float gscale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (live && !live[i]) return;
  const float pi = p[i];
  const float gi = g[i] * gscale + wd * pi;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
  p[i] = pi - lr_bc1 * mi / denom;
}

// Graph-capturable Adam: the step counter and learning rate live on the device, so a captured launch stays valid for every
// replay.  hp = {lr, beta1, beta2}; sched (2 floats of scratch) receives lr/(1-beta1^t) and 1/sqrt(1-beta2^t).
__global__ void adam_sched_kernel(int* step, const float* hp, float* sched, const unsigned* fault, const float* gfault) {
  if ((fault && *fault != 0u) || (gfault && *gfault != 0.f)) return;        // a faulted step is not applied (see adam_flat_dev_kernel): it does not count either
  const int t = *step + 1;
  *step = t;
  const double bc1 = 1.0 - pow((double)hp[1], (double)t), bc2 = 1.0 - pow((double)hp[2], (double)t);
  sched[0] = (float)((double)hp[0] / bc1);
  sched[1] = (float)(1.0 / sqrt(bc2));
}

__global__ void adam_flat_dev_kernel(float* p, const float* g, float* m, float* v, const uint8_t* live, long n, const float* sched,
                                     const float* hp, float eps, float wd, const float* gscale_dev, float gscale,
                                     const unsigned* fault, const float* gfault) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // the sticky fault word of this step's kernels is set (a persistent chain gave up at a bounded wait, a label was out of range):
  // the gradients are not to be trusted, leave parameters and moments untouched; the host raises when it next reads the word
  if ((fault && *fault != 0u) || (gfault && *gfault != 0.f)) return;     // gfault: the ranks' fault flags, summed by the all-reduce
  if (live && !live[i]) return;
  const float b1 = hp[1], b2 = hp[2];
  const float gs = gscale_dev ? gscale / *gscale_dev : gscale;
  const float pi = p[i];
  const float gi = g[i] * gs + wd * pi;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = pi - sched[0] * mi / (sqrtf(vi) * sched[1] + eps);
}

// buf[0..n) = g[0..n) * (*cnt) ; buf[n] = *cnt ; buf[n+1] = (*fault != 0)      (pack for the single data-parallel all-reduce)
__global__ void dp_pack_kernel(float* buf, const float* g, const float* cnt, long n, const unsigned* fault) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const float c = *cnt;
  if (i < n) buf[i] = g[i] * c;
  if (i == 0) { buf[n] = c; buf[n + 1] = (fault && *fault != 0u) ? 1.f : 0.f; }
}


// ================================================================================================ dropout
// x[r, c] *= keep(site, idx0 + r*cols + c) ? 1/(1-p) : 0, in place (forward activations and backward gradients alike);
// out != nullptr: write the factor itself instead (mser_dropout_scale, for a checker that needs the mask as data).
__global__ void dropout_apply_kernel(float* x, float* out, long rows, int cols, long ld, const uint32_t* rng, uint32_t site, float p,
                                     uint32_t idx0, int draw16) {
  const DropKey k = drop_key(rng, site, p);
  const long n = rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long r = e / cols;
    const int c = (int)(e - r * cols);
    const float f = draw16 ? drop_scale16(k, idx0 + (uint32_t)e) : drop_scale(k, idx0 + (uint32_t)e);
    if (out) out[e] = f;
    else x[r * ld + c] *= f;
  }
}
__global__ void rng_advance_kernel(uint32_t* rng) { rng[1] += 1u; }

}  // namespace mser

using namespace mser;

extern "C" {

int mser_softmax_rows(float* S, int64_t rows, int32_t n, int64_t ld, const float* mul, const uint8_t* mask, int32_t mask_on,
                      float fill, mser_stream_t stream) {
  MSER_REQUIRE(S && n > 0 && n <= 1024, "mser_softmax_rows: need 0 < n <= 1024 (got %d)", n);
  if (rows <= 0) return 0;
  dim3 grid(cdiv(rows, WPB)), block(64 * WPB);
  hipStream_t s = (hipStream_t)stream;
  const int nv = cdiv(n, 64);
#define SM(NV) hipLaunchKernelGGL((softmax_rows_kernel<NV>), grid, block, 0, s, S, (long)rows, n, (long)ld, mul, mask, mask_on, fill)
  if (nv <= 1) SM(1); else if (nv <= 2) SM(2); else if (nv <= 4) SM(4); else if (nv <= 8) SM(8); else SM(16);
#undef SM
  return check_launch("mser_softmax_rows");
}

int mser_softmax_bwd_rows(const float* P, float* dP, int64_t rows, int32_t n, int64_t ld, const float* mul,
                          mser_stream_t stream) {
  MSER_REQUIRE(P && dP && n > 0 && n <= 1024, "mser_softmax_bwd_rows: need 0 < n <= 1024 (got %d)", n);
  if (rows <= 0) return 0;
  dim3 grid(cdiv(rows, WPB)), block(64 * WPB);
  hipStream_t s = (hipStream_t)stream;
  const int nv = cdiv(n, 64);
#define SM(NV) hipLaunchKernelGGL((softmax_bwd_rows_kernel<NV>), grid, block, 0, s, P, dP, (long)rows, n, (long)ld, mul)
  if (nv <= 1) SM(1); else if (nv <= 2) SM(2); else if (nv <= 4) SM(4); else if (nv <= 8) SM(8); else SM(16);
#undef SM
  return check_launch("mser_softmax_bwd_rows");
}

int mser_add_layernorm_fwd(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* gamma, const float* beta,
                           float* y, float* sum_out, float* mean, float* rstd, int64_t rows, int32_t D, float eps,
                           mser_stream_t stream) {
  MSER_REQUIRE(x && gamma && beta && y && mean && rstd, "mser_add_layernorm_fwd: null pointer");
  MSER_REQUIRE(D > 0 && D <= 256, "mser_add_layernorm_fwd: D=%d unsupported (<=256)", D);
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3(cdiv(rows, WPB)), dim3(64 * WPB), 0, (hipStream_t)stream, x, (long)ldx, res,
                     (long)ldres, gamma, beta, y, sum_out, mean, rstd, (long)rows, D, eps);
  return check_launch("mser_add_layernorm_fwd");
}

int mser_layernorm_bwd(const float* dy, const float* xsum, const float* mean, const float* rstd, const float* gamma, float* dx,
                       float* dgamma, float* dbeta, int64_t rows, int32_t D, mser_stream_t stream) {
  MSER_REQUIRE(dy && xsum && mean && rstd && gamma && dx && dgamma && dbeta, "mser_layernorm_bwd: null pointer");
  MSER_REQUIRE(D > 0 && D <= 256, "mser_layernorm_bwd: D=%d unsupported (<=256)", D);
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(rows, LN_ROWS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, dy, xsum, mean,
                     rstd, gamma, dx, dgamma, dbeta, (long)rows, D);
  return check_launch("mser_layernorm_bwd");
}

int mser_colsum_acc(const float* X, int64_t rows, int32_t n, int64_t ld, float* out, mser_stream_t stream) {
  MSER_REQUIRE(X && out, "mser_colsum_acc: null pointer");
  if (rows <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(n, 64), cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream, X, (long)rows, n,
                     (long)ld, out);
  return check_launch("mser_colsum_acc");
}

int mser_relu_bwd(float* dY, const float* Y, int64_t count, mser_stream_t stream) {
  MSER_REQUIRE(dY && Y, "mser_relu_bwd: null pointer");
  if (count <= 0) return 0;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, dY, Y, (long)count);
  return check_launch("mser_relu_bwd");
}

int mser_add_rows(float* out, int64_t ldo, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t D,
                  mser_stream_t stream) {
  MSER_REQUIRE(out && a, "mser_add_rows: null pointer");
  if (rows <= 0 || D <= 0) return 0;
  hipLaunchKernelGGL(add_rows_kernel, dim3(cdiv(rows * D, 256)), dim3(256), 0, (hipStream_t)stream, out, (long)ldo, a, (long)lda,
                     b, (long)ldb, (long)rows, D);
  return check_launch("mser_add_rows");
}

int mser_dropout_apply(float* x, int64_t rows, int32_t cols, int64_t ld, const uint32_t* rng, uint32_t site, float p,
                       uint32_t idx0, mser_stream_t stream) {
  MSER_REQUIRE(x && rng, "mser_dropout_apply: null pointer");
  MSER_REQUIRE(p >= 0.f && p < 1.f && cols > 0 && ld >= cols, "mser_dropout_apply: bad arguments (p=%f cols=%d ld=%ld)", p, cols, (long)ld);
  if (rows <= 0) return 0;
  const long n = rows * cols;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(std::min<long>(cdiv(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, x, (float*)nullptr,
                     (long)rows, cols, (long)ld, rng, site, p, idx0, 0);
  return check_launch("mser_dropout_apply");
}

int mser_dropout_scale(float* out, int64_t n, const uint32_t* rng, uint32_t site, float p, uint32_t idx0, int32_t draw_bits,
                       mser_stream_t stream) {
  MSER_REQUIRE(out && rng, "mser_dropout_scale: null pointer");
  MSER_REQUIRE(draw_bits == 32 || draw_bits == 16, "mser_dropout_scale: draw_bits=%d (32 or 16)", draw_bits);
  MSER_REQUIRE(p >= 0.f && p < 1.f, "mser_dropout_scale: p=%f", p);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(std::min<long>(cdiv(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, (float*)nullptr, out,
                     (long)n, 1, 1L, rng, site, p, idx0, draw_bits == 16 ? 1 : 0);
  return check_launch("mser_dropout_scale");
}

int mser_rng_advance(uint32_t* rng, mser_stream_t stream) {
  MSER_REQUIRE(rng, "mser_rng_advance: null pointer");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng);
  return check_launch("mser_rng_advance");
}

int mser_ingest_features(const float* r1, const float* r2, const float* r3, const float* r4, const float* acouf, float* x,
                         int64_t rows, int32_t d_r, int32_t d_a, mser_stream_t stream) {
  MSER_REQUIRE(r1 && r2 && r3 && r4 && acouf && x, "mser_ingest_features: null pointer");
  MSER_REQUIRE(d_r > 0 && d_a >= 0, "mser_ingest_features: bad widths d_r=%d d_a=%d", d_r, d_a);
  if (rows <= 0) return 0;
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const bool vec = (d_r % 4 == 0) && (d_a % 4 == 0) && al(r1) && al(r2) && al(r3) && al(r4) && al(acouf) && al(x);
  if (vec) {
    const long n = rows * ((d_r + d_a) / 4);
    hipLaunchKernelGGL(ingest_kernel<4>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, r1, r2, r3, r4, acouf, x, (long)rows, d_r, d_a);
  } else {
    const long n = rows * (d_r + d_a);
    hipLaunchKernelGGL(ingest_kernel<1>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, r1, r2, r3, r4, acouf, x, (long)rows, d_r, d_a);
  }
  return check_launch("mser_ingest_features");
}

int mser_confusion_update(const float* lp, const int64_t* label, const float* mask, int64_t rows, int32_t C, double* conf,
                          int64_t* pred_out, mser_stream_t stream) {
  MSER_REQUIRE(lp && label && mask && conf, "mser_confusion_update: null pointer");
  MSER_REQUIRE(C > 0 && C <= CONF_MAXC, "mser_confusion_update: C=%d unsupported (<= %d)", C, CONF_MAXC);
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(confusion_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, lp, (const long*)label, mask,
                     (long)rows, C, conf, (long*)pred_out);
  return check_launch("mser_confusion_update");
}

int mser_scale_acc_dot(float* acc, int64_t ldacc, const float* t, int64_t ldt, const float* x, int64_t ldx, const float* s_dev,
                       float* ds, int64_t rows, int32_t D, mser_stream_t stream) {
  MSER_REQUIRE(acc && t && (!ds || x), "mser_scale_acc_dot: null pointer");
  if (rows <= 0 || D <= 0) return 0;
  const int blocks = (int)fmin((double)cdiv(rows * D, 1024), 128.0);      // one same-address float atomic per block
  hipLaunchKernelGGL(scale_acc_dot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, (long)ldacc, t, (long)ldt, x,
                     (long)ldx, s_dev, ds, (long)rows, D);
  return check_launch("mser_scale_acc_dot");
}

int mser_build_reverse_index(const float* umask, int32_t B, int32_t L, int32_t* lens, int32_t* rev, mser_stream_t stream) {
  MSER_REQUIRE(umask && lens && rev && B > 0 && L > 0, "mser_build_reverse_index: bad arguments");
  hipLaunchKernelGGL(reverse_index_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, umask, B, L, lens, rev);
  return check_launch("mser_build_reverse_index");
}

int mser_reverse_by_length(const float* X, int64_t ldx, const int32_t* rev, float* out, int64_t ldo, int32_t L, int32_t B,
                           int32_t D, mser_stream_t stream) {
  MSER_REQUIRE(X && rev && out, "mser_reverse_by_length: null pointer");
  const long total = (long)L * B * D;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(reverse_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, X, (long)ldx, rev, out,
                     (long)ldo, L, B, D);
  return check_launch("mser_reverse_by_length");
}

int mser_build_slot_tables(const float* qmask, const int32_t* rev, int32_t T, int32_t B, int32_t* party, int32_t* perm,
                           int32_t* n0, float* qm_out, mser_stream_t stream) {
  MSER_REQUIRE(qmask && party && perm && n0 && qm_out, "mser_build_slot_tables: null pointer");
  MSER_REQUIRE(B > 0 && B <= 1024 && T > 0, "mser_build_slot_tables: need 0 < B <= 1024 (got %d)", B);
  const int threads = cdiv(B, 64) * 64;
  hipLaunchKernelGGL(slot_tables_kernel, dim3(T), dim3(threads), 0, (hipStream_t)stream, qmask, rev, T, B, party, perm, n0, qm_out);
  return check_launch("mser_build_slot_tables");
}

int mser_logsoftmax_tb_fwd(const float* y, float* lp, int32_t L, int32_t B, int32_t C, mser_stream_t stream) {
  MSER_REQUIRE(y && lp && C > 0, "mser_logsoftmax_tb_fwd: bad arguments");
  hipLaunchKernelGGL(logsoftmax_tb_fwd_kernel, dim3(cdiv((long)L * B, 256)), dim3(256), 0, (hipStream_t)stream, y, lp, L, B, C);
  return check_launch("mser_logsoftmax_tb_fwd");
}

int mser_logsoftmax_tb_bwd(const float* dlp, const float* lp, float* dy, int32_t L, int32_t B, int32_t C, mser_stream_t stream) {
  MSER_REQUIRE(dlp && lp && dy && C > 0, "mser_logsoftmax_tb_bwd: bad arguments");
  hipLaunchKernelGGL(logsoftmax_tb_bwd_kernel, dim3(cdiv((long)L * B, 256)), dim3(256), 0, (hipStream_t)stream, dlp, lp, dy, L, B, C);
  return check_launch("mser_logsoftmax_tb_bwd");
}

int mser_masked_nll_fwd(const float* pred, const int64_t* target, const float* mask, int64_t rows, int32_t C, float* loss_out,
                        mser_stream_t stream) {
  MSER_REQUIRE(pred && target && mask && loss_out && rows > 0, "mser_masked_nll_fwd: bad arguments");
  hipLaunchKernelGGL(masked_nll_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, (const long*)target, mask, (long)rows,
                     C, loss_out);
  return check_launch("mser_masked_nll_fwd");
}

int mser_masked_nll_bwd(const int64_t* target, const float* mask, const float* loss_out, const float* gscale_dev, float* dpred,
                        int64_t rows, int32_t C, mser_stream_t stream) {
  MSER_REQUIRE(target && mask && loss_out && dpred && rows > 0, "mser_masked_nll_bwd: bad arguments");
  hipLaunchKernelGGL(masked_nll_bwd_kernel, dim3(cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)target, mask,
                     loss_out, gscale_dev, dpred, (long)rows, C);
  return check_launch("mser_masked_nll_bwd");
}

int mser_masked_loss_fwd(const float* pred, const int64_t* target, const float* mask, const float* weight, int32_t is_ce,
                         int64_t rows, int32_t C, float* loss_out, uint32_t* fault, mser_stream_t stream) {
  MSER_REQUIRE(pred && target && mask && loss_out && rows > 0 && C > 0, "mser_masked_loss_fwd: bad arguments");
  hipLaunchKernelGGL(masked_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, (const long*)target, mask, weight,
                     is_ce, (long)rows, C, loss_out, (unsigned*)fault);
  return check_launch("mser_masked_loss_fwd");
}

int mser_masked_loss_bwd(const float* pred, const int64_t* target, const float* mask, const float* weight, int32_t is_ce,
                         const float* loss_out, const float* gscale_dev, float* dpred, int64_t rows, int32_t C,
                         mser_stream_t stream) {
  MSER_REQUIRE(pred && target && mask && loss_out && dpred && rows > 0 && C > 0, "mser_masked_loss_bwd: bad arguments");
  hipLaunchKernelGGL(masked_loss_bwd_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, pred, (const long*)target, mask,
                     weight, is_ce, loss_out, gscale_dev, dpred, (long)rows, C);
  return check_launch("mser_masked_loss_bwd");
}

int mser_adam_flat(float* p, const float* g, float* m, float* v, const uint8_t* live, int64_t n, int32_t step, float lr, float beta1,
                   float beta2, float eps, float wd, float gscale, mser_stream_t stream) {
  MSER_REQUIRE(p && g && m && v && step >= 1, "mser_adam_flat: bad arguments");
  if (n <= 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, live, (long)n,
                     (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, wd, gscale);
  return check_launch("mser_adam_flat");
}

int mser_adam_flat_dev(float* p, const float* g, float* m, float* v, const uint8_t* live, int64_t n, int32_t* step_dev,
                       const float* hp_dev, float* sched_dev, float eps, float wd, const float* gscale_div_dev, float gscale,
                       const uint32_t* fault, const float* gfault, mser_stream_t stream) {
  MSER_REQUIRE(p && g && m && v && step_dev && hp_dev && sched_dev, "mser_adam_flat_dev: bad arguments");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adam_sched_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, hp_dev, sched_dev, (const unsigned*)fault, gfault);
  hipLaunchKernelGGL(adam_flat_dev_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, live, (long)n, sched_dev,
                     hp_dev, eps, wd, gscale_div_dev, gscale, (const unsigned*)fault, gfault);
  return check_launch("mser_adam_flat_dev");
}

int mser_dp_pack(float* buf, const float* g, const float* cnt_dev, int64_t n, const uint32_t* fault, mser_stream_t stream) {
  MSER_REQUIRE(buf && g && cnt_dev && n > 0, "mser_dp_pack: bad arguments");
  hipLaunchKernelGGL(dp_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, buf, g, cnt_dev, (long)n, (const unsigned*)fault);
  return check_launch("mser_dp_pack");
}

}  // extern "C"
