// Internal helpers shared by the HIP translation units of libmser.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

namespace mser {

// thread-local last-error slot behind mser_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

#define MSER_CHECK_HIP(expr)                                                   \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      mser::set_error("%s:%d %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return (int)_e;                                                          \
    }                                                                          \
  } while (0)

#define MSER_REQUIRE(cond, ...)                                                \
  do {                                                                         \
    if (!(cond)) {                                                             \
      mser::set_error(__VA_ARGS__);                                            \
      return -1;                                                               \
    }                                                                          \
  } while (0)

#define MSER_TRY(expr)                                                         \
  do {                                                                         \
    int _r = (expr);                                                           \
    if (_r != 0) return _r;                                                    \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- optional per-kernel timing with HIP events (bench.py's live roofline measurement; off by default) ----------------------
// mser_prof_enable(id, max) arms ONE kernel id (include/mser.h MSER_PROF_*); a ProfScope around a launch of that id records a
// (start, stop) event pair on the launch stream; mser_prof_collect sums them.  Defined in api.cpp.
struct Prof {
  int kernel_id = 0;            // 0 = off
  int cap = 0, used = 0;
  hipEvent_t* ev = nullptr;     // 2*cap events: (start, stop) pairs
};
extern Prof g_prof;
struct ProfScope {
  bool on;
  hipStream_t s;
  ProfScope(int id, hipStream_t st) : on(g_prof.kernel_id == id && g_prof.used < g_prof.cap), s(st) {
    if (on) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s); ++g_prof.used; }
  }
};

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// accurate sigmoid / tanh in fp32 (expf/tanhf from the device libm; parity target is 1e-4 on logits,
// so no fast-math approximations on the recurrent path)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- counter-based dropout (include/mser.h "Dropout") -----------------------------------------------------------------------
// keep(site, idx) is a pure function of (seed, step, site, idx): the forward and the backward of a step evaluate the same mask
// without storing it, and mser_dropout_scale writes it out for a checker.  rng = {seed, step} in device memory (the step word is
// advanced on the device, so a captured graph draws new masks at every replay).  bits(idx) = mix32(idx ^ key), one round of a
// full-avalanche 32-bit mix per element; key = mix(seed, site) ^ mix(step, site).  The rank-1 attention sites, whose masks are
// evaluated inside the recurrent chains ([B, H, H] elements per step), draw 16 bits per element: one mix per two elements.
struct DropKey { uint32_t k0, thr; float scale; };
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {      // drop iff bits < thr
  const double v = (double)p * 4294967296.0;
  return v <= 0.0 ? 0u : (v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v);
}
__device__ __forceinline__ DropKey drop_key(const uint32_t* rng, uint32_t site, float p) {
  DropKey k;
  const uint32_t seed = rng[0], step = rng[1];
  k.k0 = mix32(seed ^ (site * 0x9E3779B9U) ^ 0x2545F491U) ^ mix32(step * 0x85EBCA6BU + site + 1U);
  k.thr = drop_threshold(p);
  k.scale = 1.0f / (1.0f - p);
  return k;
}
__device__ __forceinline__ float drop_scale(const DropKey& k, uint32_t idx) {
  return mix32(idx ^ k.k0) >= k.thr ? k.scale : 0.f;
}
// 16-bit draws: elements 2w and 2w+1 share the word mix32(w ^ key) (low / high half); p is resolved to 1/65536
__device__ __forceinline__ uint32_t drop_word16(const DropKey& k, uint32_t idx) { return mix32((idx >> 1) ^ k.k0); }
__device__ __forceinline__ float drop_half16(const DropKey& k, uint32_t word, uint32_t odd) {
  return ((odd ? word >> 16 : word & 0xFFFFu) >= (k.thr >> 16)) ? k.scale : 0.f;
}
__device__ __forceinline__ float drop_scale16(const DropKey& k, uint32_t idx) { return drop_half16(k, drop_word16(k, idx), idx & 1u); }

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

}  // namespace mser
