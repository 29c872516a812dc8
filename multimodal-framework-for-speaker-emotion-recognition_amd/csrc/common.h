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

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

}  // namespace mser
