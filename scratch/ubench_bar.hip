// Barrier variants among co-resident workgroups (scratch; prices the hand-off of the recurrent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int LINE = 32;
// variant 0: one counter; 1: 8 replica counters; 2: per-WG flags (one set), wave poll; 3: per-WG flags in 4 sets; 4: flags in 8 sets
template <int V>
__global__ __launch_bounds__(512) void k_bar(unsigned* sync, float* buf, int iters, int payload, unsigned long long* tstamp) {
  extern __shared__ float smem[];
  __shared__ int ok;
  const unsigned nwg = gridDim.x, w = blockIdx.x;
  float acc = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (payload) __hip_atomic_store((gu32*)(buf + ((it & 1) * nwg + w) * 512 + threadIdx.x), __float_as_uint((float)it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned gen = (unsigned)(it + 1);
    if (V == 0) {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add((gu32*)sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load((const gu32*)sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg * gen) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) break; }
      }
    } else if (V == 1) {
      if (threadIdx.x < 8) __hip_atomic_fetch_add((gu32*)sync + threadIdx.x * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) {
        const unsigned* c = sync + (w % 8) * LINE;
        unsigned spins = 0;
        while (__hip_atomic_load((const gu32*)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg * gen) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) break; }
      }
    } else {
      constexpr int R = V == 2 ? 1 : (V == 3 ? 4 : 8);
      // flags[set][wg]: every set is 64 words = 2 lines
      if (threadIdx.x < R) __hip_atomic_store((gu32*)sync + threadIdx.x * 64 + w, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x < 64) {
        const unsigned* f = sync + (w % R) * 64 + threadIdx.x;
        const bool mine = threadIdx.x < nwg;
        unsigned spins = 0;
        while (true) {
          const unsigned v = mine ? __hip_atomic_load((const gu32*)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gen;
          if (__builtin_amdgcn_ballot_w64(v < gen) == 0ull) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) break;
        }
      }
    }
    __syncthreads();
    if (payload) {
      // every workgroup reads 16 bytes from every other workgroup's fresh slab (all-to-all exchange)
      if (threadIdx.x < nwg * 4) acc += __uint_as_float(__hip_atomic_load((const gu32*)(buf + ((it & 1) * nwg + (threadIdx.x >> 2)) * 512 + (w * 4 + (threadIdx.x & 3)) % 512), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) tstamp[w] = t1 - t0;
  if (acc == -1.f) buf[0] = acc;
  (void)ok;
}
template <int V>
int run(unsigned* sync, float* buf, unsigned long long* ts, int nwg, int payload) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  CK(hipFuncSetAttribute((const void*)k_bar<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(sync, 0, 16384));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_bar<V>, dim3(nwg), dim3(512), 84 * 1024, 0, sync, buf, iters, payload, ts);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("variant %d  nwg %2d payload %d: %.2f us per barrier\n", V, nwg, payload, best * 1000 / iters);
  return 0;
}
int main() {
  unsigned* sync; float* buf; unsigned long long* ts;
  CK(hipMalloc(&sync, 16384)); CK(hipMalloc(&buf, 2 * 64 * 512 * 4)); CK(hipMalloc(&ts, 256 * 8));
  for (int nwg : {32, 64}) for (int payload : {0, 1}) {
    if (run<0>(sync, buf, ts, nwg, payload)) return 1;
    if (run<1>(sync, buf, ts, nwg, payload)) return 1;
    if (run<2>(sync, buf, ts, nwg, payload)) return 1;
    if (run<3>(sync, buf, ts, nwg, payload)) return 1;
    if (run<4>(sync, buf, ts, nwg, payload)) return 1;
  }
  return 0;
}
