// Micro-benchmarks that price the building blocks of the recurrent chains on this MI355X (scratch; not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned int gu32;

__global__ void k_empty(int* p) { if (p && threadIdx.x == 999999) *p = 1; }

__global__ void k_chase(const int* next, int* out, int n, int sc1) {
  if (threadIdx.x != 0) return;
  int i = blockIdx.x * 64;
  for (int k = 0; k < n; ++k) {
    if (sc1) i = (int)__hip_atomic_load((const gu32*)(next + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else i = next[i];
  }
  out[blockIdx.x] = i;
}

__global__ void k_clock(unsigned long long* out) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (int i = 0; i < 200000; ++i) x = x * 1.0000001f + 0.5f;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}

// persistent barrier loop among gridDim.x workgroups; mode bit0: payload (each WG writes 1024 floats sc1, reads another WG's)
__global__ __launch_bounds__(1024) void k_barrier(unsigned* cnt, float* buf, int iters, int mode, unsigned long long* tstamp) {
  __shared__ int ok;
  const unsigned nwg = gridDim.x;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (mode & 1) {
      __hip_atomic_store((gu32*)(buf + ((it & 1) * nwg + blockIdx.x) * 1024 + threadIdx.x), __float_as_uint((float)it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      const unsigned target = nwg * (unsigned)(it + 1);
      while (__hip_atomic_load((const gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (mode & 2) __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) break;
      }
      ok = 1;
    }
    __syncthreads();
    if (mode & 1) {
      const unsigned src = (blockIdx.x + 1) % nwg;
      acc += __uint_as_float(__hip_atomic_load((const gu32*)(buf + ((it & 1) * nwg + src) * 1024 + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) tstamp[blockIdx.x] = t1 - t0;
  if (acc == -1.f) buf[0] = acc;
}

int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int* dflag; CK(hipMalloc(&dflag, 4));
  float ms;
  // A/B: empty-kernel launch cadence
  for (int cfg = 0; cfg < 4; ++cfg) {
    int thr = (cfg & 1) ? 256 : 1024; size_t lds = (cfg & 2) ? 70 * 1024 : 0;
    CK(hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(64), dim3(thr), lds, 0, dflag);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(64), dim3(thr), lds, 0, dflag);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel 64 WG x %4d thr, lds %6zu B: %.2f us per launch (eager back-to-back)\n", thr, lds, ms * 1000 / 2000);
  }
  // graph of 256 empty kernels
  {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 256; ++i) hipLaunchKernelGGL(k_empty, dim3(64), dim3(1024), 70 * 1024, st, dflag);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph of 256 empty kernels (64x1024, 70KB): %.2f us per kernel\n", ms * 1000 / 2560);
  }
  // C: pointer chase
  {
    const int N = 1 << 18;   // 1 MB of ints
    std::vector<int> h(N);
    for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 7919 + 12345) % N);
    int *dn, *dout; CK(hipMalloc(&dn, N * 4)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(dn, h.data(), N * 4, hipMemcpyHostToDevice));
    for (int sc1 = 0; sc1 < 2; ++sc1) for (int rep = 0; rep < 2; ++rep) {
      const int n = 2000;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, dn, dout, n, sc1);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("pointer chase (%s, rep %d): %.3f us per dependent load\n", sc1 ? "sc1" : "plain", rep, ms * 1000 / n);
    }
  }
  // E: clock
  {
    unsigned long long* d; CK(hipMalloc(&d, 64)); unsigned long long hh[3];
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d);
      CK(hipMemcpy(hh, d, 24, hipMemcpyDeviceToHost));
      printf("clock: %.0f MHz (s_memtime/s_memrealtime*100MHz)\n", (double)hh[0] / (double)hh[1] * 100.0);
    }
  }
  // D: barrier cost
  {
    unsigned* cnt; float* buf; unsigned long long* ts; CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&buf, 2 * 256 * 1024 * 4)); CK(hipMalloc(&ts, 256 * 8));
    for (int nwg : {2, 8, 32, 64}) for (int mode = 0; mode < 4; ++mode) {
      const int iters = 2000;
      CK(hipMemset(cnt, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_barrier, dim3(nwg), dim3(1024), 0, 0, cnt, buf, iters, mode, ts);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("barrier among %2d WGs (payload %d, sleep %d): %.2f us per barrier\n", nwg, mode & 1, (mode >> 1) & 1, ms * 1000 / iters);
    }
  }
  return 0;
}
