// Does co-locating the workgroups of a counter barrier on ONE XCD make the hand-off cheaper?  (scratch; design input)
// 256 workgroups (one per CU through the LDS request) take a ticket on their XCD; the 32 participants are either the 32
// workgroups of XCD 0 ("same") or tickets 0..3 of every XCD ("spread").  Protocol as in recurrent.hip: sc1 payload stores,
// vmcnt(0) + barrier, replicated counter adds, poll, sc1 payload loads (all-to-all: every participant reads 16 B from every other).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int LINE = 32, REP = 8;
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

__global__ __launch_bounds__(512) void k(unsigned* tickets, unsigned* sync, float* buf, int iters, int same, int payload_floats, unsigned* census) {
  extern __shared__ float smem[];
  __shared__ unsigned s_xcc, s_tk;
  if (threadIdx.x == 0) {
    s_xcc = xcc_id();
    s_tk = __hip_atomic_fetch_add((gu32*)(tickets + s_xcc * LINE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned xcc = s_xcc, tk = s_tk;
  if (threadIdx.x == 0) atomicAdd(census + xcc, 1u);
  int w = -1;                                   // participant index 0..31
  // same = number of XCDs the 32 participants are spread over (1, 2, 4, 8)
  const unsigned per = 32u / (unsigned)same;
  if (xcc < (unsigned)same && tk < per) w = (int)(xcc * per + tk);
  if (w < 0) return;
  const unsigned nwg = 32;
  float acc = 0.f;
  const int pf = payload_floats;                // floats stored per workgroup per iteration (<= 512)
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x < pf) __hip_atomic_store((gu32*)(buf + ((it & 1) * nwg + w) * 512 + threadIdx.x), __float_as_uint((float)it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned gen = (unsigned)(it + 1);
    if (threadIdx.x < REP) __hip_atomic_fetch_add((gu32*)sync + threadIdx.x * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
      const unsigned* c = sync + (w % REP) * LINE;
      unsigned spins = 0;
      while (__hip_atomic_load((const gu32*)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg * gen) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) break; }
    }
    __syncthreads();
    if (pf && threadIdx.x < nwg * 4)
      acc += __uint_as_float(__hip_atomic_load((const gu32*)(buf + ((it & 1) * nwg + (threadIdx.x >> 2)) * 512 + (w * 4 + (threadIdx.x & 3)) % pf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  if (acc == -1.f) buf[0] = acc;
}

int main() {
  unsigned *tickets, *sync, *census; float* buf;
  CK(hipMalloc(&tickets, 8 * LINE * 4)); CK(hipMalloc(&sync, 16384)); CK(hipMalloc(&census, 64)); CK(hipMalloc(&buf, 2 * 32 * 512 * 4));
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int pf : {0, 512}) for (int same : {1, 2, 4, 8}) {
    float best = 1e9;
    unsigned h[16];
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(tickets, 0, 8 * LINE * 4)); CK(hipMemset(sync, 0, 16384)); CK(hipMemset(census, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 84 * 1024, 0, tickets, sync, buf, iters, same, pf, census);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      CK(hipMemcpy(h, census, 64, hipMemcpyDeviceToHost));
    }
    printf("payload %3d floats/WG  over %d XCD(s): %.2f us per barrier+exchange   (census per XCD:", pf, same, best * 1000 / iters);
    for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
    printf(")\n");
  }
  return 0;
}
