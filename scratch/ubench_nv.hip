// All-to-all exchange time of the self-validating hand-off as a function of the group size N and the slab size per workgroup
// (scratch; design input: is a chain of two independent 16-workgroup groups cheaper per seam than one 32-workgroup group?).
// G independent groups of N workgroups run side by side (G*N workgroups launched, dispatch-order placement); per exchange every
// workgroup stores `sf` floats and polls the N*sf floats of its group until none is the sentinel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;
constexpr unsigned SENT = 0x7fc0dead;
__global__ __launch_bounds__(512) void k_ex(float* buf, unsigned bytes, int iters, int N, int sf, unsigned* fail) {
  extern __shared__ float smem[];
  const unsigned g = blockIdx.x / N, w = blockIdx.x % N, tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const unsigned per_it = gridDim.x * sf;                 // floats per exchange, all groups
  const unsigned gsz = (unsigned)N * sf;                  // floats of one group per exchange
  float carry = (float)(blockIdx.x + 1) * 1e-3f;
  for (int it = 0; it < iters; ++it) {
    const unsigned base = (unsigned)it * per_it + g * gsz;
    if (tid < (unsigned)sf) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(carry + tid * 1e-6f), r, (base + w * sf + tid) * 4, 0, AUX_SC1);
    float acc = 0.f;
    const int nld = (int)((gsz / 4 + 511) / 512);         // 16-byte loads per thread
    u32x4 v[8];
    unsigned spins = 0;
    while (true) {
      bool bad = false;
#pragma unroll
      for (int p = 0; p < 8; ++p)
        if (p < nld) {
          const unsigned o = (tid + p * 512) * 4;
          v[p] = o < gsz ? __builtin_amdgcn_raw_buffer_load_b128(r, (base + o) * 4, 0, AUX_SC1) : u32x4{0, 0, 0, 0};
          bad |= v[p].x == SENT || v[p].y == SENT || v[p].z == SENT || v[p].w == SENT;
        }
      if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;
      if (++spins > (1u << 20)) { *fail = 1; break; }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) if (p < nld) acc += __uint_as_float(v[p].x) + __uint_as_float(v[p].w);
    acc += __shfl_xor(acc, 1, 64);
    carry = carry * 0.5f + acc * 1e-4f;
    __syncthreads();
  }
  if (carry == -1.f) buf[0] = carry;
}
int main() {
  const int iters = 2000;
  const size_t bytes = (size_t)iters * 128 * 512 * 4;
  float* buf; unsigned* fail;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&fail, 4));
  CK(hipFuncSetAttribute((const void*)k_ex, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { int G, N, sf; } cfg[] = {{1, 32, 256}, {1, 32, 128}, {2, 32, 128}, {1, 16, 256}, {1, 16, 128}, {2, 16, 128}, {4, 16, 128}, {4, 16, 64}, {8, 8, 128}, {4, 8, 256}};
  for (auto c : cfg) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(fail, 0, 4));
      CK(hipMemsetD32((hipDeviceptr_t)buf, SENT, (size_t)iters * c.G * c.N * c.sf));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_ex, dim3(c.G * c.N), dim3(512), 84 * 1024, 0, buf, (unsigned)bytes, iters, c.N, c.sf, fail);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    unsigned f = 0; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    printf("%d group(s) x %2d workgroups, %4d B per workgroup (%5.1f KB read per workgroup): %.3f us per exchange%s\n", c.G, c.N, c.sf * 4,
           c.N * c.sf * 4 / 1024.0, best * 1000 / iters, f ? "  (TIME-OUT!)" : "");
  }
  return 0;
}
