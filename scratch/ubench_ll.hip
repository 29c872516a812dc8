// Hand-off variants for the all-to-all seam of the recurrent chains (scratch; design input for round 2).
// N co-resident workgroups; per exchange every workgroup publishes a 1 KB slab (32 rows x 8 units, like one LSTHM role's h slice)
// and then needs all N slabs (N KB).  The next slab depends on what was read, so the exchanges form a dependent chain like the time loop.
//   V0: today's protocol -- sc1 stores, s_waitcnt vmcnt(0), workgroup barrier, replicated counter add, one lane polls, workgroup
//       barrier, sc1 loads.
//   V1: self-validating payload -- slabs are indexed by the exchange number and pre-filled with a sentinel bit pattern that no
//       finite value takes; producers just store (sc1), consumers load (sc1) and re-load until none of their words is the
//       sentinel.  No counter, no drain wait, no barrier on the critical path.
//   V3 / V4: V1 / V0 with the slab stored by ONE wave instruction of 16-byte stores instead of 256 scalar stores.
//   V2: 8-byte {value, tag} pairs (tag = exchange number), slabs ping-pong by parity: no pre-fill needed, twice the bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int LINE = 32, AUX_SC1 = 16;
constexpr unsigned SENT = 0x7fc0dead;          // a quiet-NaN payload no arithmetic produces

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
// PL = 0: nwg workgroups, dispatch-order placement (spread over the XCDs).  PL = 1: 256 workgroups launched, the first `nwg` that
// land on XCD 0 take part (the whole exchange group on ONE XCD); payload stores are then PLAIN (they stay in that XCD's L2) and the
// loads keep their sc1 bit (L1-bypassing, L2-served).
template <int V, int PL>
__global__ __launch_bounds__(512) void k_ex(unsigned* sync, float* buf, unsigned bytes, int iters, unsigned long long* tstamp, float* sink,
                                            unsigned* fail, unsigned* tickets, unsigned nwg_arg) {
  extern __shared__ float smem[];
  __shared__ int s_w;
  unsigned nwg = gridDim.x, w = blockIdx.x;
  const unsigned tid = threadIdx.x;
  if (PL) {
    if (tid == 0) {
      const unsigned x = xcc_id();
      const unsigned tk = x == 0 ? __hip_atomic_fetch_add((gu32*)tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffu;
      s_w = tk < nwg_arg ? (int)tk : -1;
    }
    __syncthreads();
    if (s_w < 0) return;
    w = (unsigned)s_w;
    nwg = nwg_arg;
  }
  constexpr int ST_AUX = PL ? 0 : AUX_SC1;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const unsigned slab_f = nwg * 256;            // floats per exchange (all workgroups' slabs)
  float carry = (float)(w + 1) * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    // ---- publish my slab (256 threads x 1 float)
    if (V == 2) {
      if (tid < 256) {
        u32x2 v; v.x = __float_as_uint(carry + tid * 1e-6f); v.y = (unsigned)(it + 1);
        __builtin_amdgcn_raw_buffer_store_b64(v, r, (((it & 1) * slab_f + w * 256 + tid) * 8), 0, ST_AUX);
      }
    } else if (V == 3 || V == 4) {
      // the same 1 KB slab as ONE wave instruction of 16-byte stores (a scalar sc1 store is one fabric write each)
      const unsigned base = (V == 3 ? (unsigned)it : (unsigned)(it & 1)) * slab_f;
      if (tid < 64) {
        u32x4 v;
        v.x = __float_as_uint(carry + (4 * tid) * 1e-6f); v.y = __float_as_uint(carry + (4 * tid + 1) * 1e-6f);
        v.z = __float_as_uint(carry + (4 * tid + 2) * 1e-6f); v.w = __float_as_uint(carry + (4 * tid + 3) * 1e-6f);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + w * 256 + tid * 4) * 4, 0, ST_AUX);
      }
    } else {
      const unsigned base = ((V == 1 || V == 5 || V == 6) ? (unsigned)it : (unsigned)(it & 1)) * slab_f;
      if (tid < 256) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(carry + tid * 1e-6f), r, (base + w * 256 + tid) * 4, 0, ST_AUX);
    }
    float acc = 0.f;
    if (V == 0 || V == 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned gen = (unsigned)(it + 1);
      if (tid < 8) __hip_atomic_fetch_add((gu32*)sync + tid * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 0) {
        const unsigned* c = sync + (w % 8) * LINE;
        unsigned spins = 0;
        while (__hip_atomic_load((const gu32*)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg * gen) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) { *fail = 1; break; }
        }
      }
      __syncthreads();
      const unsigned base = (unsigned)(it & 1) * slab_f;
      for (unsigned o = tid * 8; o < slab_f; o += 512 * 8) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, (base + o) * 4, 0, AUX_SC1);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, (base + o) * 4 + 16, 0, AUX_SC1);
        acc += __uint_as_float(a.x) + __uint_as_float(a.w) + __uint_as_float(b.y) + __uint_as_float(b.z);
      }
    } else if (V == 1 || V == 3 || V == 5 || V == 6) {
      // all of this thread's words are requested at once (NPASS x 32 bytes in flight), then validated together
      const unsigned base = (unsigned)it * slab_f;
      if (V == 5) {
        // indicator: wave 0, lane i polls the first 16 bytes of producer i's slab (64 lanes cover up to 64 producers); the slab's other
        // words are NOT ordered behind it, so the full read below still validates every word -- it just rarely has to retry
        if (tid < 64) {
          unsigned spins = 0;
          while (true) {
            u32x4 a = {1, 1, 1, 1};
            if (tid < nwg) a = __builtin_amdgcn_raw_buffer_load_b128(r, (base + tid * 256 + 252) * 4, 0, AUX_SC1);
            const bool bad = a.x == SENT || a.y == SENT || a.z == SENT || a.w == SENT;
            if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) { *fail = 1; break; }
          }
        }
        __syncthreads();
      }
      constexpr int NPASS = 4;                 // up to 64 workgroups: 16384 floats / (512 threads x 8)
      const int np = (int)(slab_f / (512 * 8));
      u32x4 a[NPASS], b[NPASS];
      unsigned spins = 0;
      while (true) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
          if (p < np) {
            a[p] = __builtin_amdgcn_raw_buffer_load_b128(r, (base + tid * 8 + p * 4096) * 4, 0, AUX_SC1);
            b[p] = __builtin_amdgcn_raw_buffer_load_b128(r, (base + tid * 8 + p * 4096) * 4 + 16, 0, AUX_SC1);
          }
        bool bad = false;
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
          if (p < np)
            bad |= a[p].x == SENT || a[p].y == SENT || a[p].z == SENT || a[p].w == SENT || b[p].x == SENT || b[p].y == SENT ||
                   b[p].z == SENT || b[p].w == SENT;
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;
        if (V == 6) __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 20)) { *fail = 1; break; }
      }
#pragma unroll
      for (int p = 0; p < NPASS; ++p)
        if (p < np) acc += __uint_as_float(a[p].x) + __uint_as_float(a[p].w) + __uint_as_float(b[p].y) + __uint_as_float(b[p].z);
    } else {
      const unsigned base = (unsigned)(it & 1) * slab_f;
      const unsigned gen = (unsigned)(it + 1);
      for (unsigned o = tid * 8; o < slab_f; o += 512 * 8) {
        u32x4 v[4];
        unsigned spins = 0;
        while (true) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, (base + o + 2 * k) * 8, 0, AUX_SC1);
          bool bad = false;
#pragma unroll
          for (int k = 0; k < 4; ++k) bad |= v[k].y != gen || v[k].w != gen;
          if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;
          if (++spins > (1u << 20)) { *fail = 1; break; }
        }
        acc += __uint_as_float(v[0].x) + __uint_as_float(v[1].z) + __uint_as_float(v[2].z) + __uint_as_float(v[3].x);
      }
    }
    // a little dependent work + the workgroup-wide combine the real epilogue has
    acc += __shfl_xor(acc, 1, 64);
    carry = carry * 0.5f + acc * 1e-4f;
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) tstamp[w] = t1 - t0;
  if (carry == -1.f) sink[0] = carry;
}

template <int V, int PL>
int run(unsigned* sync, float* buf, size_t bytes, unsigned long long* ts, float* sink, unsigned* fail, unsigned* tickets, int nwg, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k_ex<V, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(sync, 0, 16384));
    CK(hipMemset(fail, 0, 4));
    CK(hipMemset(tickets, 0, 4));
    if (V == 1 || V == 3 || V == 5 || V == 6) CK(hipMemsetD32((hipDeviceptr_t)buf, SENT, (size_t)iters * nwg * 256));
    else CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_ex<V, PL>), dim3(PL ? 256 : nwg), dim3(512), 84 * 1024, 0, sync, buf, (unsigned)bytes, iters, ts, sink, fail, tickets, (unsigned)nwg);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned f = 0; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
  unsigned tk = 0; CK(hipMemcpy(&tk, tickets, 4, hipMemcpyDeviceToHost));
  printf("V%d %s nwg %2d: %.3f us per exchange%s", V, PL ? "one-XCD" : "spread ", nwg, best * 1000 / iters, f ? "  (TIME-OUT!)" : "");
  if (PL) printf("   (workgroups seen on XCD 0: %u)", tk);
  printf("\n");
  return 0;
}
int main() {
  const int iters = 2000;
  unsigned* sync; float* buf; unsigned long long* ts; float* sink; unsigned* fail; unsigned* tickets;
  const size_t bytes = (size_t)iters * 64 * 256 * 4 * 2;
  CK(hipMalloc(&sync, 16384)); CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&ts, 256 * 8)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&fail, 4));
  CK(hipMalloc(&tickets, 64));
  for (int nwg : {32, 64}) {
    if (run<0, 0>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<1, 0>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<2, 0>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<3, 0>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<4, 0>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<5, 0>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<6, 0>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
  }
  for (int nwg : {16, 32}) {
    if (run<0, 1>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<1, 1>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<2, 1>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<5, 1>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
    if (run<6, 1>(sync, buf, bytes, ts, sink, fail, tickets, nwg, iters)) return 1;
  }
  return 0;
}
