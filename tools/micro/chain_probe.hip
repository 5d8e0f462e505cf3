// What does a chain of DEPENDENT small launches cost per link on this GPU, and can the link be made cheaper than the
// in-order queue makes it?  (The BPTT loop of the training step is 85 such links; round-5 question.)
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
// Every kernel reads the previous kernel's output (x -> fma(x, 1.0001, 1)), so a broken order shows in the result.
// Modes:
//   plain      one stream, ordinary launches: the queue's barrier bit orders the kernels
//   flags      the same + the device-side flag protocol (cost of the protocol itself)
//   anyorder   one stream, hipExtLaunchKernel(..., hipExtAnyOrderLaunch): no barrier bit; order by device flags only
//   (capital initial: the same with protocol 1 below)
//   two        two streams, kernel k on stream k & 1, no events between them; order by device flags only (kernel k + 1 is
//              resident and spinning while kernel k runs)
// Flag protocol: a finishing workgroup does fence(release, agent) + atomic add on flags[k]; a starting workgroup of kernel
// k + 1 spins (bounded) until flags[k] == gridDim.x, then fence(acquire, agent).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// PROTO 0: plain loads / stores, release = fence (buffer_wbl2 sc1), acquire = fence (buffer_inv sc1)
// PROTO 1: the data itself moves with agent-scope (sc1) loads and stores -- nothing cached incoherently, no fences, only a wait
//          for the stores before the flag
template <int PROTO>
__global__ __launch_bounds__(256) void link_kernel(const float* __restrict__ in, float* __restrict__ out, int n, unsigned* wait_flag,
                                                   unsigned wait_val, unsigned* sig_flag, unsigned* err) {
  if (wait_flag) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_val) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000u) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    if (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // (invalidates what this XCD's L2 holds of other XCDs' lines)
  }
  if (PROTO == 0) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = __fmaf_rn(in[i], 1.0001f, 1.f);
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      __hip_atomic_store(out + i, __fmaf_rn(__hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 1.0001f, 1.f), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  if (sig_flag) {
    if (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sig_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char** argv) {
  const int K = 200, GRID = 256;
  hipStream_t sA, sB;
  CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
  hipEvent_t e0, e1, eB;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&eB, hipEventDisableTiming));
  unsigned *flags, *err;
  CK(hipMalloc(&flags, (K + 1) * sizeof(unsigned)));
  CK(hipMalloc(&err, 4));
  for (int n : {640 * 512, 640 * 512 * 12}) {
    float *b0, *b1;
    CK(hipMalloc(&b0, (size_t)n * 4)); CK(hipMalloc(&b1, (size_t)n * 4));
    float want = 0.5f;
    for (int k = 0; k < K; ++k) want = fmaf(want, 1.0001f, 1.f);
    for (const char* mode : {"plain", "flags", "anyorder", "two", "Flags", "Anyorder", "Two", "plain"}) {
      const char m0 = mode[0] | 0x20;
      const bool wt = !(mode[0] & 0x20);            // capital: PROTO 1
      const bool use_flags = m0 != 'p', any = m0 == 'a', two = m0 == 't';
      std::vector<float> ms;
      float got = 0.f;
      unsigned herr = 0;
      for (int rep = 0; rep < 7; ++rep) {
        std::vector<float> init(n, 0.5f);
        CK(hipMemcpy(b0, init.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(flags, 0, (K + 1) * sizeof(unsigned), sA));
        CK(hipMemsetAsync(err, 0, 4, sA));
        CK(hipStreamSynchronize(sA));
        CK(hipEventRecord(e0, sA));
        if (two) { CK(hipEventRecord(eB, sA)); CK(hipStreamWaitEvent(sB, eB, 0)); }
        for (int k = 0; k < K; ++k) {
          const float* in = (k & 1) ? b1 : b0;
          float* out = (k & 1) ? b0 : b1;
          unsigned* wf = use_flags && k > 0 ? flags + (k - 1) : nullptr;
          unsigned wv = GRID;
          unsigned* sf = use_flags ? flags + k : nullptr;
          hipStream_t s = two && (k & 1) ? sB : sA;
          if (any) {
            int nn = n;
            void* args[] = {(void*)&in, (void*)&out, (void*)&nn, (void*)&wf, (void*)&wv, (void*)&sf, (void*)&err};
            CK(hipExtLaunchKernel(wt ? (const void*)link_kernel<1> : (const void*)link_kernel<0>, dim3(GRID), dim3(256), args, 0, s, nullptr, nullptr, k > 0 ? hipExtAnyOrderLaunch : 0));
          } else {
            if (wt) hipLaunchKernelGGL(link_kernel<1>, dim3(GRID), dim3(256), 0, s, in, out, n, wf, wv, sf, err);
            else hipLaunchKernelGGL(link_kernel<0>, dim3(GRID), dim3(256), 0, s, in, out, n, wf, wv, sf, err);
          }
        }
        if (two) { CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0)); }
        CK(hipEventRecord(e1, sA));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
        CK(hipMemcpy(&got, (K & 1) ? b1 : b0, 4, hipMemcpyDeviceToHost));
        float last;
        CK(hipMemcpy(&last, ((K & 1) ? b1 : b0) + n - 1, 4, hipMemcpyDeviceToHost));
        if (last != got) got = NAN;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      }
      std::sort(ms.begin(), ms.end());
      printf("n = %8d floats  %-9s %7.2f us per link (median of 7 chains of %d; min %.2f)   result %s%s\n", n, mode, ms[3] * 1e3 / K, K, ms[0] * 1e3 / K,
             got == want ? "correct" : "WRONG", herr ? "  SPIN TIMED OUT" : "");
    }
    CK(hipFree(b0)); CK(hipFree(b1));
  }
  return 0;
}
