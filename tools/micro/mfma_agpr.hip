// micro-benchmark: cycles per v_mfma_f32_16x16x32_bf16 with the B operand in an AGPR vs a VGPR (4 interleaved chains)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, const u32x4* in) {
  u32x4 a = in[threadIdx.x], b0 = in[threadIdx.x + 256], b1 = in[threadIdx.x + 512], b2 = in[threadIdx.x + 768], b3 = in[threadIdx.x + 1024];
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      if (MODE == 0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "a"(b0));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "a"(b1));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "a"(b2));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "a"(b3));
      } else if (MODE == 1) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b0));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b1));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b2));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b3));
      } else {   // accumulators in AGPRs too
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "a"(b0));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "a"(b1));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "a"(b2));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c3) : "v"(a), "a"(b3));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[2048 + blockIdx.x] = r1 - r0;
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (c0.x + c1.x + c2.x + c3.x == 12345.f) out[1000] = 1;
}
int main() {
  unsigned long long* out; u32x4* in;
  hipMalloc(&out, 8192 * 8); hipMalloc(&in, 2048 * 16); hipMemset(in, 0x3c, 2048 * 16);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, in);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, in);
      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, in);
      hipDeviceSynchronize();
    }
    unsigned long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    unsigned long long hr[256]; hipMemcpy(hr, out + 2048, sizeof(hr), hipMemcpyDeviceToHost);
    double sr = 0; for (int i = 0; i < 256; ++i) sr += hr[i];
    printf("mode %d (%s): %.1f s_memtime ticks per MFMA, %.1f ns per MFMA (s_memrealtime, 100 MHz) -- 8 passes at 2.4 GHz would be 13.3 ns\n", mode,
           mode == 0 ? "B in AGPR, acc VGPR" : mode == 1 ? "B in VGPR, acc VGPR" : "B and acc in AGPR", s / 256 / (256.0 * 48), sr / 256 * 10.0 / (256.0 * 48));
  }
  return 0;
}
