// Stand-alone timing of attention-step forward kernel variants at the benchmark shapes (N = 640 caption rows, R = 36 regions,
// A = H = 512, bf16), HBM-cold (8 rotating p_att / att' pairs = 377 MB > the 256 MiB Infinity Cache), one hipEvent pair per
// launch, trimmed mean.  Build + run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/attn_variants.hip -o /tmp/attn_variants && /tmp/attn_variants
// Variants:
//   base<NW>     the shipped structure (attention.hip attn_fwd_fast_kernel): scores -> LDS -> sync -> every wave the 36-way
//                softmax -> context partials -> LDS -> sync -> NW-way sum
//   stream<NW>   loads only (+ a trivial use of every byte): what the grid costs without the arithmetic
//   online<NW>   each wave keeps (max, sum, partial context) of ITS regions (online softmax); one sync; merge of NW partials
//   multi<NW,G>  online<NW> with G workgroups per CU-slot looping over rows (next row's loads in flight during this row's math)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>

typedef __bf16 bf16_t;
constexpr int N = 640, R = 36, A = 512, H = 512;

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ float row16_sum(float v) {
#define DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  DPP_ADD(0xB1); DPP_ADD(0x4E); DPP_ADD(0x141); DPP_ADD(0x140);
#undef DPP_ADD
  return v;
}
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

struct P { const float* att_h; const bf16_t* p_att; const bf16_t* att; const float* w; float* alpha; bf16_t* ctx; };

template <bool NT> __device__ __forceinline__ uint4 ld16(const bf16_t* q) {
  if (NT) { typedef __attribute__((ext_vector_type(4))) unsigned u4; const u4 v = __builtin_nontemporal_load((const u4*)q); return make_uint4(v.x, v.y, v.z, v.w); }
  return *(const uint4*)q;
}
template <int NW, bool NT = false>
__global__ __launch_bounds__(NW * 64) void base_kernel(P p) {
  constexpr int UB = (36 + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_e = sm; float* s_red = s_e + 4 * R;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* pa = p.p_att + (size_t)n * R * A + lane * 8;
  const bf16_t* pt = p.att + (size_t)n * R * H + lane * 8;
  uint4 vp[UB], va[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) vp[u] = ld16<NT>(pa + (size_t)min(wave + u * NW, R - 1) * A);
#pragma unroll
  for (int u = 0; u < UB; ++u) va[u] = ld16<NT>(pt + (size_t)min(wave + u * NW, R - 1) * H);
  float ah[8], w[8];
  { const float4 a0 = *(const float4*)(p.att_h + (size_t)n * A + lane * 8), a1 = *(const float4*)(p.att_h + (size_t)n * A + lane * 8 + 4);
    ah[0] = a0.x; ah[1] = a0.y; ah[2] = a0.z; ah[3] = a0.w; ah[4] = a1.x; ah[5] = a1.y; ah[6] = a1.z; ah[7] = a1.w;
    const float4 w0 = *(const float4*)(p.w + lane * 8), w1 = *(const float4*)(p.w + lane * 8 + 4);
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w; }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int r = wave + u * NW;
    float f[8]; unpack8(vp[u], f);
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) part += w[j] * tanh_fast(f[j] + ah[j]);
    part = row16_sum(part);
    if ((lane & 15) == 0 && r < R) s_e[r * 4 + (lane >> 4)] = part;
  }
  __syncthreads();
  float e = -INFINITY;
  if (lane < R) { const float4 q = *(const float4*)(s_e + lane * 4); e = (q.x + q.y) + (q.z + q.w); }
  const float mx = wave_max(e);
  const float ex = lane < R ? __builtin_amdgcn_exp2f((e - mx) * 1.4426950408889634f) : 0.f;
  const float wgt = ex * __builtin_amdgcn_rcpf(wave_sum(ex));
  if (wave == 0 && lane < R) p.alpha[(size_t)n * R + lane] = wgt;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int r = wave + u * NW;
    float al = __shfl(wgt, r < R ? r : 0, 64);
    if (r >= R) al = 0.f;
    float f[8]; unpack8(va[u], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += al * f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_red[wave * H + lane * 8 + j] = acc[j];
  __syncthreads();
  for (int h = tid; h < H; h += NW * 64) {
    float a2 = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) a2 += s_red[wv * H + h];
    p.ctx[(size_t)n * H + h] = (bf16_t)a2;
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void stream_kernel(P p) {
  constexpr int UB = (36 + NW - 1) / NW;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* pa = p.p_att + (size_t)n * R * A + lane * 8;
  const bf16_t* pt = p.att + (size_t)n * R * H + lane * 8;
  uint4 vp[UB], va[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) vp[u] = *(const uint4*)(pa + (size_t)min(wave + u * NW, R - 1) * A);
#pragma unroll
  for (int u = 0; u < UB; ++u) va[u] = *(const uint4*)(pt + (size_t)min(wave + u * NW, R - 1) * H);
  unsigned x = 0;
#pragma unroll
  for (int u = 0; u < UB; ++u) x ^= vp[u].x ^ vp[u].y ^ vp[u].z ^ vp[u].w ^ va[u].x ^ va[u].y ^ va[u].z ^ va[u].w;
  if (x == 0x12345678u) p.alpha[(size_t)n * R] = 1.f;
}

// one row: online softmax per wave, merge at the end.  regs: the row's chunks (UB + UB) x 4
template <int NW>
__device__ __forceinline__ void online_row(const P& p, int n, const uint4 (&vp)[(36 + NW - 1) / NW], const uint4 (&va)[(36 + NW - 1) / NW], float* sm) {
  constexpr int UB = (36 + NW - 1) / NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* s_ms = sm;                 // [NW][2] max, sum
  float* s_red = sm + 2 * NW;       // [NW][H]
  float* s_e = s_red + NW * H;      // [R] raw scores (for alpha)
  float ah[8], w[8];
  { const float4 a0 = *(const float4*)(p.att_h + (size_t)n * A + lane * 8), a1 = *(const float4*)(p.att_h + (size_t)n * A + lane * 8 + 4);
    ah[0] = a0.x; ah[1] = a0.y; ah[2] = a0.z; ah[3] = a0.w; ah[4] = a1.x; ah[5] = a1.y; ah[6] = a1.z; ah[7] = a1.w;
    const float4 w0 = *(const float4*)(p.w + lane * 8), w1 = *(const float4*)(p.w + lane * 8 + 4);
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w; }
  float e[UB];
  float mloc = -INFINITY;
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int r = wave + u * NW;
    float f[8]; unpack8(vp[u], f);
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) part += w[j] * tanh_fast(f[j] + ah[j]);
    part = wave_sum(part);
    e[u] = r < R ? part : -INFINITY;
    mloc = fmaxf(mloc, e[u]);
    if (lane == 0 && r < R) s_e[r] = part;
  }
  float acc[8], sloc = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const float wg = __builtin_amdgcn_exp2f((e[u] - mloc) * 1.4426950408889634f);
    sloc += wg;
    float f[8]; unpack8(va[u], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += wg * f[j];
  }
  if (lane == 0) { s_ms[2 * wave] = mloc; s_ms[2 * wave + 1] = sloc; }
  *(float4*)(s_red + wave * H + lane * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *(float4*)(s_red + wave * H + lane * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int wv = 0; wv < NW; ++wv) M = fmaxf(M, s_ms[2 * wv]);
  float S = 0.f, sc[NW];
#pragma unroll
  for (int wv = 0; wv < NW; ++wv) { sc[wv] = __builtin_amdgcn_exp2f((s_ms[2 * wv] - M) * 1.4426950408889634f); S += sc[wv] * s_ms[2 * wv + 1]; }
  const float inv = __builtin_amdgcn_rcpf(S);
  for (int h = tid; h < H; h += NW * 64) {
    float a2 = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) a2 += sc[wv] * s_red[wv * H + h];
    p.ctx[(size_t)n * H + h] = (bf16_t)(a2 * inv);
  }
  if (tid < R) p.alpha[(size_t)n * R + tid] = __builtin_amdgcn_exp2f((s_e[tid] - M) * 1.4426950408889634f) * inv;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void online_kernel(P p) {
  constexpr int UB = (36 + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* pa = p.p_att + (size_t)n * R * A + lane * 8;
  const bf16_t* pt = p.att + (size_t)n * R * H + lane * 8;
  uint4 vp[UB], va[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) vp[u] = *(const uint4*)(pa + (size_t)min(wave + u * NW, R - 1) * A);
#pragma unroll
  for (int u = 0; u < UB; ++u) va[u] = *(const uint4*)(pt + (size_t)min(wave + u * NW, R - 1) * H);
  online_row<NW>(p, n, vp, va, sm);
}

// grid = G workgroups, each walks rows n = blockIdx.x, + G, ...; the next row's chunks are requested before this row's math
template <int NW>
__global__ __launch_bounds__(NW * 64) void multi_kernel(P p) {
  constexpr int UB = (36 + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, G = gridDim.x;
  uint4 vp[UB], va[UB], np_[UB], na[UB];
  int n = blockIdx.x;
  {
    const bf16_t* pa = p.p_att + (size_t)n * R * A + lane * 8;
    const bf16_t* pt = p.att + (size_t)n * R * H + lane * 8;
#pragma unroll
    for (int u = 0; u < UB; ++u) vp[u] = *(const uint4*)(pa + (size_t)min(wave + u * NW, R - 1) * A);
#pragma unroll
    for (int u = 0; u < UB; ++u) va[u] = *(const uint4*)(pt + (size_t)min(wave + u * NW, R - 1) * H);
  }
  for (; n < N; n += G) {
    const int nn = n + G < N ? n + G : n;
    {
      const bf16_t* pa = p.p_att + (size_t)nn * R * A + lane * 8;
      const bf16_t* pt = p.att + (size_t)nn * R * H + lane * 8;
#pragma unroll
      for (int u = 0; u < UB; ++u) np_[u] = *(const uint4*)(pa + (size_t)min(wave + u * NW, R - 1) * A);
#pragma unroll
      for (int u = 0; u < UB; ++u) na[u] = *(const uint4*)(pt + (size_t)min(wave + u * NW, R - 1) * H);
    }
    online_row<NW>(p, n, vp, va, sm);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UB; ++u) { vp[u] = np_[u]; va[u] = na[u]; }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static void timeit(const char* name, F launch, int iters = 96) {
  for (int i = 0; i < 16; ++i) launch(i % 8);
  std::vector<hipEvent_t> a(iters), b(iters);
  for (int i = 0; i < iters; ++i) { CK(hipEventCreate(&a[i])); CK(hipEventCreate(&b[i])); }
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a[i], 0)); launch(i % 8); CK(hipEventRecord(b[i], 0)); }
  CK(hipDeviceSynchronize());
  std::vector<float> d(iters);
  for (int i = 0; i < iters; ++i) CK(hipEventElapsedTime(&d[i], a[i], b[i]));
  std::sort(d.begin(), d.end());
  double s = 0; int c = 0;
  for (int i = iters / 8; i < iters - iters / 8; ++i) { s += d[i]; ++c; }
  // back-to-back: one event pair around all launches
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch(i % 8);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float tot; CK(hipEventElapsedTime(&tot, e0, e1));
  const double bytes = (double)N * (R * A + R * H + 2 * H + R) * 2;
  const double us = s / c * 1e3, usb = tot / iters * 1e3;
  printf("%-22s event-pair %6.2f us (%.3f of 8 TB/s)   back-to-back %6.2f us (%.3f)   min %6.2f us\n", name, us, bytes / us / 1e3 / 8000.0,
         usb, bytes / usb / 1e3 / 8000.0, d[0] * 1e3);
}

int main() {
  std::vector<bf16_t*> pa(8), at(8);
  const size_t el = (size_t)N * R * A;
  std::vector<unsigned short> host(el);
  for (size_t i = 0; i < el; ++i) host[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 24));   // bf16 values around 0.01..0.03
  for (int k = 0; k < 8; ++k) {
    CK(hipMalloc(&pa[k], el * 2)); CK(hipMalloc(&at[k], el * 2));
    CK(hipMemcpy(pa[k], host.data(), el * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(at[k], host.data(), el * 2, hipMemcpyHostToDevice));
  }
  float *att_h, *w, *alpha, *alpha2; bf16_t *ctx, *ctx2;
  CK(hipMalloc(&att_h, N * A * 4)); CK(hipMalloc(&w, A * 4)); CK(hipMalloc(&alpha, N * R * 4)); CK(hipMalloc(&alpha2, N * R * 4));
  CK(hipMalloc(&ctx, N * H * 2)); CK(hipMalloc(&ctx2, N * H * 2));
  std::vector<float> hh(N * A), hw(A);
  for (int i = 0; i < N * A; ++i) hh[i] = 0.5f * sinf(0.37f * i);
  for (int i = 0; i < A; ++i) hw[i] = 0.3f * cosf(1.7f * i);
  CK(hipMemcpy(att_h, hh.data(), N * A * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), A * 4, hipMemcpyHostToDevice));
  auto mk = [&](int k, float* al, bf16_t* cx) { return P{att_h, pa[k], at[k], w, al, cx}; };
#define LDS_BASE(NW) (sizeof(float) * (4 * R + 4 + NW * H))
#define LDS_ONL(NW) (sizeof(float) * (2 * NW + NW * H + R + 4))
  CK(hipFuncSetAttribute((const void*)base_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  // correctness of online vs base on buffer 0
  hipLaunchKernelGGL(base_kernel<12>, dim3(N), dim3(768), LDS_BASE(12), 0, mk(0, alpha, ctx));
  hipLaunchKernelGGL(online_kernel<8>, dim3(N), dim3(512), LDS_ONL(8), 0, mk(0, alpha2, ctx2));
  CK(hipDeviceSynchronize());
  {
    std::vector<float> a1(N * R), a2(N * R); std::vector<unsigned short> c1(N * H), c2(N * H);
    CK(hipMemcpy(a1.data(), alpha, N * R * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a2.data(), alpha2, N * R * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c1.data(), ctx, N * H * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), ctx2, N * H * 2, hipMemcpyDeviceToHost));
    double da = 0, dc = 0;
    for (int i = 0; i < N * R; ++i) da = std::max(da, (double)fabsf(a1[i] - a2[i]));
    for (int i = 0; i < N * H; ++i) { unsigned u1 = (unsigned)c1[i] << 16, u2 = (unsigned)c2[i] << 16; float f1, f2; memcpy(&f1, &u1, 4); memcpy(&f2, &u2, 4); dc = std::max(dc, (double)fabsf(f1 - f2)); }
    printf("online vs base: max |d alpha| %.2e  max |d ctx| %.2e\n", da, dc);
  }
  timeit("base<12>", [&](int k) { hipLaunchKernelGGL(base_kernel<12>, dim3(N), dim3(768), LDS_BASE(12), 0, mk(k, alpha, ctx)); });
  timeit("base<12> nt", [&](int k) { hipLaunchKernelGGL((base_kernel<12, true>), dim3(N), dim3(768), LDS_BASE(12), 0, mk(k, alpha, ctx)); });
  timeit("base<16>", [&](int k) { hipLaunchKernelGGL(base_kernel<16>, dim3(N), dim3(1024), LDS_BASE(16), 0, mk(k, alpha, ctx)); });
  timeit("base<16> nt", [&](int k) { hipLaunchKernelGGL((base_kernel<16, true>), dim3(N), dim3(1024), LDS_BASE(16), 0, mk(k, alpha, ctx)); });
  timeit("base<10>", [&](int k) { hipLaunchKernelGGL(base_kernel<10>, dim3(N), dim3(640), LDS_BASE(10), 0, mk(k, alpha, ctx)); });
  timeit("base<10> nt", [&](int k) { hipLaunchKernelGGL((base_kernel<10, true>), dim3(N), dim3(640), LDS_BASE(10), 0, mk(k, alpha, ctx)); });
  timeit("base<9>", [&](int k) { hipLaunchKernelGGL(base_kernel<9>, dim3(N), dim3(576), LDS_BASE(9), 0, mk(k, alpha, ctx)); });
  timeit("base<8>", [&](int k) { hipLaunchKernelGGL(base_kernel<8>, dim3(N), dim3(512), LDS_BASE(8), 0, mk(k, alpha, ctx)); });
  timeit("base<4>", [&](int k) { hipLaunchKernelGGL(base_kernel<4>, dim3(N), dim3(256), LDS_BASE(4), 0, mk(k, alpha, ctx)); });
  timeit("stream<12>", [&](int k) { hipLaunchKernelGGL(stream_kernel<12>, dim3(N), dim3(768), 0, 0, mk(k, alpha, ctx)); });
  timeit("stream<8>", [&](int k) { hipLaunchKernelGGL(stream_kernel<8>, dim3(N), dim3(512), 0, 0, mk(k, alpha, ctx)); });
  timeit("stream<4>", [&](int k) { hipLaunchKernelGGL(stream_kernel<4>, dim3(N), dim3(256), 0, 0, mk(k, alpha, ctx)); });
  timeit("online<12>", [&](int k) { hipLaunchKernelGGL(online_kernel<12>, dim3(N), dim3(768), LDS_ONL(12), 0, mk(k, alpha, ctx)); });
  timeit("online<8>", [&](int k) { hipLaunchKernelGGL(online_kernel<8>, dim3(N), dim3(512), LDS_ONL(8), 0, mk(k, alpha, ctx)); });
  timeit("online<6>", [&](int k) { hipLaunchKernelGGL(online_kernel<6>, dim3(N), dim3(384), LDS_ONL(6), 0, mk(k, alpha, ctx)); });
  timeit("online<4>", [&](int k) { hipLaunchKernelGGL(online_kernel<4>, dim3(N), dim3(256), LDS_ONL(4), 0, mk(k, alpha, ctx)); });
  return 0;
}
