/* A plain-C client of libuic_hip.so: no Python, no torch, no C++ -- only include/uic_hip.h and the HIP runtime for
 * device memory.  Compiled and run by tests/test_gpu_c_abi.py on the GPU box:
 *     gcc -std=c11 tests/c_abi_client.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lunpaired_image_captioning_amd -luic_hip \
 *         -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_client
 * Exercises: the workspace-size queries, an argument error with its message, uic_linear (f32), uic_linear_f32a, uic_attention_fwd (f32),
 * uic_adam_step and the fused captioner training step (uic_topdown_refresh_weights + uic_topdown_xe_train_step) against
 * loops / closed forms written here, on the caller's own stream.  Prints "C ABI OK" and exits 0. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "uic_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_UIC(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, uic_last_error_string()); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 65536.0f - 0.5f; }

static void* to_dev(const void* h, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (h && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  if (uic_version() < 100) { fprintf(stderr, "bad version\n"); return 1; }

  /* size queries + an argument error */
  uic_topdown_dims d;
  memset(&d, 0, sizeof(d));
  d.N = 640; d.R = 36; d.D = 2048; d.Dfc = 2048; d.H = 512; d.E = 512; d.A = 512; d.V1 = 9488; d.T = 17;
  d.dtype = UIC_DTYPE_BF16; d.drop_p = 0.5f;
  size_t ws = uic_topdown_workspace_bytes(&d), dv = uic_topdown_derived_bytes(&d);
  if (ws < ((size_t)1 << 30) || dv == 0) { fprintf(stderr, "workspace %zu derived %zu\n", ws, dv); return 1; }
  d.H = 510;
  if (uic_topdown_workspace_bytes(&d) != 0 || !strstr(uic_last_error_string(), "multiples of 8")) { fprintf(stderr, "no error for H=510\n"); return 1; }
  if (uic_adam_step(NULL, NULL, NULL, NULL, 8, 1e-3f, 0.9f, 0.999f, 1e-8f, 1, 1.0f, stream) >= 0) { fprintf(stderr, "null pointers accepted\n"); return 1; }

  /* uic_linear, f32: C[M,N] = A[M,K] B[N,K]^T + bias, ReLU */
  unsigned seed = 7;
  {
    const int M = 70, N = 48, K = 64;
    float *A = malloc(sizeof(float) * M * K), *B = malloc(sizeof(float) * N * K), *bias = malloc(sizeof(float) * N), *C = malloc(sizeof(float) * M * N);
    for (int i = 0; i < M * K; ++i) A[i] = frand(&seed);
    for (int i = 0; i < N * K; ++i) B[i] = frand(&seed);
    for (int i = 0; i < N; ++i) bias[i] = frand(&seed);
    float *dA = to_dev(A, sizeof(float) * M * K), *dB = to_dev(B, sizeof(float) * N * K), *db = to_dev(bias, sizeof(float) * N), *dC = to_dev(NULL, sizeof(float) * M * N);
    if (!dA || !dB || !db || !dC) return 2;
    CHECK_UIC(uic_linear(UIC_DTYPE_F32, M, N, K, dA, K, dB, K, dC, N, db, 1 | 4, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(C, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double acc = bias[n];
        for (int k = 0; k < K; ++k) acc += (double)A[m * K + k] * B[n * K + k];
        if (acc < 0) acc = 0;
        double e = fabs(acc - C[m * N + n]);
        if (e > worst) worst = e;
      }
    if (worst > 1e-4) { fprintf(stderr, "uic_linear off by %g\n", worst); return 1; }
    hipFree(dA); hipFree(dB); hipFree(db); hipFree(dC); free(A); free(B); free(bias); free(C);
  }

  /* uic_linear_f32a: f32 input rounded inside the GEMM, bf16 weights, f32 output + the bf16 image of the input.
   * bf16 <-> f32 by hand (round to nearest even), the product against a double loop over the ROUNDED operands. */
  {
    const int M = 384, N = 256, K = 256;
    float *A = malloc(sizeof(float) * M * K), *C = malloc(sizeof(float) * M * N);
    unsigned short *Bh = malloc(2 * N * K), *img = malloc(2 * M * K);
    float *Br = malloc(sizeof(float) * N * K), *Ar = malloc(sizeof(float) * M * K);
    for (int i = 0; i < M * K; ++i) A[i] = frand(&seed) * 3.0f;
    for (int i = 0; i < N * K; ++i) {
      float f = frand(&seed) * 0.25f; unsigned u; memcpy(&u, &f, 4);
      u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; Bh[i] = (unsigned short)u; u <<= 16; memcpy(&Br[i], &u, 4);
    }
    for (int i = 0; i < M * K; ++i) {
      unsigned u; memcpy(&u, &A[i], 4);
      u = ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16) << 16; memcpy(&Ar[i], &u, 4);
    }
    float* dA = to_dev(A, sizeof(float) * M * K);
    void *dB = to_dev(Bh, 2 * N * K), *dI = to_dev(NULL, 2 * M * K);
    float* dC = to_dev(NULL, sizeof(float) * M * N);
    if (!dA || !dB || !dI || !dC) return 2;
    CHECK_UIC(uic_linear_f32a(M, N, K, dA, K, dB, K, dC, N, NULL, 4, dI, K, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(C, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(img, dI, 2 * M * K, hipMemcpyDeviceToHost));
    for (int i = 0; i < M * K; ++i) {
      unsigned u; memcpy(&u, &Ar[i], 4);
      if (img[i] != (unsigned short)(u >> 16)) { fprintf(stderr, "uic_linear_f32a: bf16 image differs at %d\n", i); return 1; }
    }
    double worst = 0;
    for (int m = 0; m < M; m += 7)
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)Ar[m * K + k] * Br[n * K + k];
        double e = fabs(acc - C[m * N + n]);
        if (e > worst) worst = e;
      }
    if (worst > 2e-3) { fprintf(stderr, "uic_linear_f32a off by %g\n", worst); return 1; }
    if (uic_linear_f32a(M, N, 192, dA, K, dB, K, dC, N, NULL, 4, NULL, K, stream) >= 0) { fprintf(stderr, "uic_linear_f32a accepted K = 192\n"); return 1; }
    /* a row stride of the f32 input that is a multiple of 4 floats but not of 8 (the header's contract: lda % 4 == 0): same bits */
    {
      const int lda = K + 4;
      float* A2 = malloc(sizeof(float) * M * lda);
      float* C2 = malloc(sizeof(float) * M * N);
      for (int m = 0; m < M; ++m) {
        for (int k = 0; k < K; ++k) A2[m * lda + k] = A[m * K + k];
        for (int k = K; k < lda; ++k) A2[m * lda + k] = 1e30f;
      }
      float* dA2 = to_dev(A2, sizeof(float) * M * lda);
      if (!dA2) return 2;
      CHECK_UIC(uic_linear_f32a(M, N, K, dA2, lda, dB, K, dC, N, NULL, 4, NULL, K, stream));
      CHECK_HIP(hipStreamSynchronize(stream));
      CHECK_HIP(hipMemcpy(C2, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
      if (memcmp(C, C2, sizeof(float) * M * N) != 0) { fprintf(stderr, "uic_linear_f32a: lda = K + 4 changes the result\n"); return 1; }
      if (uic_linear_f32a(M, N, K, dA2, K + 2, dB, K, dC, N, NULL, 4, NULL, K, stream) >= 0) { fprintf(stderr, "uic_linear_f32a accepted lda %% 4 != 0\n"); return 1; }
      hipFree(dA2); free(A2); free(C2);
    }
    hipFree(dA); hipFree(dB); hipFree(dI); hipFree(dC); free(A); free(C); free(Bh); free(img); free(Br); free(Ar);
  }

  /* uic_attention_fwd, f32: alpha = softmax_r(w . tanh(p_att[r] + att_h) + b), masked + renormalised; ctx = sum_r alpha_r att[r] */
  {
    const int N = 5, R = 7, A = 16, H = 24;
    float *att_h = malloc(sizeof(float) * N * A), *p_att = malloc(sizeof(float) * N * R * A), *att = malloc(sizeof(float) * N * R * H);
    float *w = malloc(sizeof(float) * A), b = 0.3f, *mask = malloc(sizeof(float) * N * R), *alpha = malloc(sizeof(float) * N * R), *ctx = malloc(sizeof(float) * N * H);
    for (int i = 0; i < N * A; ++i) att_h[i] = frand(&seed);
    for (int i = 0; i < N * R * A; ++i) p_att[i] = frand(&seed);
    for (int i = 0; i < N * R * H; ++i) att[i] = frand(&seed);
    for (int i = 0; i < A; ++i) w[i] = frand(&seed);
    for (int n = 0; n < N; ++n) for (int r = 0; r < R; ++r) mask[n * R + r] = r < R - n ? 1.f : 0.f;   /* ragged region counts */
    float *d_att_h = to_dev(att_h, sizeof(float) * N * A), *d_p = to_dev(p_att, sizeof(float) * N * R * A), *d_att = to_dev(att, sizeof(float) * N * R * H);
    float *d_w = to_dev(w, sizeof(float) * A), *d_b = to_dev(&b, sizeof(float)), *d_mask = to_dev(mask, sizeof(float) * N * R);
    float *d_alpha = to_dev(NULL, sizeof(float) * N * R), *d_ctx = to_dev(NULL, sizeof(float) * N * H);
    CHECK_UIC(uic_attention_fwd(UIC_DTYPE_F32, N, R, A, H, d_att_h, d_p, d_att, d_w, d_b, d_mask, d_alpha, d_ctx, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(alpha, d_alpha, sizeof(float) * N * R, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ctx, d_ctx, sizeof(float) * N * H, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int n = 0; n < N; ++n) {
      double e[16], mx = -1e30, sum = 0, msum = 0;
      for (int r = 0; r < R; ++r) {
        double s = b;
        for (int a = 0; a < A; ++a) s += w[a] * tanh((double)p_att[(n * R + r) * A + a] + att_h[n * A + a]);
        e[r] = s; if (s > mx) mx = s;
      }
      for (int r = 0; r < R; ++r) { e[r] = exp(e[r] - mx); sum += e[r]; }
      for (int r = 0; r < R; ++r) { e[r] = e[r] / sum * mask[n * R + r]; msum += e[r]; }
      for (int r = 0; r < R; ++r) { e[r] /= msum; double er = fabs(e[r] - alpha[n * R + r]); if (er > worst) worst = er; }
      for (int h = 0; h < H; ++h) {
        double c = 0;
        for (int r = 0; r < R; ++r) c += e[r] * att[(n * R + r) * H + h];
        double er = fabs(c - ctx[n * H + h]); if (er > worst) worst = er;
      }
    }
    if (worst > 1e-5) { fprintf(stderr, "uic_attention_fwd off by %g\n", worst); return 1; }
  }

  /* uic_adam_step: two steps of torch.optim.Adam on 1000 floats */
  {
    const int n = 1000;
    float *p = malloc(sizeof(float) * n), *g = malloc(sizeof(float) * n), *out = malloc(sizeof(float) * n);
    double *rp = malloc(sizeof(double) * n), *m = calloc(n, sizeof(double)), *v = calloc(n, sizeof(double));
    for (int i = 0; i < n; ++i) { p[i] = frand(&seed); g[i] = frand(&seed); rp[i] = p[i]; }
    float *dp = to_dev(p, sizeof(float) * n), *dg = to_dev(g, sizeof(float) * n), *dm = to_dev(NULL, sizeof(float) * n), *dvv = to_dev(NULL, sizeof(float) * n);
    CHECK_HIP(hipMemsetAsync(dm, 0, sizeof(float) * n, stream));
    CHECK_HIP(hipMemsetAsync(dvv, 0, sizeof(float) * n, stream));
    const double lr = 1e-2, b1 = 0.9, b2 = 0.999, eps = 1e-8;
    for (int step = 1; step <= 2; ++step) {
      CHECK_UIC(uic_adam_step(dp, dg, dm, dvv, n, (float)lr, (float)b1, (float)b2, (float)eps, step, 1.0f, stream));
      for (int i = 0; i < n; ++i) {
        m[i] = b1 * m[i] + (1 - b1) * g[i];
        v[i] = b2 * v[i] + (1 - b2) * (double)g[i] * g[i];
        const double mh = m[i] / (1 - pow(b1, step)), vh = v[i] / (1 - pow(b2, step));
        rp[i] -= lr * mh / (sqrt(vh) + eps);
      }
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(out, dp, sizeof(float) * n, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < n; ++i) { double e = fabs(out[i] - rp[i]); if (e > worst) worst = e; }
    if (worst > 1e-5) { fprintf(stderr, "uic_adam_step off by %g\n", worst); return 1; }
  }
  /* the fused training step of the captioner from C: with all-zero weights every step predicts the uniform distribution,
   * so loss = ln(V1) and d loss / d logit.bias[v] = sum_rows mask * (1/V1 - [label == v]) / sum(mask) */
  {
    uic_topdown_dims dd;
    memset(&dd, 0, sizeof(dd));
    dd.N = 6; dd.R = 5; dd.D = 64; dd.Dfc = 64; dd.H = 32; dd.E = 32; dd.A = 32; dd.V1 = 51; dd.T = 7;
    dd.dtype = UIC_DTYPE_F32; dd.drop_p = 0.f;
    const int N = dd.N, R = dd.R, D = dd.D, H = dd.H, E = dd.E, A = dd.A, V1 = dd.V1, T = dd.T, ld = T + 1;
    const size_t sizes[21] = {(size_t)V1 * E, (size_t)H * D, H, (size_t)H * D, H, (size_t)V1 * H, V1, (size_t)A * H, A,
                              (size_t)4 * H * (E + 2 * H), (size_t)4 * H * H, 4 * H, 4 * H, (size_t)4 * H * 2 * H, (size_t)4 * H * H, 4 * H, 4 * H,
                              (size_t)A * H, A, A, 1};
    uic_topdown_weights W, G;
    memset(&W, 0, sizeof(W));
    memset(&G, 0, sizeof(G));
    float** wp = (float**)&W;      /* the first 21 members are the float* tensors in state_dict order */
    float** gp = (float**)&G;
    for (int i = 0; i < 21; ++i) {
      wp[i] = to_dev(NULL, sizeof(float) * sizes[i]);
      gp[i] = to_dev(NULL, sizeof(float) * sizes[i]);
      if (!wp[i] || !gp[i]) return 2;
      CHECK_HIP(hipMemsetAsync(wp[i], 0, sizeof(float) * sizes[i], stream));
    }
    float *fc = malloc(sizeof(float) * N * D), *att = malloc(sizeof(float) * N * R * D), *masks = calloc((size_t)N * ld, sizeof(float));
    long long* labels = calloc((size_t)N * ld, sizeof(long long));
    for (int i = 0; i < N * D; ++i) fc[i] = frand(&seed);
    for (int i = 0; i < N * R * D; ++i) att[i] = frand(&seed);
    double den = 0;
    for (int n = 0; n < N; ++n) {
      const int len = 2 + n % 4;                       /* caption length; the mask covers len + 2 positions (dataloader.py:283-286) */
      for (int t = 1; t <= len; ++t) labels[n * ld + t] = 1 + (n * 7 + t * 3) % (V1 - 1);
      for (int t = 0; t < len + 2 && t < ld; ++t) masks[n * ld + t] = 1.f;
      for (int t = 1; t <= T; ++t) den += masks[n * ld + t];
    }
    uic_topdown_batch b;
    memset(&b, 0, sizeof(b));
    b.fc_feats = to_dev(fc, sizeof(float) * N * D);
    b.att_feats = to_dev(att, sizeof(float) * N * R * D);
    b.labels = to_dev(labels, sizeof(long long) * N * ld); b.ld_labels = ld;
    b.masks = to_dev(masks, sizeof(float) * N * ld); b.ld_masks = ld;
    void* wsp = to_dev(NULL, uic_topdown_workspace_bytes(&dd));
    void* drv = to_dev(NULL, uic_topdown_derived_bytes(&dd));
    float* out = to_dev(NULL, 2 * sizeof(float));
    if (!b.fc_feats || !b.att_feats || !b.labels || !b.masks || !wsp || !drv || !out) return 2;
    CHECK_UIC(uic_topdown_refresh_weights(&dd, &W, drv, stream));
    CHECK_UIC(uic_topdown_xe_train_step(&dd, &W, drv, &b, T, 1, 123u, wsp, NULL, out, out + 1, &G, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    float h_out[2], *gb = malloc(sizeof(float) * V1);
    CHECK_HIP(hipMemcpy(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(gb, G.logit_b, sizeof(float) * V1, hipMemcpyDeviceToHost));
    if (fabs(h_out[0] - log((double)V1)) > 1e-5 || fabs(h_out[1] - den) > 1e-6) {
      fprintf(stderr, "xe_train_step: loss %g (want %g), sum(mask) %g (want %g)\n", h_out[0], log((double)V1), h_out[1], den);
      return 1;
    }
    double worst = 0;
    for (int v = 0; v < V1; ++v) {
      double want = 0;
      for (int n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t) want += masks[n * ld + t + 1] * (1.0 / V1 - (labels[n * ld + t + 1] == v ? 1.0 : 0.0));
      want /= den;
      const double e = fabs(want - gb[v]);
      if (e > worst) worst = e;
    }
    if (worst > 1e-6) { fprintf(stderr, "xe_train_step: d logit.bias off by %g\n", worst); return 1; }
  }
  printf("C ABI OK\n");
  return 0;
}
