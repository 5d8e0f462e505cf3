// Persistent decode loop of the pivot NMT decoder for gfx950 (MI355X): ONE launch runs all target steps of
// NMT_Models.Decoder.forward's loop body (P/models/NMT_Models.py:228-262) -- StackedLSTM (O/modules/StackedRNN.py:20-34: the
// layers' LSTM cells with dropout between them), dot GlobalAttention (O/modules/GlobalAttention.py:84-177: scores against
// the hoisted context x W_in, softmax, weighted context), linear_out + tanh + dropout = the next step's input feed -- instead of
// layers + 2 dependent launches per step (124 of the 827 launches of a configs[2] training step).
//
// Same decomposition and exchange protocol as the captioner's generic persistent kernel (rnn_persist.hip, header comment;
// rnn_persist_common.h): inside a row group workgroup `rank` owns 16 hidden units = 64 gate columns of every layer and 16
// columns of linear_out for all rows of the group, and rows rank, rank + 32, ... in the attention phase; phases end in a
// bounded group barrier; exchanged vectors (h of every layer, its dropped copy, the attention context, the output) are read
// with sc1 loads and -- in the placement-independent SAFE mode -- stored write-through.
// Two kernels.  nmt_dec_ws_kernel (batch <= 128, 1 or 2 layers -- configs[2]): WEIGHT-STATIONARY.  8 row groups of <= 16 rows,
// one per XCD; a workgroup's 288 KB of weight slices are loaded ONCE per launch: layer 0's 64 gate columns x K 1024 as an MFMA
// B-fragment image in LDS (128 KB, [wave][k-step][gate][lane][16 B]), layer 1's (128 KB) and linear_out's 16 columns (32 KB) in
// the registers of the 8 waves (K split over the waves: wave w holds k-steps w, w + 8, w + 16, w + 24 -- 80 registers per lane);
// per phase only the group's <= 16 activation rows move (16 KB from the XCD's L2), the 8 partial tiles are summed through the
// remaining 32 KB of LDS.  Why: re-reading the weights every step is what bounds the generic form -- with 8 groups every XCD
// pulls all 9 MB through its 4 MB L2 every step (72 MB per step from the Infinity Cache: measured 36 us per step, no better
// than the launch chain's 40), and ONE group of 32 workgroups spread over the XCDs (weights then L2-resident, exchange through
// the SAFE protocol) measured 58 us per step.
// nmt_dec_persist_kernel (batch <= 640, any layer count): the generic form, weights re-read every step, 8 groups of <= 80 rows.
// bf16 operands, rnn_size 512, source length <= 64; anything else keeps the per-step launches.
#include "rnn_persist_common.h"

namespace {

#define NMT_SITE_DEC(l, t) (2000u + (unsigned)(l) * 256u + (unsigned)(t))     // (as csrc/nmt.hip)
#define NMT_SITE_OUT(t) (4000u + (unsigned)(t))

constexpr int NMT_MAXR = 8;           // source positions per wave: S <= NWAVE * NMT_MAXR = 64

// out = dropout(tanh(linear_out([c ; rnn_output]))) (GlobalAttention.py:165-167, NMT_Models.py:258-259): 16 columns of the group's rows
template <typename T, bool SAFE>
__device__ __forceinline__ void linout_phase(Ctx& c, const T* cvec, const T* q, const T* w, T* out_pre, T* out, float drop_p,
                                             unsigned seed, unsigned site) {
  const bool owner = c.wave < c.MT;
  const int a = c.u0 + c.l15;
  f32x4 acc[MT_MAX][1];
  zero_acc<1>(acc);
  const void* const As[2] = {cvec, q};
  const void* const Bs[2] = {w, w + HH};
  const int ldb[2] = {2 * HH, 2 * HH};
  gemm_ksplit<T, 2, 1, false>(c, acc, As, Bs, ldb);
  f32x4* red = (f32x4*)c.smem;      // [wave][tile][lane]
#pragma unroll
  for (int i = 0; i < MT_MAX; ++i)
    if (i < c.MT) red[(c.wave * MT_MAX + i) * 64 + c.lane] = acc[i][0];
  __syncthreads();
  if (owner) {
    f32x4 s = red[(0 * MT_MAX + c.wave) * 64 + c.lane];
#pragma unroll
    for (int w2 = 1; w2 < NWAVE; ++w2) s += red[(w2 * MT_MAX + c.wave) * 64 + c.lane];
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * c.wave + 4 * c.lq + r;
      if (rr < c.nrow) {
        const unsigned o = (unsigned)((c.rbegin + rr) * HH + a);
        const float v = uic_tanh<T>(s[r]);
        out_pre[o] = uic_from_f<T>(v);                        // read again only in the backward pass
        st_x<SAFE>(out + o, drop_p > 0.f ? v * uic_drop_scale(seed, site, o, drop_p, inv_keep) : v);
      }
    }
  }
  __syncthreads();
}

// dot attention of the rows this workgroup takes in the attention phase: scores = ctxw[s, b, :] . q, softmax over the source
// positions (no padding mask in the training forward, as in the reference), c = sum_s a[s] ctx[s, b, :].  Every global load of
// a row is issued at its top (csrc/nmt.hip's gattn_fwd_fast_kernel, here with 8 waves and the query read past the L1).
template <bool SAFE>
__device__ __forceinline__ void nmt_attn_phase(const Ctx& c, const UicNmtDecParams& p, const bf16_t* q_all, float* attn_t, bf16_t* cvec_t) {
  const int S = p.S, B = p.B;
  float* s_t = (float*)c.smem + 64;        // [HH]   (the first words stay free: the barrier's flag lives there)
  float* s_a = s_t + HH;                   // [64]
  float* s_red = s_a + 64;                 // [NWAVE][HH]
  const bf16_t* ctx = (const bf16_t*)p.ctx;
  const float* ctxw = p.ctxw;
  for (int rr = c.rank; rr < c.nrow; rr += PW) {
    const int b = c.rbegin + rr;
    uint4 cr[NMT_MAXR];
    float4 wr[NMT_MAXR][2];
#pragma unroll
    for (int u = 0; u < NMT_MAXR; ++u) {
      const int sp = c.wave + NWAVE * u;
      const size_t r = ((size_t)(sp < S ? sp : S - 1) * B + b) * HH;
      cr[u] = *(const uint4*)(ctx + r + c.lane * 8);
      wr[u][0] = *(const float4*)(ctxw + r + c.lane * 4);
      wr[u][1] = *(const float4*)(ctxw + r + (c.lane + 64) * 4);
    }
    if (c.wave == 0) {                     // the query row: written by other workgroups of the group in this launch
      const u32x4 qv = bload<true>(rsrc_of(q_all), (unsigned)((b * HH + c.lane * 8) * 2), 0);
      float f[8];
      uic_unpack<bf16_t>(__builtin_bit_cast(uint4, qv), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) s_t[c.lane * 8 + k] = f[k];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NMT_MAXR; ++u) {
      const int sp = c.wave + NWAVE * u;
      if (sp < S) {
        const float* v0 = s_t + c.lane * 4; const float* v1 = s_t + (c.lane + 64) * 4;
        float pr = 0.f;
        pr += wr[u][0].x * v0[0]; pr += wr[u][0].y * v0[1]; pr += wr[u][0].z * v0[2]; pr += wr[u][0].w * v0[3];
        pr += wr[u][1].x * v1[0]; pr += wr[u][1].y * v1[1]; pr += wr[u][1].z * v1[2]; pr += wr[u][1].w * v1[3];
        pr = uic_wave_sum(pr);
        if (c.lane == 0) s_a[sp] = pr;
      }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int sp = 0; sp < S; ++sp) mx = fmaxf(mx, s_a[sp]);
    float sum = 0.f;
    for (int sp = 0; sp < S; ++sp) sum += expf(s_a[sp] - mx);
    const float inv = 1.f / sum;
    __syncthreads();
    if (c.tid < S) {
      const float a = expf(s_a[c.tid] - mx) * inv;
      s_a[c.tid] = a;
      attn_t[(size_t)b * S + c.tid] = a;
    }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int u = 0; u < NMT_MAXR; ++u) {
      const int sp = c.wave + NWAVE * u;
      if (sp < S) {
        float f[8];
        uic_unpack<bf16_t>(cr[u], f);
        const float a = s_a[sp];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += a * f[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_red[c.wave * HH + c.lane * 8 + k] = acc[k];
    __syncthreads();
    {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) v += s_red[w * HH + c.tid];      // (NTH == HH: one column per thread)
      st_x<SAFE>(cvec_t + (size_t)b * HH + c.tid, v);
    }
    __syncthreads();
  }
}

template <bool SAFE>
__device__ __forceinline__ void nmt_dec_steps(const UicNmtDecParams& p, Ctx& c) {
  typedef bf16_t T;
  const int B = p.B;
  const size_t BH = (size_t)B * HH;
  const size_t rb = (size_t)c.rbegin * HH;
  for (int t = 0; t < p.Td; ++t) {
    // (see rnn_persist.hip's run_steps: keeps hipcc from hoisting every phase's per-lane addresses out of the loop and spilling them)
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    asm volatile("" : "+s"(c.wave), "+s"(c.u0), "+s"(c.rbegin), "+s"(c.nrow), "+s"(c.MT));
    const T* x = nullptr;
    for (int l = 0; l < p.NL; ++l) {                          // StackedLSTM.forward
      T* h_prev = (T*)p.hd[l] + (size_t)t * BH;
      T* h_new = h_prev + BH;
      const bool inter = l + 1 < p.NL;                        // dropout between layers only
      T* hdrop = inter ? (T*)p.hdrop[l] + (size_t)t * BH : nullptr;
      T* gates = (T*)p.gates_d[l] + (size_t)t * B * 4 * HH;
      const float* cprev = p.cd[l] + (size_t)t * BH;
      float* cnew = p.cd[l] + (size_t)(t + 1) * BH;
      if (l == 0) {   // [emb_t ; input feed] -> the embedding share (+ both biases) is p.gx_d0, the input feed is the previous output
        const void* const As[2] = {(const T*)p.out_all + (size_t)t * BH + rb, h_prev + rb};
        const void* const Bs[2] = {p.w_ih[0], p.w_hh[0]};
        const int ldb[2] = {p.ld_ih[0], HH};
        const float* gx = p.gx_d0 + (size_t)t * B * 4 * HH;
        auto pre = [&](unsigned idx4, unsigned) { return gx[idx4]; };
        lstm_phase<T, SAFE, 2>(c, As, Bs, ldb, pre, cprev, cnew, h_new, hdrop, gates, B, inter ? p.drop_p : 0.f, p.seed, NMT_SITE_DEC(0, t));
      } else {
        const void* const As[2] = {x + rb, h_prev + rb};
        const void* const Bs[2] = {p.w_ih[l], p.w_hh[l]};
        const int ldb[2] = {p.ld_ih[l], HH};
        const float* b1 = p.b_ih[l];
        const float* b2 = p.b_hh[l];
        auto pre = [&](unsigned, unsigned col) { return b1[col] + b2[col]; };
        lstm_phase<T, SAFE, 2>(c, As, Bs, ldb, pre, cprev, cnew, h_new, hdrop, gates, B, inter ? p.drop_p : 0.f, p.seed, NMT_SITE_DEC(l, t));
      }
      x = inter ? hdrop : h_new;
      if (!group_barrier(c)) return;
    }
    const T* q = (const T*)p.hd[p.NL - 1] + (size_t)(t + 1) * BH;        // rnn_output = the top layer's h
    T* cvec = (T*)p.cvec_all + (size_t)t * BH;
    nmt_attn_phase<SAFE>(c, p, q, p.attn_all + (size_t)t * B * p.S, cvec);
    if (!group_barrier(c)) return;
    linout_phase<T, SAFE>(c, cvec + rb, q + rb, (const T*)p.attn_out_w, (T*)p.out_pre + (size_t)t * BH, (T*)p.out_all + (size_t)(t + 1) * BH,
                          p.drop_p, p.seed, NMT_SITE_OUT(t));
    if (!group_barrier(c)) return;
  }
}

// ---------------------------------------------------------------------------------------------------
// weight-stationary form (header comment): one 16-row tile per group, NL in {1, 2}
constexpr int WS_W0_BYTES = NWAVE * 4 * 4 * 1024;     // layer 0: [wave][k-step j][gate][lane] x 16 B = 128 KB
constexpr int WS_SCR_BYTES = 32 * 1024;
constexpr int WS_LDS_BYTES = WS_W0_BYTES + WS_SCR_BYTES;

// fragment (16 B per lane) of weight row `row` (leading dimension ld), K offset kk elements: the MFMA B operand of 16 units
__device__ __forceinline__ u32x4 ws_wfrag(const void* w, int ld, int row, int kk, int lq) {
  return *(const u32x4*)((const bf16_t*)w + (size_t)row * ld + kk + lq * 8);
}

template <bool SAFE>
__device__ __forceinline__ void nmt_dec_ws_steps(const UicNmtDecParams& p, Ctx& c, char* lds) {
  typedef bf16_t T;
  const int B = p.B;
  const size_t BH = (size_t)B * HH;
  const size_t rb = (size_t)c.rbegin * HH;
  u32x4* w0 = (u32x4*)lds;                       // layer 0's image; c.smem (the scratch) lies behind it
  // ---- the weight slices, once per launch.  k-step ks = wave + 8 j of the 32 (K = 1024 = two 512-wide segments)
  u32x4 w1[4][4], wo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ks = c.wave + NWAVE * j;
    const int seg = ks >> 4, kk = (ks & 15) * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = g * HH + c.u0 + c.l15;
      w0[((c.wave * 4 + j) * 4 + g) * 64 + c.lane] = seg == 0 ? ws_wfrag(p.w_ih[0], p.ld_ih[0], row, kk, c.lq) : ws_wfrag(p.w_hh[0], HH, row, kk, c.lq);
      if (p.NL > 1) w1[j][g] = seg == 0 ? ws_wfrag(p.w_ih[1], p.ld_ih[1], row, kk, c.lq) : ws_wfrag(p.w_hh[1], HH, row, kk, c.lq);
      else w1[j][g] = u32x4{0u, 0u, 0u, 0u};
    }
    wo[j] = ws_wfrag(p.attn_out_w, 2 * HH, c.u0 + c.l15, seg * HH + kk, c.lq);
  }
  __syncthreads();
  const int arow = c.l15 < c.nrow ? c.l15 : c.nrow - 1;      // rows past the group's share re-read its last row (results unused)
  const bool owner = c.wave == 0;
  const unsigned u = (unsigned)(c.u0 + c.l15);
  f32x4* red = (f32x4*)c.smem;                   // [wave][gate][lane]

  // A fragments of this wave's four k-steps from two [nrow, 512] slabs of exchanged rows (sc1 loads), all issued at once
  auto load_a = [&](const T* a0, const T* a1, u32x4 (&af)[4]) {
    const __amdgpu_buffer_rsrc_t r0 = rsrc_of(a0), r1 = rsrc_of(a1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ks = c.wave + NWAVE * j;
      const unsigned voff = (unsigned)((arow * HH + (ks & 15) * 32 + c.lq * 8) * 2);
      af[j] = (ks >> 4) == 0 ? bload<true>(r0, voff, 0) : bload<true>(r1, voff, 0);
    }
  };
  // one LSTM cell of the group's rows and this workgroup's 16 units; wfrag(j, g): the resident B fragment
  auto lstm = [&](const T* a0, const T* a1, auto wfrag, auto pre, const float* c_prev, float* c_out, T* h_out, T* h_drop, T* gates_out,
                  float drop_p, unsigned site) {
    unsigned nn[4];
    float pv[4][4], cp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * c.lq + r;
      nn[r] = (unsigned)((c.rbegin + (rr < c.nrow ? rr : c.nrow - 1)) * HH);
    }
    if (owner) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cp[r] = c_prev[nn[r] + u];
#pragma unroll
        for (int g = 0; g < 4; ++g) pv[r][g] = pre(4u * nn[r] + (unsigned)(g * HH) + u, (unsigned)(g * HH) + u);
      }
    }
    u32x4 af[4];
    load_a(a0, a1, af);
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = Mma<T>::run(af[j], wfrag(j, g), acc[g]);
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(c.wave * 4 + g) * 64 + c.lane] = acc[g];
    __syncthreads();
    if (owner) {
      f32x4 s[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s[g] = red[(0 * 4 + g) * 64 + c.lane];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) s[g] += red[(w * 4 + g) * 64 + c.lane];
      }
      const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * c.lq + r;
        if (rr < c.nrow) {
          const unsigned o = nn[r] + u;
          const float gi = uic_sigmoid_t<T>(s[0][r] + pv[r][0]);
          const float gf = uic_sigmoid_t<T>(s[1][r] + pv[r][1]);
          const float gg = uic_tanh<T>(s[2][r] + pv[r][2]);
          const float go = uic_sigmoid_t<T>(s[3][r] + pv[r][3]);
          const float cn = gf * cp[r] + gi * gg;
          const float h = go * uic_tanh<T>(cn);
          c_out[o] = cn;
          st_x<SAFE>(h_out + o, h);
          if (h_drop) st_x<SAFE>(h_drop + o, drop_p > 0.f ? h * uic_drop_scale(p.seed, site, o, drop_p, inv_keep) : h);
          const unsigned og = 4u * nn[r] + u;                  // read again only in the backward pass
          __builtin_nontemporal_store(uic_from_f<T>(gi), gates_out + og);
          __builtin_nontemporal_store(uic_from_f<T>(gf), gates_out + og + HH);
          __builtin_nontemporal_store(uic_from_f<T>(gg), gates_out + og + 2 * HH);
          __builtin_nontemporal_store(uic_from_f<T>(go), gates_out + og + 3 * HH);
        }
      }
    }
    __syncthreads();
  };

  for (int t = 0; t < p.Td; ++t) {
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    const bool two = p.NL > 1;
    T* h0_prev = (T*)p.hd[0] + (size_t)t * BH;
    T* h0_new = h0_prev + BH;
    T* hdrop0 = two ? (T*)p.hdrop[0] + (size_t)t * BH : nullptr;
    {  // layer 0: [input feed ; h_0] against the LDS image; the embedding share (+ both biases) is p.gx_d0
      const float* gx = p.gx_d0 + (size_t)t * B * 4 * HH;
      lstm((const T*)p.out_all + (size_t)t * BH + rb, h0_prev + rb,
           [&](int j, int g) { return w0[((c.wave * 4 + j) * 4 + g) * 64 + c.lane]; },
           [&](unsigned idx4, unsigned) { return gx[idx4]; },
           p.cd[0] + (size_t)t * BH, p.cd[0] + (size_t)(t + 1) * BH, h0_new, hdrop0, (T*)p.gates_d[0] + (size_t)t * B * 4 * HH,
           two ? p.drop_p : 0.f, NMT_SITE_DEC(0, t));
    }
    if (!group_barrier(c)) return;
    const T* q = h0_new;
    if (two) {  // layer 1: [dropped h_0 ; h_1] against the register-resident slice
      T* h1_prev = (T*)p.hd[1] + (size_t)t * BH;
      const float* b1 = p.b_ih[1];
      const float* b2 = p.b_hh[1];
      lstm(hdrop0 + rb, h1_prev + rb, [&](int j, int g) { return w1[j][g]; }, [&](unsigned, unsigned col) { return b1[col] + b2[col]; },
           p.cd[1] + (size_t)t * BH, p.cd[1] + (size_t)(t + 1) * BH, h1_prev + BH, (T*)nullptr, (T*)p.gates_d[1] + (size_t)t * B * 4 * HH,
           0.f, 0u);
      q = h1_prev + BH;
      if (!group_barrier(c)) return;
    }
    T* cvec = (T*)p.cvec_all + (size_t)t * BH;
    nmt_attn_phase<SAFE>(c, p, q, p.attn_all + (size_t)t * B * p.S, cvec);
    if (!group_barrier(c)) return;
    {  // out = dropout(tanh(linear_out([c ; q])))
      u32x4 af[4];
      load_a(cvec + rb, q + rb, af);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = Mma<T>::run(af[j], wo[j], acc);
      red[c.wave * 64 + c.lane] = acc;
      __syncthreads();
      if (owner) {
        f32x4 sres = red[c.lane];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) sres += red[w * 64 + c.lane];
        const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
        T* out_pre = (T*)p.out_pre + (size_t)t * BH;
        T* out = (T*)p.out_all + (size_t)(t + 1) * BH;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 4 * c.lq + r;
          if (rr < c.nrow) {
            const unsigned o = (unsigned)((c.rbegin + rr) * HH) + u;
            const float v = uic_tanh<T>(sres[r]);
            out_pre[o] = uic_from_f<T>(v);
            st_x<SAFE>(out + o, p.drop_p > 0.f ? v * uic_drop_scale(p.seed, NMT_SITE_OUT(t), o, p.drop_p, inv_keep) : v);
          }
        }
      }
      __syncthreads();
    }
    if (!group_barrier(c)) return;
  }
}

__global__ __launch_bounds__(NTH) void nmt_dec_ws_kernel(const UicNmtDecParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem + WS_W0_BYTES, c);
  if (mode == 0) return;
  if (mode == 2) nmt_dec_ws_steps<true>(p, c, smem);
  else nmt_dec_ws_steps<false>(p, c, smem);
}

__global__ __launch_bounds__(NTH) void nmt_dec_persist_kernel(const UicNmtDecParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem, c);
  if (mode == 0) return;
  if (mode == 2) nmt_dec_steps<true>(p, c);
  else nmt_dec_steps<false>(p, c);
}

}  // namespace

bool uic_nmt_dec_persist_eligible(int dtype, int B, int S, int H, int NL) {
  if (dtype != UIC_BF16 || H != HH || S < 1 || S > NWAVE * NMT_MAXR || B < 1 || B > 8 * 16 * MT_MAX || NL < 1 || NL > UIC_NMT_MAX_LAYERS) return false;
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    cus = prop.multiProcessorCount;
  }
  return cus == 8 * PW;        // one workgroup per CU, 32 per XCD
}

int uic_nmt_dec_persist_launch(const UicNmtDecParams& p, hipStream_t s) {
  UIC_REQUIRE(p.sync && p.Td > 0 && p.B > 0 && p.Nrows == p.B && p.row0 == 0, "nmt_dec_persist: bad arguments");
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)nmt_dec_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "hipFuncSetAttribute(nmt dec persist)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)nmt_dec_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES), "hipFuncSetAttribute(nmt dec ws)"));
    configured = true;
  }
  UicPersistGateScope gate;       // (its workgroups have to be resident together: never beside another persistent launch)
  UIC_TRY(gate.enter(s));
  UIC_TRY(uic_check_hip(hipMemsetAsync(p.sync, 0, (size_t)SY_WORDS * 4, s), "hipMemsetAsync(nmt dec sync)"));
  if (p.B <= 8 * 16 && p.NL <= 2) hipLaunchKernelGGL(nmt_dec_ws_kernel, dim3(8 * PW), dim3(NTH), WS_LDS_BYTES, s, p);     // one 16-row tile per group
  else hipLaunchKernelGGL(nmt_dec_persist_kernel, dim3(8 * PW), dim3(NTH), LDS_BYTES, s, p);
  UIC_LAUNCH_CHECK("nmt_dec_persist_kernel");
  return gate.leave();
}
