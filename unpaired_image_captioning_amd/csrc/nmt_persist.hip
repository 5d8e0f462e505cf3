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
  u32x4 w1[4][4];
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
  }
  __syncthreads();
  const int arow = c.l15 < c.nrow ? c.l15 : c.nrow - 1;      // rows past the group's share re-read its last row (results unused)
  const unsigned u = (unsigned)(c.u0 + c.l15);
  f32x4* red = (f32x4*)c.smem;                   // [wave][gate][lane]
  // Element-wise work is spread over waves 0-3: of the [16 rows x 16 units] tile a lane holds rows 4 lq + r in an accumulator,
  // wave w < 4 finishes r = w -- row 4 lq + w, unit u -- and keeps that element's cell states in registers from step to step.
  const bool fin = c.wave < 4;
  const int rw = 4 * c.lq + (c.wave & 3);
  const bool valid = fin && rw < c.nrow;
  const unsigned nnw = (unsigned)((c.rbegin + (rw < c.nrow ? rw : c.nrow - 1)) * HH);
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  float cs0 = p.cd[0][nnw + u], cs1 = p.NL > 1 ? p.cd[1][nnw + u] : 0.f;
  // layer 0's share of the gate pre-activations that does not depend on the recurrence (embedding + both biases, made by a batched
  // GEMM): requested one step ahead -- it comes from HBM, and requested where it is used its latency is on the step's critical path
  float gxn[4];
  auto load_gx = [&](int t) {
    const float* gx = p.gx_d0 + (size_t)t * B * 4 * HH;
#pragma unroll
    for (int g = 0; g < 4; ++g) gxn[g] = gx[4u * nnw + (unsigned)(g * HH) + u];
  };
  load_gx(0);

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
  // one LSTM cell of the group's rows and this workgroup's 16 units; wfrag(j, g): the resident B fragment; pv: this element's
  // recurrence-independent share; cst: its cell state (in / out)
  auto lstm = [&](const T* a0, const T* a1, auto wfrag, const float (&pv)[4], float& cst, float* c_out, T* h_out, T* h_drop, T* gates_out,
                  float drop_p, unsigned site) {
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
    if (fin) {
      float sg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = ((const float*)(red + (0 * 4 + g) * 64 + c.lane))[c.wave];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) v += ((const float*)(red + (w * 4 + g) * 64 + c.lane))[c.wave];
        sg[g] = v;
      }
      if (valid) {
        const unsigned o = nnw + u;
        const float gi = uic_sigmoid_t<T>(sg[0] + pv[0]);
        const float gf = uic_sigmoid_t<T>(sg[1] + pv[1]);
        const float gg = uic_tanh<T>(sg[2] + pv[2]);
        const float go = uic_sigmoid_t<T>(sg[3] + pv[3]);
        const float cn = gf * cst + gi * gg;
        const float h = go * uic_tanh<T>(cn);
        cst = cn;
        c_out[o] = cn;
        st_x<SAFE>(h_out + o, h);
        if (h_drop) st_x<SAFE>(h_drop + o, drop_p > 0.f ? h * uic_drop_scale(p.seed, site, o, drop_p, inv_keep) : h);
        const unsigned og = 4u * nnw + u;                  // read again only in the backward pass
        __builtin_nontemporal_store(uic_from_f<T>(gi), gates_out + og);
        __builtin_nontemporal_store(uic_from_f<T>(gf), gates_out + og + HH);
        __builtin_nontemporal_store(uic_from_f<T>(gg), gates_out + og + 2 * HH);
        __builtin_nontemporal_store(uic_from_f<T>(go), gates_out + og + 3 * HH);
      }
    }
    __syncthreads();
  };
  // The attention phase's operands of the row this workgroup takes (at most one: <= 16 rows per group, 32 workgroups): the
  // encoder's contexts and their linear_in image.  Requested between the two halves of the group barrier in front of the phase
  // -- they do not depend on the exchange, so their latency passes while the workgroup waits for the others.  (Kept in registers
  // for the whole launch they are 96 registers: with 8 waves x 256 the LSTM phases spill.)
  const bool att_wg = c.rank < c.nrow;
  uint4 cr[NMT_MAXR];
  float4 wr[NMT_MAXR][2];
  auto att_load = [&]() {
    const int b = c.rbegin + c.rank;
#pragma unroll
    for (int uu = 0; uu < NMT_MAXR; ++uu) {
      const int sp = c.wave + NWAVE * uu;
      const size_t r = ((size_t)(sp < p.S ? sp : p.S - 1) * B + b) * HH;
      cr[uu] = *(const uint4*)((const T*)p.ctx + r + c.lane * 8);
      wr[uu][0] = *(const float4*)(p.ctxw + r + c.lane * 4);
      wr[uu][1] = *(const float4*)(p.ctxw + r + (c.lane + 64) * 4);
    }
  };
  // dot attention of that row (GlobalAttention.py:120-160; csrc/nmt.hip gattn_fwd_fast_kernel): every wave reads the query in the
  // layout of its own products, takes source positions wave + 8 u, and redoes the <= 64-way softmax in its lanes (no LDS loops)
  auto attention = [&](const T* q_all, float* attn_t, T* cvec_t) {
    const int S = p.S;
    const int b = c.rbegin + c.rank;
    float* s_a = (float*)c.smem + 64;        // [64]   (the first words stay free: the barrier's flag lives there)
    float* s_red = s_a + 64;                 // [NWAVE][HH]
    const __amdgpu_buffer_rsrc_t rq = rsrc_of(q_all);
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 qa = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rq, (unsigned)((b * HH + c.lane * 4) * 2), 0, 16));
    const u32x2 qb = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rq, (unsigned)((b * HH + (c.lane + 64) * 4) * 2), 0, 16));
    const float v0[4] = {__uint_as_float(qa.x << 16), __uint_as_float(qa.x & 0xffff0000u), __uint_as_float(qa.y << 16), __uint_as_float(qa.y & 0xffff0000u)};
    const float v1[4] = {__uint_as_float(qb.x << 16), __uint_as_float(qb.x & 0xffff0000u), __uint_as_float(qb.y << 16), __uint_as_float(qb.y & 0xffff0000u)};
#pragma unroll
    for (int uu = 0; uu < NMT_MAXR; ++uu) {
      const int sp = c.wave + NWAVE * uu;
      if (sp < S) {
        float pr = 0.f;
        pr += wr[uu][0].x * v0[0]; pr += wr[uu][0].y * v0[1]; pr += wr[uu][0].z * v0[2]; pr += wr[uu][0].w * v0[3];
        pr += wr[uu][1].x * v1[0]; pr += wr[uu][1].y * v1[1]; pr += wr[uu][1].z * v1[2]; pr += wr[uu][1].w * v1[3];
        pr = uic_wave_sum(pr);
        if (c.lane == 0) s_a[sp] = pr;
      }
    }
    __syncthreads();
    const float x = c.lane < S ? s_a[c.lane] : -INFINITY;
    const float mx = uic_wave_max(x);
    const float e = c.lane < S ? expf(x - mx) : 0.f;
    const float a_l = e * (1.f / uic_wave_sum(e));
    if (c.wave == 0 && c.lane < S) attn_t[(size_t)b * S + c.lane] = a_l;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int uu = 0; uu < NMT_MAXR; ++uu) {
      const int sp = c.wave + NWAVE * uu;
      const float a = __shfl(a_l, sp & 63, 64);
      if (sp < S) {
        float f[8];
        uic_unpack<bf16_t>(cr[uu], f);
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
  };

#define FW_STAMP(i) do { if (dbg && c.tid == 0) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
  for (int t = 0; t < p.Td; ++t) {
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * p.Td + t) * 16 : nullptr;
    FW_STAMP(0);
    const bool two = p.NL > 1;
    T* h0_prev = (T*)p.hd[0] + (size_t)t * BH;
    T* h0_new = h0_prev + BH;
    T* hdrop0 = two ? (T*)p.hdrop[0] + (size_t)t * BH : nullptr;
    {  // layer 0: [input feed ; h_0] against the LDS image; the embedding share (+ both biases) is p.gx_d0
      const float pv[4] = {gxn[0], gxn[1], gxn[2], gxn[3]};
      lstm((const T*)p.out_all + (size_t)t * BH + rb, h0_prev + rb,
           [&](int j, int g) { return w0[((c.wave * 4 + j) * 4 + g) * 64 + c.lane]; }, pv, cs0,
           p.cd[0] + (size_t)(t + 1) * BH, h0_new, hdrop0, (T*)p.gates_d[0] + (size_t)t * B * 4 * HH,
           two ? p.drop_p : 0.f, NMT_SITE_DEC(0, t));
    }
    FW_STAMP(1);
    group_arrive(c);
    // (the next step's input-GEMM rows -- f32 from HBM -- are requested BEHIND the arrival: in front of it the arrival's
    // s_waitcnt vmcnt(0), which is there for this phase's stores, waited for them too, every step)
    load_gx(t + 1 < p.Td ? t + 1 : t);           // (unconditional: a load behind a run-time branch makes hipcc wait for it at the branch's join)
    if (!two && att_wg) att_load();
    if (!group_wait(c, (int*)c.smem)) return;
    FW_STAMP(2);
    const T* q = h0_new;
    if (two) {  // layer 1: [dropped h_0 ; h_1] against the register-resident slice
      T* h1_prev = (T*)p.hd[1] + (size_t)t * BH;
      float pb1[4];                                // both biases of this element's gate columns (L2 hits, requested before the A fragments)
#pragma unroll
      for (int g = 0; g < 4; ++g) pb1[g] = p.b_ih[1][g * HH + u] + p.b_hh[1][g * HH + u];
      lstm(hdrop0 + rb, h1_prev + rb, [&](int j, int g) { return w1[j][g]; }, pb1, cs1,
           p.cd[1] + (size_t)(t + 1) * BH, h1_prev + BH, (T*)nullptr, (T*)p.gates_d[1] + (size_t)t * B * 4 * HH, 0.f, 0u);
      q = h1_prev + BH;
      FW_STAMP(3);
      group_arrive(c);
      if (att_wg) att_load();
      if (!group_wait(c, (int*)c.smem)) return;
      FW_STAMP(4);
    }
    T* cvec = (T*)p.cvec_all + (size_t)t * BH;
    if (att_wg) attention(q, p.attn_all + (size_t)t * B * p.S, cvec);
    FW_STAMP(5);
    group_arrive(c);
    // linear_out's slice (4 KB per workgroup, L2-resident) is re-read every step while the workgroup waits for the others: as
    // 16 resident registers it made the LSTM phases spill
    u32x4 wo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ks = c.wave + NWAVE * j;
      wo[j] = ws_wfrag(p.attn_out_w, 2 * HH, c.u0 + c.l15, (ks >> 4) * HH + (ks & 15) * 32, c.lq);
    }
    if (!group_wait(c, (int*)c.smem)) return;
    FW_STAMP(6);
    {  // out = dropout(tanh(linear_out([c ; q])))
      u32x4 af[4];
      load_a(cvec + rb, q + rb, af);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = Mma<T>::run(af[j], wo[j], acc);
      red[c.wave * 64 + c.lane] = acc;
      __syncthreads();
      if (fin) {
        float sres = ((const float*)(red + c.lane))[c.wave];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) sres += ((const float*)(red + w * 64 + c.lane))[c.wave];
        if (valid) {
          const unsigned o = nnw + u;
          const float v = uic_tanh<T>(sres);
          ((T*)p.out_pre)[(size_t)t * BH + o] = uic_from_f<T>(v);
          st_x<SAFE>((T*)p.out_all + (size_t)(t + 1) * BH + o, p.drop_p > 0.f ? v * uic_drop_scale(p.seed, NMT_SITE_OUT(t), o, p.drop_p, inv_keep) : v);
        }
      }
      __syncthreads();
    }
    FW_STAMP(7);
    if (!group_barrier(c)) return;
    FW_STAMP(8);
  }
#undef FW_STAMP
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

// ---------------------------------------------------------------------------------------------------
// BPTT of that loop as ONE launch (weight-stationary, batch <= 128, 2 layers): per target step, latest first,
//   A  d_pre = (d out_t + d feed) * dropout mask * (1 - out_pre^2)  [computed by every workgroup on the fly as the A operand],
//      d [c ; q] = d_pre W_out: this workgroup's 16 columns of d c (exchanged) and of d q (kept in registers)
//   B  attention backward of the row this workgroup takes: d score, d q += sum_s d score[s] ctxw[s] (exchanged)
//   C  top layer: cell backward of the own 16 units (exchanged: d gates), then d [x_1 ; h_1(t-1)] = d gates W_1 -- own columns of
//      both halves stay in registers (d x_1 = the gradient of layer 0's dropped h, d h_1(t-1) = the next step's carry)
//   D  layer 0: cell backward, d gates exchanged, d [feed ; h_0(t-1)] = d gates W_0: d feed exchanged (everybody's next step A)
// -- tanh_drop_bwd, the d[c;q] GEMM, gattn_bwd_step, 2 x (lstm_bwd + GEMM) of csrc/nmt.hip's chain: 7 launches per step.
// Five group barriers per step.  Resident: W_0^T's 32 rows x K 2048 (this workgroup's columns of both halves) as an LDS image
// (128 KB), W_1^T's (128 KB) and W_out^T's (32 KB) in registers.  Exchange slabs are indexed by step (never rewritten inside a
// launch: a reader's L2 cannot hold a stale line of them).
// Four waves with the whole 512-entry register file of their SIMD each (as rnn_persist.hip's weight-stationary kernel): 160
// resident weight registers per lane leave room for the phases' operands -- 8 waves x 256 registers spill.
constexpr int BW_NW = 4;
constexpr int BW_NTH = BW_NW * 64;
constexpr int BW_W0_BYTES = BW_NW * 16 * 2 * 1024;     // [wave][k-step j < 16][half][lane] x 16 B = 128 KB

template <bool SAFE, int MAXR>
__device__ __forceinline__ void nmt_dec_bwd_steps(const UicNmtDecBwdParams& p, Ctx& c, char* lds) {
  typedef bf16_t T;
  const int B = p.B, S = p.S;
  const size_t BH = (size_t)B * HH;
  const size_t rb = (size_t)c.rbegin * HH;
  u32x4* w0t = (u32x4*)lds;
  // ---- resident weight slices.  Rows n = half * 512 + u0 + l15 of the transposed weights, K contiguous.
  u32x4 w1t[16][2], wot[4][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kk = (c.wave + BW_NW * j) * 32;                 // K = 4 x 512: 64 k-steps, 16 per wave
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = h * HH + c.u0 + c.l15;
      w0t[((c.wave * 16 + j) * 2 + h) * 64 + c.lane] = ws_wfrag(p.w0T, 4 * HH, row, kk, c.lq);
      w1t[j][h] = ws_wfrag(p.w1T, 4 * HH, row, kk, c.lq);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)                                  // K = 512: 16 k-steps, 4 per wave
#pragma unroll
    for (int h = 0; h < 2; ++h) wot[j][h] = ws_wfrag(p.woutT, HH, h * HH + c.u0 + c.l15, (c.wave + BW_NW * j) * 32, c.lq);
  __syncthreads();
  const int arow = c.l15 < c.nrow ? c.l15 : c.nrow - 1;
  const unsigned u = (unsigned)(c.u0 + c.l15);
  f32x4* red = (f32x4*)c.smem;                                 // [wave][half][lane]
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  // Everything element-wise is spread over the four waves: of the [16 rows x 16 units] tile a lane holds rows 4 lq + r in an
  // accumulator, wave w finishes r = w -- row 4 lq + w, unit u -- and carries that element's state from step to step.
  const int rw = 4 * c.lq + c.wave;
  const bool valid = rw < c.nrow;
  const unsigned nnw = (unsigned)((c.rbegin + (valid ? rw : c.nrow - 1)) * HH);
  float dh0_rec = 0.f, dh1_rec = 0.f, dc0 = 0.f, dc1 = 0.f;

  // sum of the waves' partial [16 x 16] tiles of both halves (fixed order); wave w receives component w
  auto reduce2 = [&](const f32x4 (&acc)[2], float (&out)[2]) {
    red[(c.wave * 2 + 0) * 64 + c.lane] = acc[0];
    red[(c.wave * 2 + 1) * 64 + c.lane] = acc[1];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v = ((const float*)(red + h * 64 + c.lane))[c.wave];
#pragma unroll
      for (int w = 1; w < BW_NW; ++w) v += ((const float*)(red + (w * 2 + h) * 64 + c.lane))[c.wave];
      out[h] = v;
    }
    __syncthreads();
  };
  // d gates [nrow, 4 x 512] (exchanged) x the resident slice -> this workgroup's columns of both halves
  auto dgemm = [&](const T* dg, auto wfrag, float (&out)[2], auto after_loads) {
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(dg);
    u32x4 af[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) af[j] = bload<true>(ra, (unsigned)((arow * 4 * HH + (c.wave + BW_NW * j) * 32 + c.lq * 8) * 2), 0);
    after_loads();                                  // (requests that may return after the fragments: loads come back in issue order)
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[h] = Mma<T>::run(af[j], wfrag(j, h), acc[h]);
    reduce2(acc, out);
  };
  // Backward of one LSTM cell's gate math for this wave's element (csrc/pointwise.hip lstm_bwd_kernel): d gates stored (exchanged,
  // and read again by the weight-gradient GEMMs), dc carried.  Its operands -- the forward pass's activated gates and cell states --
  // do not depend on anything this launch computes: cell_load requests them at the top of the step, two group barriers before
  // they are used (requested where they are used, each of them is a memory latency on the step's critical path).
  struct CellIn { float gi, gf, gg, go, cc, cp; };
  auto cell_load = [&](int l, int t, CellIn& q) {
    const T* G = (const T*)p.gates_d[l] + (size_t)t * B * 4 * HH;
    const unsigned og = 4u * nnw + u;
    q.gi = uic_to_f(G[og]); q.gf = uic_to_f(G[og + HH]); q.gg = uic_to_f(G[og + 2 * HH]); q.go = uic_to_f(G[og + 3 * HH]);
    q.cc = p.cd[l][(size_t)(t + 1) * BH + nnw + u];
    q.cp = p.cd[l][(size_t)t * BH + nnw + u];
  };
  auto cell_bwd = [&](int l, int t, const CellIn& q, float dh, float& dc) {
    T* D = (T*)p.dg_d[l] + (size_t)t * B * 4 * HH;
    if (valid) {
      const unsigned og = 4u * nnw + u;
      const float tc = uic_tanh<T>(q.cc);
      const float d = dc + dh * q.go * (1.f - tc * tc);
      const float d_o = dh * tc;
      st_x<SAFE>(D + og, d * q.gg * q.gi * (1.f - q.gi));
      st_x<SAFE>(D + og + HH, d * q.cp * q.gf * (1.f - q.gf));
      st_x<SAFE>(D + og + 2 * HH, d * q.gi * (1.f - q.gg * q.gg));
      st_x<SAFE>(D + og + 3 * HH, d_o * q.go * (1.f - q.go));
      dc = d * q.gf;
    }
  };
  // The attention backward's operands of the row this workgroup takes (phase B) are the same at every step: the encoder's
  // contexts and their linear_in image.  With <= 32 source positions they stay in registers for the whole launch.
  constexpr bool HOIST = MAXR <= 8;
  const bool att_wg = c.rank < c.nrow;
  uint4 cr[MAXR];                          // (S <= BW_NW * MAXR source positions)
  float4 wr[MAXR][2];
  auto att_load = [&]() {
    const int b = c.rbegin + c.rank;
    const T* ctx = (const T*)p.ctx;
#pragma unroll
    for (int uu = 0; uu < MAXR; ++uu) {
      const int sp = c.wave + BW_NW * uu;
      const size_t r = ((size_t)(sp < S ? sp : S - 1) * B + b) * HH;
      cr[uu] = *(const uint4*)(ctx + r + c.lane * 8);
      wr[uu][0] = *(const float4*)(p.ctxw + r + c.lane * 4);
      wr[uu][1] = *(const float4*)(p.ctxw + r + (c.lane + 64) * 4);
    }
  };
  if (HOIST && att_wg) att_load();
  // Phase A's operands that do not depend on the exchange -- the generator's d out (f32, HBM) and the forward pass's tanh outputs
  // -- are requested a step ahead, behind phase D2's fragment loads.
  float4 ao0[4], ao1[4];
  uint4 aop[4];
  auto a_prefetch = [&](int t) {
    const float* d_out = p.d_out_all + (size_t)t * BH;
    const T* out_pre = (const T*)p.out_pre + (size_t)t * BH;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned e = (unsigned)((c.rbegin + arow) * HH + (c.wave + BW_NW * j) * 32 + c.lq * 8);
      ao0[j] = *(const float4*)(d_out + e);
      ao1[j] = *(const float4*)(d_out + e + 4);
      aop[j] = *(const uint4*)(out_pre + e);
    }
  };
  constexpr bool PREA = MAXR <= 8;          // (with 64 source positions' operands in flight in phase B there are no registers for it)
  if (PREA) a_prefetch(p.Td - 1);
  float att_al = 0.f;                      // phase B: the forward pass's attention weight of source position `lane` of this workgroup's row

#define BW_STAMP(i) do { if (dbg && c.tid == 0) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
  for (int t = p.Td - 1; t >= 0; --t) {
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    unsigned long long* dbg = p.dbg ? p.dbg + ((size_t)blockIdx.x * p.Td + t) * 16 : nullptr;
    BW_STAMP(0);
    const bool last = t == p.Td - 1;
    CellIn q1, q0;
    cell_load(1, t, q1);
    cell_load(0, t, q0);
    float dq_lin = 0.f;
    {  // ---- phase A
      T* d_pre = (T*)p.d_pre_all + (size_t)t * BH;
      if (!PREA) a_prefetch(t);
      const __amdgpu_buffer_rsrc_t rf = rsrc_of(p.dfeed_x + (size_t)(last ? t : t + 1) * BH);
      u32x4 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = (c.wave + BW_NW * j) * 32 + c.lq * 8;
        const unsigned e = (unsigned)((c.rbegin + arow) * HH + k0);
        const float4 o0 = ao0[j], o1 = ao1[j];
        float g[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        {
          // (loaded at the last step too -- from that step's own slab, a valid address -- and dropped by a select: behind `if (!last)` hipcc
          // waited for each pair of loads at the branch's join, four L2 round trips in a row)
          const u32x4 f0 = bload<true>(rf, e * 4u, 0), f1 = bload<true>(rf, e * 4u + 16u, 0);
          g[0] += last ? 0.f : __uint_as_float(f0.x); g[1] += last ? 0.f : __uint_as_float(f0.y);
          g[2] += last ? 0.f : __uint_as_float(f0.z); g[3] += last ? 0.f : __uint_as_float(f0.w);
          g[4] += last ? 0.f : __uint_as_float(f1.x); g[5] += last ? 0.f : __uint_as_float(f1.y);
          g[6] += last ? 0.f : __uint_as_float(f1.z); g[7] += last ? 0.f : __uint_as_float(f1.w);
        }
        float op[8];
        uic_unpack<T>(aop[j], op);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (p.drop_p > 0.f) g[k] *= uic_drop_scale(p.seed, NMT_SITE_OUT(t), e + (unsigned)k, p.drop_p, inv_keep);
          g[k] *= 1.f - op[k] * op[k];
        }
        const uint4 pk = uic_pack<T>(g);
        af[j] = __builtin_bit_cast(u32x4, pk);
        // d_pre is also linear_out's weight-gradient operand: the workgroup that owns these 8 columns stores them
        if ((k0 >> 4) == (c.u0 >> 4) && c.l15 < c.nrow) *(uint4*)(d_pre + e) = pk;
      }
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[h] = Mma<T>::run(af[j], wot[j][h], acc[h]);
      float out[2];
      reduce2(acc, out);
      float* d_cq = p.d_cq_all + (size_t)t * B * 2 * HH;
      if (valid) st_x<SAFE>(d_cq + 2u * nnw + u, out[0]);                 // d c (attention backward, deferred accumulation)
      dq_lin = out[1];
    }
    BW_STAMP(1);
    group_arrive(c);
    if (att_wg) {                          // (not exchanged: requested while the workgroup waits for the others)
      if (!HOIST) att_load();
      att_al = c.lane < S ? p.attn_all[((size_t)t * B + (c.rbegin + c.rank)) * S + c.lane] : 0.f;
    }
    if (!group_wait(c, (int*)c.smem)) return;
    BW_STAMP(2);
    if (att_wg) {  // ---- phase B: attention backward of row `rank` of the group (csrc/nmt.hip gattn_bwd_step_fast_kernel)
      const int b = c.rbegin + c.rank;
      float* s_da = (float*)c.smem + 64;       // [64]
      float* s_red = s_da + 64;                // [BW_NW][HH]
      // d c of the row (written by the other workgroups in phase A): every wave reads its own copy in the layout of its products
      float dcv[8];
      {
        const __amdgpu_buffer_rsrc_t rd = rsrc_of(p.d_cq_all + (size_t)t * B * 2 * HH);
        const unsigned o = (unsigned)((b * 2 * HH + c.lane * 8) * 4);
        const u32x4 d0 = bload<true>(rd, o, 0), d1 = bload<true>(rd, o + 16u, 0);
        dcv[0] = __uint_as_float(d0.x); dcv[1] = __uint_as_float(d0.y); dcv[2] = __uint_as_float(d0.z); dcv[3] = __uint_as_float(d0.w);
        dcv[4] = __uint_as_float(d1.x); dcv[5] = __uint_as_float(d1.y); dcv[6] = __uint_as_float(d1.z); dcv[7] = __uint_as_float(d1.w);
      }
#pragma unroll
      for (int uu = 0; uu < MAXR; ++uu) {
        const int sp = c.wave + BW_NW * uu;
        if (sp < S) {
          float f[8];
          uic_unpack<T>(cr[uu], f);
          float pr = 0.f;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) pr += f[kk] * dcv[kk];
          pr = uic_wave_sum(pr);
          if (c.lane == 0) s_da[sp] = pr;
        }
      }
      __syncthreads();
      // softmax backward in the lanes (S <= 64): d score[s] = a[s] (d a[s] - sum_s' a[s'] d a[s'])
      const float da_l = c.lane < S ? s_da[c.lane] : 0.f;
      const float wbar = uic_wave_sum(att_al * da_l);
      const float ds_l = att_al * (da_l - wbar);
      if (c.wave == 0 && c.lane < S) p.dscore_all[((size_t)t * B + b) * S + c.lane] = ds_l;
      float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int uu = 0; uu < MAXR; ++uu) {
        const int sp = c.wave + BW_NW * uu;
        const float a = __shfl(ds_l, sp & 63, 64);
        if (sp < S) {
          a0[0] += a * wr[uu][0].x; a0[1] += a * wr[uu][0].y; a0[2] += a * wr[uu][0].z; a0[3] += a * wr[uu][0].w;
          a1[0] += a * wr[uu][1].x; a1[1] += a * wr[uu][1].y; a1[2] += a * wr[uu][1].z; a1[3] += a * wr[uu][1].w;
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { s_red[c.wave * HH + c.lane * 4 + kk] = a0[kk]; s_red[c.wave * HH + (c.lane + 64) * 4 + kk] = a1[kk]; }
      __syncthreads();
#pragma unroll
      for (int half = 0; half < 2; ++half) {                   // (HH = 2 BW_NTH: two columns per thread)
        const int col = c.tid + half * BW_NTH;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < BW_NW; ++w) v += s_red[w * HH + col];
        st_x<SAFE>(p.dq_att_x + (size_t)t * BH + (size_t)b * HH + col, v);
      }
      __syncthreads();
    }
    BW_STAMP(3);
    if (!group_barrier(c)) return;
    BW_STAMP(4);
    float dx1 = 0.f;
    {  // ---- phase C1: top layer's cell backward
      const __amdgpu_buffer_rsrc_t rq = rsrc_of(p.dq_att_x + (size_t)t * BH);
      const float dh = dq_lin + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, (nnw + u) * 4u, 0, 16)) + dh1_rec;
      cell_bwd(1, t, q1, dh, dc1);
    }
    BW_STAMP(5);
    if (!group_barrier(c)) return;
    BW_STAMP(6);
    {  // ---- phase C2
      float out[2];
      dgemm((const T*)p.dg_d[1] + (size_t)t * B * 4 * HH + 4 * rb, [&](int j, int h) { return w1t[j][h]; }, out, [] {});
      dx1 = out[0]; dh1_rec = out[1];
    }
    {  // ---- phase D1: layer 0's cell backward (its h went through the inter-layer dropout)
      float v = dx1;
      if (p.drop_p > 0.f) v *= uic_drop_scale(p.seed, NMT_SITE_DEC(0, t), nnw + u, p.drop_p, inv_keep);
      cell_bwd(0, t, q0, v + dh0_rec, dc0);
    }
    BW_STAMP(7);
    if (!group_barrier(c)) return;
    BW_STAMP(8);
    {  // ---- phase D2
      float out[2];
      dgemm((const T*)p.dg_d[0] + (size_t)t * B * 4 * HH + 4 * rb, [&](int j, int h) { return w0t[((c.wave * 16 + j) * 2 + h) * 64 + c.lane]; }, out,
            [&] { if (PREA) a_prefetch(t > 0 ? t - 1 : 0); });      // (PREA is a compile-time constant; no run-time branch around the loads)
      if (valid) st_x<SAFE>(p.dfeed_x + (size_t)t * BH + nnw + u, out[0]);
      dh0_rec = out[1];
    }
    BW_STAMP(9);
    if (!group_barrier(c)) return;
    BW_STAMP(10);
  }
#undef BW_STAMP
  // what the encoder's backward pass starts from: d h_l(-1) in the second halves of the [B, 2 x 512] hand-over buffers, d c_l(-1)
  if (valid) {
    p.dh_init[0][2u * nnw + HH + u] = dh0_rec;
    p.dh_init[1][2u * nnw + HH + u] = dh1_rec;
    p.dc_init[0][nnw + u] = dc0;
    p.dc_init[1][nnw + u] = dc1;
  }
}

template <int MAXR>
__global__ __launch_bounds__(BW_NTH) void nmt_dec_bwd_kernel(const UicNmtDecBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem + BW_W0_BYTES, c);
  if (mode == 0) return;
  if (mode == 2) nmt_dec_bwd_steps<true, MAXR>(p, c, smem);
  else nmt_dec_bwd_steps<false, MAXR>(p, c, smem);
}

// ---------------------------------------------------------------------------------------------------
// Encoder layer, forward (struct UicNmtEncParams in uic_common.h).  Per iteration ONE group barrier: the recurrent GEMM
// [<= 16 rows] x [K 256] x [64 gate columns] against a register-resident slice (8 k-steps, one per wave: 16 registers), partial
// tiles summed through LDS, cell update by waves 0-3 (a row of every 4-row group each), h exchanged.  30 iterations replace 2 x 30 dependent launches.
constexpr int ENC_HD = HH / 2;

template <bool SAFE>
__device__ __forceinline__ void nmt_enc_fwd_steps(const UicNmtEncParams& p, Ctx& c) {
  typedef bf16_t T;
  const int B = p.B, S = p.S;
  const int dir = c.rank >> 4;
  const int u0 = (c.rank & 15) * 16;                 // first of this workgroup's 16 units inside its direction
  const unsigned u = (unsigned)(u0 + c.l15);
  u32x4 wf[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) wf[g] = ws_wfrag(p.w_hh[dir], ENC_HD, g * ENC_HD + u0 + c.l15, c.wave * 32, c.lq);
  const int arow = c.l15 < c.nrow ? c.l15 : c.nrow - 1;
  // the cell update is spread over waves 0-3: wave w finishes accumulator component w, i.e. row 4 lq + w, unit u, and keeps that
  // element's cell state in a register
  const bool fin = c.wave < 4;
  const int rw = 4 * c.lq + (c.wave & 3);
  const int row = c.rbegin + (rw < c.nrow ? rw : c.nrow - 1);
  f32x4* red = (f32x4*)c.smem;
  const float* gx = p.gx[dir];
  float* cst = p.c[dir];
  T* gates = (T*)p.gates[dir];
  T* xo = (T*)p.x_out;
  // this element's share of W_ih x + b (f32, HBM): requested one iteration ahead, behind the iteration's A fragment
  float pvn[4];
  auto load_pv = [&](int k) {
    const int st = dir == 0 ? k : S - 1 - k;
#pragma unroll
    for (int g = 0; g < 4; ++g) pvn[g] = gx[((size_t)st * B + row) * (4 * ENC_HD) + g * ENC_HD + u];
  };
  load_pv(0);
  float cs = cst[((size_t)(dir == 0 ? 0 : S + 1) * B + row) * ENC_HD + u];      // (the zeroed slot in front of the first step)
  for (int k = 0; k < S; ++k) {
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    const int st = dir == 0 ? k : S - 1 - k;
    const int prev = dir == 0 ? st : st + 2;         // slot of the previous state in this direction
    const int alive = p.nb[st];
    const u32x4 af = bload<true>(rsrc_of(xo + (size_t)prev * B * HH), (unsigned)(((c.rbegin + arow) * HH + dir * ENC_HD + c.wave * 32 + c.lq * 8) * 2), 0);
    const float pv[4] = {pvn[0], pvn[1], pvn[2], pvn[3]};
    // (unconditional, the last iteration re-reads its own row: behind `if (k + 1 < S)` hipcc waited for these HBM loads at the branch's
    // join -- vmcnt(0) in front of the MFMAs below -- and the prefetch hid nothing)
    load_pv(k + 1 < S ? k + 1 : k);
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = Mma<T>::run(af, wf[g], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(c.wave * 4 + g) * 64 + c.lane] = acc[g];
    __syncthreads();
    if (fin) {
      float sg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = ((const float*)(red + g * 64 + c.lane))[c.wave];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) v += ((const float*)(red + (w * 4 + g) * 64 + c.lane))[c.wave];
        sg[g] = v;
      }
      if (rw < c.nrow && row < alive) {
        const float gi = uic_sigmoid_t<T>(sg[0] + pv[0]);
        const float gf = uic_sigmoid_t<T>(sg[1] + pv[1]);
        const float gg = uic_tanh<T>(sg[2] + pv[2]);
        const float go = uic_sigmoid_t<T>(sg[3] + pv[3]);
        const float cn = gf * cs + gi * gg;
        cs = cn;
        cst[((size_t)(st + 1) * B + row) * ENC_HD + u] = cn;
        st_x<SAFE>(xo + ((size_t)(st + 1) * B + row) * HH + dir * ENC_HD + u, go * uic_tanh<T>(cn));
        T* G = gates + ((size_t)st * B + row) * (4 * ENC_HD) + u;
        __builtin_nontemporal_store(uic_from_f<T>(gi), G);
        __builtin_nontemporal_store(uic_from_f<T>(gf), G + ENC_HD);
        __builtin_nontemporal_store(uic_from_f<T>(gg), G + 2 * ENC_HD);
        __builtin_nontemporal_store(uic_from_f<T>(go), G + 3 * ENC_HD);
      }
    }
    __syncthreads();
    if (!group_barrier(c)) return;
  }
}

__global__ __launch_bounds__(NTH) void nmt_enc_fwd_kernel(const UicNmtEncParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem, c);
  if (mode == 0) return;
  if (mode == 2) nmt_enc_fwd_steps<true>(p, c);
  else nmt_enc_fwd_steps<false>(p, c);
}

// Encoder layer, BPTT: per iteration the cell backward of the own units (d h = d_top + the carried d h, d c carried), d gates
// exchanged, ONE group barrier, then the carried d h of the next iteration = d gates [<= 16 rows, 4 x 256] x the resident W_hh^T slice
// (16 columns x K 1024: 32 k-steps, 4 per wave).  Replaces (lstm_bwd + GEMM) x S x 2 directions.
template <bool SAFE>
__device__ __forceinline__ void nmt_enc_bwd_steps(const UicNmtEncParams& p, Ctx& c) {
  typedef bf16_t T;
  const int B = p.B, S = p.S;
  const int dir = c.rank >> 4;
  const int u0 = (c.rank & 15) * 16;
  const unsigned u = (unsigned)(u0 + c.l15);
  u32x4 wt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wt[j] = ws_wfrag(p.w_hh[dir], 4 * ENC_HD, u0 + c.l15, (c.wave + NWAVE * j) * 32, c.lq);
  const int arow = c.l15 < c.nrow ? c.l15 : c.nrow - 1;
  // element-wise work spread over waves 0-3 (wave w: accumulator component w = row 4 lq + w, unit u); d h and d c of that
  // element are carried in registers
  const bool fin = c.wave < 4;
  const int rw = 4 * c.lq + (c.wave & 3);
  const int row = c.rbegin + (rw < c.nrow ? rw : c.nrow - 1);
  f32x4* red = (f32x4*)c.smem;
  const float* cst = p.c[dir];
  const T* gates = (const T*)p.gates[dir];
  T* dgs = (T*)p.dgates[dir];
  float dh = 0.f, dc = 0.f;
  if (fin) {
    dh = p.dh_init[(size_t)row * p.ld_dh_init + dir * ENC_HD + u];
    dc = p.dc_init[(size_t)row * p.ld_dc_init + dir * ENC_HD + u];
  }
  // The cell backward's operands (the forward pass's gates and cell states, the gradient from the layer above) do not depend on
  // this launch: requested one iteration ahead, behind the iteration's A fragments -- requested where they are used they are a
  // memory latency per iteration on the critical path.
  struct Ops { float gi, gf, gg, go, cc, cp, dtop; };
  auto load_ops = [&](int k, Ops& q) {
    const int st = dir == 0 ? k : S - 1 - k;
    const int prev = dir == 0 ? st : st + 2;
    const T* G = gates + ((size_t)st * B + row) * (4 * ENC_HD) + u;
    q.gi = uic_to_f(G[0]); q.gf = uic_to_f(G[ENC_HD]); q.gg = uic_to_f(G[2 * ENC_HD]); q.go = uic_to_f(G[3 * ENC_HD]);
    q.cc = cst[((size_t)(st + 1) * B + row) * ENC_HD + u];
    q.cp = cst[((size_t)prev * B + row) * ENC_HD + u];
    q.dtop = p.d_top[((size_t)st * B + row) * HH + dir * ENC_HD + u];
  };
  Ops q = {}, qn = {};
  if (fin) load_ops(S - 1, q);
  for (int k = S - 1; k >= 0; --k) {
    asm volatile("" : "+v"(c.lane), "+v"(c.l15), "+v"(c.lq), "+v"(c.tid));
    const int st = dir == 0 ? k : S - 1 - k;
    const int alive = p.nb[st];
    if (fin && rw < c.nrow && row < alive) {
      const float dht = q.dtop + dh;
      const float tc = uic_tanh<T>(q.cc);
      const float d = dc + dht * q.go * (1.f - tc * tc);
      const float d_o = dht * tc;
      T* D = dgs + ((size_t)st * B + row) * (4 * ENC_HD) + u;
      st_x<SAFE>(D, d * q.gg * q.gi * (1.f - q.gi));
      st_x<SAFE>(D + ENC_HD, d * q.cp * q.gf * (1.f - q.gf));
      st_x<SAFE>(D + 2 * ENC_HD, d * q.gi * (1.f - q.gg * q.gg));
      st_x<SAFE>(D + 3 * ENC_HD, d_o * q.go * (1.f - q.go));
      dc = d * q.gf;
    }
    if (!group_barrier(c)) return;
    {
      const __amdgpu_buffer_rsrc_t ra = rsrc_of(dgs + (size_t)st * B * 4 * ENC_HD);
      u32x4 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = bload<true>(ra, (unsigned)(((c.rbegin + arow) * 4 * ENC_HD + (c.wave + NWAVE * j) * 32 + c.lq * 8) * 2), 0);
      // (every wave, every iteration -- waves 4-7 and the last iteration load operands nobody uses: behind `if (fin && k > 0)` hipcc
      // waited for these HBM loads at the branch's join, in front of the MFMAs)
      load_ops(k > 0 ? k - 1 : 0, qn);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = Mma<T>::run(af[j], wt[j], acc);
      red[c.wave * 64 + c.lane] = acc;
      __syncthreads();
      if (fin) {
        float sres = ((const float*)(red + c.lane))[c.wave];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) sres += ((const float*)(red + w * 64 + c.lane))[c.wave];
        if (c.rbegin + rw < alive) dh = sres;       // (a row that is not alive yet keeps its initial carry)
      }
      __syncthreads();
      q = qn;
    }
  }
}

__global__ __launch_bounds__(NTH) void nmt_enc_bwd_kernel(const UicNmtEncParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  const int mode = setup_ctx(p, smem, c);
  if (mode == 0) return;
  if (mode == 2) nmt_enc_bwd_steps<true>(p, c);
  else nmt_enc_bwd_steps<false>(p, c);
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
  if (!p.sync_zeroed) UIC_TRY(uic_check_hip(hipMemsetAsync(p.sync, 0, (size_t)SY_WORDS * 4, s), "hipMemsetAsync(nmt dec sync)"));
  if (p.B <= 8 * 16 && p.NL <= 2) hipLaunchKernelGGL(nmt_dec_ws_kernel, dim3(8 * PW), dim3(NTH), WS_LDS_BYTES, s, p);     // one 16-row tile per group
  else hipLaunchKernelGGL(nmt_dec_persist_kernel, dim3(8 * PW), dim3(NTH), LDS_BYTES, s, p);
  UIC_LAUNCH_CHECK("nmt_dec_persist_kernel");
  return gate.leave();
}

bool uic_nmt_dec_bwd_persist_eligible(int dtype, int B, int S, int H, int NL) {
  return NL == 2 && B <= 8 * 16 && uic_nmt_dec_persist_eligible(dtype, B, S, H, NL);
}

int uic_nmt_dec_bwd_persist_launch(const UicNmtDecBwdParams& p, hipStream_t s) {
  UIC_REQUIRE(p.sync && p.Td > 0 && p.B > 0 && p.B <= 8 * 16 && p.Nrows == p.B && p.row0 == 0, "nmt_dec_bwd_persist: bad arguments");
  static bool configured = false;
  if (!configured) {
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)nmt_dec_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, BW_W0_BYTES + WS_SCR_BYTES), "hipFuncSetAttribute(nmt dec bwd)"));
    UIC_TRY(uic_check_hip(hipFuncSetAttribute((const void*)nmt_dec_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, BW_W0_BYTES + WS_SCR_BYTES), "hipFuncSetAttribute(nmt dec bwd)"));
    configured = true;
  }
  UicPersistGateScope gate;
  UIC_TRY(gate.enter(s));
  if (!p.sync_zeroed) UIC_TRY(uic_check_hip(hipMemsetAsync(p.sync, 0, (size_t)SY_WORDS * 4, s), "hipMemsetAsync(nmt dec bwd sync)"));
  if (p.S <= BW_NW * 8) hipLaunchKernelGGL(nmt_dec_bwd_kernel<8>, dim3(8 * PW), dim3(BW_NTH), BW_W0_BYTES + WS_SCR_BYTES, s, p);
  else hipLaunchKernelGGL(nmt_dec_bwd_kernel<16>, dim3(8 * PW), dim3(BW_NTH), BW_W0_BYTES + WS_SCR_BYTES, s, p);
  UIC_LAUNCH_CHECK("nmt_dec_bwd_kernel");
  return gate.leave();
}

bool uic_nmt_enc_persist_eligible(int dtype, int B, int S, int H) {
  return S <= UIC_NMT_ENC_MAX_S && B <= 8 * 16 && uic_nmt_dec_persist_eligible(dtype, B, S, H, 1);
}

namespace {
template <typename K>
int enc_launch(K kernel, const char* what, const UicNmtEncParams& p, hipStream_t s) {
  UIC_REQUIRE(p.sync && p.S > 0 && p.S <= UIC_NMT_ENC_MAX_S && p.B > 0 && p.B <= 8 * 16 && p.Nrows == p.B && p.row0 == 0, "%s: bad arguments", what);
  UicPersistGateScope gate;
  UIC_TRY(gate.enter(s));
  if (!p.sync_zeroed) UIC_TRY(uic_check_hip(hipMemsetAsync(p.sync, 0, (size_t)SY_WORDS * 4, s), "hipMemsetAsync(nmt enc sync)"));
  hipLaunchKernelGGL(kernel, dim3(8 * PW), dim3(NTH), WS_SCR_BYTES, s, p);
  UIC_LAUNCH_CHECK(what);
  return gate.leave();
}
}  // namespace
int uic_nmt_enc_fwd_persist_launch(const UicNmtEncParams& p, hipStream_t s) { return enc_launch(nmt_enc_fwd_kernel, "nmt_enc_fwd_kernel", p, s); }
int uic_nmt_enc_bwd_persist_launch(const UicNmtEncParams& p, hipStream_t s) { return enc_launch(nmt_enc_bwd_kernel, "nmt_enc_bwd_kernel", p, s); }
