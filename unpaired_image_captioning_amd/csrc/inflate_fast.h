// A raw-deflate (RFC 1951) decoder for the loader's one use: whole zip members of a known uncompressed size, decoded in one call
// into one contiguous buffer (np.savez_compressed members of the reference's per-image feature files, scripts/make_bu_data.py:55).
// Host code.  zlib's inflate() does this at ~310-480 MB/s per core on bottom-up features (dense positive floats: almost every
// symbol is a literal); with the whole input and output in memory the decoder can keep 56+ bits in a 64-bit buffer, refill it
// without a branch, resolve a symbol with ONE lookup in an 11-bit table and take up to three literals per refill.
// Anything unusual -- a malformed tree, a distance before the start of the output, input or output running out -- makes
// uic_inflate_fast return false and the caller falls back to zlib, which then reports the error (or decodes what this did not).
#pragma once
#include <cstdint>
#include <cstring>
#include <mutex>

namespace uic_inflate {

// table entry: bits 0-5 bits to consume at this step (the shift count is taken as `e & 63`, which x86-64 and AArch64 shifts apply by
// themselves: one instruction less in the lookup -> shift -> lookup chain that bounds the decoder), bits 6-8 kind, bits 9-13
// extra-bit count (length / distance) or the subtable's index width (pointer), bits 16-31 literal byte / base length / base
// distance / subtable offset
enum : uint32_t { K_LIT = 0u << 6, K_LEN = 1u << 6, K_EOB = 2u << 6, K_SUB = 3u << 6, K_BAD = 4u << 6, K_MASK = 7u << 6 };
constexpr int X_SHIFT = 9;
constexpr int LIT_TB = 11, DIST_TB = 8, CL_TB = 7;
constexpr int LIT_CAP = (1 << LIT_TB) + 288 * 16, DIST_CAP = (1 << DIST_TB) + 32 * 128, CL_CAP = 1 << CL_TB;

inline uint32_t rev_bits(uint32_t c, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1); c >>= 1; }
  return r;
}

// Canonical Huffman code of `n` symbols with lengths lens[] (0 = unused) -> lookup table with `tb` primary index bits, LSB-first.
// sym_entry(s) gives the kind / extra / value part of symbol s's entry.  Returns false for over-subscribed or (except for the
// one-code case RFC 1951 allows for distances) incomplete codes and when the table would not fit `cap` entries.
template <typename F>
inline bool build_table(const uint8_t* lens, int n, int tb, uint32_t* table, int cap, F sym_entry) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  count[0] = 0;
  int used = 0, left = 1;
  for (int l = 1; l <= 15; ++l) {
    left = (left << 1) - count[l];
    if (left < 0) return false;                    // over-subscribed
    used += count[l];
  }
  if (used == 0) return false;
  if (left > 0 && !(used == 1)) return false;      // incomplete (a single code of any length is tolerated: its other half stays K_BAD)
  uint32_t next[16];
  {
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
  }
  const int P = 1 << tb;
  for (int i = 0; i < P; ++i) table[i] = K_BAD | 1;
  // pass 1: the widest code under every primary prefix that has long codes
  uint8_t subw[1 << LIT_TB];
  memset(subw, 0, (size_t)P);
  uint32_t codes[320];
  {
    uint32_t nx[16];
    memcpy(nx, next, sizeof(nx));
    for (int s = 0; s < n; ++s) {
      const int l = lens[s];
      if (!l) continue;
      const uint32_t r = rev_bits(nx[l]++, l);
      codes[s] = r;
      if (l > tb) { const int w = l - tb; if (w > subw[r & (P - 1)]) subw[r & (P - 1)] = (uint8_t)w; }
    }
  }
  int top = P;
  for (int p = 0; p < P; ++p) {
    if (!subw[p]) continue;
    const int size = 1 << subw[p];
    if (top + size > cap) return false;
    table[p] = K_SUB | (uint32_t)tb | ((uint32_t)subw[p] << X_SHIFT) | ((uint32_t)top << 16);
    for (int i = 0; i < size; ++i) table[top + i] = K_BAD | 1;
    top += size;
  }
  // pass 2: fill
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t r = codes[s], e = sym_entry(s);
    if (l <= tb) {
      for (uint32_t i = r; i < (uint32_t)P; i += 1u << l) table[i] = e | (uint32_t)l;
    } else {
      const uint32_t ptr = table[r & (P - 1)];
      if ((ptr & K_MASK) != K_SUB) return false;   // (a short code owns this prefix: over-subscription the count test cannot see)
      const int w = (int)((ptr >> X_SHIFT) & 31), base = (int)(ptr >> 16), sl = l - tb;
      for (uint32_t i = r >> tb; i < (1u << w); i += 1u << sl) table[base + i] = e | (uint32_t)sl;
    }
  }
  return true;
}

inline uint32_t litlen_entry(int s) {
  static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  if (s < 256) return K_LIT | ((uint32_t)s << 16);
  if (s == 256) return K_EOB;
  if (s > 285) return K_BAD;
  return K_LEN | ((uint32_t)extra[s - 257] << X_SHIFT) | ((uint32_t)base[s - 257] << 16);
}
inline uint32_t dist_entry(int s) {
  static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  if (s > 29) return K_BAD;
  return K_LEN | ((uint32_t)extra[s] << X_SHIFT) | ((uint32_t)base[s] << 16);
}

struct Tables {
  uint32_t lit[LIT_CAP];
  uint32_t dist[DIST_CAP];
};

inline const Tables* fixed_tables() {
  static Tables t;
  static bool ok = false;
  static std::once_flag once;
  std::call_once(once, [] {
    uint8_t ll[288], dl[32];
    for (int i = 0; i < 144; ++i) ll[i] = 8;
    for (int i = 144; i < 256; ++i) ll[i] = 9;
    for (int i = 256; i < 280; ++i) ll[i] = 7;
    for (int i = 280; i < 288; ++i) ll[i] = 8;
    for (int i = 0; i < 32; ++i) dl[i] = 5;
    ok = build_table(ll, 288, LIT_TB, t.lit, LIT_CAP, litlen_entry) && build_table(dl, 32, DIST_TB, t.dist, DIST_CAP, dist_entry);
  });
  return ok ? &t : nullptr;
}

// ---- decoder state of one stream.  src must be readable up to src + n + 16 (the caller pads its buffer): the bit buffer is
// refilled with unaligned 8-byte loads.
struct Stream {
  const unsigned char* src; const unsigned char* in; const unsigned char* in_end;
  unsigned char* dst; unsigned char* out; unsigned char* out_end;
  uint64_t bitbuf; unsigned bitcnt;
  int bfinal;
  const uint32_t* lit; const uint32_t* dt;       // tables of the block being decoded
  Tables* work;
  size_t n;
  void start(const unsigned char* s, size_t n_, unsigned char* d, size_t m, Tables* w) {
    src = in = s; in_end = s + n_; dst = out = d; out_end = d + m; bitbuf = 0; bitcnt = 0; bfinal = 0; lit = dt = nullptr; work = w; n = n_;
  }
  // the stream decoded exactly its output and did not read past its input
  bool finished_ok() const {
    if ((size_t)(in - src) * 8 < bitcnt) return false;
    return out == out_end && (size_t)(in - src) * 8 - bitcnt <= n * 8;
  }
};

#define UIC_REFILL_(in, bitbuf, bitcnt)                 \
  do {                                                  \
    uint64_t w_;                                        \
    memcpy(&w_, in, 8);                                 \
    bitbuf |= w_ << bitcnt;                             \
    in += (63 - bitcnt) >> 3;                           \
    bitcnt |= 56;                                       \
  } while (0)
#define UIC_TAKE_(bitbuf, bitcnt, nb) (bitbuf >>= (nb), bitcnt -= (nb))

// Headers up to the next Huffman block (stored blocks are copied on the way).  1: S.lit / S.dt are set, decode symbols; 0: the
// final block was a stored one -- the stream is through; -1: malformed / out of space.
inline int next_block(Stream& S) {
  const unsigned char* in = S.in;
  uint64_t bitbuf = S.bitbuf;
  unsigned bitcnt = S.bitcnt;
#define UIC_REFILL() UIC_REFILL_(in, bitbuf, bitcnt)
#define UIC_TAKE(nb) UIC_TAKE_(bitbuf, bitcnt, nb)
#define UIC_SAVE() (S.in = in, S.bitbuf = bitbuf, S.bitcnt = bitcnt)
  for (;;) {
    if (in > S.in_end + 8) return -1;
    UIC_REFILL();
    S.bfinal = (int)(bitbuf & 1);
    const int type = (int)((bitbuf >> 1) & 3);
    UIC_TAKE(3);
    if (type == 0) {
      // stored: to the next byte boundary of the INPUT, LEN, NLEN, bytes
      UIC_TAKE(bitcnt & 7);
      const unsigned char* p = in - (bitcnt >> 3);
      if (p + 4 > S.in_end) return -1;
      const unsigned len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
      if ((len ^ nlen) != 0xffffu) return -1;
      p += 4;
      if (p + len > S.in_end || S.out + len > S.out_end) return -1;
      memcpy(S.out, p, len);
      S.out += len;
      in = p + len;
      bitbuf = 0;
      bitcnt = 0;
      if (S.bfinal) { UIC_SAVE(); return 0; }
      continue;
    }
    if (type == 1) {
      const Tables* T = fixed_tables();
      if (!T) return -1;
      S.lit = T->lit; S.dt = T->dist;
      UIC_SAVE();
      return 1;
    }
    if (type != 2) return -1;
    const int hlit = (int)(bitbuf & 31) + 257, hdist = (int)((bitbuf >> 5) & 31) + 1, hclen = (int)((bitbuf >> 10) & 15) + 4;
    UIC_TAKE(14);
    if (hlit > 286 || hdist > 30) return -1;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    UIC_REFILL();
    for (int i = 0; i < hclen; ++i) {
      if (i == 14) UIC_REFILL();                       // (14 x 3 = 42 bits of the 56 used)
      cl[order[i]] = (uint8_t)(bitbuf & 7);
      UIC_TAKE(3);
    }
    uint32_t clt[CL_CAP];
    if (!build_table(cl, 19, CL_TB, clt, CL_CAP, [](int s) { return K_LIT | ((uint32_t)s << 16); })) return -1;
    uint8_t lens[320];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
      if (in > S.in_end + 8) return -1;
      UIC_REFILL();
      const uint32_t e = clt[bitbuf & (CL_CAP - 1)];
      if ((e & K_MASK) != K_LIT) return -1;
      UIC_TAKE(e & 63);
      const int s = (int)(e >> 16);
      if (s < 16) { lens[i++] = (uint8_t)s; continue; }
      int rep, v = 0;
      if (s == 16) { if (i == 0) return -1; v = lens[i - 1]; rep = 3 + (int)(bitbuf & 3); UIC_TAKE(2); }
      else if (s == 17) { rep = 3 + (int)(bitbuf & 7); UIC_TAKE(3); }
      else { rep = 11 + (int)(bitbuf & 127); UIC_TAKE(7); }
      if (i + rep > total) return -1;
      while (rep--) lens[i++] = (uint8_t)v;
    }
    if (lens[256] == 0) return -1;                     // no end-of-block code
    if (!build_table(lens, hlit, LIT_TB, S.work->lit, LIT_CAP, litlen_entry)) return -1;
    if (!build_table(lens + hlit, hdist, DIST_TB, S.work->dist, DIST_CAP, dist_entry)) {
      // a block of literals only may give the distance code no usable length at all: every entry stays "bad", a match is an error
      bool none = true;
      for (int k = 0; k < hdist; ++k) none = none && lens[hlit + k] == 0;
      if (!none) return -1;
      for (int k = 0; k < (1 << DIST_TB); ++k) S.work->dist[k] = K_BAD | 1;
    }
    S.lit = S.work->lit; S.dt = S.work->dist;
    UIC_SAVE();
    return 1;
  }
#undef UIC_SAVE
#undef UIC_TAKE
#undef UIC_REFILL
}

// ONE symbol of any kind with every check, bit buffer just refilled (>= 56 bits: a 15-bit length code + 5 extra bits + a 15-bit
// distance code + 13 extra bits are 48).  0: go on; 1: end of block; -1: malformed / out of space.
#define UIC_SLOW_SYMBOL(X, result)                                                                                       \
  do {                                                                                                                   \
    uint32_t e_ = X##lit[X##bitbuf & ((1u << LIT_TB) - 1)];                                                              \
    if ((e_ & K_MASK) == K_SUB) {                                                                                        \
      UIC_TAKE_(X##bitbuf, X##bitcnt, e_ & 63);                                                                          \
      e_ = X##lit[(e_ >> 16) + (X##bitbuf & ((1u << ((e_ >> X_SHIFT) & 31)) - 1))];                                      \
    }                                                                                                                    \
    const uint32_t kind_ = e_ & K_MASK;                                                                                  \
    if (kind_ == K_LIT) {                                                                                                \
      if (X##out >= X##out_end) { result = -1; break; }                                                                  \
      UIC_TAKE_(X##bitbuf, X##bitcnt, e_ & 63);                                                                          \
      *X##out++ = (unsigned char)(e_ >> 16);                                                                             \
      result = 0;                                                                                                        \
      break;                                                                                                             \
    }                                                                                                                    \
    if (kind_ == K_EOB) { UIC_TAKE_(X##bitbuf, X##bitcnt, e_ & 63); result = 1; break; }                                 \
    if (kind_ != K_LEN) { result = -1; break; }                                                                          \
    UIC_TAKE_(X##bitbuf, X##bitcnt, e_ & 63);                                                                            \
    const unsigned xb_ = (e_ >> X_SHIFT) & 31;                                                                           \
    const size_t len_ = (e_ >> 16) + (size_t)(X##bitbuf & ((1u << xb_) - 1));                                            \
    UIC_TAKE_(X##bitbuf, X##bitcnt, xb_);                                                                                \
    uint32_t d_ = X##dt[X##bitbuf & ((1u << DIST_TB) - 1)];                                                              \
    if ((d_ & K_MASK) == K_SUB) {                                                                                        \
      UIC_TAKE_(X##bitbuf, X##bitcnt, d_ & 63);                                                                          \
      d_ = X##dt[(d_ >> 16) + (X##bitbuf & ((1u << ((d_ >> X_SHIFT) & 31)) - 1))];                                       \
    }                                                                                                                    \
    if ((d_ & K_MASK) != K_LEN) { result = -1; break; }                                                                  \
    UIC_TAKE_(X##bitbuf, X##bitcnt, d_ & 63);                                                                            \
    const unsigned db_ = (d_ >> X_SHIFT) & 31;                                                                           \
    const size_t dist_ = (d_ >> 16) + (size_t)(X##bitbuf & ((1u << db_) - 1));                                           \
    UIC_TAKE_(X##bitbuf, X##bitcnt, db_);                                                                                \
    if (dist_ > (size_t)(X##out - X##dst) || len_ > (size_t)(X##out_end - X##out)) { result = -1; break; }               \
    const unsigned char* from_ = X##out - dist_;                                                                         \
    if (dist_ >= len_) memcpy(X##out, from_, len_);                                                                      \
    else for (size_t k_ = 0; k_ < len_; ++k_) X##out[k_] = from_[k_];      /* (overlapping: a run) */                    \
    X##out += len_;                                                                                                      \
    result = 0;                                                                                                          \
  } while (0)

#define UIC_LOAD_STREAM(X, S)                                                                                            \
  const unsigned char* X##in = (S).in; const unsigned char* const X##in_end = (S).in_end;                                \
  unsigned char* X##out = (S).out; unsigned char* const X##out_end = (S).out_end; unsigned char* const X##dst = (S).dst; \
  uint64_t X##bitbuf = (S).bitbuf; unsigned X##bitcnt = (S).bitcnt;                                                      \
  const uint32_t* X##lit = (S).lit; const uint32_t* X##dt = (S).dt
#define UIC_STORE_STREAM(X, S) ((S).in = X##in, (S).out = X##out, (S).bitbuf = X##bitbuf, (S).bitcnt = X##bitcnt)
#define UIC_LOOK(X) X##lit[X##bitbuf & ((1u << LIT_TB) - 1)]
#define UIC_LITERAL(X, e) (UIC_TAKE_(X##bitbuf, X##bitcnt, (e) & 63), *X##out++ = (unsigned char)((e) >> 16))

// Symbols of the current Huffman block of ONE stream until its end-of-block code.  0: block done; -1: malformed / out of space.
inline int run_block(Stream& S) {
  UIC_LOAD_STREAM(a, S);
  int r = 0;
  for (;;) {
    if (ain > ain_end + 8) { r = -1; break; }
    UIC_REFILL_(ain, abitbuf, abitcnt);
    uint32_t e = UIC_LOOK(a);
    if ((e & K_MASK) == K_LIT && aout + 3 <= aout_end) {   // up to three literals on one refill (a literal of the primary table is <= 11 bits)
      UIC_LITERAL(a, e);
      e = UIC_LOOK(a);
      if ((e & K_MASK) == K_LIT) {
        UIC_LITERAL(a, e);
        e = UIC_LOOK(a);
        if ((e & K_MASK) == K_LIT) { UIC_LITERAL(a, e); continue; }
      }
      UIC_REFILL_(ain, abitbuf, abitcnt);
    }
    UIC_SLOW_SYMBOL(a, r);
    if (r != 0) break;
  }
  UIC_STORE_STREAM(a, S);
  return r == 1 ? 0 : -1;
}

// One stream, start to end.  Decodes exactly `m` bytes into dst or returns false (dst then holds garbage).
inline bool uic_inflate_fast(const unsigned char* src, size_t n, unsigned char* dst, size_t m, Tables* work) {
  Stream S;
  S.start(src, n, dst, m, work);
  for (;;) {
    const int b = next_block(S);
    if (b < 0) return false;
    if (b == 0) break;
    if (run_block(S) < 0) return false;
    if (S.bfinal) break;
  }
  return S.finished_ok();
}

// TWO streams side by side.  A deflate stream is one chain of dependent steps -- table lookup -> shift -> next lookup, ~8 cycles per
// symbol however wide the core is -- so a thread that decodes two members in lock-step does nearly twice the work per second:
// the two chains share the core's issue slots.  (Feature files are almost all literals: the lock-step loop is the pair of
// three-literal runs; anything else takes one checked symbol per stream and comes back.)  ok[i]: stream i decoded exactly.
inline void uic_inflate_fast_pair(Stream& A, Stream& B, bool ok[2]) {
  int ra = next_block(A), rb = next_block(B);
  while (ra == 1 && rb == 1) {
    UIC_LOAD_STREAM(a, A);
    UIC_LOAD_STREAM(b, B);
    int sa = 0, sb = 0;                                  // 0: in the block; 1: the block ended; -1: failed
    for (;;) {
      if (ain > ain_end + 8) { sa = -1; break; }
      if (bin > bin_end + 8) { sb = -1; break; }
      UIC_REFILL_(ain, abitbuf, abitcnt);
      UIC_REFILL_(bin, bbitbuf, bbitcnt);
      uint32_t ea = UIC_LOOK(a), eb = UIC_LOOK(b);
      if (((ea | eb) & K_MASK) == K_LIT && aout + 3 <= aout_end && bout + 3 <= bout_end) {
        UIC_LITERAL(a, ea); UIC_LITERAL(b, eb);
        ea = UIC_LOOK(a); eb = UIC_LOOK(b);
        if (((ea | eb) & K_MASK) == K_LIT) {
          UIC_LITERAL(a, ea); UIC_LITERAL(b, eb);
          ea = UIC_LOOK(a); eb = UIC_LOOK(b);
          if (((ea | eb) & K_MASK) == K_LIT) { UIC_LITERAL(a, ea); UIC_LITERAL(b, eb); }
        }
        continue;                                        // (whatever was not a literal is looked up again behind the next refill)
      }
      UIC_SLOW_SYMBOL(a, sa);
      if (sa != 0) break;
      UIC_SLOW_SYMBOL(b, sb);
      if (sb != 0) break;
    }
    UIC_STORE_STREAM(a, A);
    UIC_STORE_STREAM(b, B);
    if (sa < 0) ra = -1;
    else if (sa == 1) ra = A.bfinal ? 0 : next_block(A);
    if (sb < 0) rb = -1;
    else if (sb == 1) rb = B.bfinal ? 0 : next_block(B);
  }
  // what is left of either stream, alone
  for (int k = 0; k < 2; ++k) {
    Stream& S = k ? B : A;
    int r = k ? rb : ra;
    while (r == 1) {
      if (run_block(S) < 0) { r = -1; break; }
      r = S.bfinal ? 0 : next_block(S);
    }
    ok[k] = r == 0 && S.finished_ok();
  }
}

#undef UIC_LITERAL
#undef UIC_LOOK
#undef UIC_STORE_STREAM
#undef UIC_LOAD_STREAM
#undef UIC_SLOW_SYMBOL
#undef UIC_TAKE_
#undef UIC_REFILL_

}  // namespace uic_inflate
