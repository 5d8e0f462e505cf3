// Host side of the input pipeline: the per-image feature files (.npy, and .npz as np.savez / np.savez_compressed write
// them -- scripts/make_bu_data.py:55-57) read by a thread team STRAIGHT INTO the caller's pinned staging buffer, at the
// offsets the batch-assembly kernel (loader.hip) wants them.  The reference reads them with np.load in 4 DataLoader worker
// processes (P/misc/dataloader/dataloader.py:309,319,331,351-356) and re-copies them three times on the way to the device
// (worker -> pickle -> np.stack / replication -> torch.from_numpy().cuda()).
//
// Two calls: uic_loader_scan parses the headers (shapes are needed to lay the batch out), uic_loader_read moves the data.
// No device code in this file.
#include "uic_common.h"
#include "inflate_fast.h"

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int IO_MAX_THREADS = 128;
constexpr int IO_INFO = 6;            // per file: ndim, d0, d1, data offset in the (uncompressed) member, zip method, member offset in the file

struct FileBytes {
  std::vector<unsigned char> buf;
  bool read_all(const char* path, size_t limit) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return false; }
    size_t want = (size_t)st.st_size;
    if (limit && want > limit) want = limit;
    buf.resize(want);
    size_t got = 0;
    while (got < want) {
      const ssize_t r = pread(fd, buf.data() + got, want - got, (off_t)got);
      if (r <= 0) break;
      got += (size_t)r;
    }
    close(fd);
    buf.resize(got);
    return got == want;
  }
};

uint32_t le16(const unsigned char* p) { return p[0] | (p[1] << 8); }
uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// Locate member `member`.npy at the head of a zip (np.savez writes one member per array, ours has one): method and the
// offset of its data.  A plain .npy: method -1, offset 0.
bool locate(const std::vector<unsigned char>& b, const char* member, int* method, size_t* off, std::string* why) {
  if (b.size() >= 30 && b[0] == 'P' && b[1] == 'K' && b[2] == 3 && b[3] == 4) {
    const uint32_t nlen = le16(&b[26]), xlen = le16(&b[28]);
    *method = (int)le16(&b[8]);
    const std::string want = std::string(member ? member : "") + ".npy";
    if (b.size() < 30 + nlen || std::string((const char*)&b[30], nlen) != want) { *why = "first zip member is not " + want; return false; }
    if (*method != 0 && *method != 8) { *why = "zip method is neither stored nor deflate"; return false; }
    *off = 30 + nlen + xlen;
    return true;
  }
  *method = -1;
  *off = 0;
  return true;
}

// The .npy header at p[0..n): shape and the offset of the data (numpy/lib/format.py).  float32, C order only.
bool parse_npy(const unsigned char* p, size_t n, int64_t* ndim, int64_t* d0, int64_t* d1, size_t* data_off, std::string* why) {
  if (n < 12 || memcmp(p, "\x93NUMPY", 6) != 0) { *why = "not an .npy image"; return false; }
  size_t hlen, start;
  if (p[6] == 1) { hlen = le16(p + 8); start = 10; }
  else if (p[6] == 2 || p[6] == 3) { hlen = le32(p + 8); start = 12; }
  else { *why = "unknown .npy version"; return false; }
  if (start + hlen > n) { *why = ".npy header longer than the bytes scanned"; return false; }
  const std::string h((const char*)p + start, hlen);
  const size_t dpos = h.find("'descr'");
  if (dpos == std::string::npos || h.find("'<f4'", dpos) == std::string::npos || h.find("'<f4'", dpos) > dpos + 12) {
    *why = "dtype is not float32 ('<f4'): the assembly kernel restates the reference's float32 arithmetic";
    return false;
  }
  if (h.find("'fortran_order': False") == std::string::npos) { *why = "fortran_order is not False"; return false; }
  size_t s = h.find("'shape'");
  if (s == std::string::npos || (s = h.find('(', s)) == std::string::npos) { *why = "no shape in the .npy header"; return false; }
  int64_t dims[2] = {1, 1};
  int nd = 0;
  for (++s; s < h.size() && h[s] != ')';) {
    if (h[s] >= '0' && h[s] <= '9') {
      int64_t v = 0;
      while (s < h.size() && h[s] >= '0' && h[s] <= '9') v = v * 10 + (h[s++] - '0');
      if (nd >= 2) { *why = "more than 2 dimensions"; return false; }
      dims[nd++] = v;
    } else {
      ++s;
    }
  }
  *ndim = nd; *d0 = dims[0]; *d1 = dims[1];
  *data_off = start + hlen;
  return true;
}

// inflate raw deflate data src[0..n) : skip `skip` bytes of output, then write `want` bytes to dst (dst may be NULL when
// want == 0 and only `head` -- the first head_cap bytes -- is wanted).
bool inflate_member(const unsigned char* src, size_t n, size_t skip, unsigned char* dst, size_t want, unsigned char* head,
                    size_t head_cap, size_t* head_got) {
  // one inflate state per reader thread for the life of the pool (inflateInit2 allocates ~40 KB: per file and from 128 threads it
  // was mmap / page-fault traffic on the process's address-space lock, which is what kept more threads from helping)
  struct ZState {
    z_stream z; bool ready = false;
    ~ZState() { if (ready) inflateEnd(&z); }
  };
  static thread_local ZState zs;
  z_stream& z = zs.z;
  if (!zs.ready) {
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) return false;
    zs.ready = true;
  } else if (inflateReset2(&z, -15) != Z_OK) {
    return false;
  }
  z.next_in = const_cast<unsigned char*>(src);
  z.avail_in = (uInt)n;
  bool ok = true;
  if (head) {
    z.next_out = head; z.avail_out = (uInt)head_cap;
    const int r = inflate(&z, Z_SYNC_FLUSH);
    ok = r == Z_OK || r == Z_STREAM_END || r == Z_BUF_ERROR;
    *head_got = head_cap - z.avail_out;
  } else {
    static thread_local std::vector<unsigned char> scratch;
    if (scratch.size() < (skip ? skip : 1)) scratch.resize(skip ? skip : 1);
    z.next_out = scratch.data(); z.avail_out = (uInt)skip;
    while (ok && z.avail_out > 0) {
      const int r = inflate(&z, Z_NO_FLUSH);
      if (r == Z_STREAM_END) break;
      ok = r == Z_OK;
    }
    ok = ok && z.avail_out == 0;
    z.next_out = dst; z.avail_out = (uInt)want;
    while (ok && z.avail_out > 0) {
      const int r = inflate(&z, Z_NO_FLUSH);
      if (r == Z_STREAM_END) break;
      ok = r == Z_OK;
    }
    ok = ok && z.avail_out == 0;
  }
  return ok;
}

// Deflated members through the decoder of inflate_fast.h, one or TWO at a time (two streams in lock-step on one thread decode
// nearly twice as fast as one after the other: inflate_fast.h).  Per member: `skip` header bytes + `want` data bytes decoded into
// a thread-local buffer (matches may reach back into the .npy header, so both must be contiguous while decoding), the data
// copied to dst.  done[k] = false: not decoded (the caller tries zlib).  UIC_LOADER_ZLIB=1: always false.
struct FastMember { FileBytes* f; size_t off, skip, want; unsigned char* dst; };
void inflate_members_fast(FastMember* mem, int count, bool* done) {
  static const bool zlib_only = [] { const char* e = getenv("UIC_LOADER_ZLIB"); return e && *e && *e != '0'; }();
  done[0] = false;
  if (count > 1) done[1] = false;
  if (zlib_only) return;
  static thread_local std::vector<unsigned char> full[2];
  static thread_local std::unique_ptr<uic_inflate::Tables> work[2];
  uic_inflate::Stream S[2];
  for (int k = 0; k < count; ++k) {
    if (!work[k]) work[k].reset(new uic_inflate::Tables);
    const size_t n = mem[k].f->buf.size() - mem[k].off;
    mem[k].f->buf.resize(mem[k].f->buf.size() + 16, 0);  // the decoder refills its bit buffer with 8-byte loads
    if (full[k].size() < mem[k].skip + mem[k].want) full[k].resize(mem[k].skip + mem[k].want);
    S[k].start(mem[k].f->buf.data() + mem[k].off, n, full[k].data(), mem[k].skip + mem[k].want, work[k].get());
  }
  if (count == 2) {
    uic_inflate::uic_inflate_fast_pair(S[0], S[1], done);
  } else {
    done[0] = uic_inflate::uic_inflate_fast(S[0].src, S[0].n, S[0].dst, mem[0].skip + mem[0].want, work[0].get());
  }
  for (int k = 0; k < count; ++k) {
    mem[k].f->buf.resize(mem[k].f->buf.size() - 16);
    if (done[k]) memcpy(mem[k].dst, full[k].data() + mem[k].skip, mem[k].want);
  }
}

// The reader team is a process-wide pool of worker threads that lives as long as the process (round 6): a std::thread team per
// call cost ~1 ms per batch at 32 threads and made more threads a loss for stored files; with the pool a batch of DEFLATED files
// (what np.savez_compressed / make_bu_data.py writes: ~1.4 ms of inflate per 36 x 2048 file) spreads over up to 128 workers.
// One job at a time (the loader's read-ahead thread and the caller's thread serialise on job_mu); the caller works too.
class Pool {
 public:
  void run(int nt, const std::function<void(int)>& body) {
    std::lock_guard<std::mutex> job(job_mu_);
    {
      std::unique_lock<std::mutex> lk(mu_);
      while ((int)workers_.size() < nt - 1) {
        const int idx = (int)workers_.size();
        workers_.emplace_back([this, idx] { loop(idx); });
        pin(workers_.back(), idx);
        workers_.back().detach();                      // (they sleep on the condition variable until the process ends)
      }
      body_ = &body;
      want_ = nt - 1;
      active_ = nt - 1;
      ++gen_;
    }
    cv_.notify_all();
    body(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return active_ == 0; });
    body_ = nullptr;
  }

 private:
  // Every worker on a CPU of its own (round 6).  The team sleeps on one condition variable and is woken by ONE thread: the kernel's
  // wake-affine placement then queues the woken workers next to the waker and leaves it to the periodic load balancer to spread
  // them -- a 128-image batch of deflated files took 7.9 ms on a 256-thread host where one image inflates in 1.2 ms, and on 8 CPUs
  // teams of 2-4 threads ran no faster than one thread until the balancer had moved them (tools/loader_threads.py).  Worker i
  // goes to allowed CPU (first + 1 + i) mod n, first = a per-process offset so that several ranks on one host do not stack up.
  // UIC_LOADER_NO_PIN=1 in the environment leaves placement to the kernel (a host whose CPUs are shared out by other means).
  void pin(std::thread& th, int idx) {
    static const bool off = [] { const char* e = getenv("UIC_LOADER_NO_PIN"); return e && *e && *e != '0'; }();
    if (off) return;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    int cpus[CPU_SETSIZE], n = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &allowed)) cpus[n++] = c;
    if (n < 2) return;
    const int first = (int)(((unsigned)getpid() * 2654435761u) >> 8) % n;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpus[(first + 1 + idx) % n], &one);
    pthread_setaffinity_np(th.native_handle(), sizeof(one), &one);     // (best effort: a refusal leaves the thread where it is)
  }
  void loop(int idx) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* body = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (idx < want_) body = body_;
      }
      if (body) {
        (*body)(idx + 1);
        std::lock_guard<std::mutex> lk(mu_);
        if (--active_ == 0) done_.notify_all();
      }
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* body_ = nullptr;
  int want_ = 0, active_ = 0;
  unsigned long gen_ = 0;
};
Pool& pool() {
  static Pool* p = new Pool();                          // never destroyed: its detached workers may outlive static destructors
  return *p;
}

template <typename F>
int run_team(int n, int n_threads, F&& work, std::string* first_error) {
  std::atomic<int> next(0), failed(0);
  std::string errors[IO_MAX_THREADS];
  int nt = n_threads < 1 ? 1 : (n_threads > IO_MAX_THREADS ? IO_MAX_THREADS : n_threads);
  if (nt > n) nt = n;
  const std::function<void(int)> body = [&](int t) {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n || failed.load()) return;
      std::string why;
      if (!work(i, &why)) { errors[t] = why; failed.store(1); return; }
    }
  };
  if (nt <= 1) body(0);
  else pool().run(nt, body);
  if (failed.load())
    for (int t = 0; t < nt; ++t)
      if (!errors[t].empty()) { *first_error = errors[t]; return 1; }
  return 0;
}

}  // namespace

extern "C" {

int uic_loader_scan(const char* const* paths, int32_t n, const char* member, int64_t* info, int32_t n_threads) {
  UIC_REQUIRE(paths && info && n >= 1, "loader_scan: bad arguments (n=%d)", n);
  std::string err;
  const int rc = run_team(n, n_threads, [&](int i, std::string* why) {
    FileBytes f;
    if (!f.read_all(paths[i], 4096)) { *why = std::string("cannot read ") + paths[i]; return false; }
    int method; size_t off; std::string w;
    if (!locate(f.buf, member, &method, &off, &w)) { *why = std::string(paths[i]) + ": " + w; return false; }
    unsigned char head[1024];
    const unsigned char* p = f.buf.data() + off;
    size_t avail = f.buf.size() > off ? f.buf.size() - off : 0;
    if (method == 8) {
      size_t got = 0;
      if (!inflate_member(p, avail, 0, nullptr, 0, head, sizeof(head), &got)) { *why = std::string(paths[i]) + ": inflate failed"; return false; }
      p = head; avail = got;
    }
    int64_t* o = info + (size_t)i * IO_INFO;
    size_t data_off;
    if (!parse_npy(p, avail, &o[0], &o[1], &o[2], &data_off, &w)) { *why = std::string(paths[i]) + ": " + w; return false; }
    o[3] = (int64_t)data_off; o[4] = method; o[5] = (int64_t)off;
    return true;
  }, &err);
  UIC_REQUIRE(rc == 0, "loader_scan: %s", err.c_str());
  return UIC_OK;
}

int uic_loader_read(const char* const* paths, int32_t n, const int64_t* info, void* const* dst, int32_t n_threads) {
  UIC_REQUIRE(paths && info && dst && n >= 1, "loader_read: bad arguments (n=%d)", n);
  // tasks: deflated members in PAIRS (one thread decodes two streams in lock-step), first -- they are the long ones --, then every
  // stored member / plain .npy by itself
  std::vector<std::pair<int, int>> tasks;
  {
    int held = -1;
    for (int i = 0; i < n; ++i) {
      if (info[(size_t)i * IO_INFO + 4] != 8) continue;
      if (held < 0) held = i;
      else { tasks.emplace_back(held, i); held = -1; }
    }
    if (held >= 0) tasks.emplace_back(held, -1);
    for (int i = 0; i < n; ++i)
      if (info[(size_t)i * IO_INFO + 4] != 8) tasks.emplace_back(i, -1);
  }
  std::string err;
  const int rc = run_team((int)tasks.size(), n_threads, [&](int t, std::string* why) {
    const int i = tasks[t].first;
    const int64_t* o = info + (size_t)i * IO_INFO;
    const size_t want = (size_t)(o[1] * o[2]) * sizeof(float), data_off = (size_t)o[3], off = (size_t)o[5];
    if (o[4] == 8) {                                   // deflate: the whole file, inflated past the header into place
      static thread_local FileBytes fb[2];             // (kept across files: a fresh 270-KB vector per file is an mmap + page faults)
      const int idx[2] = {i, tasks[t].second};
      const int count = idx[1] >= 0 ? 2 : 1;
      FastMember mem[2];
      for (int k = 0; k < count; ++k) {
        const int64_t* ok_ = info + (size_t)idx[k] * IO_INFO;
        if (!fb[k].read_all(paths[idx[k]], 0) || fb[k].buf.size() <= (size_t)ok_[5]) { *why = std::string("cannot read ") + paths[idx[k]]; return false; }
        mem[k] = FastMember{&fb[k], (size_t)ok_[5], (size_t)ok_[3], (size_t)(ok_[1] * ok_[2]) * sizeof(float), (unsigned char*)dst[idx[k]]};
      }
      bool done[2] = {false, false};
      inflate_members_fast(mem, count, done);
      for (int k = 0; k < count; ++k) {
        if (done[k]) continue;                         // (declined: zlib decodes it, or names the file that is broken)
        if (!inflate_member(fb[k].buf.data() + mem[k].off, fb[k].buf.size() - mem[k].off, mem[k].skip, mem[k].dst, mem[k].want, nullptr, 0, nullptr)) {
          *why = std::string(paths[idx[k]]) + ": inflate failed or the member is shorter than its header says";
          return false;
        }
      }
      return true;
    }
    const int fd = open(paths[i], O_RDONLY);           // stored member / plain .npy: pread straight into the staging buffer
    if (fd < 0) { *why = std::string("cannot open ") + paths[i]; return false; }
    size_t got = 0;
    while (got < want) {
      const ssize_t r = pread(fd, (char*)dst[i] + got, want - got, (off_t)(off + data_off + got));
      if (r <= 0) break;
      got += (size_t)r;
    }
    close(fd);
    if (got != want) { *why = std::string(paths[i]) + ": file is shorter than its header says"; return false; }
    return true;
  }, &err);
  UIC_REQUIRE(rc == 0, "loader_read: %s", err.c_str());
  return UIC_OK;
}

// One raw deflate stream src[0..n) -> exactly m bytes at dst, through the loader's own decoder (fast != 0; returns 1 when it declines
// the stream) or through zlib (fast == 0).  Test hook: tests/test_dataloader_host.py compares the two on every block type.
int uic_loader_inflate(const void* src, size_t n, void* dst, size_t m, int32_t fast) {
  UIC_REQUIRE(src && (dst || m == 0), "loader_inflate: null pointer");
  if (fast) {
    std::vector<unsigned char> padded(n + 16, 0);
    memcpy(padded.data(), src, n);
    std::unique_ptr<uic_inflate::Tables> work(new uic_inflate::Tables);
    return uic_inflate::uic_inflate_fast(padded.data(), n, (unsigned char*)dst, m, work.get()) ? 0 : 1;
  }
  return inflate_member((const unsigned char*)src, n, 0, (unsigned char*)dst, m, nullptr, 0, nullptr) ? 0 : 1;
}
// Two streams decoded in lock-step by one thread, as uic_loader_read does with pairs of deflated members.  Returns a bit mask:
// bit k set = stream k was declined.
int uic_loader_inflate_pair(const void* src0, size_t n0, void* dst0, size_t m0, const void* src1, size_t n1, void* dst1, size_t m1) {
  UIC_REQUIRE(src0 && src1 && (dst0 || m0 == 0) && (dst1 || m1 == 0), "loader_inflate_pair: null pointer");
  std::vector<unsigned char> p0(n0 + 16, 0), p1(n1 + 16, 0);
  memcpy(p0.data(), src0, n0);
  memcpy(p1.data(), src1, n1);
  std::unique_ptr<uic_inflate::Tables> w0(new uic_inflate::Tables), w1(new uic_inflate::Tables);
  uic_inflate::Stream A, B;
  A.start(p0.data(), n0, (unsigned char*)dst0, m0, w0.get());
  B.start(p1.data(), n1, (unsigned char*)dst1, m1, w1.get());
  bool ok[2];
  uic_inflate::uic_inflate_fast_pair(A, B, ok);
  return (ok[0] ? 0 : 1) | (ok[1] ? 0 : 2);
}

}  // extern "C"
