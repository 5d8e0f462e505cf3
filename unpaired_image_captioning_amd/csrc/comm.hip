// uic_comm_*: the data-parallel gradient exchange of the hot path (what torch.nn.DataParallel's reduce-add does at
// P/trainer.py:74,88-89) as plain C-ABI calls on RCCL, for callers that do not go through torch.distributed.  One
// communicator per process = per GPU; the sum runs in place on the caller's stream (enqueue only).  librccl is loaded with
// dlopen at the first call, so libuic_hip.so itself keeps no link-time dependency on it (and a process that already has a
// librccl loaded -- PyTorch's -- shares that copy).
#include "uic_common.h"
#include "../../include/uic_hip.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

struct UniqueId { char bytes[UIC_COMM_ID_BYTES]; };   // ncclUniqueId: 128 opaque bytes, passed by value

typedef int (*get_unique_id_fn)(UniqueId*);
typedef int (*comm_init_rank_fn)(void**, int, UniqueId, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*reduce_scatter_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*group_fn)(void);
typedef int (*comm_destroy_fn)(void*);
typedef const char* (*get_error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  all_reduce_fn all_reduce = nullptr;
  reduce_scatter_fn reduce_scatter = nullptr;
  all_gather_fn all_gather = nullptr;
  group_fn group_start = nullptr, group_end = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  get_error_string_fn get_error_string = nullptr;
  bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl(Rccl** out) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.tried) {
    g_rccl.tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.handle) break;
    }
    if (g_rccl.handle) {
      g_rccl.get_unique_id = (get_unique_id_fn)dlsym(g_rccl.handle, "ncclGetUniqueId");
      g_rccl.comm_init_rank = (comm_init_rank_fn)dlsym(g_rccl.handle, "ncclCommInitRank");
      g_rccl.all_reduce = (all_reduce_fn)dlsym(g_rccl.handle, "ncclAllReduce");
      g_rccl.reduce_scatter = (reduce_scatter_fn)dlsym(g_rccl.handle, "ncclReduceScatter");
      g_rccl.all_gather = (all_gather_fn)dlsym(g_rccl.handle, "ncclAllGather");
      g_rccl.group_start = (group_fn)dlsym(g_rccl.handle, "ncclGroupStart");
      g_rccl.group_end = (group_fn)dlsym(g_rccl.handle, "ncclGroupEnd");
      g_rccl.comm_destroy = (comm_destroy_fn)dlsym(g_rccl.handle, "ncclCommDestroy");
      g_rccl.get_error_string = (get_error_string_fn)dlsym(g_rccl.handle, "ncclGetErrorString");
    }
  }
  UIC_REQUIRE(g_rccl.handle && g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy &&
                  g_rccl.reduce_scatter && g_rccl.all_gather && g_rccl.group_start && g_rccl.group_end,
              "uic_comm: librccl.so could not be loaded (%s)", g_rccl.handle ? "symbols missing" : dlerror());
  *out = &g_rccl;
  return UIC_OK;
}

int check_rccl(Rccl* r, int rc, const char* what) {
  if (rc == 0) return UIC_OK;
  uic_set_error("%s failed: %s (ncclResult %d)", what, r->get_error_string ? r->get_error_string(rc) : "?", rc);
  return 1000 + rc;
}

}  // namespace

extern "C" {

int uic_comm_unique_id(void* id_out) {
  UIC_REQUIRE(id_out, "uic_comm_unique_id: null pointer");
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  UniqueId id;
  memset(&id, 0, sizeof(id));
  UIC_TRY(check_rccl(r, r->get_unique_id(&id), "ncclGetUniqueId"));
  memcpy(id_out, id.bytes, UIC_COMM_ID_BYTES);
  return UIC_OK;
}

int uic_comm_init(int32_t rank, int32_t world, const void* id, void** comm_out) {
  UIC_REQUIRE(id && comm_out, "uic_comm_init: null pointer");
  UIC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "uic_comm_init: rank %d outside [0, %d)", rank, world);
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  UniqueId uid;
  memcpy(uid.bytes, id, UIC_COMM_ID_BYTES);
  void* comm = nullptr;
  UIC_TRY(check_rccl(r, r->comm_init_rank(&comm, world, uid, rank), "ncclCommInitRank"));
  *comm_out = comm;
  return UIC_OK;
}

int uic_comm_allreduce(void* comm, void* buf, size_t count, int32_t dtype, void* stream) {
  UIC_REQUIRE(comm && (buf || count == 0), "uic_comm_allreduce: null pointer");
  UIC_REQUIRE(dtype == UIC_F32 || dtype == UIC_BF16, "uic_comm_allreduce: bad dtype %d", dtype);
  if (count == 0) return UIC_OK;
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  const int nccl_float = 7, nccl_bfloat16 = 9, nccl_sum = 0;      // ncclDataType_t / ncclRedOp_t values of rccl.h
  return check_rccl(r, r->all_reduce(buf, buf, count, dtype == UIC_F32 ? nccl_float : nccl_bfloat16, nccl_sum, comm, (hipStream_t)stream),
                    "ncclAllReduce");
}

static int nccl_type(int32_t dtype) { return dtype == UIC_F32 ? 7 : 9; }      // ncclFloat32 / ncclBfloat16 of rccl.h

int uic_comm_reduce_scatter(void* comm, const void* sendbuf, void* recvbuf, size_t recvcount, int32_t dtype, void* stream) {
  UIC_REQUIRE(comm && ((sendbuf && recvbuf) || recvcount == 0), "uic_comm_reduce_scatter: null pointer");
  UIC_REQUIRE(dtype == UIC_F32 || dtype == UIC_BF16, "uic_comm_reduce_scatter: bad dtype %d", dtype);
  if (recvcount == 0) return UIC_OK;
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  return check_rccl(r, r->reduce_scatter(sendbuf, recvbuf, recvcount, nccl_type(dtype), 0 /* ncclSum */, comm, (hipStream_t)stream), "ncclReduceScatter");
}

int uic_comm_allgather(void* comm, const void* sendbuf, void* recvbuf, size_t sendcount, int32_t dtype, void* stream) {
  UIC_REQUIRE(comm && ((sendbuf && recvbuf) || sendcount == 0), "uic_comm_allgather: null pointer");
  UIC_REQUIRE(dtype == UIC_F32 || dtype == UIC_BF16, "uic_comm_allgather: bad dtype %d", dtype);
  if (sendcount == 0) return UIC_OK;
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  return check_rccl(r, r->all_gather(sendbuf, recvbuf, sendcount, nccl_type(dtype), comm, (hipStream_t)stream), "ncclAllGather");
}

int uic_comm_group_start(void) {
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  return check_rccl(r, r->group_start(), "ncclGroupStart");
}
int uic_comm_group_end(void) {
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  return check_rccl(r, r->group_end(), "ncclGroupEnd");
}

int uic_comm_destroy(void* comm) {
  if (!comm) return UIC_OK;
  Rccl* r = nullptr;
  UIC_TRY(load_rccl(&r));
  return check_rccl(r, r->comm_destroy(comm), "ncclCommDestroy");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// A stand-in for one all-reduce of `bytes` bytes on a box with ONE GPU (tools/comm_proxy.py): RCCL's all-reduce kernels are a
// co-resident load of a few workgroups that move every byte out and back over xGMI, so the stand-in is `workgroups` (8-32)
// 256-thread workgroups that stream buf -> scratch -> buf (two passes over the bytes, nothing changes).  16 workgroups move
// ~0.38 GB/ms, the order of an 8-GPU ring's bus bandwidth.  A measurement aid, not a collective.
namespace {
typedef __attribute__((ext_vector_type(4))) unsigned u32x4c;
__global__ __launch_bounds__(256) void comm_proxy_kernel(u32x4c* buf, u32x4c* scratch, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) scratch[i] = __builtin_nontemporal_load(buf + i);
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) buf[i] = __builtin_nontemporal_load(scratch + i);
}
__global__ __launch_bounds__(256) void comm_proxy_oneway_kernel(const u32x4c* src, u32x4c* dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}
}  // namespace
extern "C" int uic_comm_proxy(void* buf, void* scratch, size_t bytes, int32_t workgroups, void* stream) {
  UIC_REQUIRE(buf && scratch && bytes % 16 == 0 && workgroups >= 1 && workgroups <= 1024, "comm_proxy: bad arguments");
  if (bytes == 0) return UIC_OK;
  hipLaunchKernelGGL(comm_proxy_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (u32x4c*)buf, (u32x4c*)scratch, bytes / 16);
  UIC_LAUNCH_CHECK("comm_proxy_kernel");
  return UIC_OK;
}
// The stand-in for a reduce-scatter or an all-gather of `bytes` bytes: ONE pass over the bytes (an all-reduce moves every byte
// out and back, each of its two halves moves it once), src -> dst at the same per-workgroup rate as uic_comm_proxy.
extern "C" int uic_comm_proxy_oneway(const void* src, void* dst, size_t bytes, int32_t workgroups, void* stream) {
  UIC_REQUIRE(src && dst && bytes % 16 == 0 && workgroups >= 1 && workgroups <= 1024, "comm_proxy_oneway: bad arguments");
  if (bytes == 0) return UIC_OK;
  hipLaunchKernelGGL(comm_proxy_oneway_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4c*)src, (u32x4c*)dst, bytes / 16);
  UIC_LAUNCH_CHECK("comm_proxy_oneway_kernel");
  return UIC_OK;
}
