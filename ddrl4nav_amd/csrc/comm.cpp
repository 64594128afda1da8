// RCCL binding of the one collective on the data path: a SUM all-reduce of the flat fp32 gradient
// arena (+ DDRL_STATS_FLOATS loss tail) per PPO iteration, between ddrl_ppo_iter and ddrl_clip_adam_step.  The reference has
// no multi-GPU path (`# TODO support mutil GPU CARD`, USTC_lab/server/backward.py:167).
//
// librccl is resolved at run time (dlopen): libddrl_hip.so carries no link dependency on it, a single-GPU host without RCCL
// still loads the library, and inside a PyTorch process the RCCL that torch already loaded is the one that gets used.
//
// Pure host code (no HIP header: the stream crosses this file as the opaque pointer it is), so that `make asan` can build it with
// g++ -fsanitize=address,undefined next to easybytes.cpp.
#include <dlfcn.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/ddrl.h"

namespace {

// the subset of the RCCL / NCCL ABI this file needs (rccl.h: ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0)
struct nccl_unique_id {
  char internal[128];
};
typedef void* nccl_comm_t;
typedef void* hipStream_t;  // ihipStream_t* in the HIP headers: opaque here
typedef int (*fn_get_unique_id)(nccl_unique_id*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, nccl_unique_id, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_get_version)(int*);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_broadcast broadcast = nullptr;
  fn_get_version get_version = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {  // an already loaded copy (e.g. PyTorch's) first
      r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (r.handle) break;
    }
    for (int i = 0; !r.handle && i < 3; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) return;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
    r.broadcast = (fn_broadcast)dlsym(r.handle, "ncclBroadcast");
    r.get_version = (fn_get_version)dlsym(r.handle, "ncclGetVersion");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.broadcast;
  });
  return r;
}

}  // namespace

struct ddrl_comm {
  nccl_comm_t comm;
  int32_t rank, world;
};

extern "C" {

// First-contact diagnostics (bench.py --preflight): WHICH librccl this library resolved (the file the ncclAllReduce symbol lives in: inside
// a PyTorch process that is torch's bundled copy, not /opt/rocm's) and its version code (ncclGetVersion: major * 10000 + minor * 100 +
// patch).  No communicator and no GPU needed.  DDRL_ERR_UNSUPPORTED when no librccl could be loaded.
int32_t ddrl_comm_info(char* path_out, int64_t cap, int32_t* version_out) {
  if ((cap > 0 && !path_out) || cap < 0) return DDRL_ERR_INVALID_ARG;
  if (cap > 0) path_out[0] = 0;
  if (version_out) *version_out = 0;
  Rccl& r = rccl();
  if (!r.ok) return DDRL_ERR_UNSUPPORTED;
  Dl_info info;
  if (cap > 0 && dladdr((void*)r.all_reduce, &info) != 0 && info.dli_fname) {
    std::strncpy(path_out, info.dli_fname, (size_t)cap - 1);
    path_out[cap - 1] = 0;
  }
  if (version_out && r.get_version) {
    int v = 0;
    if (r.get_version(&v) == 0) *version_out = (int32_t)v;
  }
  return DDRL_OK;
}

int32_t ddrl_comm_unique_id(uint8_t* out128) {
  if (!out128) return DDRL_ERR_INVALID_ARG;
  Rccl& r = rccl();
  if (!r.ok) return DDRL_ERR_UNSUPPORTED;
  nccl_unique_id id;
  if (r.get_unique_id(&id) != 0) return DDRL_ERR_HIP;
  std::memcpy(out128, id.internal, 128);
  return DDRL_OK;
}

int32_t ddrl_comm_create(const uint8_t* id128, int32_t rank, int32_t world, ddrl_comm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return DDRL_ERR_INVALID_ARG;
  Rccl& r = rccl();
  if (!r.ok) return DDRL_ERR_UNSUPPORTED;
  ddrl_comm* c = new (std::nothrow) ddrl_comm();
  if (!c) return DDRL_ERR_NO_MEMORY;
  nccl_unique_id id;
  std::memcpy(id.internal, id128, 128);
  if (r.comm_init_rank(&c->comm, world, id, rank) != 0) {
    delete c;
    return DDRL_ERR_HIP;
  }
  c->rank = rank;
  c->world = world;
  *out = c;
  return DDRL_OK;
}

int32_t ddrl_comm_destroy(ddrl_comm* c) {
  if (!c) return DDRL_ERR_INVALID_ARG;
  rccl().comm_destroy(c->comm);
  delete c;
  return DDRL_OK;
}

int32_t ddrl_allreduce_f32(ddrl_comm* c, float* buf, int64_t count, void* stream) {
  if (!c || !buf || count < 1) return DDRL_ERR_INVALID_ARG;
  return rccl().all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, (hipStream_t)stream) == 0
             ? DDRL_OK
             : DDRL_ERR_HIP;
}

int32_t ddrl_broadcast_f32(ddrl_comm* c, float* buf, int64_t count, int32_t root, void* stream) {
  if (!c || !buf || count < 1 || root < 0 || root >= c->world) return DDRL_ERR_INVALID_ARG;
  return rccl().broadcast(buf, buf, (size_t)count, 7, root, c->comm, (hipStream_t)stream) == 0 ? DDRL_OK : DDRL_ERR_HIP;
}

}  // extern "C"
