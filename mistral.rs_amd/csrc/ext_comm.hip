// ext_comm.hip -- RCCL over xGMI for tensor parallelism: one process per GPU, sum all-reduce of the row-parallel partial
// outputs on the runner's own stream (graph-capturable), replacing the cudarc-NCCL calls of
// mistralrs-quant/src/distributed/mod.rs:244-303,511-809 (Comm::from_device -> ncclCommInitRank, all_reduce(Sum)).
// librccl is dlopen'd (preferring the copy the process already loaded, i.e. torch's) so single-GPU use has no RCCL dependency.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

extern "C" const char *mrs_last_error(void);
namespace mrs_host { int fail(const char *fmt, ...); }

namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef int (*get_uid_fn)(nccl_uid *);
typedef int (*init_rank_fn)(void **, int, nccl_uid, int);
typedef int (*all_reduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*destroy_fn)(void *);
typedef const char *(*errstr_fn)(int);
typedef int (*count_fn)(void *, int *);
struct Rccl { void *h = nullptr; get_uid_fn uid; init_rank_fn init; all_reduce_fn ar; destroy_fn destroy; errstr_fn err; count_fn count; };
Rccl *rccl() {
  static Rccl r;
  if (r.h) return &r;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *n : names) if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;  // already in the process (torch)
  if (!r.h) for (const char *n : names) if ((r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!r.h) { mrs_host::fail("RCCL: cannot dlopen librccl.so (%s)", dlerror()); return nullptr; }
  r.uid = (get_uid_fn)dlsym(r.h, "ncclGetUniqueId"); r.init = (init_rank_fn)dlsym(r.h, "ncclCommInitRank");
  r.ar = (all_reduce_fn)dlsym(r.h, "ncclAllReduce"); r.destroy = (destroy_fn)dlsym(r.h, "ncclCommDestroy");
  r.err = (errstr_fn)dlsym(r.h, "ncclGetErrorString"); r.count = (count_fn)dlsym(r.h, "ncclCommCount");
  if (!r.uid || !r.init || !r.ar || !r.destroy) { mrs_host::fail("RCCL: missing symbols in librccl"); r.h = nullptr; return nullptr; }
  return &r;
}
int check(Rccl *r, int rc, const char *what) { return rc == 0 ? 0 : mrs_host::fail("RCCL %s failed: %s", what, r->err ? r->err(rc) : "?"); }
}  // namespace

extern "C" int mrs_comm_unique_id(void *out128) {
  Rccl *r = rccl();
  if (!r) return -1;
  nccl_uid id;
  if (check(r, r->uid(&id), "ncclGetUniqueId")) return -1;
  memcpy(out128, &id, 128);
  return 0;
}
extern "C" void *mrs_comm_init(const void *id128, int rank, int world) {
  Rccl *r = rccl();
  if (!r) return nullptr;
  nccl_uid id;
  memcpy(&id, id128, 128);
  void *comm = nullptr;
  if (check(r, r->init(&comm, world, id, rank), "ncclCommInitRank")) return nullptr;
  return comm;
}
// in-place sum all-reduce of f32 (ncclFloat32 = 7, ncclSum = 0)
extern "C" int mrs_comm_all_reduce_sum_f32(void *comm, float *buf, size_t count, void *stream) {
  Rccl *r = rccl();
  if (!r || !comm) return mrs_host::fail("RCCL communicator not initialised");
  return check(r, r->ar(buf, buf, count, 7, 0, comm, (hipStream_t)stream), "ncclAllReduce");
}
// ranks of the communicator as RCCL sees them (ncclCommCount); -1 on error
extern "C" int mrs_comm_nranks(void *comm) {
  Rccl *r = rccl();
  int n = -1;
  if (!r || !comm || !r->count || r->count(comm, &n) != 0) return -1;
  return n;
}
extern "C" void mrs_comm_destroy(void *comm) {
  Rccl *r = rccl();
  if (r && comm) r->destroy(comm);
}
