/* oracle/hip_host/comm_shim.c -- TEST INFRASTRUCTURE ONLY.  Stand-in for ext_comm.hip (RCCL) inside the host emulation: the communicator is a
 * record of (rank, world) and the sum all-reduce is delegated to a callback the test process registers -- tests/test_distributed.py plugs in
 * torch.distributed's gloo all_reduce, so the REAL tensor-parallel code of the C++ runner (sharded weights, local head counts, the fused
 * scaled-residual all-reduce of every row-parallel projection) runs in two CPU processes.  Without a callback the all-reduce refuses (-1). */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
typedef int (*hiphost_all_reduce_cb)(float *buf, size_t count);
static hiphost_all_reduce_cb g_all_reduce = 0;
struct hiphost_comm { int rank, world; };
void hiphost_set_all_reduce(hiphost_all_reduce_cb cb) { g_all_reduce = cb; }
int mrs_comm_unique_id(void *out128) { memset(out128, 0x5a, 128); return 0; }
void *mrs_comm_init(const void *id128, int rank, int world) {
  struct hiphost_comm *c = (struct hiphost_comm *)malloc(sizeof *c);
  (void)id128;
  if (c) { c->rank = rank; c->world = world; }
  return c;
}
int mrs_comm_all_reduce_sum_f32(void *comm, float *buf, size_t count, void *stream) {
  (void)stream;
  if (!comm || !g_all_reduce) return -1;
  return g_all_reduce(buf, count);
}
int mrs_comm_nranks(void *comm) { return comm ? ((struct hiphost_comm *)comm)->world : -1; }
void mrs_comm_destroy(void *comm) { free(comm); }
