// ext_p2p.hip -- one-shot all-reduce over peer-mapped mailboxes for the decode-sized messages of tensor parallelism.
//
// Role: SumAllReduce after every row-parallel projection (mistralrs-quant/src/distributed/layers.rs:965-975 -> ncclAllReduce,
// distributed/mod.rs:584-587).  In decode the message is [b, hidden] f32 = 16-256 KiB, 2 x layers times per token: a ring all-reduce is
// latency-bound there (2 (N-1) hops), while xGMI is a full point-to-point mesh (7 links per GPU), so every rank can write its whole vector
// straight into every peer's HBM in ONE hop and sum locally:
//   post:    rank r stores its values as 8-byte granules {f32 bits, sequence number} into slot r of EVERY rank's mailbox (its own included)
//            -- one naturally aligned 8-byte store per value, so a reader can never see a value without its tag (no flag, no fence, no
//            ordering assumption about the fabric);
//   reduce:  each rank polls the `world` slots of its own mailbox until every granule carries the current sequence number and adds them in
//            rank order -- every rank computes bit-identical sums (the reference's NCCL ring does not guarantee that).
// Round 3: a call runs as up to 8 workgroups (the posting side drives the xGMI links from several CUs), neighbouring granules leave as one 16-byte
// store, a granule that never arrives turns its element into NaN, and mailboxes come from mrs_p2p_alloc_mailbox (fine-grained / uncached memory).
// Two mailbox halves alternate (sequence parity): a rank that is already posting all-reduce n+1 cannot overwrite granules a slower peer still
// reads for n, and nobody can start n+2 before every peer has posted n+1, i.e. finished reading n.
// One workgroup per call (the message is a few thousand values; latency, not bandwidth), no host work: the sequence number lives in device
// memory and is advanced by the kernel, so the call is graph-capturable like ncclAllReduce.  Messages above `max_elems` are refused (-2):
// the caller keeps RCCL for those (prefill: [T, hidden]).  Mailboxes are plain device allocations shared with hipIpcGetMemHandle /
// hipIpcOpenMemHandle (one process per GPU); a spin that sees no progress gives up and raises the error word instead of hanging.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace mrs_host { int fail(const char *fmt, ...); }

namespace mrs {
namespace p2p {

constexpr int MAX_WORLD = 16, NT = 1024, MAX_WG = 8;
struct Comm {
  int rank, world;
  unsigned long long *mail[MAX_WORLD];  // mail[r]: base of rank r's mailbox, [2 parity][world src][max_elems] granules
  unsigned *state;                      // own device memory: [0] sequence number of the last finished call, [1] error flag, [2] workgroup arrivals of the running call
  size_t max_elems;
};

#ifndef MRS_P2P_SYSTEM_IO
#define MRS_P2P_STORE(P, V) __hip_atomic_store((P), (V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define MRS_P2P_LOAD(P) __hip_atomic_load((P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#endif
// two neighbouring granules as ONE 16-byte system-scope store (round 3: half the xGMI write transactions; a granule validates itself, so the pair
// needs no atomicity beyond each 8-byte half -- the guide observes none torn, and a torn pair would still be two valid granules)
__device__ __forceinline__ void store2(unsigned long long *p, unsigned long long g0, unsigned long long g1) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 v = {g0, g1};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
  MRS_P2P_STORE(p, g0);
  MRS_P2P_STORE(p + 1, g1);
#endif
}

// the slice of the message this workgroup owns (even boundaries: pairs stay together)
__device__ __forceinline__ void wg_slice(size_t count, size_t &i0, size_t &i1) {
  const size_t per = ((count + gridDim.x - 1) / gridDim.x + 1) & ~(size_t)1;
  i0 = (size_t)blockIdx.x * per;
  i1 = i0 + per < count ? i0 + per : count;
  if (i0 > count) i0 = count;
}
constexpr unsigned long long SPIN_TICKS = 200000000ull;  // 2 s at 100 MHz: tensor-parallel ranks are separate PROCESSES -- lazy module loads, graph capture of 80 layers or a GC pause skew them
                                                         // by far more than a hop (10 ms made a late but healthy peer look dead)
__device__ __forceinline__ void post(const Comm &c, const float *buf, size_t count, unsigned seq) {
  const size_t half = (size_t)(seq & 1) * c.world * c.max_elems;
  size_t i0, i1;
  wg_slice(count, i0, i1);
  const bool pairs = (c.max_elems & 1) == 0;
  for (size_t i = i0 + 2 * (size_t)threadIdx.x; i < i1; i += 2 * NT) {
    const unsigned long long g0 = (unsigned long long)__float_as_uint(buf[i]) | ((unsigned long long)seq << 32);
    const bool two = i + 1 < i1;
    const unsigned long long g1 = two ? (unsigned long long)__float_as_uint(buf[i + 1]) | ((unsigned long long)seq << 32) : 0ull;
    for (int r = 0; r < c.world; ++r) {
      unsigned long long *p = c.mail[r] + half + (size_t)c.rank * c.max_elems + i;
      if (two && pairs) store2(p, g0, g1);
      else { MRS_P2P_STORE(p, g0); if (two) MRS_P2P_STORE(p + 1, g1); }
    }
  }
}
// returns false if a granule never arrived (bounded spin); such elements become NaN (advisor, round 2: a timed-out sum must not look like data)
__device__ __forceinline__ bool reduce(const Comm &c, float *buf, size_t count, unsigned seq) {
  const size_t half = (size_t)(seq & 1) * c.world * c.max_elems;
  size_t i0, i1;
  wg_slice(count, i0, i1);
  // state[1] != 0: a granule has timed out before (in this call or an earlier one, in any workgroup): nobody waits again -- every remaining element becomes NaN at
  // once and later calls return immediately until the host clears or drops the route (a dead peer used to cost the full bound per element)
  bool ok = *(volatile unsigned *)(c.state + 1) == 0u;
  for (size_t i = i0 + threadIdx.x; i < i1; i += NT) {
    float sum = 0.f;
    bool mine = ok;
    for (int r = 0; mine && r < c.world; ++r) {
      const unsigned long long *p = c.mail[c.rank] + half + (size_t)r * c.max_elems + i;
      unsigned long long g = MRS_P2P_LOAD(p);
      if ((unsigned)(g >> 32) != seq) {
        // bounded by TIME, not by iterations: the 100 MHz s_memrealtime clock.  The host reads the error word at its next synchronisation point, reduces it (MAX)
        // over the ranks and drops the route on ALL of them in the same step (Llama.p2p_error, bench.py)
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        do {
          if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS || ((++spins & 63u) == 0u && *(volatile unsigned *)(c.state + 1) != 0u)) { mine = false; break; }
          __builtin_amdgcn_s_sleep(1);
          g = MRS_P2P_LOAD(p);
        } while ((unsigned)(g >> 32) != seq);
        if (!mine) *(volatile unsigned *)(c.state + 1) = 1u;  // tell the other threads / workgroups now, not at the end of the call
      }
      sum += __uint_as_float((unsigned)g);  // rank order: identical bits on every rank
    }
    buf[i] = mine ? sum : __uint_as_float(0x7fc00000u);
    ok = ok && mine;
  }
  return ok;
}
// the LAST workgroup of a call advances the sequence number: every workgroup read state[0] before it arrived, so the number cannot move under a
// workgroup that has not started yet
__device__ __forceinline__ void finish(const Comm &c, unsigned seq, bool ok) {
  if (!ok) c.state[1] = 1;
  __syncthreads();  // every thread of this workgroup has read state[0] and finished its elements
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(c.state + 2, 1u) == gridDim.x - 1) { c.state[2] = 0; __threadfence(); c.state[0] = seq; }
  }
}

__global__ void __launch_bounds__(NT) all_reduce_kernel(const Comm c, float *buf, size_t count) {
  const unsigned seq = *(volatile unsigned *)c.state + 1;
  post(c, buf, count, seq);
  finish(c, seq, reduce(c, buf, count, seq));
}
// the two halves as separate launches (tests on a sequential schedule: post on every rank, then reduce on every rank)
__global__ void __launch_bounds__(NT) post_kernel(const Comm c, const float *buf, size_t count) { post(c, buf, count, *(volatile unsigned *)c.state + 1); }
__global__ void __launch_bounds__(NT) reduce_kernel(const Comm c, float *buf, size_t count) {
  const unsigned seq = *(volatile unsigned *)c.state + 1;
  finish(c, seq, reduce(c, buf, count, seq));
}

// several ranks of one address space in ONE grid (single-GPU test of the real polling path: the workgroups of a grid are co-resident, kernels on
// different streams of one device need not be): blockIdx.y = rank, blockIdx.x = the rank's workgroups
struct Group { Comm c[MAX_WORLD]; float *buf[MAX_WORLD]; };
__global__ void __launch_bounds__(NT) all_reduce_group_kernel(const Group g, size_t count) {
  const Comm &c = g.c[blockIdx.y];
  float *buf = g.buf[blockIdx.y];
  const unsigned seq = *(volatile unsigned *)c.state + 1;
  post(c, buf, count, seq);
  finish(c, seq, reduce(c, buf, count, seq));
}
// workgroups of one call: ~2048 values each, at most MAX_WG (the posting side drives the 7 xGMI links from several CUs at once)
static inline int wgs_for(size_t count) { const size_t w = (count + 2047) / 2048; return (int)(w < 1 ? 1 : (w > MAX_WG ? MAX_WG : w)); }

}  // namespace p2p
}  // namespace mrs

using mrs::p2p::Comm;

extern "C" size_t mrs_p2p_mailbox_bytes(int world, size_t max_elems) { return world > 0 ? (size_t)2 * world * max_elems * 8 + 256 : 0; }
// A mailbox must be FINE-GRAINED / uncached device memory: peers write into it over xGMI while the owner's kernel polls it, and HIP only guarantees
// in-kernel visibility of peer writes for such allocations (advisor, round 2: a torch.zeros mailbox is coarse-grained; the local L2 may keep a stale line
// of a half that was read two calls earlier).  Zeroed; free with mrs_p2p_free_mailbox.
extern "C" void *mrs_p2p_alloc_mailbox(size_t bytes) {
  void *p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess || !p) {
    p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) { mrs_host::fail("mrs_p2p_alloc_mailbox: no fine-grained device memory (%zu bytes)", bytes); return nullptr; }
  }
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); mrs_host::fail("mrs_p2p_alloc_mailbox: memset failed"); return nullptr; }
  return p;
}
extern "C" void mrs_p2p_free_mailbox(void *p) { if (p) (void)hipFree(p); }
// hipIpcGetMemHandle / hipIpcOpenMemHandle of a mailbox (64-byte handles, exchanged by the host framework); HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack
extern "C" int mrs_ipc_get_handle(void *dev_ptr, void *out64) {
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, dev_ptr) != hipSuccess) return mrs_host::fail("hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
  memcpy(out64, &h, sizeof h);
  return 0;
}
extern "C" void *mrs_ipc_open_handle(const void *in64) {
  hipIpcMemHandle_t h;
  memcpy(&h, in64, sizeof h);
  void *p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { mrs_host::fail("hipIpcOpenMemHandle failed"); return nullptr; }
  return p;
}
extern "C" int mrs_ipc_close_handle(void *p) { return hipIpcCloseMemHandle(p) == hipSuccess ? 0 : -1; }

// mailboxes[r] = rank r's mailbox as addressable from THIS process (own allocation for r == rank, an opened IPC handle otherwise), each
// mrs_p2p_mailbox_bytes(world, max_elems) bytes and zeroed before the first call on any rank
extern "C" void *mrs_p2p_create(int rank, int world, void *const *mailboxes, size_t max_elems) {
  if (world < 1 || world > mrs::p2p::MAX_WORLD || rank < 0 || rank >= world || !mailboxes || !max_elems) { mrs_host::fail("mrs_p2p_create: bad arguments"); return nullptr; }
  Comm *c = (Comm *)calloc(1, sizeof(Comm));
  if (!c) return nullptr;
  c->rank = rank; c->world = world; c->max_elems = max_elems;
  for (int r = 0; r < world; ++r) {
    if (!mailboxes[r]) { free(c); mrs_host::fail("mrs_p2p_create: mailbox %d missing", r); return nullptr; }
    c->mail[r] = (unsigned long long *)mailboxes[r];
  }
  c->state = (unsigned *)((char *)mailboxes[rank] + (size_t)2 * world * max_elems * 8);  // the tail of the own mailbox
  return c;
}
extern "C" void mrs_p2p_destroy(void *comm) { free(comm); }
extern "C" size_t mrs_p2p_max_elems(void *comm) { return comm ? ((Comm *)comm)->max_elems : 0; }
// in-place sum all-reduce of buf[count] f32 on `stream`; -2: message larger than the mailboxes (use RCCL)
extern "C" int mrs_p2p_all_reduce_sum_f32(void *comm, float *buf, size_t count, void *stream) {
  if (!comm) return mrs_host::fail("p2p all-reduce: no communicator");
  const Comm &c = *(Comm *)comm;
  if (count > c.max_elems) return -2;
  if (!count) return 0;
  hipLaunchKernelGGL(mrs::p2p::all_reduce_kernel, dim3(mrs::p2p::wgs_for(count)), dim3(mrs::p2p::NT), 0, (hipStream_t)stream, c, buf, count);
  return 0;
}
extern "C" int mrs_p2p_post(void *comm, const float *buf, size_t count, void *stream) {
  if (!comm || count > ((Comm *)comm)->max_elems) return -2;
  hipLaunchKernelGGL(mrs::p2p::post_kernel, dim3(mrs::p2p::wgs_for(count)), dim3(mrs::p2p::NT), 0, (hipStream_t)stream, *(Comm *)comm, buf, count);
  return 0;
}
extern "C" int mrs_p2p_reduce(void *comm, float *buf, size_t count, void *stream) {
  if (!comm || count > ((Comm *)comm)->max_elems) return -2;
  hipLaunchKernelGGL(mrs::p2p::reduce_kernel, dim3(mrs::p2p::wgs_for(count)), dim3(mrs::p2p::NT), 0, (hipStream_t)stream, *(Comm *)comm, buf, count);
  return 0;
}
// error word of the last calls (1: a granule never arrived -- a peer is not running the same sequence of all-reduces); blocking read
extern "C" int mrs_p2p_error(void *comm) {
  if (!comm) return -1;
  unsigned st[2] = {0, 0};
  if (hipMemcpy(st, ((Comm *)comm)->state, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)st[1];
}
// test entry: n communicators of one address space run the one-kernel all-reduce concurrently as the n workgroups of a single grid
extern "C" int mrs_p2p_all_reduce_group(void *const *comms, float *const *bufs, int n, size_t count, void *stream) {
  if (!comms || !bufs || n < 1 || n > mrs::p2p::MAX_WORLD) return -1;
  mrs::p2p::Group g{};
  for (int i = 0; i < n; ++i) {
    if (!comms[i] || count > ((Comm *)comms[i])->max_elems) return -2;
    g.c[i] = *(Comm *)comms[i];
    g.buf[i] = bufs[i];
  }
  hipLaunchKernelGGL(mrs::p2p::all_reduce_group_kernel, dim3(mrs::p2p::wgs_for(count), n), dim3(mrs::p2p::NT), 0, (hipStream_t)stream, g, count);
  return 0;
}
