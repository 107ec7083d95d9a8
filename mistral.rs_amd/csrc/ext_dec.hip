// ext_dec.hip -- MI355X decode engine: the GEMV phases of a decode step on the core of dec_core.cuh, the decode-layout repack, and the
// C entry points (include/mrs_hip_ext.h, mrs_dec_*).
//
// A decode step of one Llama layer is five phases (reference call sequence: mistralrs-core/src/models/llama.rs:68-157, 243-260):
//   qkv      RMSNorm(h) -> Q8_K/Q8_0 activations in LDS -> q, k, v GEMV rows -> RoPE -> q (f32), k / v into the paged cache
//   attn     decode attention over the cache (paged_attention.cuh) -> f32
//   o_proj   quantize(attn) -> GEMV -> h = h * s + W_o . attn
//   gate/up  RMSNorm(h) -> quantize -> gate and up rows -> act(gate) * up (f32)
//   down     quantize(act) -> GEMV -> h = h * s + W_d . act
// Every phase is the same kernel body: fill the weight ring, run the activation prologue (norm + quantize: ONE pass over <= 57 KB of f32),
// stream the wave's rows, apply the phase's epilogue per finished row.  Arithmetic: header of dec_core.cuh.
#include "dec_attn.cuh"
#include "dec_core.cuh"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include "../../include/mistralrs_paged_attn.h"

namespace mrs {
namespace dec {

#ifndef MRS_WAIT_VMCNT0
#define MRS_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

enum : int { EPI_STORE = 0, EPI_RESID = 1, EPI_GLU = 2, EPI_QKV = 3, EPI_RESID2 = 4 };

struct GemvArgs {
  Mat m[3];
  int nrows[3];  // logical rows of the phase per tensor (== m[i].n except for stacked experts)
  int K;
  const float *x; int ldx; const float *norm_w; float eps;
  float *out; int out_stride; float resid_scale;
  int activation;
  float *q_out; void *k_cache, *v_cache; const int64_t *slot_mapping; const int32_t *positions; const float *cos_t, *sin_t;
  int head_dim, rot_pairs, num_kv_heads, block_size, cache_x, kv_f16;
  int units, units_per_wave;
  int neox;     // EPI_QKV: rows of q / k are stored in PAIR order (original rows i, i + head_dim / 2 of a head adjacent): rotate-half RoPE; results go back to i, i + head_dim / 2
  int upw3[3];  // EPI_QKV: RoPE pairs per wave of q, k, v (the byte-heaviest tensors get the shorter runs; 0 = units_per_wave)
  int wstart[4];  // EPI_QKV: first wave of q, k, v and the total
  const int32_t *expert_sel;  // stacked experts [E * nrows][K]: rows of expert e start at e * nrows (nullptr = dense)
  const float *acc_scale;     // RESID: out = out * resid_scale + (*acc_scale) * W.x  (routing weight)
  int ablate;                 // experiments (MRS_DEC_ABLATE): 1 = skip the activation prologue's arithmetic, 2 = skip the accumulate (loads only)
  int slots, slot_out_stride;  // GLU with several experts of ONE token in a launch (MoE top-k): unit u -> slot u / nrows[0], expert expert_sel[slot], output out + slot * slot_out_stride
  const void *x_img;          // activations already quantized by the producer (decode_attn_fused_kernel): the LDS image of NCOLS columns, byte for byte
  int staged;                 // 1: the one-column prologue runs in stages between the ring loads (dec_core.cuh ActStager); set by the launcher
  unsigned long long *tl;     // experiment builds (-DMRS_DEC_TIMELINE): 8 s_memrealtime stamps per workgroup of this launch (scripts/exp/timeline.py)
};

// -DMRS_DEC_TIMELINE: lane 0 of wave `w` writes the 100 MHz constant clock into slot i of its workgroup's record
#ifdef MRS_DEC_TIMELINE
#define MRS_TL(a, w, i) do { if ((a).tl && tid_opaque() == (w) * 64) (a).tl[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MRS_TLW(a, i) do { if ((a).tl && (tid_opaque() & 63) == 0) (a).tl[blockIdx.x * 32 + (i) + (tid_opaque() >> 6)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MRS_TL(a, w, i) do { } while (0)
#define MRS_TLW(a, i) do { } while (0)
#endif

#define MRS_DEC_TYPE_SWITCH(t, ...)                              \
  switch (t) {                                                   \
  case T_Q4_K: { constexpr int TT = T_Q4_K; __VA_ARGS__ } break; \
  case T_Q5_K: { constexpr int TT = T_Q5_K; __VA_ARGS__ } break; \
  case T_Q6_K: { constexpr int TT = T_Q6_K; __VA_ARGS__ } break; \
  case T_Q8_0: { constexpr int TT = T_Q8_0; __VA_ARGS__ } break; \
  default: break;                                                \
  }

// one stream() call of gemv_phase: the staged prologue (one column, launch-per-phase, f32 activations) or the plain one
#define MRS_DEC_STREAM(TYPE_EXPR, NC, SEGCOL, sg, epi, skip)                                                                         \
  MRS_DEC_TYPE_SWITCH(TYPE_EXPR, {                                                                                                   \
    ActStager<Tile<TT>::DEPTH> stg_{smem, red, a.x, a.norm_w, a.eps, K, act_mode_for(TT), a.tl ? a.tl + blockIdx.x * 32 : nullptr, can_stage, 1.0f, 1.0f};                          \
    auto pro2 = [&](const AP &p_) -> Act {                                                                                       \
      if (!stg_.staged) return pro(p_);                                                                                              \
      MRS_TLW(a, 1);                                                                                                                 \
      const Act r_ = stg_.finish(p_);                                                                                                \
      MRS_TL(a, 0, 10);                                                                                                              \
      return r_;                                                                                                                     \
    };                                                                                                                               \
    stream<TT, NC, SEGCOL>(sg, K, pre, pro2, epi, skip, &stg_);                                                                      \
  })

// MRS_DEC_AGENT_IO (build experiment, off): in the persistent step, write the vectors handed to other CUs through at agent scope (sc1) and read them at
// agent scope, with MRS_DEC_NOFENCE dropping the release / acquire fences of the phase barrier.  Measured: no faster, attention not covered.
#ifndef MRS_DEC_AGENT_IO
#define MRS_DEC_AGENT_IO 0
#endif
template <bool AGENT> __device__ __forceinline__ void st_out(float *p, float v) {
  if constexpr (AGENT && MRS_DEC_AGENT_IO) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool AGENT> __device__ __forceinline__ float ld_out(const float *p) {
  if constexpr (AGENT && MRS_DEC_AGENT_IO) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
__device__ __forceinline__ float rl(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

// LATE / sync: the persistent step kernel cannot read the activations before the producer phase has finished on every CU: the ring is filled
// first, then sync() waits for the grid, then the activations are loaded at agent scope and quantized.  Launch-per-phase: LATE = false, the
// activation loads go out before the ring (they are at the head of the wave's in-order load queue) and sync() is empty.
struct NoSync { __device__ __forceinline__ void operator()() const {} };
template <int NCOLS, int EPI, bool LATE = false, class Sync = NoSync, class AP = ActPre>
__device__ __forceinline__ void gemv_phase(const GemvArgs &a, char *smem, float *red, Sync sync = Sync()) {
  const int tid0 = tid_opaque();
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), lane = tid0 & 63;  // readfirstlane: lets hipcc keep everything derived from the wave index in SGPRs
  const int gw = blockIdx.x * NW + wave;
  const int K = a.K;
  MRS_TL(a, 0, 0);
  // the activation image is staged once per workgroup
  auto pre = [&]() -> AP {
    if constexpr (LATE) return AP{};
    // a pre-quantized image goes through the same registers: its 16-byte pieces sit at tid * 16 + j * NT * 16, like the f32 vector's
    else return act_issue<false, AP>(a.x_img ? (const float *)a.x_img : a.x, a.x_img ? nullptr : a.norm_w, a.x_img ? (int)(act_bytes(K, NCOLS) / 4) : K);
  };
  auto pro = [&](const AP &p) -> Act {
    if constexpr (LATE) {
      sync();
      const AP late = act_issue<MRS_DEC_AGENT_IO != 0, AP>(a.x, a.norm_w, K);
      return act_finish<NCOLS, MRS_DEC_AGENT_IO != 0, AP>(smem, red, late, a.x, a.ldx, a.norm_w, a.eps, K, act_mode_for(a.m[0].type));
    } else {
      if (a.x_img) {
#pragma unroll
        for (int j = 0; j < AP::NV; ++j)
          if ((size_t)(tid0 * 16 + j * NT * 16) < act_bytes(K, NCOLS)) *(v4u *)(smem + tid0 * 16 + j * NT * 16) = p.xv[j];
        __syncthreads();
        return Act{smem, (const float *)(smem + (size_t)NCOLS * K), (const int *)(smem + (size_t)NCOLS * K + (size_t)NCOLS * (K / 32) * 4), K};
      }
      if (a.ablate & 1) { __syncthreads(); return Act{smem, (const float *)(smem + (size_t)NCOLS * K), (const int *)(smem + (size_t)NCOLS * K + (size_t)NCOLS * (K / 32) * 4), K}; }
      MRS_TLW(a, 1);  // ring issued (per wave: slots 1..8)
      const Act r = act_finish<NCOLS, false, AP>(smem, red, p, a.x, a.ldx, a.norm_w, a.eps, K, act_mode_for(a.m[0].type), a.tl ? a.tl + blockIdx.x * 32 : nullptr);
      MRS_TL(a, 0, 10);  // prologue done (after its last barrier)
      return r;
    }
  };
  const bool can_stage = NCOLS == 1 && !LATE && a.staged != 0 && !a.x_img && !(a.ablate & 1);
  int u0 = min(gw * a.units_per_wave, a.units), u1 = min(u0 + a.units_per_wave, a.units);
  int slot = 0;
  if (a.slots > 1) {  // the launcher made units_per_wave a divisor of nrows: a wave never straddles two experts
    slot = min(u0 / a.nrows[0], a.slots - 1);  // a tail wave without units (u0 == units) must not index expert_sel[slots] (advisor, round 2)
    u0 -= slot * a.nrows[0]; u1 -= slot * a.nrows[0];
  }
  const int eoff = a.expert_sel ? a.expert_sel[slot] * a.nrows[0] : 0;

  if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
    Segs sg{};
    sg.nseg = 1; sg.mat[0] = a.m[0]; sg.row0[0] = eoff + u0; sg.nrows[0] = u1 - u0;
    const float ascale = a.acc_scale ? *a.acc_scale : 1.0f;
    float hold[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) hold[c] = 0.0f;
    if constexpr (EPI == EPI_RESID) {  // residual values up front (lane i <-> the wave's row i): no dependent load between a row's sum and its store
      if (lane < u1 - u0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) hold[c] = ld_out<LATE>(a.out + (size_t)c * a.out_stride + u0 + lane);
      }
    }
    auto epi = [&](int, int row, const float(&sum)[NCOLS]) {
      const int r = row - eoff;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        float *o = a.out + (size_t)c * a.out_stride + r;
        if constexpr (EPI == EPI_RESID) {
          const float old = rl(hold[c], r - u0);
          if (lane == 0) st_out<LATE>(o, old * a.resid_scale + sum[c] * ascale);
        } else {
          if (lane == 0) st_out<LATE>(o, sum[c]);
        }
      }
    };
    MRS_DEC_STREAM(a.m[0].type, NCOLS, false, sg, epi, (a.ablate & 2) != 0)
  } else if constexpr (EPI == EPI_RESID2) {
    // MoE down of the two experts of one token in one launch (NCOLS == 2 = the two experts' activation vectors): a wave streams rows [u0, u1) of expert
    // sel[0] against column 0, then the same rows of expert sel[1] against column 1, and writes (out * resid_scale + w0 s0) * 1 + w1 s1 -- the two
    // roundings of two consecutive EPI_RESID launches, bit for bit
    static_assert(EPI != EPI_RESID2 || NCOLS == 2, "two experts = two activation columns");
    Segs sg{};
    sg.nseg = 2; sg.mat[0] = sg.mat[1] = a.m[0];
    sg.row0[0] = a.expert_sel[0] * a.nrows[0] + u0; sg.row0[1] = a.expert_sel[1] * a.nrows[0] + u0; sg.nrows[0] = sg.nrows[1] = u1 - u0;
    const float w0 = a.acc_scale[0], w1 = a.acc_scale[1];
    float hold = 0.0f, s0save = 0.0f;
    if (lane < u1 - u0) hold = ld_out<LATE>(a.out + u0 + lane);
    auto epi = [&](int seg, int row, const float(&sum)[1]) {
      const int i = row - (seg == 0 ? sg.row0[0] : sg.row0[1]);
      if (seg == 0) {
        s0save = lane == i ? sum[0] : s0save;
      } else {
        const float s0 = rl(s0save, i), old = rl(hold, i);
        const float h1 = old * a.resid_scale + s0 * w0;
        if (lane == 0) st_out<LATE>(a.out + u0 + i, h1 * 1.0f + sum[0] * w1);
      }
    };
    MRS_DEC_STREAM(a.m[0].type, 1, true, sg, epi, false)
  } else if constexpr (EPI == EPI_GLU) {
    Segs sg{};
    sg.nseg = 2; sg.mat[0] = a.m[0]; sg.mat[1] = a.m[1];
    sg.row0[0] = sg.row0[1] = eoff + u0; sg.nrows[0] = sg.nrows[1] = u1 - u0;
    float gsave[NCOLS];  // lane i keeps gate row i of the wave until the matching up row arrives
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) gsave[c] = 0.0f;
    auto epi = [&](int seg, int row, const float(&sum)[NCOLS]) {
      const int i = row - eoff - u0;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        if (seg == 0) {
          gsave[c] = lane == i ? sum[c] : gsave[c];
        } else {
          const float g = rl(gsave[c], i);
          if (lane == 0) st_out<LATE>(a.out + (size_t)slot * a.slot_out_stride + (size_t)c * a.out_stride + (row - eoff), (a.activation == 0 ? silu_engine(g) : glu_act(g, a.activation)) * sum[c]);
        }
      }
    };
    MRS_DEC_STREAM(a.m[0].type, NCOLS, false, sg, epi, (a.ablate & 2) != 0)
  } else {  // EPI_QKV: units are RoPE pairs (2i, 2i+1); waves [wstart[i], wstart[i+1]) take tensor i (q, k, v), so a wave never straddles two tensors
    // (selects, not a.m[mi]: in the persistent kernel the arguments are a local struct and a dynamic index would push it into scratch memory)
    const int mi = gw >= a.wstart[2] ? 2 : (gw >= a.wstart[1] ? 1 : 0);
    const int npairs = (mi == 0 ? a.nrows[0] : (mi == 1 ? a.nrows[1] : a.nrows[2])) >> 1;
    const int w0 = mi == 0 ? a.wstart[0] : (mi == 1 ? a.wstart[1] : a.wstart[2]);
    const int upw_sel = mi == 0 ? a.upw3[0] : (mi == 1 ? a.upw3[1] : a.upw3[2]);
    const int upw_i = upw_sel > 0 ? upw_sel : a.units_per_wave;
    const int p0 = min((gw - w0) * upw_i, npairs), p1 = min(p0 + upw_i, npairs);
    const int r0 = 2 * p0;
    // epilogue operands up front: lane i <-> pair i of the wave
    float pcs[NCOLS], psn[NCOLS];
    int64_t slots[NCOLS];
    {
      const int pair_i = ((r0 + 2 * min(lane, max(p1 - p0 - 1, 0))) % a.head_dim) >> 1;
      const bool rot = mi < 2 && pair_i < a.rot_pairs;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const size_t ti = (size_t)a.positions[c] * a.rot_pairs + (rot ? pair_i : 0);
        const float cs = a.cos_t[ti], sn = a.sin_t[ti];
        pcs[c] = rot ? cs : 1.0f;  // identity rotation for v and unrotated dims (x*1 - y*0 = x exactly)
        psn[c] = rot ? sn : 0.0f;
        slots[c] = mi == 0 ? 0 : a.slot_mapping[c];
      }
    }
    float prev[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) prev[c] = 0.0f;
    auto epi = [&](int, int row, const float(&sum)[NCOLS]) {  // row = local row of tensor mi
      if ((row & 1) == 0) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) prev[c] = sum[c];
        return;
      }
      const int lr = row - 1;          // even row of the pair
      const int pi = (lr - r0) >> 1;   // pair index inside the wave
      const int head = lr / a.head_dim, dd = lr % a.head_dim;
      // where the two results live: adjacent dims (interleaved RoPE), or dims i and i + head_dim / 2 when the rows were stored in pair order (v: never)
      const bool nx = a.neox && mi < 2;
      const int d0 = nx ? dd >> 1 : dd, d1 = nx ? d0 + (a.head_dim >> 1) : dd + 1;
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) {
        const float cs = rl(pcs[c], pi), sn = rl(psn[c], pi);
        float x, y;
        rope_pair<float>(prev[c], sum[c], cs, sn, x, y);
        if (lane == 0) {
          if (mi == 0) {
            st_out<LATE>(a.q_out + (size_t)c * a.nrows[0] + head * a.head_dim + d0, x);
            st_out<LATE>(a.q_out + (size_t)c * a.nrows[0] + head * a.head_dim + d1, y);
          } else {
            const int64_t slot = slots[c];
            if (slot >= 0) {
              const int64_t blk = slot / a.block_size, off = slot % a.block_size;
              uint16_t *kc = (uint16_t *)a.k_cache, *vc = (uint16_t *)a.v_cache;
              const uint16_t xb = a.kv_f16 ? float_to_half_bits(x) : float_to_bf16_bits(x), yb = a.kv_f16 ? float_to_half_bits(y) : float_to_bf16_bits(y);
              if (mi == 1) {
                const int X = a.cache_x;
                const int64_t hb = (blk * a.num_kv_heads + head) * (a.head_dim / X);
                kc[(hb + d0 / X) * a.block_size * X + off * X + d0 % X] = xb;
                kc[(hb + d1 / X) * a.block_size * X + off * X + d1 % X] = yb;  // interleaved: d1 = d0 + 1, same 16-byte group
              } else {
                const int64_t o = ((blk * a.num_kv_heads + head) * a.head_dim + dd) * a.block_size + off;
                vc[o] = xb;
                vc[o + a.block_size] = yb;
              }
            }
          }
        }
      }
    };
    Segs sg{};
    sg.nseg = 1; sg.mat[0] = mi == 0 ? a.m[0] : (mi == 1 ? a.m[1] : a.m[2]); sg.row0[0] = r0; sg.nrows[0] = 2 * (p1 - p0);
    MRS_DEC_STREAM(sg.mat[0].type, NCOLS, false, sg, epi, (a.ablate & 2) != 0)
  }
}

// SMALL: rows of <= 2 register-resident pieces per thread (ActPreSmall): the one-column launches of 4096-wide rows
template <int NCOLS, int EPI, bool SMALL = false>
__global__ void __launch_bounds__(NT) dec_gemv_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[8 * NW];  // RMSNorm partials: [column][wave]
  gemv_phase<NCOLS, EPI, false, NoSync, typename std::conditional<SMALL, ActPreSmall, ActPre>::type>(a, smem, red);
  MRS_TLW(a, 11);  // per-wave end: slots 11..18
}

// ------------------------------------------------------------------------------------------------ persistent decode step
// ONE launch runs a range of phases of the decode step on a grid of resident workgroups (one per CU): embedding, then per layer
// qkv / attention splits / attention merge / o_proj / gate-up / down, then final norm + lm_head.  Between phases every workgroup arrives on a
// monotonic device-scope counter and the next phase's prologue waits for it -- but the next phase's WEIGHT RING is filled before the wait, so
// HBM keeps streaming across the seam (what a kernel boundary cannot do: DESIGN.md section 4.5).  Cross-CU visibility: agent-scope release
// (after the phase's stores) -> relaxed counter -> agent-scope acquire (guide: MI355X_MICROARCH "Workgroup dispatch ... visibility").
// Every spin is bounded: a grid that is not fully resident gives up, flags sync[1] and lets the launch finish (garbage out, no hang).
// Every phase reads its arguments from a table in device memory that the host fills once (mrs_dec_build_step_table): the kernel bodies then
// see them exactly as a stand-alone kernel sees its kernarg segment (scalar loads, dynamic indices allowed) -- a struct assembled inside the
// kernel would live in scratch memory and put scratch loads into the weight stream's in-order queue.
enum : int { PH_EMBED = 0, PH_QKV = 1, PH_ATTN = 2, PH_MERGE = 3, PH_RESID = 4, PH_GLU = 5, PH_STORE = 6 };
struct PhaseDesc {
  int kind, group, kv_f16, embd_type;
  GemvArgs g;
  AttnArgs t;
  const uint8_t *embd; const int32_t *input_ids;  // PH_EMBED: g.out = h, g.K = hidden
};
struct StepArgs {
  const PhaseDesc *table;
  unsigned *sync;  // [0] arrivals, [1] error flag; [0] is zeroed before every launch
  int phase_begin, phase_end;  // phase ids: 0 = embedding, 1 + 6 l + {0 qkv, 1 attention splits, 2 attention merge, 3 o_proj, 4 gate/up, 5 down}, 1 + 6 L = lm_head
};

__device__ __forceinline__ void grid_arrive(unsigned *ctr) {
  MRS_WAIT_VMCNT0();  // this wave's stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
#ifndef MRS_DEC_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    MRS_WAIT_VMCNT0();  // the write-back must be complete before the arrival becomes visible (hipcc may drop its own wait)
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void grid_wait(unsigned *sync, unsigned target) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22) || __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {  // not all workgroups resident (or a peer gave up)
        __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
#ifndef MRS_DEC_NOFENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  __syncthreads();
}

struct GridSync {
  unsigned *sync; unsigned target; bool wait;
  __device__ __forceinline__ void operator()() const { if (wait) grid_wait(sync, target); }
};
// HG = query heads per work item (a GQA group of G runs as G / HG items: the K/V chunk is read G / HG times, the wave keeps HG heads of state)
template <int NCOLS, int HG, class CT>
__device__ __forceinline__ void attn_phase(const AttnArgs &t, int G, char *smem, bool merge) {
  const int wave = __builtin_amdgcn_readfirstlane(tid_opaque() >> 6);
  // item i -> workgroup i % grid, wave i / grid: the first `grid` items land on different CUs
  const int nwg = gridDim.x;
  if (!merge) {
    float *q_s = (float *)smem + wave * (HG * 128 + HG * 32), *p_s = q_s + HG * 128;
    const int sub = G / HG;  // items per kv head and split
    const int items = NCOLS * t.num_kv_heads * sub * t.max_splits;
    for (int i = wave * nwg + blockIdx.x; i < items; i += NW * nwg) {
      const int split = i % t.max_splits, r = i / t.max_splits, sb = r % sub, kvh = (r / sub) % t.num_kv_heads, seq = r / (sub * t.num_kv_heads);
      attn_split_item<HG, CT>(t, kvh, kvh * G + sb * HG, seq, split, q_s, p_s);
    }
  } else {
    const int items = NCOLS * t.num_heads;
    for (int i = wave * nwg + blockIdx.x; i < items; i += NW * nwg) attn_merge_item(t, i % t.num_heads, i / t.num_heads);
  }
}

template <int TYPE> __device__ __forceinline__ void dequant_units(const uint8_t *row, int K, float *o) {
  for (int s = tid_opaque(); s < K / 32; s += NT) {
    const Slice sl = load_slice<TYPE>(row, s);
    int ra, rb;
    slice_runs<TYPE>(s, ra, rb);
    const int8_t *qa = (const int8_t *)&sl.qa, *qb = (const int8_t *)&sl.qb;
#pragma unroll
    for (int j = 0; j < 16; ++j) { o[ra * 16 + j] = sl.sa * (float)qa[j] - sl.oa; o[rb * 16 + j] = sl.sb * (float)qb[j] - sl.ob; }
  }
}

#ifndef MRS_DEC_NO_STEP
template <int NCOLS>
__global__ void __launch_bounds__(NT) dec_step_kernel(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[8 * NW];  // RMSNorm partials: [column][wave]
  const int nwg = gridDim.x;
  for (int p = a.phase_begin; p < a.phase_end; ++p) {
    const GridSync sync{a.sync, (unsigned)(p - a.phase_begin) * (unsigned)nwg, p > a.phase_begin};
    const PhaseDesc &d = a.table[p];
    switch (d.kind) {
    case PH_EMBED: {  // embedding rows -> h (QuantMethod::embedding_forward): workgroup c takes token c
      if ((int)blockIdx.x < NCOLS) {
        const int K = d.g.K;
        const int64_t id = d.input_ids[blockIdx.x];
        float *o = d.g.out + (size_t)blockIdx.x * K;
        const int t0 = tid_opaque();
        if (d.embd_type == 0) { const float *src = (const float *)d.embd + id * K; for (int i = t0; i < K; i += NT) o[i] = src[i]; }
        else if (d.embd_type == 1) { const uint16_t *src = (const uint16_t *)d.embd + id * K; for (int i = t0; i < K; i += NT) o[i] = half_bits_to_float(src[i]); }
        else if (d.embd_type == 30) { const uint16_t *src = (const uint16_t *)d.embd + id * K; for (int i = t0; i < K; i += NT) o[i] = bf16_bits_to_float(src[i]); }
        else {
          switch (d.embd_type) {
          case T_Q4_K: dequant_units<T_Q4_K>(d.embd + (size_t)id * (K / 256) * 144, K, o); break;
          case T_Q5_K: dequant_units<T_Q5_K>(d.embd + (size_t)id * (K / 256) * 176, K, o); break;
          case T_Q6_K: dequant_units<T_Q6_K>(d.embd + (size_t)id * (K / 256) * 210, K, o); break;
          default: dequant_units<T_Q8_0>(d.embd + (size_t)id * (K / 32) * 34, K, o); break;
          }
        }
      }
    } break;
    case PH_QKV: gemv_phase<NCOLS, EPI_QKV, true, GridSync>(d.g, smem, red, sync); break;
    case PH_RESID: gemv_phase<NCOLS, EPI_RESID, true, GridSync>(d.g, smem, red, sync); break;
    case PH_GLU: gemv_phase<NCOLS, EPI_GLU, true, GridSync>(d.g, smem, red, sync); break;
    case PH_STORE: gemv_phase<NCOLS, EPI_STORE, true, GridSync>(d.g, smem, red, sync); break;
    case PH_ATTN:
      sync();
      if (d.group == 1) { if (d.kv_f16) attn_phase<NCOLS, 1, f16_t>(d.t, 1, smem, false); else attn_phase<NCOLS, 1, bf16_t>(d.t, 1, smem, false); }
      else { if (d.kv_f16) attn_phase<NCOLS, 2, f16_t>(d.t, d.group, smem, false); else attn_phase<NCOLS, 2, bf16_t>(d.t, d.group, smem, false); }
      break;
    default:  // PH_MERGE
      sync();
      attn_phase<NCOLS, 1, bf16_t>(d.t, d.group, smem, true);
      break;
    }
    if (p + 1 < a.phase_end) grid_arrive(a.sync);
  }
}

#endif  // MRS_DEC_NO_STEP

// ------------------------------------------------------------------------------------------------ repack GGUF blocks -> decode layout
// one thread per block (superblock, or 32-block for Q8_0); block i of the row-major GGUF tensor = (row, sb)
template <int TYPE>
__global__ void __launch_bounds__(256) repack_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const Planes p, long long nblocks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nblocks) return;
  if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K) {
    const uint8_t *b = src + i * Fmt<TYPE>::TS;
    *(uint32_t *)(dst + p.hd + i * 4) = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    const uint8_t *sc12 = b + 4;
    uint8_t sc[8], mn[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {  // get_scale_min_k4 (marlin_gguf_affine_repack.cu:200-210)
      if (g < 4) { sc[g] = sc12[g] & 63; mn[g] = sc12[g + 4] & 63; }
      else { sc[g] = (sc12[g + 4] & 15) | ((sc12[g - 4] >> 6) << 4); mn[g] = (sc12[g + 4] >> 4) | ((sc12[g] >> 6) << 4); }
    }
    for (int c = 0; c < 4; ++c) {
      uint8_t *h = dst + p.hs + i * 16 + c * 4;
      h[0] = sc[2 * c]; h[1] = sc[2 * c + 1]; h[2] = mn[2 * c]; h[3] = mn[2 * c + 1];
    }
    const uint8_t *qs = b + (TYPE == T_Q5_K ? 48 : 16);
    for (int j = 0; j < 128; j += 4) *(uint32_t *)(dst + p.q + i * 128 + j) = (uint32_t)qs[j] | ((uint32_t)qs[j + 1] << 8) | ((uint32_t)qs[j + 2] << 16) | ((uint32_t)qs[j + 3] << 24);
    if constexpr (TYPE == T_Q5_K) {
      const uint8_t *qh = b + 16;
      for (int s = 0; s < 8; ++s) {  // slice s = (quarter c, half hp): low nibbles = weights c*64 + hp*16 + a, high = + 32; fifth bits = qh[hp*16 + a] bits 2c, 2c+1
        const int c = s >> 1, hp = s & 1;
        uint32_t w = 0;
        for (int k = 0; k < 4; ++k)
          for (int j = 0; j < 4; ++j) {
            const uint32_t v = qh[hp * 16 + 4 * k + j];
            w |= ((v >> (2 * c)) & 1u) << (8 * j + k);
            w |= ((v >> (2 * c + 1)) & 1u) << (8 * j + 4 + k);
          }
        *(uint32_t *)(dst + p.x + i * 32 + s * 4) = w;
      }
    }
  } else if constexpr (TYPE == T_Q6_K) {
    const uint8_t *b = src + i * 210;
    for (int j = 0; j < 128; j += 2) *(uint16_t *)(dst + p.q + i * 128 + j) = (uint16_t)(b[j] | (b[j + 1] << 8));
    const uint8_t *qh = b + 128;
    const int8_t *scs = (const int8_t *)(b + 192);
    for (int u = 0; u < 4; ++u) {
      const int h = u >> 1, j = u & 1;
      for (int e = 0; e < 16; ++e) {
        const uint32_t v0 = qh[h * 32 + e], v1 = qh[h * 32 + 16 + e];
        dst[p.x + i * 64 + u * 16 + e] = (uint8_t)(((v0 >> (2 * j)) & 3) | (((v1 >> (2 * j)) & 3) << 2) | (((v0 >> (2 * j + 4)) & 3) << 4) | (((v1 >> (2 * j + 4)) & 3) << 6));
      }
      const int r = 8 * h + 2 * j;
      uint8_t *hs = dst + p.hs + i * 16 + u * 4;
      hs[0] = (uint8_t)scs[r]; hs[1] = (uint8_t)scs[r + 1]; hs[2] = (uint8_t)scs[r + 4]; hs[3] = (uint8_t)scs[r + 5];
    }
    *(uint16_t *)(dst + p.hd + i * 2) = (uint16_t)(b[208] | (b[209] << 8));
  } else {  // Q8_0
    const uint8_t *b = src + i * 34;
    *(uint16_t *)(dst + p.hd + i * 2) = (uint16_t)(b[0] | (b[1] << 8));
    for (int j = 0; j < 32; j += 2) *(uint16_t *)(dst + p.q + i * 32 + j) = (uint16_t)(b[2 + j] | (b[3 + j] << 8));
  }
}

static bool make_mat(Mat &m, const void *planes, int type, long long n, long long k) {
  if (!planes || !dec_type(type) || n <= 0 || k <= 0 || k % 32 || (type != T_Q8_0 && k % 256)) return false;
  const Planes p = plane_layout(type, n, k);
  if (p.total >= 0x7fffff00ull) return false;  // one 31-bit buffer descriptor per tensor
  m.base = (const uint8_t *)planes; m.off_x = (unsigned)p.x; m.off_hs = (unsigned)p.hs; m.off_hd = (unsigned)p.hd; m.bytes = (unsigned)p.total;
  m.type = type; m.n = (int)n; m.k = (int)k;
  return true;
}

// ------------------------------------------------------------------------------------------------ short-context decode attention, one launch
// Split-KV attention + merge + Q8_K quantization of the result in ONE kernel for contexts of <= FUSED_MAX_CTX tokens (the launch-per-phase path
// otherwise spends 6.8 + 4.8 us per layer in two latency-bound kernels, and o_proj re-quantizes the f32 result in every workgroup):
// a workgroup = HG = 2 query heads of one kv head (256 output values = exactly one Q8_K superblock of the attention vector), 12 waves (168 VGPRs: 16 waves spill); wave w
// takes the 32-token blocks w, w + 12, ... one at a time (attn_split_core with the same per-block arithmetic as the split kernel at bpw = 1),
// partials go to LDS instead of HBM; waves 0 / 1 merge the two heads (attn_merge_core: the merge kernel's order), wave 0 quantizes the 256
// values exactly as o_proj's prologue would (quantize4: same lane <-> element mapping) and writes the activation IMAGE (q | d | bsums in the
// LDS layout of dec_core.cuh) that dec_gemv_kernel copies into LDS.  Bits: identical to decode_attn_wave_kernel<.., false> + merge + prologue.
constexpr int FUSED_MAX_CTX = 1024;
template <int HG, class CT, int FUSED_NW>
__global__ void __launch_bounds__(FUSED_NW * 64) decode_attn_fused_kernel(const AttnArgs a, uint8_t *img, float *out_f32, int G, int ns_cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *po = (float *)smem;                              // [HG][ns_cap][128]
  float *pm = po + (size_t)HG * ns_cap * 128, *pl = pm + HG * ns_cap;  // [HG][ns_cap]
  float *merged = pl + HG * ns_cap;                       // [HG * 128]
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *q_s = merged + HG * 128 + wave * (HG * 128 + HG * 32), *p_s = q_s + HG * 128;
  const int sub = G / HG, kvh = (int)blockIdx.x / sub, head0 = kvh * G + ((int)blockIdx.x % sub) * HG, seq = blockIdx.y;
  const int nblk = min(((int)a.context_lens[seq] + 31) / 32, ns_cap);
  for (int b = wave; b < nblk; b += FUSED_NW)
    attn_split_core<HG, CT>(a, kvh, head0, seq, b, b + 1, q_s, p_s, [&](int g, float o0, float o1, float m, float l) {
      float *o = po + ((size_t)g * ns_cap + b) * 128;
      o[lane] = o0; o[lane + 64] = o1;
      if (lane == 0) { pm[g * ns_cap + b] = m; pl[g * ns_cap + b] = l; }
    });
  __syncthreads();
  if (wave < HG) {
    float v0, v1;
    attn_merge_core(nblk, pm + wave * ns_cap, pl + wave * ns_cap, po + (size_t)wave * ns_cap * 128, v0, v1);
    merged[wave * 128 + lane] = v0; merged[wave * 128 + lane + 64] = v1;
    if (out_f32) { float *o = out_f32 + ((size_t)seq * a.num_heads + head0 + wave) * 128; o[lane] = v0; o[lane + 64] = v1; }
  }
  __syncthreads();
  if (wave == 0) {  // HG * 128 = 256 values = superblock head0 / 2 of column seq
    const int K = a.num_heads * 128, ncols = gridDim.y, sb = head0 >> 1;
    const float4 v = *(const float4 *)(merged + lane * 4);
    char *qc = (char *)img + (size_t)seq * K;
    float *dc = (float *)(img + (size_t)ncols * K) + (size_t)seq * (K / 32);
    int *bsc = (int *)(img + (size_t)ncols * K + (size_t)ncols * (K / 32) * 4) + (size_t)seq * (K / 16);
    const int e = sb * 256 + lane * 4, piece = e >> 4;
    quantize4(v, e, ((piece ^ sb_mask(sb)) << 4) | ((lane & 3) << 2), true, ACT_Q8K, qc, dc, bsc);
  }
}

// ------------------------------------------------------------------------------------------------ decode attention, split + last-arriver merge
// Round 3.  Grid (kv heads, sequences, ceil(max splits / 4)), 4 waves = 4 splits per workgroup, no barrier in the split phase (as
// decode_attn_wave_kernel).  TICKET variant: instead of a second launch for the merge (4.9 us + a kernel boundary per layer), every workgroup publishes its
// partials write-through at agent scope, drains its stores and takes a ticket on the (sequence, kv head) counter; the workgroup that draws the
// last ticket merges all G heads of the kv head: wave w = query heads 2w, 2w + 1 (256 output values = ONE Q8_K superblock of the attention
// vector), lane = 4 consecutive dims, sequential over the splits (attn_merge_core's order).  It writes the f32 result and -- what o_proj's
// prologue would otherwise recompute in all 256 workgroups -- the Q8_K activation image that dec_gemv_kernel copies into LDS (quantize4: same
// lane <-> element mapping as the prologue, so the same bits).  The counter resets itself; hand-off recipe: MI355X guide, "handoff-flag"
// (sc1 payload -> vmcnt(0) -> agent atomic; consumer: returned atomic -> sc1 loads).
struct Attn2Args {
  AttnArgs t;
  unsigned *ticket;  // [seqs][kv heads], zero at rest
  uint8_t *img;      // Q8_K image of [seqs] columns of num_heads * 128 values, or nullptr (odd GQA groups: the caller's o_proj quantizes)
};
// merge of one (sequence, kv head): wave w <-> query heads (2w, 2w + 1) of the group (G == 1: wave 0, one head in lanes 0..31), lane = 4 consecutive
// dims, sequential over the splits.  AGENT: the partials were published by other workgroups of the SAME launch (sc1 loads); else plain loads.
template <int G, bool AGENT>
__device__ __forceinline__ void attn2_merge(const Attn2Args &a, int kvh, int seq, int ns, int wave, int lane, int ncols) {
  constexpr int HD = 128, NP = (G + 1) / 2;
  const AttnArgs &t = a.t;
  if (wave >= NP) return;
  const int head0 = kvh * G, hsel = lane >> 5, head = head0 + 2 * wave + hsel;
  const bool live = 2 * wave + hsel < G;
  const size_t pA = ((size_t)seq * t.num_heads + head0 + 2 * wave) * t.max_splits;                 // partials of head A (lanes 0..31)
  const size_t pB = ((size_t)seq * t.num_heads + head0 + min(2 * wave + 1, G - 1)) * t.max_splits;  // head B (lanes 32..63)
  auto ldw = [&](const float *p) { if constexpr (AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p; };
  const float mA = lane < ns ? ldw(t.part_m + pA + lane) : -FLT_MAX, lA = lane < ns ? ldw(t.part_l + pA + lane) : 0.f;
  const float mB = lane < ns ? ldw(t.part_m + pB + lane) : -FLT_MAX, lB = lane < ns ? ldw(t.part_l + pB + lane) : 0.f;
  // one descriptor for both heads' partials (wave-uniform); the lanes of head B add its distance.  The partial outputs of up to 16 splits are
  // requested before anything is consumed
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(t.part_o + pA * HD), (short)0, (int)((pB - pA + ns) * HD * 4), 0x00020000);
  const unsigned off0 = (unsigned)(lane & 31) * 16u + (hsel ? (unsigned)((pB - pA) * HD * 4) : 0u);
  constexpr int MB = 16;
  constexpr int AUX = AGENT ? 16 : 0;
  v4u r[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, off0 + (unsigned)i * 512u, 0, AUX);  // past the last split of head B: out of range, zeros
  const float wA = fast_exp_ref(mA - wave_max(mA)), wB = fast_exp_ref(mB - wave_max(mB));
  float s_all = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto step = [&](int j, v4u raw) {
    const float wAj = __shfl(wA, j, 64), wBj = __shfl(wB, j, 64), lAj = __shfl(lA, j, 64), lBj = __shfl(lB, j, 64);  // all lanes take part in every exchange
    const float wj = hsel ? wBj : wAj, lj = hsel ? lBj : lAj;
    const float lw = lj * wj;
    s_all = s_all + lw;
    const float4 o = as_f4(raw);
    const float t0 = o.x * wj, t1 = o.y * wj, t2 = o.z * wj, t3 = o.w * wj;
    acc.x = acc.x + t0; acc.y = acc.y + t1; acc.z = acc.z + t2; acc.w = acc.w + t3;
  };
  for (int j0 = 0; j0 < ns; j0 += MB) {
    if (j0 > 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, off0 + (unsigned)(j0 + i) * 512u, 0, AUX);
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
      if (j0 + i < ns) step(j0 + i, r[i]);  // wave-uniform
  }
  const float inv = 1.0f / s_all;
  const float4 v = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (t.out && live) *(float4 *)(t.out + ((size_t)seq * t.num_heads + head) * HD + (lane & 31) * 4) = v;
  if (a.img) {  // G even: the wave's 256 values = superblock (head0 + 2 wave) / 2 of column seq
    const int K = t.num_heads * HD, sb = (head0 + 2 * wave) >> 1;
    char *qc = (char *)a.img + (size_t)seq * K;
    float *dc = (float *)(a.img + (size_t)ncols * K) + (size_t)seq * (K / 32);
    int *bsc = (int *)(a.img + (size_t)ncols * K + (size_t)ncols * (K / 32) * 4) + (size_t)seq * (K / 16);
    const int e = sb * 256 + lane * 4, piece = e >> 4;
    quantize4(v, e, ((piece ^ sb_mask(sb)) << 4) | ((lane & 3) << 2), true, ACT_Q8K, qc, dc, bsc);
  }
}

// TICKET: one launch (the last workgroup of a (sequence, kv head) merges); else the split phase only and dec_attn2_merge_kernel follows
template <int G, class CT, bool TICKET>
__global__ void __launch_bounds__(256) dec_attn2_kernel(const Attn2Args a) {
  constexpr int HD = 128;
  __shared__ __attribute__((aligned(16))) float q_s[4][G * HD + G * 32];
  __shared__ int last_s;
  const AttnArgs &t = a.t;
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kvh = blockIdx.x, seq = blockIdx.y;
  const int split = blockIdx.z * 4 + wave, head0 = kvh * G;
  // the first page index of the split does not depend on the context length: both scalar loads leave together (they used to be a chain)
  const unsigned blk_first = t.block_tables[(size_t)seq * t.max_blocks_per_seq + min(split * t.bpw, t.max_blocks_per_seq - 1)];
  const int nblk = ((int)t.context_lens[seq] + 31) / 32;
  const int ns = (nblk + t.bpw - 1) / t.bpw, nwg = (ns + 3) / 4;
  if ((int)blockIdx.z >= nwg) return;  // workgroup-uniform: no split of this sequence lands here
  if (split < ns) {
    const int b0 = split * t.bpw, b1 = min(b0 + t.bpw, nblk);
    const int ctx_w = (int)t.context_lens[seq], lo_w = t.window > 0 && ctx_w > t.window ? ctx_w - t.window : 0;
    auto publish = [&](int g, float o0, float o1, float m, float l) {
      const size_t pi = ((size_t)seq * t.num_heads + head0 + g) * t.max_splits + split;
      if constexpr (TICKET) {
        __hip_atomic_store(t.part_o + pi * HD + lane, o0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t.part_o + pi * HD + lane + 64, o1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
          __hip_atomic_store(t.part_m + pi, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(t.part_l + pi, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        t.part_o[pi * HD + lane] = o0;
        t.part_o[pi * HD + lane + 64] = o1;
        if (lane == 0) { t.part_m[pi] = m; t.part_l[pi] = l; }
      }
    };
    if (b1 * 32 <= lo_w) {  // every block of the split lies before the sliding window: the partial a fully masked pass would give, without reading K / V
#pragma unroll
      for (int g = 0; g < G; ++g) publish(g, 0.f, 0.f, -FLT_MAX, 0.f);
    } else {
      attn_split_core<G, CT>(t, kvh, head0, seq, b0, b1, q_s[wave], q_s[wave] + G * HD, publish, (int)blk_first);
    }
  }
  if constexpr (TICKET) {
    MRS_WAIT_VMCNT0();  // this wave's partials have left the CU
    __syncthreads();
    unsigned *tk = a.ticket + (size_t)seq * t.num_kv_heads + kvh;
    if (tid == 0) last_s = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nwg - 1);
    __syncthreads();
    if (!last_s) return;
    if (tid == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every other workgroup of this (seq, kv head) has already drawn
    attn2_merge<G, true>(a, kvh, seq, ns, wave, lane, gridDim.y);
  }
}
// the merge as its own launch: grid (kv heads, sequences), (G + 1) / 2 waves
template <int G>
__global__ void __launch_bounds__(64 * ((G + 1) / 2)) dec_attn2_merge_kernel(const Attn2Args a) {
  const int tid = tid_opaque(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = ((int)a.t.context_lens[blockIdx.y] + 31) / 32;
  attn2_merge<G, false>(a, blockIdx.x, blockIdx.y, (nblk + a.t.bpw - 1) / a.t.bpw, wave, lane, gridDim.y);
}

// experiment support (-DMRS_DEC_TIMELINE builds): mrs_dec_timeline(buf, cap) hands the launchers a device buffer of cap records of 256 x 32 stamps;
// launch i of the process writes record i % cap, its epilogue kind goes to the host-side log read back with mrs_dec_timeline_log
static unsigned long long *g_tl_buf = nullptr;
static int g_tl_cap = 0, g_tl_next = 0;
static int g_tl_kind[4096];
static unsigned long long *timeline_slot(int kind) {
  if (!g_tl_buf || g_tl_cap <= 0) return nullptr;
  const int i = g_tl_next++ % g_tl_cap;
  g_tl_kind[i % 4096] = kind;
  return g_tl_buf + (size_t)i * 256 * 32;
}

template <int EPI> struct Launch {
  template <int NCOLS> static int go(GemvArgs a, hipStream_t s) {
    int upw = (a.units + 256 * NW - 1) / (256 * NW);
    if (upw < 1) upw = 1;
    if (upw > 64) upw = 64;  // epilogue operands are prefetched one unit per lane
    { static int ov = -1; if (ov < 0) { const char *e = getenv("MRS_DEC_UPW"); ov = e ? atoi(e) : 0; } if (ov > 0 && ov <= 64) upw = ov; }
    { static int ab = -1; if (ab < 0) { const char *e = getenv("MRS_DEC_ABLATE"); ab = e ? atoi(e) : 0; } a.ablate = ab; }
    a.tl = timeline_slot(EPI);
    { static int stg = -1; if (stg < 0) { const char *e = getenv("MRS_DEC_STAGED"); stg = e ? atoi(e) : 1; }
      a.staged = stg && NCOLS == 1 && EPI != EPI_RESID2 && !a.x_img && a.K <= ACT_MAXV * ACT_STRIDE && (!a.norm_w || a.K <= ACT_MAXW * ACT_STRIDE); }
    if (a.slots > 1) {  // units = slots * nrows: waves must not straddle experts
      while (upw > 1 && a.nrows[0] % upw) --upw;
      if (a.nrows[0] % upw) return -3;
    }
    a.units_per_wave = upw;
    int grid = (a.units + upw * NW - 1) / (upw * NW);
    if (EPI == EPI_QKV) {
      // pairs per wave per tensor: start from the common figure, then shorten the runs of the tensor with the most bytes per wave while the waves
      // still fit one workgroup per CU (8B: q 2 / k 1 / v 1 pairs -> 256 workgroups instead of 192, the Q6_K v rows no longer the long pole)
      static const int bal = [] { const char *e = getenv("MRS_DEC_QKV_BALANCE"); return e ? atoi(e) : 1; }();
      int u3[3] = {upw, upw, upw};
      auto waves_of = [&](const int *u) { int w = 0; for (int i = 0; i < 3; ++i) w += ((a.nrows[i] >> 1) + u[i] - 1) / u[i]; return w; };
      auto row_bytes = [&](int i) { const Planes p = plane_layout(a.m[i].type, 1, a.K); return (double)p.total; };
      for (int it = 0; bal && it < 8; ++it) {
        int best = -1; double worst = 0;
        for (int i = 0; i < 3; ++i) { const double b = u3[i] * 2 * row_bytes(i); if (u3[i] > 1 && b > worst) { worst = b; best = i; } }
        if (best < 0) break;
        int t3[3] = {u3[0], u3[1], u3[2]};
        // the heaviest first; if it does not fit, try the others in turn
        bool moved = false;
        for (int k = 0; k < 3 && !moved; ++k) {
          const int i = (best + k) % 3;
          if (t3[i] <= 1) continue;
          t3[i] -= 1;
          if (waves_of(t3) <= 256 * NW) { u3[i] = t3[i]; moved = true; } else t3[i] += 1;
        }
        if (!moved) break;
      }
      a.wstart[0] = 0;
      for (int i = 0; i < 3; ++i) { a.upw3[i] = u3[i]; a.wstart[i + 1] = a.wstart[i] + ((a.nrows[i] >> 1) + u3[i] - 1) / u3[i]; }
      grid = (a.wstart[3] + NW - 1) / NW;
    }
    size_t lds = (act_bytes(a.K, NCOLS) + 15) & ~(size_t)15;
    if (lds > 158 * 1024) return -2;
    { static long pad = -1; if (pad < 0) { const char *e = getenv("MRS_DEC_LDS_PAD"); pad = e ? atol(e) : 0; } if ((size_t)pad > lds && pad <= 158 * 1024) lds = (size_t)pad; }  // experiment: > 80 KiB forces one workgroup per CU
    if constexpr (NCOLS == 1 && EPI != EPI_RESID2) {
      static const int small_on = [] { const char *e = getenv("MRS_DEC_SMALL"); return e ? atoi(e) : 1; }();
      const size_t pieces = a.x_img ? act_bytes(a.K, NCOLS) : (size_t)a.K * 4;  // bytes that travel through the prologue's registers
      if (small_on && pieces <= (size_t)ActPreSmall::NV * NT * 16) {
        a.staged = a.staged && (!a.norm_w || a.K <= ActPreSmall::NWV * ACT_STRIDE);
        auto kern = dec_gemv_kernel<NCOLS, EPI, true>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024); attr = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);
        return 0;
      }
    }
    auto kern = dec_gemv_kernel<NCOLS, EPI, false>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024); attr = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);
    return 0;
  }
  // activation columns [c0, ...) of a batched launch: every per-column pointer moves
  static GemvArgs shift_cols(GemvArgs a, int c0) {
    if (a.x) a.x += (size_t)c0 * a.ldx;
    if (a.out) a.out += (size_t)c0 * a.out_stride;
    if (a.q_out) a.q_out += (size_t)c0 * a.nrows[0];
    if (a.positions) a.positions += c0;
    if (a.slot_mapping) a.slot_mapping += c0;
    return a;
  }
  static int run(const GemvArgs &a, int b, hipStream_t s) {
#ifdef MRS_DEC_EXP_B1  // experiment builds: batch 1 only (an eighth of the compile time)
    return b == 1 ? go<1>(a, s) : -1;
#endif
    // the activation image of all columns must fit LDS (1.375 K bytes per column): wider batches run as column groups, each a launch of its own
    // (Llama-3-70B down_proj, K = 28672: 4 columns per launch) -- advisor, round 2: such batches used to fail with -2
    if (b > 1 && !a.x_img && act_bytes(a.K, b) > 158 * 1024) {
      const int half = b / 2;
      const int rc = run(a, half, s);
      return rc ? rc : run(shift_cols(a, half), b - half, s);
    }
    switch (b) {
    case 1: return go<1>(a, s); case 2: return go<2>(a, s); case 3: return go<3>(a, s); case 4: return go<4>(a, s);
    case 5: return go<5>(a, s); case 6: return go<6>(a, s); case 7: return go<7>(a, s); case 8: return go<8>(a, s);
    default: return -1;
    }
  }
};

}  // namespace dec
}  // namespace mrs

using namespace mrs;
using namespace mrs::dec;

struct mrs_dec_mat_c { const void *planes; int type; long long n, k; };  // == mrs_dec_mat (include/mrs_hip_ext.h)

extern "C" void mrs_dec_timeline(void *buf, int cap) { g_tl_buf = (unsigned long long *)buf; g_tl_cap = cap > 4096 ? 4096 : cap; g_tl_next = 0; }
extern "C" int mrs_dec_timeline_log(int *kinds, int n) { const int m = g_tl_next < g_tl_cap ? g_tl_next : g_tl_cap; for (int i = 0; i < n && i < m; ++i) kinds[i] = g_tl_kind[i]; return g_tl_next; }
extern "C" int mrs_dec_supported(int ggml_type) { return dec_type(ggml_type) ? 1 : 0; }
extern "C" size_t mrs_dec_repack_bytes(int type, long long n, long long k) {
  if (!dec_type(type) || k % 32 || (type != T_Q8_0 && k % 256)) return 0;
  return plane_layout(type, n, k).total;
}
extern "C" int mrs_dec_repack(const void *gguf_blocks, int type, long long n, long long k, void *planes, void *stream) {
  if (!mrs_dec_repack_bytes(type, n, k) || !gguf_blocks || !planes) return -1;
  const Planes p = plane_layout(type, n, k);
  const long long nb = n * (k / (type == T_Q8_0 ? 32 : 256));
  const dim3 grid((unsigned)((nb + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
  case T_Q4_K: hipLaunchKernelGGL(repack_kernel<T_Q4_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, p, nb); break;
  case T_Q5_K: hipLaunchKernelGGL(repack_kernel<T_Q5_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, p, nb); break;
  case T_Q6_K: hipLaunchKernelGGL(repack_kernel<T_Q6_K>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, p, nb); break;
  default: hipLaunchKernelGGL(repack_kernel<T_Q8_0>, grid, dim3(256), 0, s, (const uint8_t *)gguf_blocks, (uint8_t *)planes, p, nb); break;
  }
  return 0;
}

// q, k, v projections of the decode step: h [b][ldh] f32 -> RMSNorm -> quantize -> GEMV -> RoPE (interleaved pairs) -> q_out f32 [b][nq],
// k / v into the paged cache (kv_dtype 1 = bf16, 0 = f16).  All three tensors must share the activation format (K-quants or Q8_0).
static int dec_qkv_impl(const mrs_dec_mat_c *wq, const mrs_dec_mat_c *wk, const mrs_dec_mat_c *wv, const float *h, int ldh, const float *norm_w, float eps,
                        float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                        const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, int neox, void *stream) {
  GemvArgs a{};
  a.neox = neox;
  if (neox && rot_pairs * 2 != head_dim) return -1;  // pair order (i, i + head_dim / 2) is the rotate-half pairing only when every dim rotates
  if (!wq || !wk || !wv || !make_mat(a.m[0], wq->planes, wq->type, wq->n, wq->k) || !make_mat(a.m[1], wk->planes, wk->type, wk->n, wk->k) ||
      !make_mat(a.m[2], wv->planes, wv->type, wv->n, wv->k)) return -1;
  if (wq->k != wk->k || wq->k != wv->k || ((wq->n | wk->n | wv->n | head_dim) & 1)) return -1;
  if (act_mode_for(wq->type) != act_mode_for(wk->type) || act_mode_for(wq->type) != act_mode_for(wv->type)) return -1;
  if (kv_dtype != 0 && kv_dtype != 1) return -1;
  a.nrows[0] = (int)wq->n; a.nrows[1] = (int)wk->n; a.nrows[2] = (int)wv->n; a.K = (int)wq->k;
  a.x = h; a.ldx = ldh; a.norm_w = norm_w; a.eps = eps; a.q_out = q_out; a.k_cache = k_cache; a.v_cache = v_cache; a.slot_mapping = slot_mapping;
  a.positions = positions; a.cos_t = cos_t; a.sin_t = sin_t; a.head_dim = head_dim; a.rot_pairs = rot_pairs; a.num_kv_heads = num_kv_heads;
  a.block_size = block_size; a.cache_x = 8; a.kv_f16 = kv_dtype == 0;
  a.units = (a.nrows[0] + a.nrows[1] + a.nrows[2]) / 2;
  return Launch<EPI_QKV>::run(a, b, (hipStream_t)stream);
}
extern "C" int mrs_dec_qkv(const mrs_dec_mat_c *wq, const mrs_dec_mat_c *wk, const mrs_dec_mat_c *wv, const float *h, int ldh, const float *norm_w, float eps,
                           float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                           const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, void *stream) {
  return dec_qkv_impl(wq, wk, wv, h, ldh, norm_w, eps, q_out, k_cache, v_cache, slot_mapping, positions, cos_t, sin_t, head_dim, rot_pairs, num_kv_heads, block_size,
                      kv_dtype, b, 0, stream);
}
// Rotate-half ("neox") RoPE, as safetensors Llama / Mistral checkpoints use it (RotaryEmbedding::forward with is_gpt_neox, layers.rs:2978): the planes of
// wq and wk must have been repacked from rows in PAIR order -- inside every head the original rows 0, hd/2, 1, hd/2 + 1, ... -- so that a wave still holds
// both members of a pair; the results are written back to dims i and i + hd/2 (q_out and the K pages keep the model's dim order).  wv: original order.
extern "C" int mrs_dec_qkv_neox(const mrs_dec_mat_c *wq, const mrs_dec_mat_c *wk, const mrs_dec_mat_c *wv, const float *h, int ldh, const float *norm_w, float eps,
                                float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                                const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, void *stream) {
  return dec_qkv_impl(wq, wk, wv, h, ldh, norm_w, eps, q_out, k_cache, v_cache, slot_mapping, positions, cos_t, sin_t, head_dim, rot_pairs, num_kv_heads, block_size,
                      kv_dtype, b, 1, stream);
}

// gate / up: h -> RMSNorm -> quantize -> act(W_g . x) * (W_u . x) -> act_out f32 [b][ld_out].  expert_sel != nullptr: stacked experts
// [E * n][K], the expert index is read on the device (graph-capturable routing).
extern "C" int mrs_dec_gate_up(const mrs_dec_mat_c *wg, const mrs_dec_mat_c *wu, int n, const int32_t *expert_sel, const float *h, int ldh, const float *norm_w,
                               float eps, int activation, float *act_out, int ld_out, int b, void *stream) {
  GemvArgs a{};
  if (!wg || !wu || !make_mat(a.m[0], wg->planes, wg->type, wg->n, wg->k) || !make_mat(a.m[1], wu->planes, wu->type, wu->n, wu->k)) return -1;
  if (wg->type != wu->type || wg->n != wu->n || wg->k != wu->k || n <= 0 || wg->n % n || (!expert_sel && wg->n != n)) return -1;
  a.nrows[0] = a.nrows[1] = n; a.K = (int)wg->k; a.x = h; a.ldx = ldh; a.norm_w = norm_w; a.eps = eps; a.activation = activation;
  a.out = act_out; a.out_stride = ld_out; a.units = n; a.expert_sel = expert_sel;
  return Launch<EPI_GLU>::run(a, b, (hipStream_t)stream);
}

// MoE decode: the gate / up phase of ALL top-k experts of one token in one launch (one activation prologue, one dispatch, 2 x top-k x n rows streamed):
// expert_sel [topk] on the device, act_out [topk][ld_out].  -3: the row count does not split into whole waves per expert (caller loops over mrs_dec_gate_up).
extern "C" int mrs_dec_gate_up_topk(const mrs_dec_mat_c *wg, const mrs_dec_mat_c *wu, int n, const int32_t *expert_sel, int topk, const float *h, const float *norm_w,
                                    float eps, int activation, float *act_out, int ld_out, void *stream) {
  GemvArgs a{};
  if (!wg || !wu || !expert_sel || topk < 1 || topk > 8 || !make_mat(a.m[0], wg->planes, wg->type, wg->n, wg->k) || !make_mat(a.m[1], wu->planes, wu->type, wu->n, wu->k)) return -1;
  if (wg->type != wu->type || wg->n != wu->n || wg->k != wu->k || n <= 0 || wg->n % n) return -1;
  a.nrows[0] = a.nrows[1] = n; a.K = (int)wg->k; a.x = h; a.ldx = (int)wg->k; a.norm_w = norm_w; a.eps = eps; a.activation = activation;
  a.out = act_out; a.out_stride = ld_out; a.units = n * topk; a.expert_sel = expert_sel; a.slots = topk; a.slot_out_stride = ld_out;
  return Launch<EPI_GLU>::run(a, 1, (hipStream_t)stream);
}

// MoE decode, top-2: out = (out * resid_scale + w[0] W_{sel[0]} . x[0]) + w[1] W_{sel[1]} . x[1] in one launch; x [2][ldx] = the two experts' activations,
// expert_sel / acc_scale [2] on the device.  Same bits as two mrs_dec_proj launches (mode 1: resid_scale, then 1).
extern "C" int mrs_dec_proj_top2(const mrs_dec_mat_c *w, int n, const int32_t *expert_sel, const float *x, int ldx, float *out, float resid_scale,
                                 const float *acc_scale, void *stream) {
  GemvArgs a{};
  if (!w || !expert_sel || !acc_scale || !make_mat(a.m[0], w->planes, w->type, w->n, w->k) || n <= 0 || w->n % n) return -1;
  a.nrows[0] = n; a.K = (int)w->k; a.x = x; a.ldx = ldx; a.out = out; a.out_stride = n; a.resid_scale = resid_scale; a.acc_scale = acc_scale; a.units = n;
  a.expert_sel = expert_sel;
  return Launch<EPI_RESID2>::go<2>(a, (hipStream_t)stream);
}

// plain projection: x [b][ldx] f32 (-> RMSNorm when norm_w) -> quantize -> GEMV.  mode 0: out = W.x;  mode 1: out = out * resid_scale + s * W.x
// (s = *acc_scale or 1; resid_scale = 1 / world_size under tensor parallelism, distributed/layers.rs:965-975)
extern "C" int mrs_dec_proj(const mrs_dec_mat_c *w, int n, const int32_t *expert_sel, const float *x, int ldx, const float *norm_w, float eps, float *out,
                            int ld_out, int mode, float resid_scale, const float *acc_scale, int b, void *stream) {
  GemvArgs a{};
  if (!w || !make_mat(a.m[0], w->planes, w->type, w->n, w->k) || n <= 0 || w->n % n || (!expert_sel && w->n != n)) return -1;
  a.nrows[0] = n; a.K = (int)w->k; a.x = x; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.out = out; a.out_stride = ld_out;
  a.resid_scale = resid_scale; a.acc_scale = acc_scale; a.units = n; a.expert_sel = expert_sel;
  return mode ? Launch<EPI_RESID>::run(a, b, (hipStream_t)stream) : Launch<EPI_STORE>::run(a, b, (hipStream_t)stream);
}

// o_proj & friends on activations the producer already quantized (mrs_dec_attention_q8k): x_img = the LDS image of b columns of k values
extern "C" int mrs_dec_proj_img(const mrs_dec_mat_c *w, int n, const void *x_img, float *out, int ld_out, int mode, float resid_scale, int b, void *stream) {
  GemvArgs a{};
  if (!w || !x_img || !make_mat(a.m[0], w->planes, w->type, w->n, w->k) || n <= 0 || w->n != n || act_mode_for(w->type) != ACT_Q8K) return -1;
  if (act_bytes((int)w->k, b) > (size_t)ACT_MAXV * NT * 16) return -3;  // the image is staged through the prologue's registers
  a.nrows[0] = n; a.K = (int)w->k; a.x_img = x_img; a.out = out; a.out_stride = ld_out; a.resid_scale = resid_scale; a.units = n;
  return mode ? Launch<EPI_RESID>::run(a, b, (hipStream_t)stream) : Launch<EPI_STORE>::run(a, b, (hipStream_t)stream);
}
extern "C" size_t mrs_dec_act_image_bytes(int k, int b) { return act_bytes(k, b); }
extern "C" size_t mrs_dec_proj_img_max_bytes(void) { return (size_t)ACT_MAXV * NT * 16; }
// Decode attention for short contexts in one launch: img_out = Q8_K activation image [b columns][num_heads * 128] for mrs_dec_proj_img, out_f32
// (may be NULL) = the f32 result [b][num_heads * 128].  Returns -3 when the shape is outside the kernel (max_context_len > 1024, GQA group
// not a multiple of 2, head size != 128, block size != 32): the caller uses mrs_decode_attention_f32_* + mrs_dec_proj.
extern "C" int mrs_dec_attention_q8k(void *img_out, float *out_f32, const float *q, const void *k_cache, const void *v_cache, int num_kv_heads, float scale,
                                     const uint32_t *block_tables, const uint32_t *context_lens, int block_size, int max_context_len, int num_seqs,
                                     int num_heads, int head_size, int max_blocks_per_seq, int q_stride, int kv_block_stride, int kv_head_stride,
                                     int kv_dtype, void *stream) {
  if (!img_out || block_size != 32 || head_size != 128 || num_seqs <= 0 || num_seqs > 8 || num_kv_heads <= 0 || num_heads % num_kv_heads ||
      (kv_dtype != 0 && kv_dtype != 1)) return -1;
  const int G = num_heads / num_kv_heads;
  if (G % 2 || max_context_len > FUSED_MAX_CTX || max_context_len <= 0) return -3;
  AttnArgs t{};
  t.q = q; t.k_cache = (const uint16_t *)k_cache; t.v_cache = (const uint16_t *)v_cache; t.block_tables = block_tables; t.context_lens = context_lens;
  t.num_heads = num_heads; t.num_kv_heads = num_kv_heads; t.max_blocks_per_seq = max_blocks_per_seq; t.q_stride = q_stride;
  t.kv_block_stride = kv_block_stride; t.kv_head_stride = kv_head_stride; t.bpw = 1; t.num_seqs = num_seqs; t.scale = scale;
  const int ns_cap = (max_context_len + 31) / 32;
  t.max_splits = ns_cap;
  static const int nw = [] { const char *e = getenv("MRS_DEC_ATTN_WAVES"); const int v = e ? atoi(e) : 8; return v == 12 ? 12 : 8; }();
  const size_t lds = ((size_t)2 * ns_cap * 128 + 2 * 2 * ns_cap + 2 * 128 + (size_t)nw * (2 * 128 + 2 * 32)) * 4;
  const dim3 grid(num_kv_heads * (G / 2), num_seqs);
  auto go = [&](auto kern) {
    lds_attr_once((const void *)kern, 158 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(nw * 64), lds, (hipStream_t)stream, t, (uint8_t *)img_out, out_f32, G, ns_cap);
  };
  if (kv_dtype == 1) { if (nw == 8) go(decode_attn_fused_kernel<2, bf16_t, 8>); else go(decode_attn_fused_kernel<2, bf16_t, 12>); }
  else { if (nw == 8) go(decode_attn_fused_kernel<2, f16_t, 8>); else go(decode_attn_fused_kernel<2, f16_t, 12>); }
  return 0;
}

// Decode attention of the engine, one launch (dec_attn2_kernel).  out_f32 [b][num_heads * 128] (may be NULL when an image is written);
// img_out: Q8_K activation image for mrs_dec_proj_img (written when the GQA group is even; may be NULL); ticket: [b * num_kv_heads] u32, zero
// before the first call (the kernel leaves it zero); part_*: the partials workspace of mrs_decode_attention_f32_*.  Returns 1 when the image
// was written, 0 when only out_f32 was, < 0 on a shape outside the kernel (head size 128, 32-token pages, GQA group 1 / 2 / 4 / 8).
extern "C" int mrs_dec_attention(float *out_f32, void *img_out, unsigned *ticket, float *part_o, float *part_m, float *part_l, const float *q, const void *k_cache,
                                 const void *v_cache, int num_kv_heads, float scale, const uint32_t *block_tables, const uint32_t *context_lens, int block_size,
                                 int max_context_len, int num_seqs, int num_heads, int head_size, int max_blocks_per_seq, int q_stride, int kv_block_stride,
                                 int kv_head_stride, int kv_dtype, int sliding_window, void *stream) {
  if (!ticket || !part_o || !part_m || !part_l || block_size != 32 || head_size != 128 || num_seqs <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads ||
      (kv_dtype != 0 && kv_dtype != 1) || max_context_len <= 0) return -1;
  const int G = num_heads / num_kv_heads;
  if (G != 1 && G != 2 && G != 4 && G != 8) return -1;
  const bool with_img = img_out && (G % 2 == 0) && num_seqs <= 8;
  if (!with_img && !out_f32) return -1;
  Attn2Args a{};
  AttnArgs &t = a.t;
  t.q = q; t.k_cache = (const uint16_t *)k_cache; t.v_cache = (const uint16_t *)v_cache; t.block_tables = block_tables; t.context_lens = context_lens;
  t.part_o = part_o; t.part_m = part_m; t.part_l = part_l; t.out = out_f32;
  t.num_heads = num_heads; t.num_kv_heads = num_kv_heads; t.max_blocks_per_seq = max_blocks_per_seq; t.q_stride = q_stride;
  t.kv_block_stride = kv_block_stride; t.kv_head_stride = kv_head_stride; t.num_seqs = num_seqs; t.scale = scale;
  t.window = sliding_window > 0 ? sliding_window : 0;
  const int nblk = (max_context_len + 31) / 32;
  t.bpw = nblk <= 64 ? 1 : (nblk + 63) / 64;                        // == dec_bpw() of paged_attention.hip: at most 64 splits
  { static const int force = [] { const char *e = getenv("MRS_DEC_ATTN_BPW"); return e ? atoi(e) : 0; }(); if (force > 0) t.bpw = force; }  // measurements only: the oracle's order follows the rule above
  t.max_splits = mrs_decode_attention_max_splits(max_context_len);  // stride of the partials, as in the two-launch route
  a.ticket = ticket; a.img = with_img ? (uint8_t *)img_out : nullptr;
  const int nsplit = (nblk + t.bpw - 1) / t.bpw;
  const dim3 grid(num_kv_heads, num_seqs, (nsplit + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  // MRS_DEC_ATTN_TICKET (default 1): the merge inside the split launch (last arriver); 0 = a second launch for the merge.  Measured on the MI355X
  // (profiles/round3_decode.md): the hand-off (write-through partials, drain, device-scope ticket, sc1 loads) costs about what the second launch and its
  // boundary cost -- 472.8 vs 465.4 tok/s for the whole step
  static const int one_launch = [] { const char *e = getenv("MRS_DEC_ATTN_TICKET"); return e ? atoi(e) : 1; }();
  const dim3 mgrid(num_kv_heads, num_seqs);
#define MRS_A2(GG, CT)                                                                                                   \
  do {                                                                                                                   \
    if (one_launch) hipLaunchKernelGGL((dec_attn2_kernel<GG, CT, true>), grid, dim3(256), 0, s, a);                      \
    else {                                                                                                               \
      hipLaunchKernelGGL((dec_attn2_kernel<GG, CT, false>), grid, dim3(256), 0, s, a);                                   \
      hipLaunchKernelGGL((dec_attn2_merge_kernel<GG>), mgrid, dim3(64 * ((GG + 1) / 2)), 0, s, a);                       \
    }                                                                                                                    \
  } while (0)
#define MRS_A2G(CT) switch (G) { case 1: MRS_A2(1, CT); break; case 2: MRS_A2(2, CT); break; case 4: MRS_A2(4, CT); break; default: MRS_A2(8, CT); break; }
  if (kv_dtype == 1) { MRS_A2G(bf16_t) } else { MRS_A2G(f16_t) }
#undef MRS_A2G
#undef MRS_A2
  return with_img ? 1 : 0;
}

// ---- persistent decode step (one launch for a range of phases; b = 1)
struct mrs_dec_layer_c { mrs_dec_mat_c q, k, v, o, gate, up, down; const float *attn_norm, *ffn_norm; void *k_cache, *v_cache; };
struct mrs_dec_step_args_c {
  int num_layers;
  mrs_dec_mat_c lm_head; const float *final_norm;
  const void *embd; int embd_type;
  const int32_t *input_ids;
  float *h, *q, *attn, *act, *logits, *part_o, *part_m, *part_l;
  const uint32_t *block_tables, *context_lens; const int32_t *positions; const int64_t *slot_mapping; const float *cos_t, *sin_t;
  int hidden, num_heads, num_kv_heads, head_dim, rot_pairs, ff, vocab, block_size, max_blocks_per_seq, max_context_len;
  float eps, resid_scale;
  int kv_dtype;
};
static int dec_step_grid() {
  static int g = 0;
  if (!g) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const char *e = getenv("MRS_DEC_STEP_WGS");
    g = e && atoi(e) > 0 ? atoi(e) : cus;  // one resident workgroup per CU
  }
  return g;
}
extern "C" size_t mrs_dec_step_table_bytes(int num_layers) { return (size_t)(2 + 6 * num_layers) * sizeof(PhaseDesc); }
extern "C" int mrs_dec_step_num_phases(int num_layers) { return 2 + 6 * num_layers; }
// Builds the phase table of a model on the host and copies it to `device_table` (blocking copy: load-time set-up; every pointer in `c` and
// `layers` must stay valid while the table is used).  Returns -3 when a shape needs the launch-per-phase route, -1 on bad arguments.
extern "C" int mrs_dec_build_step_table(const mrs_dec_step_args_c *c, const mrs_dec_layer_c *layers, void *device_table) {
  if (!c || !layers || !device_table || c->num_layers <= 0) return -1;
  if (c->head_dim != 128 || c->block_size != 32 || (c->kv_dtype != 0 && c->kv_dtype != 1)) return -3;
  const int G = c->num_heads / c->num_kv_heads;
  if (c->num_heads % c->num_kv_heads || (G != 1 && G != 2 && G != 4 && G != 8)) return -3;
  const int grid = dec_step_grid(), waves = grid * NW;
  const int nq = c->num_heads * c->head_dim, nkv = c->num_kv_heads * c->head_dim, d = c->hidden;
  auto upw_of = [&](int units) { return (units + waves - 1) / waves; };
  if (upw_of(c->vocab) > 64 || upw_of(c->ff) > 64 || upw_of(d) > 64) return -3;
  const int np = 2 + 6 * c->num_layers;
  PhaseDesc *tab = (PhaseDesc *)calloc((size_t)np, sizeof(PhaseDesc));
  if (!tab) return -1;
  int rc = 0;
  const int eff_max = std::min(c->max_blocks_per_seq * c->block_size, c->max_context_len);
  const int nblk = (eff_max + 31) / 32;
  AttnArgs t{};
  t.q = c->q; t.block_tables = c->block_tables; t.context_lens = c->context_lens; t.part_o = c->part_o; t.part_m = c->part_m; t.part_l = c->part_l;
  t.out = c->attn; t.num_heads = c->num_heads; t.num_kv_heads = c->num_kv_heads; t.max_blocks_per_seq = c->max_blocks_per_seq; t.q_stride = nq;
  t.kv_block_stride = c->num_kv_heads * c->head_dim * c->block_size; t.kv_head_stride = c->head_dim * c->block_size;
  t.bpw = nblk <= 64 ? 1 : (nblk + 63) / 64;                 // == dec_bpw() of paged_attention.hip
  t.max_splits = mrs_decode_attention_max_splits(eff_max);   // stride of the partials, as in the launch-per-phase route
  t.num_seqs = 1; t.scale = 1.0f / sqrtf((float)c->head_dim);
  {
    PhaseDesc &e = tab[0];
    e.kind = PH_EMBED; e.embd = (const uint8_t *)c->embd; e.embd_type = c->embd_type; e.input_ids = c->input_ids; e.g.out = c->h; e.g.K = d;
  }
  for (int l = 0; l < c->num_layers && !rc; ++l) {
    const mrs_dec_layer_c &L = layers[l];
    PhaseDesc *ph = tab + 1 + 6 * l;
    {  // qkv
      GemvArgs &g = ph[0].g;
      ph[0].kind = PH_QKV;
      if (!make_mat(g.m[0], L.q.planes, L.q.type, L.q.n, L.q.k) || !make_mat(g.m[1], L.k.planes, L.k.type, L.k.n, L.k.k) || !make_mat(g.m[2], L.v.planes, L.v.type, L.v.n, L.v.k)) rc = -1;
      if (act_mode_for(L.q.type) != act_mode_for(L.k.type) || act_mode_for(L.q.type) != act_mode_for(L.v.type) || L.q.n != nq || L.k.n != nkv || L.v.n != nkv || L.q.k != d) rc = -1;
      g.nrows[0] = nq; g.nrows[1] = nkv; g.nrows[2] = nkv; g.K = d; g.x = c->h; g.ldx = d; g.norm_w = L.attn_norm; g.eps = c->eps; g.q_out = c->q;
      g.k_cache = L.k_cache; g.v_cache = L.v_cache; g.slot_mapping = c->slot_mapping; g.positions = c->positions; g.cos_t = c->cos_t; g.sin_t = c->sin_t;
      g.head_dim = c->head_dim; g.rot_pairs = c->rot_pairs; g.num_kv_heads = c->num_kv_heads; g.block_size = c->block_size; g.cache_x = 8; g.kv_f16 = c->kv_dtype == 0;
      g.units = (nq + 2 * nkv) / 2;
      int upw = upw_of(g.units);
      for (;; ++upw) {  // one tensor per wave: grow the pairs per wave until q, k and v fit the grid's waves
        g.wstart[0] = 0;
        for (int i = 0; i < 3; ++i) g.wstart[i + 1] = g.wstart[i] + ((g.nrows[i] >> 1) + upw - 1) / upw;
        if (g.wstart[3] <= waves) break;
      }
      if (upw > 64) rc = -3;
      g.units_per_wave = upw;
    }
    for (int j = 1; j <= 2; ++j) {  // attention splits, merge
      ph[j].kind = j == 1 ? PH_ATTN : PH_MERGE; ph[j].group = G; ph[j].kv_f16 = c->kv_dtype == 0; ph[j].t = t;
      ph[j].t.k_cache = (const uint16_t *)L.k_cache; ph[j].t.v_cache = (const uint16_t *)L.v_cache;
    }
    for (int j = 3; j <= 5; j += 2) {  // o_proj, down: h = h * s + W x
      GemvArgs &g = ph[j].g;
      const mrs_dec_mat_c &w = j == 3 ? L.o : L.down;
      ph[j].kind = PH_RESID;
      if (!make_mat(g.m[0], w.planes, w.type, w.n, w.k) || w.n != d || w.k != (j == 3 ? nq : c->ff)) rc = -1;
      g.nrows[0] = d; g.K = (int)w.k; g.x = j == 3 ? c->attn : c->act; g.ldx = g.K; g.eps = c->eps; g.out = c->h; g.out_stride = d; g.resid_scale = c->resid_scale;
      g.units = d; g.units_per_wave = upw_of(d);
    }
    {  // gate / up
      GemvArgs &g = ph[4].g;
      ph[4].kind = PH_GLU;
      if (!make_mat(g.m[0], L.gate.planes, L.gate.type, L.gate.n, L.gate.k) || !make_mat(g.m[1], L.up.planes, L.up.type, L.up.n, L.up.k)) rc = -1;
      if (L.gate.type != L.up.type || L.gate.n != c->ff || L.up.n != c->ff || L.gate.k != d || L.up.k != d) rc = -1;
      g.nrows[0] = g.nrows[1] = c->ff; g.K = d; g.x = c->h; g.ldx = d; g.norm_w = L.ffn_norm; g.eps = c->eps; g.out = c->act; g.out_stride = c->ff;
      g.units = c->ff; g.units_per_wave = upw_of(c->ff);
    }
  }
  {  // final norm + lm_head
    PhaseDesc &e = tab[np - 1];
    GemvArgs &g = e.g;
    e.kind = PH_STORE;
    if (!make_mat(g.m[0], c->lm_head.planes, c->lm_head.type, c->lm_head.n, c->lm_head.k) || c->lm_head.n != c->vocab || c->lm_head.k != d) rc = -1;
    g.nrows[0] = c->vocab; g.K = d; g.x = c->h; g.ldx = d; g.norm_w = c->final_norm; g.eps = c->eps; g.out = c->logits; g.out_stride = c->vocab;
    g.units = c->vocab; g.units_per_wave = upw_of(c->vocab);
  }
  size_t lds = act_bytes(std::max(std::max(d, c->ff), nq), 1);
  if (std::max(lds, (size_t)NW * 2 * 160 * 4) > 158 * 1024) rc = rc ? rc : -3;
  if (!rc && hipMemcpy(device_table, tab, (size_t)np * sizeof(PhaseDesc), hipMemcpyHostToDevice) != hipSuccess) rc = -1;
  free(tab);
  return rc;
}
// Phases [phase_begin, phase_end) of one decode step for ONE sequence in a single launch.  max_k = the longest GEMV row of the model
// (max(hidden, ffn, heads * 128)): sizes the activation image in LDS.
extern "C" int mrs_dec_step(const void *device_table, int num_layers, int max_k, void *sync, int phase_begin, int phase_end, void *stream) {
  const int np = 2 + 6 * num_layers;
  if (!device_table || !sync || phase_begin < 0 || phase_end > np || phase_begin >= phase_end) return -1;
  StepArgs a{(const PhaseDesc *)device_table, (unsigned *)sync, phase_begin, phase_end};
  size_t lds = std::max(act_bytes(max_k, 1), (size_t)NW * 2 * 160 * 4);
  lds = (lds + 15) & ~(size_t)15;
  if (lds > 158 * 1024) return -3;
#ifdef MRS_DEC_NO_STEP
  (void)a; (void)stream;
  return -3;
#else
  hipStream_t s = (hipStream_t)stream;
  auto kern = dec_step_kernel<1>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024); attr = true; }
  if (phase_end - phase_begin > 1 && hipMemsetAsync(sync, 0, 8, s) != hipSuccess) return -1;
  hipLaunchKernelGGL(kern, dim3(dec_step_grid()), dim3(NT), lds, s, a);
  return 0;
#endif
}
