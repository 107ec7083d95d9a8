// host/runtime.cpp -- host-side model runner for the quantized decode path (C++ because the reference host
// is compiled Rust and no Rust toolchain exists here; see INTEGRATION.md for the Rust-side binding).
//
// Mirrors, with the same names and call order:
//   trait QuantMethod ............ mistralrs-quant/src/lib.rs:1515-1688
//   GgufMatMul ................... mistralrs-quant/src/gguf/mod.rs:43-52,298-323,436-479
//   try_fused_quantized_{qkv,gate_up}  mistralrs-quant/src/lib.rs:1839,1949 (fusion policy: equal dtypes, no bias)
//   RmsNorm / Mlp / CausalSelfAttention / Block / Llama  mistralrs-core/src/models/llama.rs:68-157,243-260,487-518
//   PagedAttention::forward (decode)  mistralrs-core/src/paged_attention/layers/paged_attention.rs:1477-1561
//   tensor-name bindings ......... mistralrs-core/src/gguf/normal_bindings.rs:40-220
// Two execution modes over identical numerics:
//   use_fused = 0 : the reference's launch sequence through the drop-in C-ABI symbols (launch_mmvq_*,
//                   rotary_embedding_positions, reshape_and_cache, paged attention, add_rms_norm_*)
//   use_fused = 1 : the MI355X fused kernels of ext_decode.hip (5 launches per layer)
// The runner never allocates device memory: every buffer comes from the caller (mrs_llama_buffers).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <initializer_list>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/mistralrs_core.h"
#include "../../../include/mistralrs_paged_attn.h"
#include "../../../include/mistralrs_quant.h"
#include "../../../include/mrs_hip_ext.h"

namespace mrs_host {

static thread_local std::string g_last_error;
int fail(const char *fmt, ...) {  // external linkage: shared with ext_comm.hip
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return -1;
}

// ---------------------------------------------------------------------------------------------- dtypes
enum GgmlDType : int { F32 = 0, F16 = 1, Q4_0 = 2, Q4_1 = 3, Q5_0 = 6, Q5_1 = 7, Q8_0 = 8, Q8_1 = 9, Q2K = 10, Q3K = 11,
                       Q4K = 12, Q5K = 13, Q6K = 14, Q8K = 15, BF16 = 30 };
struct DTypeInfo { int id, block, bytes; const char *tag; };
static const DTypeInfo kTypes[] = {{F32, 1, 4, "f32"}, {F16, 1, 2, "f16"}, {Q4_0, 32, 18, "q4_0"}, {Q4_1, 32, 20, "q4_1"},
                                   {Q5_0, 32, 22, "q5_0"}, {Q5_1, 32, 24, "q5_1"}, {Q8_0, 32, 34, "q8_0"}, {Q2K, 256, 84, "q2_k"},
                                   {Q3K, 256, 110, "q3_k"}, {Q4K, 256, 144, "q4_k"}, {Q5K, 256, 176, "q5_k"},
                                   {Q6K, 256, 210, "q6_k"}, {BF16, 1, 2, "bf16"}};
static const DTypeInfo *type_info(int id) {
  for (const auto &t : kTypes) if (t.id == id) return &t;
  return nullptr;
}
static bool mmvq_supports(int id) { return id == Q4_0 || id == Q4_1 || id == Q5_0 || id == Q5_1 || id == Q8_0 || (id >= Q2K && id <= Q6K); }

constexpr int MATRIX_ROW_PADDING = 512;  // fast_mmvq.rs:21
constexpr int MMVQ_MAX_BATCH = 8;        // fast_mmvq.rs:52
static int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// candle `QTensor`: packed GGUF blocks [rows][cols/blk] resident in HBM
struct QTensor {
  const void *data = nullptr;
  int dtype = -1;
  int64_t rows = 0, cols = 0;
  mutable const void *qi = nullptr;  // MFMA-order copy for the exact-integer prompt GEMM of ext_gemm_qi.hip (mrs_gemm_qi_repack; caller-owned, optional)
  mutable const void *bf16 = nullptr;  // bf16 shadow copy [rows][cols] (mrs_dequantize at load time; caller-owned, optional): the selectable bf16 prompt path then runs plain library GEMMs (ext_gemm_lt.hip)
  size_t nbytes() const { const auto *t = type_info(dtype); return t ? (size_t)rows * (cols / t->block) * t->bytes : 0; }
};

// f32 activation view [b, k] + the per-model Q8_1 scratch (the `workspace_ensure` slot of fast_mmvq.rs:70-112)
struct Scratch { void *q8; size_t q8_bytes; };

// ---------------------------------------------------------------------------------------------- QuantMethod
class QuantMethod {
 public:
  virtual ~QuantMethod() = default;
  virtual const char *name() const = 0;
  // forward_raw: a [b, K] f32 -> out [b, N] f32
  virtual int forward_raw(const float *a, int b, float *out, const Scratch &ws, hipStream_t s) const = 0;
  virtual int embedding_forward_raw(const int32_t *ids, int n, float *out, hipStream_t s) const {
    (void)ids; (void)n; (void)out; (void)s;
    return fail("%s does not support `embedding_forward`. Please raise an issue.", name());
  }
  virtual const QTensor *get_qtensor() const { return nullptr; }
  virtual bool has_bias() const { return false; }
};

typedef void (*plain_fn)(const void *, const void *, void *, int, int, int, int, int, void *);
typedef void (*glu_fn)(const void *, const void *, const void *, void *, int, int, int, int, int, int, void *);
typedef void (*qkv_fn)(const void *, const void *, const void *, const void *, void *, void *, void *, int, int, int, int, int, int, void *);

static void *lookup_quiet(const char *sym) { return dlsym(RTLD_DEFAULT, sym); }  // optional entry points (absent from the host-emulation build)
static void *lookup(const std::string &sym) {  // resolved once per (symbol) and cached
  static std::vector<std::pair<std::string, void *>> cache;
  for (auto &e : cache) if (e.first == sym) return e.second;
  void *p = dlsym(RTLD_DEFAULT, sym.c_str());
  if (!p) { fail("missing HIP symbol %s (is libmistralrsquant.so loaded?)", sym.c_str()); return nullptr; }
  cache.emplace_back(sym, p);
  return p;
}

class GgufMatMul : public QuantMethod {
 public:
  explicit GgufMatMul(const QTensor &w) : w_(w) {}
  const char *name() const override { return "gguf"; }
  const QTensor *get_qtensor() const override { return &w_; }

  // try_fast_forward (gguf/mod.rs:298-323): b in 1..8 -> MMVQ.  (b > 8 -> GEMM: prefill path, ext_gemm)
  int forward_raw(const float *a, int b, float *out, const Scratch &ws, hipStream_t s) const override {
    if (!mmvq_supports(w_.dtype)) return fail("fast_mmvq: unsupported quant dtype %d", w_.dtype);
    if (b <= 0 || b > MMVQ_MAX_BATCH) return fail("fast_mmvq: batch size %d out of supported range 1..=%d", b, MMVQ_MAX_BATCH);
    const int k = (int)w_.cols, kp = pad_to(k, MATRIX_ROW_PADDING), stride = kp / 32;
    if ((size_t)b * stride * 36 > ws.q8_bytes) return fail("fast_mmvq: Q8_1 workspace too small");
    launch_mmvq_gguf_quantize_q8_1_f32(a, ws.q8, k, kp, b, s);
    auto fn = (plain_fn)lookup(std::string("launch_mmvq_gguf_") + type_info(w_.dtype)->tag + "_f32_plain");
    if (!fn) return -1;
    fn(w_.data, ws.q8, out, k, (int)w_.rows, stride, (int)w_.rows, b, s);
    return 0;
  }
  int embedding_forward_raw(const int32_t *ids, int n, float *out, hipStream_t s) const override {
    if (mrs_embedding(w_.data, w_.dtype, ids, out, (int)w_.cols, n, s) != 0)
      return fail("gguf embedding_forward: unsupported dtype %d", w_.dtype);
    return 0;
  }

 private:
  QTensor w_;
};

// try_fused_quantized_qkv (lib.rs:1949): one shared Q8_1 activation, one launch; requires equal dtypes
static int try_fused_quantized_qkv(const QuantMethod &q, const QuantMethod &k, const QuantMethod &v, const float *x, int b,
                                   float *qo, float *ko, float *vo, const Scratch &ws, hipStream_t s) {
  const QTensor *tq = q.get_qtensor(), *tk = k.get_qtensor(), *tv = v.get_qtensor();
  if (!tq || !tk || !tv || tq->dtype != tk->dtype || tq->dtype != tv->dtype || !mmvq_supports(tq->dtype)) return 1;  // not fused
  const int kk = (int)tq->cols, kp = pad_to(kk, MATRIX_ROW_PADDING), stride = kp / 32;
  launch_mmvq_gguf_quantize_q8_1_f32(x, ws.q8, kk, kp, b, s);
  auto fn = (qkv_fn)lookup(std::string("launch_mmvq_gguf_") + type_info(tq->dtype)->tag + "_f32_fused_qkv");
  if (!fn) return -1;
  fn(tq->data, tk->data, tv->data, ws.q8, qo, ko, vo, kk, (int)tq->rows, (int)tk->rows, (int)tv->rows, stride, b, s);
  return 0;
}

// try_fused_quantized_gate_up (lib.rs:1839)
static int try_fused_quantized_gate_up(const QuantMethod &g, const QuantMethod &u, const float *x, int b, float *out, int act,
                                       const Scratch &ws, hipStream_t s) {
  const QTensor *tg = g.get_qtensor(), *tu = u.get_qtensor();
  if (!tg || !tu || tg->dtype != tu->dtype || tg->rows != tu->rows || !mmvq_supports(tg->dtype)) return 1;
  const int kk = (int)tg->cols, kp = pad_to(kk, MATRIX_ROW_PADDING), stride = kp / 32;
  launch_mmvq_gguf_quantize_q8_1_f32(x, ws.q8, kk, kp, b, s);
  auto fn = (glu_fn)lookup(std::string("launch_mmvq_gguf_") + type_info(tg->dtype)->tag + "_f32_fused_glu");
  if (!fn) return -1;
  fn(tg->data, tu->data, ws.q8, out, kk, (int)tg->rows, stride, (int)tg->rows, b, act, s);
  return 0;
}

// ---------------------------------------------------------------------------------------------- model
struct Block {
  std::unique_ptr<GgufMatMul> q_proj, k_proj, v_proj, o_proj, gate_proj, up_proj, down_proj;
  // decode-layout copies (mrs_dec_repack) for the decode engine; planes == nullptr until mrs_llama_set_dec_tensor
  mrs_dec_mat dq{}, dk{}, dv{}, dout{}, dgate{}, dup{}, ddown{}, dgate_exps{}, dup_exps{}, ddown_exps{};
  // sparse MoE FFN (cfg.num_experts > 0): router [E][hidden] f32 + experts stacked along the row axis, [E * n][k] packed blocks
  const float *router = nullptr;
  QTensor gate_exps, up_exps, down_exps;
  const float *input_layernorm = nullptr, *post_attention_layernorm = nullptr;
  void *key_cache = nullptr, *value_cache = nullptr;
};

struct Workspace {  // carve-up of the caller's scratch
  float *h, *xn, *q, *k, *v, *attn, *proj, *act;
  void *y_a, *y_b;          // Q8_1 scratch (hidden-sized rows / ffn-sized rows)
  size_t y_a_bytes, y_b_bytes;
  void *attn_ws;            // v2 partials
  float *exp_sums, *max_logits;
  void *sample_scratch;
  int32_t *moe_ids;         // [B][top_k]
  float *moe_w;             // [B][top_k]
  void *moe_y;              // [top_k] Q8_1 rows of the selected experts' activations
  float *moe_act;           // decode engine: [top_k][intermediate] f32 activations of the selected experts
  void *moe_router_scratch; // [B][experts + 1]: logits + arrival ticket of the split router (mrs_moe_router_topk_norm_split; zero at rest)
  void *attn_img;           // decode engine: Q8_K activation image of the attention result (mrs_dec_attention / mrs_dec_attention_q8k -> mrs_dec_proj_img)
  unsigned *attn_ticket;    // decode engine: [max_batch][kv heads] arrival counters of mrs_dec_attention (zero at rest)
  void *act_img;            // decode engine, batched steps: the activation image of a phase, built once (mrs_dec_act_image -> mrs_dec_*_img); hidden or ffn width, <= 8 columns
};

class Llama {
 public:
  explicit Llama(const mrs_llama_config &c) : cfg(c), blocks(c.num_layers) {}
  mrs_llama_config cfg;
  std::vector<Block> blocks;
  std::unique_ptr<GgufMatMul> wte, lm_head;
  mrs_dec_mat dlm_head{};
  const float *ln_f = nullptr;
  mrs_llama_buffers bufs{};
  Workspace ws{};
  bool have_bufs = false;
  int attn2 = [] { const char *e = getenv("MRS_DEC_ATTN2"); return e ? atoi(e) : 1; }();  // decode engine attention: 1 = split kernel with the last-arriver merge + Q8_K image for o_proj (round 3, one launch), 0 = split + merge launches
  void *comm = nullptr;  // RCCL communicator (ext_comm.hip) when cfg.world_size > 1
  void *p2p = nullptr;   // one-shot peer-mailbox all-reduce (ext_p2p.hip) for decode-sized messages

  // SumAllReduce of a row-parallel output (distributed/layers.rs:965-975), in place on the runner's stream
  int all_reduce(float *buf, size_t count, hipStream_t s) const {
    if (cfg.world_size <= 1) return 0;
    if (p2p) {  // decode-sized messages: one-shot write-to-all-peers all-reduce (ext_p2p.hip); -2 = too large for the mailboxes
      const int rc = mrs_p2p_all_reduce_sum_f32(p2p, buf, count, s);
      if (rc != -2) return rc;
    }
    if (!comm) return fail("tensor parallel world_size %d but no communicator was set (mrs_llama_set_comm)", cfg.world_size);
    return mrs_comm_all_reduce_sum_f32(comm, buf, count, s);
  }

  static size_t align(size_t v) { return (v + 255) & ~(size_t)255; }
  static size_t workspace_bytes(const mrs_llama_config &c) {
    const size_t B = c.max_batch, d = c.hidden_size, nq = (size_t)c.num_heads * c.head_dim, nkv = (size_t)c.num_kv_heads * c.head_dim;
    const size_t ya = B * (pad_to((int)std::max(d, nq), MATRIX_ROW_PADDING) / 32) * 36;
    const size_t yb = B * (pad_to(c.intermediate_size, MATRIX_ROW_PADDING) / 32) * 36;
    const size_t parts = (size_t)mrs_decode_attention_max_splits(c.max_context_len);  // >= the reference's 512-token partitions
    size_t t = 0;
    t += align(B * d * 4) * 3;            // h, xn, proj
    t += align(B * nq * 4) * 2;           // q, attn
    t += align(B * nkv * 4) * 2;          // k, v
    t += align(B * (size_t)c.intermediate_size * 4);  // act
    t += align(ya) + align(yb);
    t += align(B * c.num_heads * parts * c.head_dim * 4) + 2 * align(B * c.num_heads * parts * 4);
    t += align(B * 8);
    t += align(mrs_dec_act_image_bytes((int)nq, (int)B)) + align(B * (size_t)c.num_kv_heads * 4);
    t += align(mrs_dec_act_image_bytes(std::max((int)d, (int)c.intermediate_size), (int)std::min<size_t>(B, 8)));  // act_img
    if (c.num_experts > 0) {
      const size_t k = std::max(1, (int)c.num_experts_per_tok);
      t += align(B * k * 4) * 2 + align(k * (pad_to(c.intermediate_size, MATRIX_ROW_PADDING) / 32) * 36);
      t += align(k * (size_t)c.intermediate_size * 4);
      t += align(mrs_moe_router_split_scratch_bytes((int)B, (int)c.num_experts));
    }
    return t + 4096;
  }

  int set_buffers(const mrs_llama_buffers &b) {
    if (b.workspace_bytes < workspace_bytes(cfg)) return fail("workspace too small: %zu < %zu", b.workspace_bytes, workspace_bytes(cfg));
    bufs = b;
    const size_t B = cfg.max_batch, d = cfg.hidden_size, nq = (size_t)cfg.num_heads * cfg.head_dim, nkv = (size_t)cfg.num_kv_heads * cfg.head_dim;
    char *p = (char *)(((uintptr_t)b.workspace + 255) & ~(uintptr_t)255);
    auto take = [&](size_t n) { char *r = p; p += align(n); return r; };
    ws.h = (float *)take(B * d * 4); ws.xn = (float *)take(B * d * 4); ws.proj = (float *)take(B * d * 4);
    ws.q = (float *)take(B * nq * 4); ws.attn = (float *)take(B * nq * 4);
    ws.k = (float *)take(B * nkv * 4); ws.v = (float *)take(B * nkv * 4);
    ws.act = (float *)take(B * (size_t)cfg.intermediate_size * 4);
    ws.y_a_bytes = B * (pad_to((int)std::max(d, nq), MATRIX_ROW_PADDING) / 32) * 36;
    ws.y_b_bytes = B * (pad_to(cfg.intermediate_size, MATRIX_ROW_PADDING) / 32) * 36;
    ws.y_a = take(ws.y_a_bytes); ws.y_b = take(ws.y_b_bytes);
    const size_t parts = (size_t)mrs_decode_attention_max_splits(cfg.max_context_len);
    ws.attn_ws = take(B * cfg.num_heads * parts * cfg.head_dim * 4);
    ws.exp_sums = (float *)take(B * cfg.num_heads * parts * 4);
    ws.max_logits = (float *)take(B * cfg.num_heads * parts * 4);
    ws.sample_scratch = take(B * 8);
    ws.attn_img = take(mrs_dec_act_image_bytes((int)nq, (int)B));
    ws.attn_ticket = (unsigned *)take(B * (size_t)cfg.num_kv_heads * 4);
    ws.act_img = take(mrs_dec_act_image_bytes(std::max((int)d, (int)cfg.intermediate_size), (int)std::min<size_t>(B, 8)));
    if (cfg.num_experts > 0) {
      const size_t k = std::max(1, (int)cfg.num_experts_per_tok);
      ws.moe_ids = (int32_t *)take(B * k * 4); ws.moe_w = (float *)take(B * k * 4);
      ws.moe_y = take(k * (pad_to(cfg.intermediate_size, MATRIX_ROW_PADDING) / 32) * 36);
      ws.moe_act = (float *)take(k * (size_t)cfg.intermediate_size * 4);
      ws.moe_router_scratch = take(mrs_moe_router_split_scratch_bytes((int)B, (int)cfg.num_experts));
    }
    // zero once: Q8_1 padding blocks beyond K are never written by the fused epilogues; sample scratch must start at 0
    if (hipMemset(b.workspace, 0, b.workspace_bytes) != hipSuccess) return fail("hipMemset(workspace) failed");
    have_bufs = true;
    return 0;
  }

  int check_ready(int b) const {
    if (!have_bufs) return fail("mrs_llama_set_buffers was not called");
    if (b <= 0 || b > cfg.max_batch || b > MMVQ_MAX_BATCH) return fail("decode batch %d out of range 1..=%d", b, cfg.max_batch);
    if (!wte || !lm_head || !ln_f) return fail("model is missing token_embd / output / output_norm");
    for (size_t i = 0; i < blocks.size(); ++i) {
      const Block &bl = blocks[i];
      const bool ffn = cfg.num_experts > 0 ? (bl.router && bl.gate_exps.data && bl.up_exps.data && bl.down_exps.data)
                                           : (bl.gate_proj && bl.up_proj && bl.down_proj);
      if (!bl.q_proj || !bl.k_proj || !bl.v_proj || !bl.o_proj || !ffn || !bl.input_layernorm || !bl.post_attention_layernorm)
        return fail("layer %zu is missing tensors", i);
      if (!bl.key_cache || !bl.value_cache) return fail("layer %zu has no KV cache", i);
    }
    return 0;
  }

  bool fused_ok() const {
    if (cfg.use_fused != 1 || !cfg.rope_interleaved) return false;
    auto hot = [](const std::unique_ptr<GgufMatMul> &m) { return mrs_decode_gemv_supported(m->get_qtensor()->dtype) != 0; };
    if (!hot(lm_head)) return false;
    for (const Block &bl : blocks) {
      if (!hot(bl.q_proj) || !hot(bl.k_proj) || !hot(bl.v_proj) || !hot(bl.o_proj)) return false;
      if (cfg.intermediate_size % 32) return false;
      if (cfg.num_experts > 0) {
        if (!mrs_decode_gemv_supported(bl.gate_exps.dtype) || !mrs_decode_gemv_supported(bl.down_exps.dtype) || bl.gate_exps.dtype != bl.up_exps.dtype) return false;
        continue;
      }
      if (!hot(bl.gate_proj) || !hot(bl.down_proj)) return false;
      if (bl.gate_proj->get_qtensor()->dtype != bl.up_proj->get_qtensor()->dtype) return false;
    }
    return true;
  }

  // PagedAttention::forward, decode branch (run_decode): attention over the cache that already holds this token
  int paged_attention_decode(const Block &bl, int b, hipStream_t s) const {
    const int hd = cfg.head_dim, bs = cfg.block_size, kvh = cfg.num_kv_heads;
    const int kv_block_stride = kvh * hd * bs, kv_head_stride = hd * bs;
    const int eff_max = std::min(cfg.max_blocks_per_seq * bs, cfg.max_context_len);
    const int parts = (eff_max + 511) / 512;
    const bool use_v1 = (parts == 1 || b * cfg.num_heads > 512);  // paged_attention.rs:302-307
    mrs_paged_attention_f32_bf16(use_v1 ? 0 : 1, ws.attn, ws.exp_sums, ws.max_logits, ws.attn_ws, ws.q, bl.key_cache, bl.value_cache,
                                 nullptr, kvh, 1.0f / sqrtf((float)hd), 1.0f, bufs.block_tables, bufs.context_lens, bs, eff_max, b,
                                 cfg.num_heads, hd, cfg.max_blocks_per_seq, cfg.num_heads * hd, kv_block_stride, kv_head_stride, s, nullptr);
    return 0;
  }

  // ---- reference launch sequence (models/llama.rs:487-518 driving the C-ABI kernels one by one)
  int forward_unfused(int b, hipStream_t s) const {
    const int d = cfg.hidden_size, hd = cfg.head_dim, nq = cfg.num_heads * hd, nkv = cfg.num_kv_heads * hd;
    const Scratch sa{ws.y_a, ws.y_a_bytes}, sb{ws.y_b, ws.y_b_bytes};
    const int64_t st = (int64_t)(intptr_t)s;
    if (wte->embedding_forward_raw(bufs.input_ids, b, ws.h, s)) return -1;
    mrs_rms_norm_f32(ws.h, blocks[0].input_layernorm, ws.xn, b, d, cfg.rms_eps, st);
    for (size_t li = 0; li < blocks.size(); ++li) {
      const Block &bl = blocks[li];
      // CausalSelfAttention::forward (llama.rs:68-157): qkv_projections -> rotary -> paged attention -> o_proj
      int rc = try_fused_quantized_qkv(*bl.q_proj, *bl.k_proj, *bl.v_proj, ws.xn, b, ws.q, ws.k, ws.v, sa, s);
      if (rc < 0) return -1;
      if (rc == 1) {  // mixed dtypes (Q4_K_M: attn_v is often Q6_K): three GEMVs, as the reference falls back to
        if (bl.q_proj->forward_raw(ws.xn, b, ws.q, sa, s) || bl.k_proj->forward_raw(ws.xn, b, ws.k, sa, s) ||
            bl.v_proj->forward_raw(ws.xn, b, ws.v, sa, s)) return -1;
      }
      rotary_embedding_positions(ws.q, ws.k, (void *)bufs.cos_table, (void *)bufs.sin_table, bufs.positions,
                                 cfg.rope_interleaved ? 0 : 1, hd, b, cfg.rot_dim / 2, cfg.max_context_len, cfg.num_heads,
                                 cfg.num_kv_heads, nq, nkv, 2, st);
      reshape_and_cache(ws.k, ws.v, bl.key_cache, bl.value_cache, bufs.slot_mapping, b, cfg.num_kv_heads, hd, cfg.block_size, 8,
                        nkv, nkv, s, 2, 1, nullptr, nullptr);
      paged_attention_decode(bl, b, s);
      if (bl.o_proj->forward_raw(ws.attn, b, ws.proj, sa, s) || all_reduce(ws.proj, (size_t)b * d, s)) return -1;
      // Block::forward (llama.rs:243-260): x = attn + residual ; mlp(rms_norm(x)) + x
      add_rms_norm_f32(ws.proj, ws.h, bl.post_attention_layernorm, ws.h, ws.xn, b, d, cfg.rms_eps, st);
      // Mlp::forward -> quantized_ffn (ops.rs:5036): fused gate/up + act, then down
      rc = try_fused_quantized_gate_up(*bl.gate_proj, *bl.up_proj, ws.xn, b, ws.act, 0, sa, s);
      if (rc < 0) return -1;
      if (rc == 1) return fail("gate/up with different dtypes are not supported yet");
      if (bl.down_proj->forward_raw(ws.act, b, ws.proj, sb, s) || all_reduce(ws.proj, (size_t)b * d, s)) return -1;
      const float *next_norm = li + 1 < blocks.size() ? blocks[li + 1].input_layernorm : ln_f;
      add_rms_norm_f32(ws.proj, ws.h, next_norm, ws.h, ws.xn, b, d, cfg.rms_eps, st);
    }
    return lm_head->forward_raw(ws.xn, b, bufs.logits, sa, s);
  }

  // ---- MI355X fused sequence: 5 launches per layer (+ attention reduce / quantize)
  // experiment hook (scripts/exp): MRS_ABLATE bit mask skips kernel classes so their in-graph cost can be measured
  static int ablate() { static int v = -1; if (v < 0) { const char *e = getenv("MRS_ABLATE"); v = e ? atoi(e) : 0; } return v; }
  int forward_fused(int b, hipStream_t s) const {
    const int ab = ablate();
    const float rs = 1.0f / (float)std::max(1, (int)cfg.world_size);
    const int d = cfg.hidden_size, hd = cfg.head_dim, nq = cfg.num_heads * hd, nkv = cfg.num_kv_heads * hd, ff = cfg.intermediate_size;
    const int stride_q = pad_to(nq, MATRIX_ROW_PADDING) / 32, stride_f = pad_to(ff, MATRIX_ROW_PADDING) / 32;
    if (wte->embedding_forward_raw(bufs.input_ids, b, ws.h, s)) return -1;
    for (const Block &bl : blocks) {
      const QTensor *q = bl.q_proj->get_qtensor(), *k = bl.k_proj->get_qtensor(), *v = bl.v_proj->get_qtensor();
      if (!(ab & 1) && mrs_decode_qkv(q->data, k->data, v->data, q->dtype, k->dtype, v->dtype, nq, nkv, nkv, d, ws.h, bl.input_layernorm,
                         cfg.rms_eps, ws.q, bl.key_cache, bl.value_cache, bufs.slot_mapping, bufs.positions, bufs.cos_table,
                         bufs.sin_table, hd, cfg.rot_dim / 2, cfg.num_kv_heads, cfg.block_size, b, s))
        return fail("mrs_decode_qkv refused the layer");
      if (!(ab & 2)) {
        const int bs = cfg.block_size, kvh = cfg.num_kv_heads;
        const int eff_max = std::min(cfg.max_blocks_per_seq * bs, cfg.max_context_len);
        if (mrs_decode_attention_q8_1_f32_bf16(ws.y_a, stride_q, ws.exp_sums, ws.max_logits, ws.attn_ws, ws.q, bl.key_cache, bl.value_cache, kvh,
                                               1.0f / sqrtf((float)hd), bufs.block_tables, bufs.context_lens, bs, eff_max, b, cfg.num_heads, hd,
                                               cfg.max_blocks_per_seq, cfg.num_heads * hd, kvh * hd * bs, hd * bs, s)) {
          paged_attention_decode(bl, b, s);  // shapes outside the split kernel: reference-partitioned attention + quantize
          mrs_quantize_rows_q8_1(ws.attn, ws.y_a, nq, stride_q, b, s);
        }
      }
      const QTensor *o = bl.o_proj->get_qtensor();
      // TP: h <- h / world + W_o . attn on every rank, then ONE sum all-reduce of h gives h + sum of the partials (division by a
      // power of two is exact), so the residual add stays fused and nothing else crosses GPUs
      if (!(ab & 4) && (mrs_decode_proj_scaled(o->data, o->dtype, d, nq, ws.y_a, stride_q, ws.h, d, rs, b, s) || all_reduce(ws.h, (size_t)b * d, s))) return fail("o_proj failed: %s", g_last_error.c_str());
      if (cfg.num_experts > 0) {
        // SparseMoeBlock::forward (models/mixtral.rs:280-304): router on the normed hidden state, then per token the top-k experts'
        // fused gate/up (+SiLU*up -> Q8_1) and down GEMVs, accumulated into h with the renormalised routing weights.  Expert ids and
        // weights stay on the device (the kernels read them), so the step is graph-capturable.
        if (cfg.world_size > 1) return fail("tensor-parallel MoE runs on the decode engine (decode_engine / use_fused = 2), not on the round-1 fused kernels");
        const int E = cfg.num_experts, tk = cfg.num_experts_per_tok;
        const size_t g_stride = bl.gate_exps.nbytes() / E, d_stride = bl.down_exps.nbytes() / E;
        const size_t y_row = (size_t)stride_f * 36;
        mrs_rms_norm_f32(ws.h, bl.post_attention_layernorm, ws.xn, b, d, cfg.rms_eps, (int64_t)(intptr_t)s);
        if (mrs_moe_router_topk(ws.xn, bl.router, b, E, d, tk, 1, ws.moe_ids, ws.moe_w, nullptr, s)) return fail("moe router refused (experts %d, top-k %d)", E, tk);
        for (int t = 0; t < b; ++t) {
          float *ht = ws.h + (size_t)t * d;
          for (int sl = 0; sl < tk; ++sl)  // all of a token's expert activations are computed before h changes
            if (mrs_moe_decode_gate_up(bl.gate_exps.data, bl.up_exps.data, g_stride, ws.moe_ids + t * tk + sl, bl.gate_exps.dtype, ff, d, ht,
                                       bl.post_attention_layernorm, cfg.rms_eps, 0, (char *)ws.moe_y + sl * y_row, stride_f, s))
              return fail("moe gate/up refused");
          for (int sl = 0; sl < tk; ++sl)
            if (mrs_moe_decode_down(bl.down_exps.data, d_stride, ws.moe_ids + t * tk + sl, ws.moe_w + t * tk + sl, bl.down_exps.dtype, d, ff,
                                    (char *)ws.moe_y + sl * y_row, stride_f, ht, s))
              return fail("moe down refused");
        }
        continue;
      }
      const QTensor *g = bl.gate_proj->get_qtensor(), *u = bl.up_proj->get_qtensor(), *dn = bl.down_proj->get_qtensor();
      if (!(ab & 8) && mrs_decode_gate_up(g->data, u->data, g->dtype, ff, d, ws.h, bl.post_attention_layernorm, cfg.rms_eps, 0, ws.y_b, stride_f, b, s))
        return fail("mrs_decode_gate_up refused");
      if (!(ab & 16) && (mrs_decode_proj_scaled(dn->data, dn->dtype, d, ff, ws.y_b, stride_f, ws.h, d, rs, b, s) || all_reduce(ws.h, (size_t)b * d, s))) return fail("down_proj failed: %s", g_last_error.c_str());
    }
    const QTensor *lm = lm_head->get_qtensor();
    if (!(ab & 32) && mrs_decode_norm_proj(lm->data, lm->dtype, cfg.vocab_size, d, ws.h, ln_f, cfg.rms_eps, bufs.logits, cfg.vocab_size, b, s))
      return fail("mrs_decode_norm_proj refused");
    return 0;
  }

  // ---- decode engine (ext_dec.hip): the reference CPU path's arithmetic -- f32 activations quantized to Q8_K / Q8_0 inside the GEMV
  //      kernels (candle QMatMul::forward, gguf/mod.rs:465-478), f32 norm / RoPE / SiLU / softmax -- over decode-layout weights
  bool engine_ok() const {
    if (cfg.use_fused != 2 || cfg.head_dim != 128 || cfg.block_size != 32 || (cfg.head_dim & 1)) return false;
    if (!cfg.rope_interleaved && cfg.rot_dim != cfg.head_dim) return false;  // rotate-half RoPE: the pair-order layout of q / k covers full rotary only
    const int g = cfg.num_heads / cfg.num_kv_heads;
    if (g != 1 && g != 2 && g != 4 && g != 8) return false;
    if (!dlm_head.planes) return false;
    for (const Block &bl : blocks) {
      if (!bl.dq.planes || !bl.dk.planes || !bl.dv.planes || !bl.dout.planes) return false;
      if ((bl.dq.type == Q8_0) != (bl.dk.type == Q8_0) || (bl.dq.type == Q8_0) != (bl.dv.type == Q8_0)) return false;  // one activation format per phase
      if (cfg.num_experts > 0) {
        if (!bl.dgate_exps.planes || !bl.dup_exps.planes || !bl.ddown_exps.planes || bl.dgate_exps.type != bl.dup_exps.type) return false;
      } else if (!bl.dgate.planes || !bl.dup.planes || !bl.ddown.planes || bl.dgate.type != bl.dup.type) return false;
    }
    return true;
  }
  // chained (round 6, batch 1): ws.h already holds the embedding row of input_ids (the previous step's mrs_sample_advance_embed, or mrs_llama_embed_state after the host
  // changed the state) and lm_head folds the arg-max into its epilogue: the captured step is two launches shorter (no embedding_kernel, no argmax_partial_kernel)
  int forward_engine(int b, hipStream_t s, bool chained = false) const {
    const float rs = 1.0f / (float)std::max(1, (int)cfg.world_size);
    const int d = cfg.hidden_size, hd = cfg.head_dim, nq = cfg.num_heads * hd, ff = cfg.intermediate_size, kvd = cfg.kv_f16 ? 0 : 1;
    const int bs = cfg.block_size, kvh = cfg.num_kv_heads;
    const int eff_max = std::min(cfg.max_blocks_per_seq * bs, cfg.max_context_len);
    if (chained && b != 1) return fail("chained decode step: batch 1 only");
    if (!chained && wte->embedding_forward_raw(bufs.input_ids, b, ws.h, s)) return -1;
    // Batched steps: every GEMV workgroup would normalise + quantize all b activation columns itself (256 times the same work, 17-25 us of a 40-60 us launch at
    // b = 8); from MRS_DEC_IMG_MIN_B columns on the image of a phase is built once by mrs_dec_act_image (b workgroups) and the GEMVs copy it -- same bytes.
    static const int img_min_b = [] { const char *e = getenv("MRS_DEC_IMG_MIN_B"); return e ? atoi(e) : 2; }();
    const bool imgb = b >= img_min_b && b >= 2 && b <= 8;
    auto image = [&](const float *x, int ldx, const float *nw, int k, int wtype) { return mrs_dec_act_image(x, ldx, nw, cfg.rms_eps, k, wtype, b, ws.act_img, s); };
    // (round 6) batched steps on the matrix cores (ext_dec_mm.hip): the same launches on the MFMA-order copy of the weights the exact prompt path keeps (QTensor::qi) -- integer
    // dots on v_mfma_i32_32x32x32_i8 instead of 535 VALU per 8-column tile, the same bits.  MRS_DEC_MM=0 keeps the vector-ALU kernels; MRS_DEC_MM_MIN_B: smallest batch that takes it.
    static const int mm_min_b = [] { const char *e = getenv("MRS_DEC_MM"); if (e && atoi(e) == 0) return 1 << 30; const char *m = getenv("MRS_DEC_MM_MIN_B"); return m ? atoi(m) : 3; }();  // measured (MI355X, 8B Q4_K_M): batch 2 / 3 / 4 / 8 = 760 / 1128 / 1518 / 2722 tok/s here, 824 / 1103 / 1315 / 1685 on the vector ALU
    const bool mmb = imgb && b >= mm_min_b;
    auto qi_of = [](const std::unique_ptr<GgufMatMul> &l) -> const void * { return l && l->get_qtensor() ? l->get_qtensor()->qi : nullptr; };
    auto mm_ok = [&](const void *qi, int type, int k) { return mmb && qi && mrs_dec_mm_supported(type, k, b); };
    for (const Block &bl : blocks) {
      // rotate-half RoPE: the caller registered q / k decode planes in pair order (mrs_dec_qkv_neox; llama.py permutes the rows before the repack)
      if (imgb && cfg.rope_interleaved && mm_ok(qi_of(bl.q_proj), bl.dq.type, d) && mm_ok(qi_of(bl.k_proj), bl.dk.type, d) && mm_ok(qi_of(bl.v_proj), bl.dv.type, d)) {
        if (image(ws.h, d, bl.input_layernorm, d, bl.dq.type) ||
            mrs_dec_mm_qkv(qi_of(bl.q_proj), bl.dq.type, (int)bl.dq.n, qi_of(bl.k_proj), bl.dk.type, (int)bl.dk.n, qi_of(bl.v_proj), bl.dv.type, (int)bl.dv.n, d, ws.act_img, ws.q,
                           bl.key_cache, bl.value_cache, bufs.slot_mapping, bufs.positions, bufs.cos_table, bufs.sin_table, hd, cfg.rot_dim / 2, kvh, bs, kvd, b, s))
          return fail("mrs_dec_mm_qkv refused the layer");
      } else if (imgb) {
        if (image(ws.h, d, bl.input_layernorm, d, bl.dq.type) ||
            mrs_dec_qkv_img(&bl.dq, &bl.dk, &bl.dv, ws.act_img, ws.q, bl.key_cache, bl.value_cache, bufs.slot_mapping, bufs.positions, bufs.cos_table, bufs.sin_table, hd,
                            cfg.rot_dim / 2, kvh, bs, kvd, b, cfg.rope_interleaved ? 0 : 1, s))
          return fail("mrs_dec_qkv_img refused the layer");
      } else if ((cfg.rope_interleaved ? mrs_dec_qkv : mrs_dec_qkv_neox)(&bl.dq, &bl.dk, &bl.dv, ws.h, d, bl.input_layernorm, cfg.rms_eps, ws.q, bl.key_cache, bl.value_cache,
                                                                  bufs.slot_mapping, bufs.positions, bufs.cos_table, bufs.sin_table, hd, cfg.rot_dim / 2, kvh, bs, kvd, b, s))
        return fail("mrs_dec_qkv refused the layer");
      if (attn2) {
        // one launch: splits + last-arriver merge; even GQA groups hand o_proj the Q8_K image of the result (Q8_0 weights take Q8_0 activations: f32 result)
        const bool want_img = bl.dout.type != 8 && (cfg.num_heads / kvh) % 2 == 0 && mrs_dec_act_image_bytes(nq, b) <= mrs_dec_proj_img_max_bytes();
        const int rc2 = mrs_dec_attention(want_img ? nullptr : ws.attn, want_img ? ws.attn_img : nullptr, ws.attn_ticket, (float *)ws.attn_ws, ws.max_logits, ws.exp_sums, ws.q,
                                          bl.key_cache, bl.value_cache, kvh, 1.0f / sqrtf((float)hd), bufs.block_tables, bufs.context_lens, bs, eff_max, b, cfg.num_heads, hd,
                                          cfg.max_blocks_per_seq, nq, kvh * hd * bs, hd * bs, kvd, cfg.sliding_window, s);
        if (rc2 < 0) return fail("mrs_dec_attention refused the shape");
        const int prc = rc2 == 1 ? (mm_ok(qi_of(bl.o_proj), bl.dout.type, nq) ? mrs_dec_mm_proj(qi_of(bl.o_proj), bl.dout.type, d, nq, ws.attn_img, ws.h, d, 1, rs, b, s)
                                                                                : mrs_dec_proj_img(&bl.dout, d, ws.attn_img, ws.h, d, 1, rs, b, s))
                                 : mrs_dec_proj(&bl.dout, d, nullptr, ws.attn, nq, nullptr, 0.f, ws.h, d, 1, rs, nullptr, b, s);
        if (prc || all_reduce(ws.h, (size_t)b * d, s)) return fail("o_proj failed (%d): %s", prc, g_last_error.c_str());
      } else {
        if (mrs_decode_attention_f32_f32_bf16(ws.attn, ws.exp_sums, ws.max_logits, ws.attn_ws, ws.q, bl.key_cache, bl.value_cache, kvh, 1.0f / sqrtf((float)hd),
                                              bufs.block_tables, bufs.context_lens, bs, eff_max, b, cfg.num_heads, hd, cfg.max_blocks_per_seq, nq, kvh * hd * bs,
                                              hd * bs, kvd, s))
          return fail("mrs_decode_attention_f32 refused the shape");
        // TP: h <- h / world + W_o . attn on every rank, then ONE sum all-reduce of h (the residual add stays fused, as in forward_fused)
        if (mrs_dec_proj(&bl.dout, d, nullptr, ws.attn, nq, nullptr, 0.f, ws.h, d, 1, rs, nullptr, b, s) || all_reduce(ws.h, (size_t)b * d, s))
          return fail("o_proj failed: %s", g_last_error.c_str());
      }
      if (cfg.num_experts > 0) {
        // SparseMoeBlock::forward (models/mixtral.rs:280-304): router on the normed hidden state; per token the top-k experts' gate/up then down,
        // accumulated into h with the renormalised routing weights; expert ids / weights stay on the device
        // TP (moe/experts/mod.rs:332-339): every expert is sharded on the ffn dimension like a dense FFN; h <- h / world + sum of the local experts' partial
        // outputs, then ONE all-reduce per MoE block (the router is replicated: every rank picks the same experts)
        const int E = cfg.num_experts, tk = cfg.num_experts_per_tok;
        // (round 6) E workgroups per token + the last arriver's top-k: the same ids and weights as the one-workgroup router, bit for bit, without 8 router rows through one CU
        static const bool split_router = [] { const char *e = getenv("MRS_MOE_ROUTER_SPLIT"); return !e || atoi(e) != 0; }();
        const int rrc = split_router ? mrs_moe_router_topk_norm_split(ws.h, bl.post_attention_layernorm, cfg.rms_eps, bl.router, b, E, d, tk, 1, ws.moe_ids, ws.moe_w, ws.moe_router_scratch, s)
                                     : mrs_moe_router_topk_norm(ws.h, bl.post_attention_layernorm, cfg.rms_eps, bl.router, b, E, d, tk, 1, ws.moe_ids, ws.moe_w, s);  // norm inside the router
        if (rrc == -3) {
          mrs_rms_norm_f32(ws.h, bl.post_attention_layernorm, ws.xn, b, d, cfg.rms_eps, (int64_t)(intptr_t)s);
          if (mrs_moe_router_topk(ws.xn, bl.router, b, E, d, tk, 1, ws.moe_ids, ws.moe_w, nullptr, s)) return fail("moe router refused (experts %d, top-k %d)", E, tk);
        } else if (rrc) return fail("moe router refused (experts %d, top-k %d)", E, tk);
        for (int t = 0; t < b; ++t) {
          float *ht = ws.h + (size_t)t * d;
          // all of a token's expert activations are computed before h changes: one launch for the top-k experts (-3: shapes that do not split -> one per expert)
          const int grc = mrs_dec_gate_up_topk(&bl.dgate_exps, &bl.dup_exps, ff, ws.moe_ids + t * tk, tk, ht, bl.post_attention_layernorm, cfg.rms_eps, 0, ws.moe_act, ff, s);
          if (grc != 0 && grc != -3) return fail("moe gate/up refused (%d)", grc);
          for (int sl = 0; grc == -3 && sl < tk; ++sl)
            if (mrs_dec_gate_up(&bl.dgate_exps, &bl.dup_exps, ff, ws.moe_ids + t * tk + sl, ht, d, bl.post_attention_layernorm, cfg.rms_eps, 0,
                                ws.moe_act + (size_t)sl * ff, ff, 1, s))
              return fail("moe gate/up refused");
          // top-2 (Mixtral): both experts' down projections in one launch, same roundings as two accumulating launches
          if (tk == 2 && grc == 0 && mrs_dec_proj_top2(&bl.ddown_exps, d, ws.moe_ids + t * tk, ws.moe_act, ff, ht, rs, ws.moe_w + t * tk, s) == 0) continue;
          for (int sl = 0; sl < tk; ++sl)
            if (mrs_dec_proj(&bl.ddown_exps, d, ws.moe_ids + t * tk + sl, ws.moe_act + (size_t)sl * ff, ff, nullptr, 0.f, ht, d, 1, sl == 0 ? rs : 1.0f, ws.moe_w + t * tk + sl, 1, s))
              return fail("moe down refused");
        }
        if (all_reduce(ws.h, (size_t)b * d, s)) return fail("moe all-reduce failed: %s", g_last_error.c_str());
        continue;
      }
      if (imgb) {
        const bool mmg = mm_ok(qi_of(bl.gate_proj), bl.dgate.type, d) && qi_of(bl.up_proj);
        if (image(ws.h, d, bl.post_attention_layernorm, d, bl.dgate.type) ||
            (mmg ? mrs_dec_mm_gate_up(qi_of(bl.gate_proj), qi_of(bl.up_proj), bl.dgate.type, ff, d, ws.act_img, 0, ws.act, ff, b, s)
                 : mrs_dec_gate_up_img(&bl.dgate, &bl.dup, ff, ws.act_img, 0, ws.act, ff, b, s)))
          return fail("mrs_dec_gate_up_img refused");
      } else if (mrs_dec_gate_up(&bl.dgate, &bl.dup, ff, nullptr, ws.h, d, bl.post_attention_layernorm, cfg.rms_eps, 0, ws.act, ff, b, s)) return fail("mrs_dec_gate_up refused");
      const int drc = imgb ? (image(ws.act, ff, nullptr, ff, bl.ddown.type) ||
                              (mm_ok(qi_of(bl.down_proj), bl.ddown.type, ff) ? mrs_dec_mm_proj(qi_of(bl.down_proj), bl.ddown.type, d, ff, ws.act_img, ws.h, d, 1, rs, b, s)
                                                                               : mrs_dec_proj_img(&bl.ddown, d, ws.act_img, ws.h, d, 1, rs, b, s)))
                                                    : mrs_dec_proj(&bl.ddown, d, nullptr, ws.act, ff, nullptr, 0.f, ws.h, d, 1, rs, nullptr, b, s);
      if (drc || all_reduce(ws.h, (size_t)b * d, s)) return fail("down_proj failed: %s", g_last_error.c_str());
    }
    const int lrc = chained ? mrs_dec_proj_argmax(&dlm_head, cfg.vocab_size, ws.h, d, ln_f, cfg.rms_eps, bufs.logits, cfg.vocab_size, ws.sample_scratch, s)
                    : imgb ? (image(ws.h, d, ln_f, d, dlm_head.type) ||
                              (mm_ok(qi_of(lm_head), dlm_head.type, d) ? mrs_dec_mm_proj(qi_of(lm_head), dlm_head.type, cfg.vocab_size, d, ws.act_img, bufs.logits, cfg.vocab_size, 0, 1.0f, b, s)
                                                                        : mrs_dec_proj_img(&dlm_head, cfg.vocab_size, ws.act_img, bufs.logits, cfg.vocab_size, 0, 1.0f, b, s)))
                           : mrs_dec_proj(&dlm_head, cfg.vocab_size, nullptr, ws.h, d, ln_f, cfg.rms_eps, bufs.logits, cfg.vocab_size, 0, 1.0f, nullptr, b, s);
    if (lrc) return fail("lm_head refused");
    return 0;
  }

  // ---- prefill: T prompt tokens of one sequence through the bf16-MFMA GEMMs (role of fast_mmq::* + the prompt branch
  //      of PagedAttention::forward).  Buffers are carved from the caller's prefill workspace.
  // prompts longer than this use the 256-row-tile GEMM over bf16 slabs (its L1-bypassing A loads beat the 128-row f32 kernel even when
  // half of the tile rows are empty: TTFT 16.3 -> 10.1 ms at T = 128, 19.5 -> 8.3 ms at T = 64, 18.5 -> 7.7 ms at T = 32)
  static int prefill_big_min() { static int v = -1; if (v < 0) { const char *e = getenv("MRS_PREFILL_BIG_MIN"); v = e ? atoi(e) : 16; } return v; }
  static size_t prefill_workspace_bytes(const mrs_llama_config &c, int T) {
    const size_t t = (size_t)T, d = c.hidden_size, nq = (size_t)c.num_heads * c.head_dim, nkv = (size_t)c.num_kv_heads * c.head_dim;
    const size_t ff = c.intermediate_size;
    size_t b = 0;
    b += align(t * d * 4) * 2;        // h, xn
    b += align(t * nq * 4) * 2;       // q, attn
    b += align(t * nkv * 4) * 2;      // k, v
    b += align(t * ff * 4) * 3;       // gate, up, act
    b += align((pad_to((int)d, MATRIX_ROW_PADDING) / 32) * 36);  // Q8_1 scratch of the last-token lm_head GEMV
    b += align(mrs_qi_act_bytes(T, (int)std::max(std::max(d, nq), ff))) + align(mrs_gemm_qi_workspace_bytes(T, (int)std::max(d, nq)));  // exact prompt path: Q8_K operands of T rows (f16 quants, scales, run sums) + run sums of split launches
    if (T > prefill_big_min()) b += align(t * std::max(std::max(d, nq), ff) * 2) + align(mrs_gemm_q_bf16_workspace_bytes(T)) + align(t * ff * 2);  // bf16 activations + split-K partials + act(gate)*up slabs of the fused gate/up GEMM
    if (c.num_experts > 0) {  // MoE FFN of the prompt: routes = T * top_k rows in expert-sorted order
      const size_t tk = (size_t)std::max(1, (int)c.num_experts_per_tok), r = t * tk, E = (size_t)c.num_experts;
      b += align(r * ff * 4) * 3;                                                    // gate, up, act per route (replace the dense t * ff buffers)
      b += align(r * 4) * 3 + align((E + 1) * 4) + align(E * 4) * 2;                  // ids, weights, sorted routes; bounds, counts, cursors
      b += align(t * (pad_to((int)d, MATRIX_ROW_PADDING) / 32) * 36);                // Q8_1 rows of the normed hidden states
      b += align(r * (pad_to((int)ff, MATRIX_ROW_PADDING) / 32) * 36);               // Q8_1 rows of the routes' activations
      b += align(t * d * 4);                                                         // sum of the weighted expert outputs
      if (T > prefill_big_min()) b += align(t * d * 2) + align(r * ff * 2);         // bf16 slabs of the normed tokens and of the routes' activations
      b += align(r * 4) + align(r * d * 4) + align(mrs_qi_act_bytes((int)r, (int)d)) + align(mrs_qi_act_bytes((int)r, (int)ff));  // exact path: inverse route table, per-route down outputs, gathered / per-route operand rows
    }
    return b + 4096;
  }
  // MoE FFN of a prompt (SparseMoeBlock::forward on [T, hidden], models/mixtral.rs:280-304; the reference's prompt route is
  // FastExpertsWeights::forward_* -> moe_dispatch_build + grouped GEMM, moe/experts/backends.rs:969-1100, gguf/cuda.rs:590-640,1340-1420):
  // router top-k on the device, stable dispatch by expert, grouped gate / up GEMMs over the expert-sorted routes (every expert's weights are
  // read once per 8 of ITS routes instead of once per token), SiLU * up, grouped down GEMM that scales by the routing weight and sums the
  // top-k routes of a token with f32 atomics into a zeroed buffer (top-k = 2: order-independent), residual add.  No host sync.
  struct MoePrefillBufs { int32_t *ids, *sorted, *bounds, *counts, *cursors; float *w, *g, *u, *act, *sum; void *y_in, *y_act; void *xb, *yb; };  // xb / yb: bf16 slabs of the normed tokens / of the routes' activations (matrix-core route)
  typedef void (*moe_grouped_fn)(const void *, const void *, const int32_t *, const int32_t *, const float *, float *, int, int, int, int, int, int, void *);
  static moe_grouped_fn grouped_gemm_for(int dtype) {
    const DTypeInfo *ti = type_info(dtype);
    if (!ti || !mmvq_supports(dtype)) return nullptr;
    std::string tag = ti->tag;  // MoE symbols spell the K-quants without the underscore (q4k ...)
    const size_t us = tag.find("_k");
    if (us != std::string::npos) tag.erase(us, 1);
    return (moe_grouped_fn)lookup("launch_moe_grouped_gemm_" + tag);
  }
  int moe_ffn_prefill(const Block &bl, float *h, float *xn, int T, const MoePrefillBufs &m, hipStream_t s) const {
    const int d = cfg.hidden_size, ff = cfg.intermediate_size, E = cfg.num_experts, tk = cfg.num_experts_per_tok, routes = T * tk;
    const int kp_d = pad_to(d, MATRIX_ROW_PADDING), kp_ff = pad_to(ff, MATRIX_ROW_PADDING);
    const moe_grouped_fn gate_up = grouped_gemm_for(bl.gate_exps.dtype), down = grouped_gemm_for(bl.down_exps.dtype);
    if (!gate_up || !down || bl.up_exps.dtype != bl.gate_exps.dtype) return fail("prefill: no grouped MoE GEMM for expert dtypes %d / %d / %d", bl.gate_exps.dtype, bl.up_exps.dtype, bl.down_exps.dtype);
    mrs_rms_norm_f32(h, bl.post_attention_layernorm, xn, T, d, cfg.rms_eps, (int64_t)(intptr_t)s);
    if (mrs_moe_router_topk(xn, bl.router, T, E, d, tk, 1, m.ids, m.w, nullptr, s)) return fail("moe router refused (experts %d, top-k %d)", E, tk);
    launch_moe_dispatch(m.ids, m.bounds, m.sorted, nullptr, routes, E, tk, m.counts, m.cursors, s);
    // long prompts: the grouped GEMMs on the matrix cores (block dequant -> bf16 MFMA, the dense prompt GEMM's arithmetic) over the same dispatch
    // tables -- every expert's weights are decoded once per 256 of ITS routes.  MRS_MOE_PREFILL_MFMA=0 keeps the int8-dot grouped GEMV route.
    static const bool mfma_on = [] { const char *e = getenv("MRS_MOE_PREFILL_MFMA"); return !e || atoi(e) != 0; }();
    if (mfma_on && m.xb && m.yb && d % 64 == 0 && ff % 64 == 0 && bl.up_exps.dtype == bl.gate_exps.dtype) {
      int rc = mrs_convert_f32_bf16_slabs(xn, d, T, d, m.xb, s);
      if (!rc) rc = mrs_moe_gemm_q_bf16(bl.gate_exps.data, bl.gate_exps.dtype, ff, d, E, m.xb, T, m.bounds, m.sorted, tk, 1, nullptr, m.g, ff, routes, s);
      if (!rc) rc = mrs_moe_gemm_q_bf16(bl.up_exps.data, bl.up_exps.dtype, ff, d, E, m.xb, T, m.bounds, m.sorted, tk, 1, nullptr, m.u, ff, routes, s);
      if (!rc) rc = mrs_glu_bf16_slabs(m.g, m.u, ff, routes, ff, 0, m.yb, s);
      if (!rc && hipMemsetAsync(m.sum, 0, (size_t)T * d * 4, s) != hipSuccess) return fail("prefill: hipMemsetAsync failed");
      if (!rc) rc = mrs_moe_gemm_q_bf16(bl.down_exps.data, bl.down_exps.dtype, d, ff, E, m.yb, routes, m.bounds, m.sorted, tk, 0, m.w, m.sum, d, routes, s);
      if (!rc && all_reduce(m.sum, (size_t)T * d, s)) return fail("prefill: moe all-reduce failed: %s", g_last_error.c_str());  // TP: experts sharded on ffn
      if (!rc) return mrs_vec_add_f32(h, m.sum, (size_t)T * d, s) ? fail("prefill: residual add failed") : 0;
      if (rc != -1) return fail("prefill: grouped MoE GEMM on the matrix cores failed (%d)", rc);  // -1: a dtype outside that kernel -> the route below
    }
    launch_mmvq_gguf_quantize_q8_1_f32(xn, m.y_in, d, kp_d, T, s);
    gate_up(bl.gate_exps.data, m.y_in, m.bounds, m.sorted, nullptr, m.g, ff, d, kp_d, E, tk, 1, s);  // input_dim1 = 1: the token's row; out[sorted position]
    gate_up(bl.up_exps.data, m.y_in, m.bounds, m.sorted, nullptr, m.u, ff, d, kp_d, E, tk, 1, s);
    fused_glu_f32(m.g, m.u, m.act, (uint32_t)routes, (uint32_t)ff, (uint32_t)ff, (uint32_t)ff, 0, s);
    launch_mmvq_gguf_quantize_q8_1_f32(m.act, m.y_act, ff, kp_ff, routes, s);
    if (hipMemsetAsync(m.sum, 0, (size_t)T * d * 4, s) != hipSuccess) return fail("prefill: hipMemsetAsync failed");
    down(bl.down_exps.data, m.y_act, m.bounds, m.sorted, m.w, m.sum, d, ff, kp_ff, E, tk, 0, s);     // input_dim1 = 0: rows already in sorted order
    if (all_reduce(m.sum, (size_t)T * d, s)) return fail("prefill: moe all-reduce failed: %s", g_last_error.c_str());
    return mrs_vec_add_f32(h, m.sum, (size_t)T * d, s) ? fail("prefill: residual add failed") : 0;
  }
  // ---- prompts in the decode engine's arithmetic (round 4): every linear = Q8_K activation rows x exact-integer MFMA GEMM in the engine's f32 order
  //      (ext_gemm_qi.hip), attention = the decode kernels' per-block partials and merge per query token, RoPE / cache write / SiLU / residual adds = the decode
  //      expressions.  A prompt token's hidden states, KV pages and logits are bit for bit what a token-by-token decode produces (tests/test_prefill_exact.py);
  //      the role in the reference is the CPU prompt path (QMatMul f32 fallback per row, gguf/mod.rs:465-478; attention/backends/cpu).
  int prefill_mode = -1;  // -1: MRS_PREFILL_EXACT (default 1); 0: bf16-operand MFMA prompt GEMMs (library GEMMs on the bf16 shadow copy when every dense linear has one, else mrs_gemm_q_bf16_multi; + flash attention); 1: exact path; 2: as 0 with the fused-dequant kernels forced
  bool bf16_shadow_ok() const {
    if (cfg.num_experts > 0 || blocks.empty()) return false;
    for (const Block &bl : blocks) {
      const GgufMatMul *ls[7] = {bl.q_proj.get(), bl.k_proj.get(), bl.v_proj.get(), bl.o_proj.get(), bl.gate_proj.get(), bl.up_proj.get(), bl.down_proj.get()};
      for (const GgufMatMul *l : ls)
        if (!l || !l->get_qtensor() || !l->get_qtensor()->bf16) return false;
    }
    return true;
  }
  bool prefill_exact_ok() const {
    static const int env_want = [] { const char *e = getenv("MRS_PREFILL_EXACT"); return e ? atoi(e) : 1; }();
    const int want = prefill_mode >= 0 ? (prefill_mode == 1 ? 1 : 0) : env_want;
    // round 6: tensor-parallel shards (the row-parallel projections write h * (1 / world) + W_shard . y like the decode step's RESID epilogue, then the same ONE sum
    // all-reduce of h) and sparse-MoE layers (moe_ffn_exact) run in the engine's arithmetic too
    if (!want || cfg.use_fused != 2 || !engine_ok() || cfg.head_dim != 128 || cfg.block_size != 32) return false;
    const int G = cfg.num_heads / std::max(1, (int)cfg.num_kv_heads);
    if (cfg.num_heads % cfg.num_kv_heads || (G != 1 && G != 2 && G != 4 && G != 8)) return false;
    const bool moe = cfg.num_experts > 0;
    for (const Block &bl : blocks) {
      const GgufMatMul *ls[7] = {bl.q_proj.get(), bl.k_proj.get(), bl.v_proj.get(), bl.o_proj.get(), bl.gate_proj.get(), bl.up_proj.get(), bl.down_proj.get()};
      for (int i = 0; i < (moe ? 4 : 7); ++i)
        if (!ls[i] || !ls[i]->get_qtensor() || !ls[i]->get_qtensor()->qi) return false;
      if (moe && (!bl.gate_exps.qi || !bl.up_exps.qi || !bl.down_exps.qi || !bl.router || bl.gate_exps.dtype != bl.up_exps.dtype || cfg.intermediate_size % 32 || cfg.hidden_size % 32)) return false;
    }
    return dlm_head.planes != nullptr;
  }
  struct MoeExactBufs { int32_t *ids, *sorted, *inv, *bounds, *counts, *cursors; float *w, *y; void *qact_g, *qact_r; };  // routes = T * top_k rows in expert-sorted order
  // SparseMoeBlock::forward of a prompt in the decode step's arithmetic: the step's router kernel on every token, the tokens' Q8_K / Q8_0 rows gathered into
  // expert-sorted order, every expert's gate / up / down as one window of the exact GEMM, silu(g) * u quantized per route, the slots folded into h in slot order
  int moe_ffn_exact(const Block &bl, int T, float *h, float *g, float *u, float *act, void *qact, const MoeExactBufs &m, hipStream_t s) const {
    const int d = cfg.hidden_size, ff = cfg.intermediate_size, E = cfg.num_experts, tk = cfg.num_experts_per_tok, R = T * tk;
    const float rs = 1.0f / (float)std::max(1, (int)cfg.world_size);
    const int rrc = mrs_moe_router_topk_norm(h, bl.post_attention_layernorm, cfg.rms_eps, bl.router, T, E, d, tk, 1, m.ids, m.w, s);
    if (rrc) return fail("prefill (exact): moe router refused (%d)", rrc);  // (-3, rows beyond 64 KiB, takes the two-launch router in the decode step: not a shape of this path)
    launch_moe_dispatch(m.ids, m.bounds, m.sorted, nullptr, R, E, tk, m.counts, m.cursors, s);
    const int tg = bl.gate_exps.dtype, td = bl.down_exps.dtype;
    if (mrs_qi_quantize_for(tg, h, nullptr, d, bl.post_attention_layernorm, cfg.rms_eps, T, d, qact, nullptr, s)) return fail("prefill (exact): activation quantizer refused K=%d", d);
    if (mrs_qi_gather_rows(tg, qact, T, m.qact_g, R, d, m.sorted, tk, m.inv, s)) return fail("prefill (exact): operand gather refused");
    const size_t pg = mrs_gemm_qi_repack_bytes(tg, ff, d), pd = mrs_gemm_qi_repack_bytes(td, d, ff);  // bytes of one expert's panels
    for (int e = 0; e < E; ++e) {
      if (mrs_gemm_qi_win((const char *)bl.gate_exps.qi + (size_t)e * pg, tg, ff, d, m.qact_g, R, m.bounds + e, T, g, ff, 0, s) ||
          mrs_gemm_qi_win((const char *)bl.up_exps.qi + (size_t)e * pg, tg, ff, d, m.qact_g, R, m.bounds + e, T, u, ff, 0, s))
        return fail("prefill (exact): expert gate / up GEMM refused (ggml dtype %d)", tg);
    }
    if (mrs_qi_quantize_for(td, g, u, ff, nullptr, 0.f, R, ff, m.qact_r, act, s)) return fail("prefill (exact): activation quantizer refused K=%d", ff);
    for (int e = 0; e < E; ++e)
      if (mrs_gemm_qi_win((const char *)bl.down_exps.qi + (size_t)e * pd, td, d, ff, m.qact_r, R, m.bounds + e, T, m.y, d, 0, s)) return fail("prefill (exact): expert down GEMM refused (ggml dtype %d)", td);
    if (mrs_moe_fold_exact(h, rs, m.y, m.inv, m.w, T, d, tk, s)) return fail("prefill (exact): moe fold refused");
    return all_reduce(h, (size_t)T * d, s) ? fail("prefill (exact): moe all-reduce failed: %s", g_last_error.c_str()) : 0;
  }
  int prefill_exact(const mrs_llama_prefill_args &pa, int T, float *h, float *xn, float *q, float *k, float *v, float *attn, float *g, float *u, float *act, void *qact,
                    void *qws, size_t qws_bytes, const MoeExactBufs &mx, hipStream_t s) const {
    const int d = cfg.hidden_size, hd = cfg.head_dim, nq = cfg.num_heads * hd, nkv = cfg.num_kv_heads * hd, ff = cfg.intermediate_size;
    const int bs = cfg.block_size, kvh = cfg.num_kv_heads, kvd = cfg.kv_f16 ? 0 : 1;
    const int eff_max = std::min(cfg.max_blocks_per_seq * bs, cfg.max_context_len);
    const int64_t st = (int64_t)(intptr_t)s;
    auto lin = [&](const GgufMatMul &m, int N, int K, float *out, int acc) -> int {
      const QTensor *w = m.get_qtensor();
      return mrs_gemm_qi_ws(w->qi, w->dtype, N, K, qact, T, out, N, acc, N <= std::max(d, nq) ? qws : nullptr, qws_bytes, s) ? fail("prefill (exact): mrs_gemm_qi refused ggml dtype %d (N=%d K=%d)", w->dtype, N, K) : 0;
    };
    // o_proj / down_proj: h <- h + W . x;  tensor parallel: h <- h * (1 / world) + W_shard . x on every rank (the decode step's RESID epilogue, two roundings), then ONE
    // sum all-reduce of h (distributed/layers.rs:965-975) -- the same sums as the decode step whenever the all-reduce adds the ranks in the same order (world 2: always;
    // the peer-mailbox route: rank order for every message it takes)
    const float rs = 1.0f / (float)std::max(1, (int)cfg.world_size);
    auto row_parallel = [&](const GgufMatMul &m, int N, int K, const float *x, const float *x2, int ldx, float *xtmp) -> int {
      const int ty = m.get_qtensor()->dtype;
      if (mrs_qi_quantize_for(ty, x, x2, ldx, nullptr, 0.f, T, K, qact, xtmp, s)) return fail("prefill (exact): activation quantizer refused K=%d (ggml dtype %d)", K, ty);
      if (cfg.world_size <= 1) return lin(m, N, K, h, 1);
      if (lin(m, N, K, xn, 0) || mrs_resid_scale_add_f32(h, rs, xn, (size_t)T * N, s)) return -1;
      return all_reduce(h, (size_t)T * N, s) ? fail("prefill (exact): all-reduce failed: %s", g_last_error.c_str()) : 0;
    };
    // one activation image per vec_dot partner (gguf/mod.rs:465-478: Q8_K rows for the K-quants, Q8_0 rows for Q8_0 weights): projections that share their input share
    // the image when their weight formats ask for the same one (every model of BASELINE.json's configs: one image per group)
    auto group = [&](std::initializer_list<const GgufMatMul *> ms, std::initializer_list<int> Ns, std::initializer_list<float *> outs, int K, const float *x, const float *x2, int ldx,
                     const float *nw, float *xtmp, int acc) -> int {
      std::vector<const GgufMatMul *> m(ms);
      std::vector<int> n(Ns);
      std::vector<float *> o(outs);
      std::vector<bool> done(m.size(), false);
      for (size_t i = 0; i < m.size(); ++i) {
        if (done[i]) continue;
        const int ty = m[i]->get_qtensor()->dtype, mode = ty == Q8_0 ? 1 : 0;
        if (mrs_qi_quantize_for(ty, x, x2, ldx, nw, cfg.rms_eps, T, K, qact, xtmp, s)) return fail("prefill (exact): activation quantizer refused K=%d (ggml dtype %d)", K, ty);
        for (size_t j = i; j < m.size(); ++j)
          if (!done[j] && ((m[j]->get_qtensor()->dtype == Q8_0) ? 1 : 0) == mode) { if (lin(*m[j], n[j], K, o[j], acc)) return -1; done[j] = true; }
      }
      return 0;
    };
    if (wte->embedding_forward_raw(pa.token_ids, T, h, s)) return -1;
    for (size_t li = 0; li < blocks.size(); ++li) {
      const Block &bl = blocks[li];
      if (group({bl.q_proj.get(), bl.k_proj.get(), bl.v_proj.get()}, {nq, nkv, nkv}, {q, k, v}, d, h, nullptr, d, bl.input_layernorm, nullptr, 0)) return -1;
      rotary_embedding_positions(q, k, (void *)bufs.cos_table, (void *)bufs.sin_table, (void *)pa.positions, cfg.rope_interleaved ? 0 : 1, hd, T,
                                 cfg.rot_dim / 2, cfg.max_context_len, cfg.num_heads, cfg.num_kv_heads, nq, nkv, 2, st);
      reshape_and_cache(k, v, bl.key_cache, bl.value_cache, (int64_t *)pa.slot_mapping, T, cfg.num_kv_heads, hd, bs, 8, nkv, nkv, s, 2, kvd == 1 ? 1 : 0, nullptr, nullptr);
      if (mrs_prefill_attention_exact(q, bl.key_cache, bl.value_cache, pa.block_tables, pa.context_lens, attn, T, cfg.num_heads, kvh, hd, bs, nq, kvh * hd * bs, hd * bs,
                                      1.0f / sqrtf((float)hd), eff_max, kvd, cfg.sliding_window, pa.start_pos + T, s))
        return fail("prefill (exact): attention refused the shape");
      if (row_parallel(*bl.o_proj, d, nq, attn, nullptr, nq, nullptr)) return -1;
      if (cfg.num_experts > 0) {
        if (moe_ffn_exact(bl, T, h, g, u, act, qact, mx, s)) return -1;
        continue;
      }
      if (group({bl.gate_proj.get(), bl.up_proj.get()}, {ff, ff}, {g, u}, d, h, nullptr, d, bl.post_attention_layernorm, nullptr, 0)) return -1;
      if (row_parallel(*bl.down_proj, d, ff, g, u, ff, act)) return -1;
    }
    // ctx.logits: only the last prompt token reaches lm_head (llama.rs:514-517): the decode engine's final norm + lm_head launch
    if (mrs_dec_proj(&dlm_head, cfg.vocab_size, nullptr, h + (size_t)(T - 1) * d, d, ln_f, cfg.rms_eps, pa.logits, cfg.vocab_size, 0, 1.0f, nullptr, 1, s))
      return fail("prefill (exact): lm_head refused");
    return 0;
  }
  int prefill(const mrs_llama_prefill_args &pa, int T, hipStream_t s) const {
    if (T <= 0) return fail("prefill: T must be positive");
    const int start_pos = pa.start_pos;
    if (!wte || !lm_head || !ln_f) return fail("model is missing token_embd / output / output_norm");
    if (pa.workspace_bytes < prefill_workspace_bytes(cfg, T)) return fail("prefill workspace too small");
    const int d = cfg.hidden_size, hd = cfg.head_dim, nq = cfg.num_heads * hd, nkv = cfg.num_kv_heads * hd, ff = cfg.intermediate_size;
    const size_t t = (size_t)T;
    char *p = (char *)(((uintptr_t)pa.workspace + 255) & ~(uintptr_t)255);
    auto take = [&](size_t n) { char *r = p; p += align(n); return r; };
    float *h = (float *)take(t * d * 4), *xn = (float *)take(t * d * 4);
    float *q = (float *)take(t * nq * 4), *attn = (float *)take(t * nq * 4);
    float *k = (float *)take(t * nkv * 4), *v = (float *)take(t * nkv * 4);
    float *g = (float *)take(t * ff * 4), *u = (float *)take(t * ff * 4), *act = (float *)take(t * ff * 4);
    if (prefill_exact_ok()) {
      void *qact = take(mrs_qi_act_bytes(T, std::max(std::max(d, nq), ff)));
      const size_t qws_bytes = mrs_gemm_qi_workspace_bytes(T, std::max(d, nq));
      void *qws = take(qws_bytes);
      MoeExactBufs mx{};
      if (cfg.num_experts > 0) {  // routes = T * top_k rows in expert-sorted order; g / u / act become [routes][ff]
        const size_t tk = (size_t)cfg.num_experts_per_tok, r = t * tk, E = (size_t)cfg.num_experts;
        g = (float *)take(r * ff * 4); u = (float *)take(r * ff * 4); act = (float *)take(r * ff * 4);
        mx.ids = (int32_t *)take(r * 4); mx.w = (float *)take(r * 4); mx.sorted = (int32_t *)take(r * 4); mx.inv = (int32_t *)take(r * 4);
        mx.bounds = (int32_t *)take((E + 1) * 4); mx.counts = (int32_t *)take(E * 4); mx.cursors = (int32_t *)take(E * 4);
        mx.y = (float *)take(r * d * 4);
        mx.qact_g = take(mrs_qi_act_bytes((int)r, d)); mx.qact_r = take(mrs_qi_act_bytes((int)r, ff));
      }
      return prefill_exact(pa, T, h, xn, q, k, v, attn, g, u, act, qact, qws, qws_bytes, mx, s);
    }
    MoePrefillBufs moe{};
    if (cfg.num_experts > 0) {
      const size_t tk = (size_t)cfg.num_experts_per_tok, r = t * tk, E = (size_t)cfg.num_experts;
      moe.g = (float *)take(r * ff * 4); moe.u = (float *)take(r * ff * 4); moe.act = (float *)take(r * ff * 4);
      moe.ids = (int32_t *)take(r * 4); moe.w = (float *)take(r * 4); moe.sorted = (int32_t *)take(r * 4);
      moe.bounds = (int32_t *)take((E + 1) * 4); moe.counts = (int32_t *)take(E * 4); moe.cursors = (int32_t *)take(E * 4);
      moe.y_in = take(t * (pad_to(d, MATRIX_ROW_PADDING) / 32) * 36);
      moe.y_act = take(r * (pad_to(ff, MATRIX_ROW_PADDING) / 32) * 36);
      moe.sum = (float *)take(t * d * 4);
      if (T > prefill_big_min()) { moe.xb = take(t * d * 2); moe.yb = take(r * ff * 2); }
    }
    const int64_t st = (int64_t)(intptr_t)s;
    // T > 128: the 256-row-tile kernel over bf16 activations (converted once per GEMM group), split-K partials in `part`
    const bool big = T > prefill_big_min() && !getenv("MRS_PREFILL_SMALL_TILES");
    const size_t part_bytes = big ? mrs_gemm_q_bf16_workspace_bytes(T) : 0;
    void *xb = big ? take(t * std::max(std::max(d, nq), ff) * 2) : nullptr, *part = big ? take(part_bytes) : nullptr;
    void *xg = big ? take(t * ff * 2) : nullptr;  // output slabs of the fused gate / up GEMM (it reads xb while it writes)
    const float *xb_src = nullptr;    // which f32 buffer xb currently mirrors (within one GEMM group)
    const float *xb_ready = nullptr;  // set by a producer that wrote the slabs of that (never materialised) f32 buffer directly
    auto to_bf16 = [&](const float *x, int K) -> int {
      if (xb_src == x) return 0;
      xb_src = x;
      if (xb_ready == x) { xb_ready = nullptr; return 0; }
      return mrs_convert_f32_bf16_slabs(x, K, T, K, xb, s);
    };
    auto gemm = [&](const GgufMatMul &m, const float *x, int K, float *out, int N, int acc) -> int {
      const QTensor *w = m.get_qtensor();
      int rc;
      if (big) {
        xb_src = nullptr;
        rc = to_bf16(x, K);
        if (!rc) rc = mrs_gemm_q_bf16_multi(1, &w->data, &N, &out, &N, w->dtype, K, xb, T, acc, part, part_bytes, s);
        xb_src = nullptr;
      } else rc = mrs_gemm_q_f32(w->data, w->dtype, N, K, x, K, out, N, T, acc, s);
      if (rc) return fail("prefill: no GEMM for ggml dtype %d (K=%d)", w->dtype, K);
      return 0;
    };
    // projections that share their input run as ONE launch per weight type (fast_mmq::fused_qkv / fused_glu role)
    auto gemm_multi = [&](std::initializer_list<const GgufMatMul *> ms, const float *x, int K, std::initializer_list<float *> outs,
                          std::initializer_list<int> Ns) -> int {
      std::vector<const GgufMatMul *> m(ms);
      std::vector<float *> o(outs);
      std::vector<int> n(Ns);
      std::vector<bool> done(m.size(), false);
      xb_src = nullptr;
      for (size_t i = 0; i < m.size(); ++i) {
        if (done[i]) continue;
        const int ty = m[i]->get_qtensor()->dtype;
        const void *w[3]; float *oo[3]; int nn[3], ld[3], c = 0;
        for (size_t j = i; j < m.size(); ++j)
          if (!done[j] && m[j]->get_qtensor()->dtype == ty) {
            w[c] = m[j]->get_qtensor()->data;
            oo[c] = o[j]; nn[c] = n[j]; ld[c] = n[j]; ++c; done[j] = true;
          }
        int rc;
        if (big) {
          rc = to_bf16(x, K);
          if (!rc) rc = mrs_gemm_q_bf16_multi(c, w, nn, oo, ld, ty, K, xb, T, 0, part, part_bytes, s);
        } else rc = mrs_gemm_q_f32_multi(c, w, nn, oo, ld, ty, K, x, K, T, 0, s);
        if (rc) return fail("prefill: no GEMM for ggml dtype %d (K=%d)", ty, K);
      }
      xb_src = nullptr;
      return 0;
    };
    if (wte->embedding_forward_raw(pa.token_ids, T, h, s)) return -1;
    const int bs = cfg.block_size, kvh = cfg.num_kv_heads;
    const int eff_max = std::min(cfg.max_blocks_per_seq * bs, cfg.max_context_len);
    const int parts = (eff_max + 511) / 512;
    // ---- round 6: every dense linear has a bf16 shadow copy -> the prompt GEMMs are plain bf16 library GEMMs (ext_gemm_lt.hip: hipBLASLt), the activations bf16 ROWS;
    //      same arithmetic as the fused-dequant kernels below (weights rounded to bf16 once, f32 accumulation) in the library's summation order.  prefill_mode 2 forces the
    //      fused-dequant kernels (A / B).
    typedef int (*lt_gemm_fn)(const void *, const void *, float *, int, int, int, int, int, void *);
    typedef int (*lt_rows_fn)(const float *, int, int, int, void *, void *);
    typedef int (*lt_glu_fn)(const float *, const float *, int, int, int, void *, void *);
    typedef int (*lt_norm_fn)(const float *, const float *, int, int, float, void *, void *);
    static const lt_gemm_fn lt_gemm = (lt_gemm_fn)lookup_quiet("mrs_lt_gemm_bf16");
    static const lt_rows_fn lt_rows = (lt_rows_fn)lookup_quiet("mrs_rows_f32_to_bf16");
    static const lt_glu_fn lt_glu = (lt_glu_fn)lookup_quiet("mrs_rows_glu_bf16");
    static const lt_norm_fn lt_norm = (lt_norm_fn)lookup_quiet("mrs_rows_rms_norm_bf16");
    if (big && bf16_shadow_ok() && prefill_mode != 2 && lt_gemm && lt_rows && lt_glu && lt_norm && cfg.num_experts == 0 && cfg.head_dim == 128 && cfg.block_size == 32 &&
        d % 8 == 0 && nq % 8 == 0 && ff % 8 == 0) {
      auto lgemm_at = [&](const void *wb, int N, int K, float *out, int ldo, int acc) -> int {
        const int rc = lt_gemm(wb, xb, out, ldo, N, K, T, acc, s);
        return rc ? fail("prefill (bf16 shadow): library GEMM refused N=%d K=%d T=%d (%d)", N, K, T, rc) : 0;
      };
      auto lgemm = [&](const GgufMatMul &m, int N, int K, float *out, int acc) -> int { return lgemm_at(m.get_qtensor()->bf16, N, K, out, N, acc); };
      for (size_t li = 0; li < blocks.size(); ++li) {
        const Block &bl = blocks[li];
        if (!bl.q_proj || !bl.key_cache) return fail("layer %zu is incomplete", li);
        if (lt_norm(h, bl.input_layernorm, T, d, cfg.rms_eps, xb, s)) return fail("prefill (bf16 shadow): hidden size must be a multiple of 8");
        // q / k / v (and gate / up) shadow rows laid out back to back by the loader (llama.py) = ONE GEMM of N = nq + 2 nkv (2 ff) rows: the role of fast_mmq::fused_qkv /
        // fused_glu (one launch per shared input); the outputs are column ranges of one [T][N] buffer, which rotary / reshape_and_cache / the attention kernel take by stride
        const char *qb = (const char *)bl.q_proj->get_qtensor()->bf16, *kb = (const char *)bl.k_proj->get_qtensor()->bf16, *vb = (const char *)bl.v_proj->get_qtensor()->bf16;
        const bool qkv_fused = kb == qb + (size_t)nq * d * 2 && vb == kb + (size_t)nkv * d * 2 && (size_t)(nq + 2 * nkv) <= (size_t)ff;
        float *qp = q, *kp = k, *vp = v;
        int qs = nq, ks = nkv;
        if (qkv_fused) {
          const int nqkv = nq + 2 * nkv;
          if (lgemm_at(qb, nqkv, d, g, nqkv, 0)) return -1;  // g [T][ff] is free until the FFN
          qp = g; kp = g + nq; vp = g + nq + nkv; qs = ks = nqkv;
        } else if (lgemm(*bl.q_proj, nq, d, q, 0) || lgemm(*bl.k_proj, nkv, d, k, 0) || lgemm(*bl.v_proj, nkv, d, v, 0)) return -1;
        rotary_embedding_positions(qp, kp, (void *)bufs.cos_table, (void *)bufs.sin_table, (void *)pa.positions, cfg.rope_interleaved ? 0 : 1, hd, T,
                                   cfg.rot_dim / 2, cfg.max_context_len, cfg.num_heads, cfg.num_kv_heads, qs, ks, 2, st);
        reshape_and_cache(kp, vp, bl.key_cache, bl.value_cache, (int64_t *)pa.slot_mapping, T, cfg.num_kv_heads, hd, bs, 8, ks, ks, s, 2, 1, nullptr, nullptr);
        if (mrs_prefill_attention_window_f32_bf16(qp, bl.key_cache, bl.value_cache, pa.block_tables, attn, T, start_pos, cfg.num_heads, kvh, hd, bs, qs, nq,
                                                  kvh * hd * bs, hd * bs, 1.0f / sqrtf((float)hd), cfg.sliding_window, s) != 0)
          return fail("prefill (bf16 shadow): the MFMA flash attention refused the shape (head_dim 128, block 32)");
        if (lt_rows(attn, nq, T, nq, xb, s)) return -1;
        if (cfg.world_size > 1) { if (lgemm(*bl.o_proj, d, nq, xn, 0) || all_reduce(xn, t * d, s) || mrs_vec_add_f32(h, xn, t * d, s)) return -1; }
        else if (lgemm(*bl.o_proj, d, nq, h, 1)) return -1;
        if (lt_norm(h, bl.post_attention_layernorm, T, d, cfg.rms_eps, xb, s)) return -1;
        const char *gb = (const char *)bl.gate_proj->get_qtensor()->bf16, *ub = (const char *)bl.up_proj->get_qtensor()->bf16;
        if (ub == gb + (size_t)ff * d * 2 && u == g + t * ff) {  // one GEMM of 2 ff rows into [T][2 ff] = the g and u buffers, which the workspace holds back to back
          if (lgemm_at(gb, 2 * ff, d, g, 2 * ff, 0) || lt_glu(g, g + ff, 2 * ff, T, ff, xb, s)) return -1;
        } else if (lgemm(*bl.gate_proj, ff, d, g, 0) || lgemm(*bl.up_proj, ff, d, u, 0) || lt_glu(g, u, ff, T, ff, xb, s)) return -1;
        if (cfg.world_size > 1) { if (lgemm(*bl.down_proj, d, ff, xn, 0) || all_reduce(xn, t * d, s) || mrs_vec_add_f32(h, xn, t * d, s)) return -1; }
        else if (lgemm(*bl.down_proj, d, ff, h, 1)) return -1;
      }
      const QTensor *lm = lm_head->get_qtensor();
      if (!mrs_decode_gemv_supported(lm->dtype)) return fail("prefill (bf16 shadow): lm_head dtype %d", lm->dtype);
      if (mrs_decode_norm_proj(lm->data, lm->dtype, cfg.vocab_size, d, h + (t - 1) * d, ln_f, cfg.rms_eps, pa.logits, cfg.vocab_size, 1, s)) return fail("prefill: lm_head refused");
      return 0;
    }
    const bool use_v1 = (parts == 1 || (long)T * cfg.num_heads > 512);  // paged_attention.rs:302-307 (only consulted when the MFMA flash kernel refuses the shape)
    for (size_t li = 0; li < blocks.size(); ++li) {
      const Block &bl = blocks[li];
      if (!bl.q_proj || !bl.key_cache) return fail("layer %zu is incomplete", li);
      if (big) { if (mrs_rms_norm_bf16_slabs(h, bl.input_layernorm, T, d, cfg.rms_eps, xb, s)) return fail("prefill: hidden size must be a multiple of 64"); xb_ready = xn; }
      else mrs_rms_norm_f32(h, bl.input_layernorm, xn, T, d, cfg.rms_eps, st);
      if (gemm_multi({bl.q_proj.get(), bl.k_proj.get(), bl.v_proj.get()}, xn, d, {q, k, v}, {nq, nkv, nkv})) return -1;
      rotary_embedding_positions(q, k, (void *)bufs.cos_table, (void *)bufs.sin_table, (void *)pa.positions, cfg.rope_interleaved ? 0 : 1, hd, T,
                                 cfg.rot_dim / 2, cfg.max_context_len, cfg.num_heads, cfg.num_kv_heads, nq, nkv, 2, st);
      reshape_and_cache(k, v, bl.key_cache, bl.value_cache, (int64_t *)pa.slot_mapping, T, cfg.num_kv_heads, hd, bs, 8, nkv, nkv, s, 2, 1, nullptr, nullptr);
      // causal attention over the pages just written: MFMA flash kernel (head_dim 128 / block 32), else prompt token t = "sequence" t
      // of the decode-style kernel with context_lens[t] = pos + 1
      if (mrs_prefill_attention_window_f32_bf16(q, bl.key_cache, bl.value_cache, pa.block_tables, attn, T, start_pos, cfg.num_heads, kvh, hd, bs, nq, nq,
                                                kvh * hd * bs, hd * bs, 1.0f / sqrtf((float)hd), cfg.sliding_window, s) != 0) {
        if (cfg.sliding_window > 0) return fail("prefill: sliding-window attention needs the MFMA flash kernel's shapes (head_dim 128, block 32)");
        // fallback for head / block sizes outside the flash kernel: prompt token t = "sequence" t of the decode-style kernel.  v1 keeps all logits of
        // a sequence in LDS, so long contexts with few (token, head) pairs take v2 over 512-token partitions with the runner's partial buffers
        if (use_v1) {
          mrs_paged_attention_f32_bf16(0, attn, nullptr, nullptr, nullptr, q, bl.key_cache, bl.value_cache, nullptr, kvh, 1.0f / sqrtf((float)hd), 1.0f,
                                       pa.block_tables, pa.context_lens, bs, eff_max, T, cfg.num_heads, hd, cfg.max_blocks_per_seq, nq, kvh * hd * bs,
                                       hd * bs, s, nullptr);
        } else {
          if (T > cfg.max_batch) return fail("prefill: %d prompt tokens at max_context_len %d need the v2 attention workspace of %d sequences; use prompts of <= %d tokens or the flash-attention shapes (head_dim 128, block 32)", T, cfg.max_context_len, T, cfg.max_batch);
          mrs_paged_attention_f32_bf16(1, attn, ws.exp_sums, ws.max_logits, ws.attn_ws, q, bl.key_cache, bl.value_cache, nullptr, kvh, 1.0f / sqrtf((float)hd), 1.0f,
                                       pa.block_tables, pa.context_lens, bs, eff_max, T, cfg.num_heads, hd, cfg.max_blocks_per_seq, nq, kvh * hd * bs,
                                       hd * bs, s, nullptr);
        }
      }
      if (cfg.world_size > 1) {  // row-parallel: partial -> all-reduce -> residual add (bias-free)
        if (gemm(*bl.o_proj, attn, nq, xn, d, 0) || all_reduce(xn, t * d, s) || mrs_vec_add_f32(h, xn, t * d, s)) return -1;
      } else if (gemm(*bl.o_proj, attn, nq, h, d, 1)) return -1;
      if (cfg.num_experts > 0) {
        if (!bl.router || !bl.gate_exps.data || !bl.up_exps.data || !bl.down_exps.data) return fail("layer %zu has no experts", li);
        if (moe_ffn_prefill(bl, h, xn, T, moe, s)) return -1;
        continue;
      }
      if (big) { if (mrs_rms_norm_bf16_slabs(h, bl.post_attention_layernorm, T, d, cfg.rms_eps, xb, s)) return -1; xb_ready = xn; }
      else mrs_rms_norm_f32(h, bl.post_attention_layernorm, xn, T, d, cfg.rms_eps, st);
      // gate / up / SiLU*up in ONE launch (act(gate) * up leaves the epilogue as the down GEMM's bf16 slabs): bit-identical, but measured 1.5-2.5 % SLOWER per
      // prompt on the MI355X than two GEMM segments + the 12.7 us GLU pass (TTFT 14.15 vs 13.82 ms at 512 tokens, 47.5 vs 46.9 ms at 2048: the epilogue's expf and
      // 2-byte half-line stores sit in the un-overlapped tail of every workgroup) -> opt-in (MRS_PREFILL_FUSED_GLU=1)
      static const bool glu_fused = [] { const char *e = getenv("MRS_PREFILL_FUSED_GLU"); return e && atoi(e) != 0; }();
      const QTensor *qg = bl.gate_proj->get_qtensor(), *qu = bl.up_proj->get_qtensor(), *qd = bl.down_proj->get_qtensor();
      if (big && glu_fused && ff % 64 == 0 && qg->dtype == qu->dtype && xb_ready == xn &&
          mrs_gemm_q_bf16_glu(qg->data, qu->data, qg->dtype, ff, d, xb, T, 0, xg, s) == 0) {
        xb_ready = nullptr; xb_src = nullptr;
        float *dst = cfg.world_size > 1 ? xn : h;
        if (mrs_gemm_q_bf16_multi(1, &qd->data, &d, &dst, &d, qd->dtype, ff, xg, T, cfg.world_size > 1 ? 0 : 1, part, part_bytes, s))
          return fail("prefill: no GEMM for ggml dtype %d (K=%d)", qd->dtype, ff);
        if (cfg.world_size > 1 && (all_reduce(xn, t * d, s) || mrs_vec_add_f32(h, xn, t * d, s))) return -1;
        continue;
      }
      if (gemm_multi({bl.gate_proj.get(), bl.up_proj.get()}, xn, d, {g, u}, {ff, ff})) return -1;
      if (big && ff % 64 == 0) { if (mrs_glu_bf16_slabs(g, u, ff, T, ff, 0, xb, s)) return -1; xb_ready = act; }
      else fused_glu_f32(g, u, act, (uint32_t)T, (uint32_t)ff, (uint32_t)ff, (uint32_t)ff, 0, s);
      if (cfg.world_size > 1) {
        if (gemm(*bl.down_proj, act, ff, xn, d, 0) || all_reduce(xn, t * d, s) || mrs_vec_add_f32(h, xn, t * d, s)) return -1;
      } else if (gemm(*bl.down_proj, act, ff, h, d, 1)) return -1;
    }
    // ctx.logits: only the last prompt token reaches lm_head (llama.rs:514-517)
    const QTensor *lm = lm_head->get_qtensor();
    if (mrs_decode_gemv_supported(lm->dtype)) {
      if (mrs_decode_norm_proj(lm->data, lm->dtype, cfg.vocab_size, d, h + (t - 1) * d, ln_f, cfg.rms_eps, pa.logits, cfg.vocab_size, 1, s))
        return fail("prefill: lm_head refused");
    } else {
      mrs_rms_norm_f32(h + (t - 1) * d, ln_f, xn, 1, d, cfg.rms_eps, st);
      const Scratch sc{take((size_t)(pad_to(d, MATRIX_ROW_PADDING) / 32) * 36), (size_t)(pad_to(d, MATRIX_ROW_PADDING) / 32) * 36};
      if (lm_head->forward_raw(xn, 1, pa.logits, sc, s)) return -1;
    }
    return 0;
  }
  double prefill_flops(int T) const {
    double w = 0;
    for (const Block &bl : blocks)
      for (const auto *m : {&bl.q_proj, &bl.k_proj, &bl.v_proj, &bl.o_proj, &bl.gate_proj, &bl.up_proj, &bl.down_proj})
        if (*m) w += (double)(*m)->get_qtensor()->rows * (double)(*m)->get_qtensor()->cols;
    double f = 2.0 * T * w;
    if (cfg.num_experts > 0)  // every token visits top-k experts: gate, up, down of [ffn x hidden] each (the router's E x hidden GEMV is negligible)
      f += 2.0 * T * (double)cfg.num_experts_per_tok * 3.0 * (double)cfg.intermediate_size * (double)cfg.hidden_size * (double)blocks.size();
    if (lm_head) f += 2.0 * (double)lm_head->get_qtensor()->rows * (double)lm_head->get_qtensor()->cols;
    f += 4.0 * cfg.num_layers * cfg.num_heads * cfg.head_dim * (double)T * (double)T / 2.0;  // QK^T + PV, causal
    return f;
  }

  int forward_logits(int b, hipStream_t s) const {
    if (check_ready(b)) return -1;
    // sliding-window attention (Mistral) exists in the decode engine's split attention and in the MFMA prefill only: every other path would silently attend everything
    if (cfg.sliding_window > 0 && (cfg.use_fused != 2 || !attn2))
      return fail("sliding_window %d needs the decode engine with its default attention (use_fused = 2, MRS_DEC_ATTN2 / MRS_DEC_FUSED_ATTN / MRS_DEC_PERSIST unset)", cfg.sliding_window);
    if (cfg.use_fused == 2) {  // the engine never falls back silently: its arithmetic (Q8_K activations) differs from the Q8_1 paths
      if (!engine_ok()) return fail("decode engine: needs interleaved RoPE, head_dim 128, block 32, q4_k/q5_k/q6_k/q8_0 linears and a decode-layout copy of every linear");
      return forward_engine(b, s);
    }
    if (cfg.num_experts > 0 && !fused_ok()) return fail("MoE layers need the fused decode path (interleaved RoPE, q4_k/q5_k/q6_k/q8_0 weights, use_fused)");
    return fused_ok() ? forward_fused(b, s) : forward_unfused(b, s);
  }

  int decode_step(int b, hipStream_t s) const {
    if (forward_logits(b, s)) return -1;
    if (mrs_sample_greedy_advance(bufs.logits, cfg.vocab_size, b, bufs.input_ids, bufs.tokens_out, bufs.tokens_out_stride,
                                  bufs.step_counter, bufs.positions, bufs.context_lens, bufs.slot_mapping, bufs.block_tables,
                                  cfg.max_blocks_per_seq, cfg.block_size, ws.sample_scratch, s))
      return fail("mrs_sample_greedy_advance refused");
    return 0;
  }

  // ---- the chained greedy step (round 6): forward_engine without the embedding launch + lm_head with the arg-max in its epilogue + ONE launch that samples, advances
  //      the device-resident state and gathers the NEXT token's embedding row into ws.h.  Same logits, same token as decode_step (tests/test_dec_model.py).
  bool chained_ok(int b) const {
    static const bool on = [] { const char *e = getenv("MRS_DEC_CHAINED"); return !e || atoi(e) != 0; }();
    return on && b == 1 && cfg.use_fused == 2 && engine_ok() && wte && wte->get_qtensor();
  }
  int embed_state(int b, hipStream_t s) const {
    if (check_ready(b)) return -1;
    return wte->embedding_forward_raw(bufs.input_ids, b, ws.h, s);
  }
  int decode_step_chained(int b, hipStream_t s) const {
    if (check_ready(b)) return -1;
    if (!chained_ok(b)) return fail("chained decode step: needs the decode engine at batch 1");
    if (cfg.sliding_window > 0 && !attn2) return fail("sliding_window %d needs the decode engine with its default attention", cfg.sliding_window);
    if (forward_engine(b, s, true)) return -1;
    const QTensor *e = wte->get_qtensor();
    if (mrs_sample_advance_embed(bufs.input_ids, bufs.tokens_out, bufs.tokens_out_stride, bufs.step_counter, bufs.positions, bufs.context_lens, bufs.slot_mapping,
                                 bufs.block_tables, cfg.max_blocks_per_seq, cfg.block_size, ws.sample_scratch, e->data, e->dtype, ws.h, (int)e->cols, s))
      return fail("mrs_sample_advance_embed refused (embedding dtype %d)", e->dtype);
    return 0;
  }

  double decode_bytes(int b, int ctx) const {
    double t = 0;
    for (const Block &bl : blocks)
      for (const auto *m : {&bl.q_proj, &bl.k_proj, &bl.v_proj, &bl.o_proj, &bl.gate_proj, &bl.up_proj, &bl.down_proj})
        if (*m) t += (double)(*m)->get_qtensor()->nbytes();
    if (cfg.num_experts > 0)  // per token only the top-k experts of every layer are streamed (+ the router)
      for (const Block &bl : blocks)
        t += (double)b * cfg.num_experts_per_tok * (double)(bl.gate_exps.nbytes() + bl.up_exps.nbytes() + bl.down_exps.nbytes()) / cfg.num_experts +
             (double)cfg.num_experts * cfg.hidden_size * 4.0;
    if (lm_head) t += (double)lm_head->get_qtensor()->nbytes();
    if (wte) { const auto *ti = type_info(wte->get_qtensor()->dtype); t += (double)b * (wte->get_qtensor()->cols / ti->block) * ti->bytes; }
    t += (double)b * 2.0 * cfg.num_layers * cfg.num_kv_heads * cfg.head_dim * (double)ctx * 2.0;  // bf16 K + V
    return t;
  }
};

// GGUF tensor name -> model slot (gguf/normal_bindings.rs:40-220)
static int bind_tensor(Llama &m, const std::string &name, const void *p, int type, int64_t rows, int64_t cols) {
  const auto *ti = type_info(type);
  if (!ti) return fail("tensor %s: unsupported ggml dtype %d", name.c_str(), type);
  if (cols % ti->block) return fail("tensor %s: %lld columns is not a multiple of the block size %d", name.c_str(), (long long)cols, ti->block);
  auto norm = [&](const float *&slot) { if (type != F32) return fail("tensor %s: norm weights must be F32", name.c_str()); slot = (const float *)p; return 0; };
  auto lin = [&](std::unique_ptr<GgufMatMul> &slot, int64_t er, int64_t ec) {
    if (rows != er || cols != ec) return fail("tensor %s: shape [%lld, %lld], expected [%lld, %lld]", name.c_str(), (long long)rows, (long long)cols, (long long)er, (long long)ec);
    slot.reset(new GgufMatMul(QTensor{p, type, rows, cols}));
    return 0;
  };
  const auto &c = m.cfg;
  const int64_t d = c.hidden_size, nq = (int64_t)c.num_heads * c.head_dim, nkv = (int64_t)c.num_kv_heads * c.head_dim, ff = c.intermediate_size;
  if (name == "token_embd.weight") return lin(m.wte, c.vocab_size, d);
  if (name == "output.weight") return lin(m.lm_head, c.vocab_size, d);
  if (name == "output_norm.weight") return norm(m.ln_f);
  int layer = -1, consumed = 0;
  if (sscanf(name.c_str(), "blk.%d.%n", &layer, &consumed) == 1 && consumed > 0) {
    if (layer < 0 || layer >= c.num_layers) return fail("tensor %s: layer out of range", name.c_str());
    Block &b = m.blocks[layer];
    const std::string rest = name.substr(consumed);
    if (rest == "attn_norm.weight") return norm(b.input_layernorm);
    if (rest == "ffn_norm.weight") return norm(b.post_attention_layernorm);
    if (rest == "attn_q.weight") return lin(b.q_proj, nq, d);
    if (rest == "attn_k.weight") return lin(b.k_proj, nkv, d);
    if (rest == "attn_v.weight") return lin(b.v_proj, nkv, d);
    if (rest == "attn_output.weight") return lin(b.o_proj, d, nq);
    if (rest == "ffn_gate.weight") return lin(b.gate_proj, ff, d);
    if (rest == "ffn_up.weight") return lin(b.up_proj, ff, d);
    if (rest == "ffn_down.weight") return lin(b.down_proj, d, ff);
    // Mixtral (gguf/normal_bindings.rs, models/mixtral.rs:236-279): router + experts stacked over the leading axis
    const int64_t E = c.num_experts;
    auto exps = [&](QTensor &slot, int64_t er, int64_t ec) {
      if (E <= 0) return fail("tensor %s: the config has no experts", name.c_str());
      if (rows != E * er || cols != ec) return fail("tensor %s: shape [%lld, %lld], expected [%lld x %lld, %lld]", name.c_str(), (long long)rows, (long long)cols, (long long)E, (long long)er, (long long)ec);
      slot = QTensor{p, type, rows, cols};
      return 0;
    };
    if (rest == "ffn_gate_inp.weight") {
      if (type != F32 || rows != E || cols != d) return fail("tensor %s: the router must be F32 [%lld, %lld]", name.c_str(), (long long)E, (long long)d);
      b.router = (const float *)p;
      return 0;
    }
    if (rest == "ffn_gate_exps.weight") return exps(b.gate_exps, ff, d);
    if (rest == "ffn_up_exps.weight") return exps(b.up_exps, ff, d);
    if (rest == "ffn_down_exps.weight") return exps(b.down_exps, d, ff);
  }
  return fail("tensor %s: no binding for this name", name.c_str());
}

}  // namespace mrs_host

using mrs_host::Llama;

extern "C" const char *mrs_last_error(void) { return mrs_host::g_last_error.c_str(); }
extern "C" size_t mrs_llama_workspace_bytes(const mrs_llama_config *cfg) { return Llama::workspace_bytes(*cfg); }
extern "C" void *mrs_llama_create(const mrs_llama_config *cfg) {
  if (!cfg || cfg->num_layers <= 0 || cfg->hidden_size <= 0 || cfg->num_heads <= 0 || cfg->num_kv_heads <= 0 ||
      cfg->num_heads % cfg->num_kv_heads || cfg->head_dim <= 0 || cfg->max_batch <= 0 || cfg->max_batch > 8 ||
      cfg->rot_dim > cfg->head_dim || (cfg->rot_dim & 1) || cfg->num_experts < 0 ||
      (cfg->num_experts > 0 && (cfg->num_experts_per_tok < 1 || cfg->num_experts_per_tok > cfg->num_experts))) {
    mrs_host::fail("mrs_llama_create: invalid config");
    return nullptr;
  }
  return new Llama(*cfg);
}
extern "C" void mrs_llama_destroy(void *m) { delete (Llama *)m; }
extern "C" int mrs_llama_set_tensor(void *m, const char *name, const void *p, int type, int64_t rows, int64_t cols) {
  return mrs_host::bind_tensor(*(Llama *)m, name, p, type, rows, cols);
}
extern "C" int mrs_llama_set_dec_tensor(void *mm, const char *cname, const void *planes) {
  Llama &m = *(Llama *)mm;
  const std::string name = cname;
  auto bind = [&](mrs_dec_mat &slot, const mrs_host::QTensor *t) {
    if (!t || !t->data) return mrs_host::fail("decode layout for %s: register the tensor with mrs_llama_set_tensor first", cname);
    if (!mrs_dec_repack_bytes(t->dtype, t->rows, t->cols)) return mrs_host::fail("decode layout for %s: ggml dtype %d / shape not supported", cname, t->dtype);
    slot = mrs_dec_mat{planes, t->dtype, (long long)t->rows, (long long)t->cols};
    return 0;
  };
  auto lin = [&](mrs_dec_mat &slot, const std::unique_ptr<mrs_host::GgufMatMul> &l) { return bind(slot, l ? l->get_qtensor() : nullptr); };
  if (name == "output.weight") return lin(m.dlm_head, m.lm_head);
  int layer = -1, consumed = 0;
  if (sscanf(cname, "blk.%d.%n", &layer, &consumed) == 1 && consumed > 0 && layer >= 0 && layer < m.cfg.num_layers) {
    mrs_host::Block &b = m.blocks[layer];
    const std::string rest = name.substr(consumed);
    if (rest == "attn_q.weight") return lin(b.dq, b.q_proj);
    if (rest == "attn_k.weight") return lin(b.dk, b.k_proj);
    if (rest == "attn_v.weight") return lin(b.dv, b.v_proj);
    if (rest == "attn_output.weight") return lin(b.dout, b.o_proj);
    if (rest == "ffn_gate.weight") return lin(b.dgate, b.gate_proj);
    if (rest == "ffn_up.weight") return lin(b.dup, b.up_proj);
    if (rest == "ffn_down.weight") return lin(b.ddown, b.down_proj);
    if (rest == "ffn_gate_exps.weight") return bind(b.dgate_exps, &b.gate_exps);
    if (rest == "ffn_up_exps.weight") return bind(b.dup_exps, &b.up_exps);
    if (rest == "ffn_down_exps.weight") return bind(b.ddown_exps, &b.down_exps);
  }
  return mrs_host::fail("decode layout: tensor %s has no decode-engine role", cname);
}
// MFMA-order copy (mrs_gemm_qi_repack output, caller-owned) of a dense linear already registered with mrs_llama_set_tensor: the exact-integer prompt GEMM
// of ext_gemm_qi.hip reads it; a model whose dense linears all have one runs its prompts in the decode engine's arithmetic (Llama::prefill_exact)
extern "C" int mrs_llama_set_qi_tensor(void *mm, const char *cname, const void *planes) {
  Llama &m = *(Llama *)mm;
  const std::string name = cname;
  auto bind = [&](const std::unique_ptr<mrs_host::GgufMatMul> &l) {
    const mrs_host::QTensor *t = l ? l->get_qtensor() : nullptr;
    if (!t || !t->data) return mrs_host::fail("MFMA layout for %s: register the tensor with mrs_llama_set_tensor first", cname);
    if (!mrs_gemm_qi_repack_bytes(t->dtype, t->rows, t->cols)) return mrs_host::fail("MFMA-order copy for %s: ggml dtype %d / shape not supported", cname, t->dtype);
    t->qi = planes;
    return 0;
  };
  int layer = -1, consumed = 0;
  if (sscanf(cname, "blk.%d.%n", &layer, &consumed) == 1 && consumed > 0 && layer >= 0 && layer < m.cfg.num_layers) {
    mrs_host::Block &b = m.blocks[layer];
    const std::string rest = name.substr(consumed);
    if (rest == "attn_q.weight") return bind(b.q_proj);
    if (rest == "attn_k.weight") return bind(b.k_proj);
    if (rest == "attn_v.weight") return bind(b.v_proj);
    if (rest == "attn_output.weight") return bind(b.o_proj);
    if (rest == "ffn_gate.weight") return bind(b.gate_proj);
    if (rest == "ffn_up.weight") return bind(b.up_proj);
    if (rest == "ffn_down.weight") return bind(b.down_proj);
    auto bind_exps = [&](const mrs_host::QTensor &t) {  // stacked experts [E * n][k]: expert e's panels start at e * n / 32 (n % 32 == 0 is checked at use)
      if (!t.data) return mrs_host::fail("MFMA layout for %s: register the tensor with mrs_llama_set_tensor first", cname);
      if (!mrs_gemm_qi_repack_bytes(t.dtype, t.rows, t.cols)) return mrs_host::fail("MFMA-order copy for %s: ggml dtype %d / shape not supported", cname, t.dtype);
      t.qi = planes;
      return 0;
    };
    if (rest == "ffn_gate_exps.weight") return bind_exps(b.gate_exps);
    if (rest == "ffn_up_exps.weight") return bind_exps(b.up_exps);
    if (rest == "ffn_down_exps.weight") return bind_exps(b.down_exps);
  }
  if (name == "output.weight") return bind(m.lm_head);  // (round 6) lm_head of the batched decode steps on the matrix cores (ext_dec_mm.hip); prompts do not read it
  return mrs_host::fail("MFMA layout: tensor %s has no prompt-GEMM role", cname);
}
// bf16 shadow copy (mrs_dequantize(..., out_dtype = 30), caller-owned) of a dense linear already registered with mrs_llama_set_tensor: with one for every dense linear the
// selectable bf16 prompt path (mrs_llama_set_prefill_mode(model, 0)) runs plain bf16 library GEMMs (csrc/ext_gemm_lt.hip)
extern "C" int mrs_llama_set_bf16_tensor(void *mm, const char *cname, const void *rows_bf16) {
  Llama &m = *(Llama *)mm;
  const std::string name = cname;
  int layer = -1, consumed = 0;
  if (sscanf(cname, "blk.%d.%n", &layer, &consumed) == 1 && consumed > 0 && layer >= 0 && layer < m.cfg.num_layers) {
    mrs_host::Block &b = m.blocks[layer];
    const std::string rest = name.substr(consumed);
    const std::unique_ptr<mrs_host::GgufMatMul> *slot = rest == "attn_q.weight" ? &b.q_proj : rest == "attn_k.weight" ? &b.k_proj : rest == "attn_v.weight" ? &b.v_proj
        : rest == "attn_output.weight" ? &b.o_proj : rest == "ffn_gate.weight" ? &b.gate_proj : rest == "ffn_up.weight" ? &b.up_proj : rest == "ffn_down.weight" ? &b.down_proj : nullptr;
    if (slot) {
      const mrs_host::QTensor *t = *slot ? (*slot)->get_qtensor() : nullptr;
      if (!t || !t->data) return mrs_host::fail("bf16 shadow for %s: register the tensor with mrs_llama_set_tensor first", cname);
      t->bf16 = rows_bf16;
      return 0;
    }
  }
  return mrs_host::fail("bf16 shadow: tensor %s has no prompt-GEMM role", cname);
}
extern "C" int mrs_llama_bf16_shadow_ok(void *m) { return ((Llama *)m)->bf16_shadow_ok() ? 1 : 0; }
extern "C" int mrs_llama_set_mode(void *m, int use_fused) {
  if (use_fused < 0 || use_fused > 2) return mrs_host::fail("mrs_llama_set_mode: 0, 1 or 2");
  ((Llama *)m)->cfg.use_fused = use_fused;
  return 0;
}
extern "C" int mrs_llama_set_kv_cache(void *m, int layer, void *k, void *v) {
  Llama &l = *(Llama *)m;
  if (layer < 0 || layer >= l.cfg.num_layers) return mrs_host::fail("kv cache: layer %d out of range", layer);
  l.blocks[layer].key_cache = k; l.blocks[layer].value_cache = v;
  return 0;
}
extern "C" int mrs_llama_set_buffers(void *m, const mrs_llama_buffers *b) { return ((Llama *)m)->set_buffers(*b); }
extern "C" int mrs_llama_decode_step(void *m, int b, void *stream) { return ((Llama *)m)->decode_step(b, (hipStream_t)stream); }
extern "C" int mrs_llama_decode_step_chained(void *m, int b, void *stream) { return ((Llama *)m)->decode_step_chained(b, (hipStream_t)stream); }
extern "C" int mrs_llama_embed_state(void *m, int b, void *stream) { return ((Llama *)m)->embed_state(b, (hipStream_t)stream); }
extern "C" int mrs_llama_chained_ok(void *m, int b) { return ((Llama *)m)->chained_ok(b) ? 1 : 0; }
extern "C" int mrs_llama_forward_logits(void *m, int b, void *stream) { return ((Llama *)m)->forward_logits(b, (hipStream_t)stream); }
extern "C" double mrs_llama_decode_bytes(void *m, int b, int ctx) { return ((Llama *)m)->decode_bytes(b, ctx); }
extern "C" size_t mrs_llama_prefill_workspace_bytes(const mrs_llama_config *cfg, int T) { return Llama::prefill_workspace_bytes(*cfg, T); }
extern "C" int mrs_llama_prefill(void *m, const mrs_llama_prefill_args *a, int T, void *stream) {
  Llama &l = *(Llama *)m;
  if (!l.have_bufs) return mrs_host::fail("mrs_llama_set_buffers was not called");
  return l.prefill(*a, T, (hipStream_t)stream);
}
extern "C" double mrs_llama_prefill_flops(void *m, int T) { return ((Llama *)m)->prefill_flops(T); }
extern "C" int mrs_llama_prefill_is_exact(void *m) { return ((Llama *)m)->prefill_exact_ok() ? 1 : 0; }
extern "C" int mrs_llama_set_prefill_mode(void *m, int exact) { ((Llama *)m)->prefill_mode = exact; return 0; }
extern "C" int mrs_llama_set_comm(void *m, void *comm) { ((Llama *)m)->comm = comm; return 0; }
extern "C" int mrs_llama_set_p2p(void *m, void *p2p) { ((Llama *)m)->p2p = p2p; return 0; }
// Error word of the peer-mailbox route (blocking device read: call where the host synchronises anyway).  Non-zero = a granule never arrived within the bounded spin and the
// affected sums are NaN.  The word is PER RANK and the route must change on EVERY rank in the same step (a rank on RCCL next to a rank on p2p pairs collectives of
// different steps): this call only READS -- the host reduces the word (MAX) over the tensor-parallel ranks, and when the maximum is non-zero every rank calls
// mrs_llama_set_p2p(model, NULL), re-captures its decode graph and repeats the steps since its last check (Llama.p2p_sync_error does exactly this).
extern "C" int mrs_llama_check_p2p(void *m) {
  Llama *l = (Llama *)m;
  if (!l->p2p) return 0;
  return mrs_p2p_error(l->p2p);
}
