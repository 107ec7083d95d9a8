/*
 * include/mrs_hip_ext.h -- C ABI of libmrs_hip_ext.so: MI355X-native additions that have no counterpart
 * symbol in the reference (fused decode kernels + the host-side model runner written in C++ because the
 * reference host is compiled Rust and there is no Rust toolchain here).
 *
 * Everything is plain pointers + sizes, asynchronous on the given hipStream_t (void *stream), never
 * allocates device memory.  Functions returning int use 0 = ok, <0 = refused (mrs_last_error() says why).
 */
#ifndef MRS_HIP_EXT_H
#define MRS_HIP_EXT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- fused decode kernels (ext_decode.hip)
 * Activations/residual stream f32, weights raw GGUF blocks (types q4_k q5_k q6_k q8_0), KV cache bf16 in the
 * reference's paged layout.  Each call replaces a run of reference launches (models/llama.rs:68-157,243-260):
 *   mrs_decode_qkv        = RmsNorm + quantize_q8_1 + fused_qkv GEMV + rotary_embedding_positions + reshape_and_cache
 *   mrs_decode_gate_up    = RmsNorm + quantize_q8_1 + fused_glu GEMV + quantize_q8_1 (of the GLU output)
 *   mrs_decode_proj       = plain GEMV (+ residual add)
 *   mrs_decode_norm_proj  = RmsNorm + quantize_q8_1 + plain GEMV (final norm + lm_head)
 * and produces bit-identical results to that sequence of C-ABI launches. */
int mrs_decode_gemv_supported(int ggml_type);
int mrs_decode_qkv(const void *wq, const void *wk, const void *wv, int tq, int tk, int tv, int nq, int nk, int nv, int K,
                   const float *h, const float *norm_w, float eps, float *q_out, void *k_cache, void *v_cache,
                   const int64_t *slot_mapping, const int32_t *positions, const float *cos_t, const float *sin_t,
                   int head_dim, int rot_pairs, int num_kv_heads, int block_size, int b, void *stream);
int mrs_decode_gate_up(const void *wg, const void *wu, int type, int n, int K, const float *h, const float *norm_w, float eps,
                       int activation, void *y_out, int y_out_stride, int b, void *stream);
int mrs_decode_proj(const void *w, int type, int n, int K, const void *y_q8_1, int stride_col_y, float *out, int out_stride,
                    int accumulate, int b, void *stream);
/* tensor parallelism: out = out * resid_scale + W.y with resid_scale = 1 / world_size, followed by ONE sum all-reduce of `out`
 * (role of RowParallelLayer::forward + SumAllReduce, distributed/layers.rs:965-975, with the residual add kept fused) */
int mrs_decode_proj_scaled(const void *w, int type, int n, int K, const void *y_q8_1, int stride_col_y, float *out, int out_stride,
                           float resid_scale, int b, void *stream);
int mrs_vec_add_f32(float *a, const float *b, size_t n, void *stream); /* a += b */
int mrs_decode_norm_proj(const void *w, int type, int n, int K, const float *h, const float *norm_w, float eps, float *out,
                         int out_stride, int b, void *stream);
/* ---------------------------------------------------------------- decode engine (ext_dec.hip, dec_gemv.cuh, dec_core2.cuh)
 * The batch <= 8 decode step in the arithmetic of the reference CPU path (GgufMatMul::forward_raw -> candle QMatMul with f32
 * activations, mistralrs-quant/src/gguf/mod.rs:465-478): activations are quantized to Q8_K (K-quants) / Q8_0 (Q8_0 weights) inside
 * the kernels, integer block dots, f32 combination in the order "ORD-U" (one term per 256-value superblock, four runs, dec_core2.cuh) that the prompt
 * GEMM (mrs_gemm_qi) shares -- north_star's parity target.  Weights are read from a DECODE LAYOUT made once at load time from the unmodified GGUF
 * blocks (same bits; tiles of 16 superblocks = 4 rows x the 4 ORD-U chunks, four lanes per superblock, every plane lane-major; Q4_K / Q5_K sub-block scales expanded
 * from 6 to 8 bits; tensors of 0xF0000000 bytes and more, and expert stacks of more than 256 experts, are refused).  K % 256 == 0. */
typedef struct { const void *planes; int type; long long n, k; } mrs_dec_mat; /* planes: mrs_dec_repack output for a [n][k] tensor */
int mrs_dec_supported(int ggml_type);                         /* q4_k q5_k q6_k q8_0 */
size_t mrs_dec_repack_bytes(int ggml_type, long long n, long long k); /* 0 = unsupported type / shape */
int mrs_dec_repack(const void *gguf_blocks, int ggml_type, long long n, long long k, void *planes, void *stream);
/* RmsNorm + q/k/v projections + interleaved RoPE + KV-cache write (kv_dtype: 1 = bf16, 0 = f16 pages); q_out f32 [b][nq].  head_dim and block_size must be
 * powers of two, head_dim >= 8 (-1 otherwise: the epilogue indexes heads, cache blocks and the cache's x-groups with shifts and masks) */
int mrs_dec_qkv(const mrs_dec_mat *wq, const mrs_dec_mat *wk, const mrs_dec_mat *wv, const float *h, int ldh, const float *norm_w, float eps,
                float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, void *stream);
/* rotate-half ("neox") RoPE: wq / wk repacked from rows in pair order (inside every head the original rows 0, hd/2, 1, hd/2 + 1, ...); results land in the
 * model's dim order (RotaryEmbedding::forward with is_gpt_neox, mistralrs-core/src/layers.rs:2978); needs rot_pairs * 2 == head_dim */
int mrs_dec_qkv_neox(const mrs_dec_mat *wq, const mrs_dec_mat *wk, const mrs_dec_mat *wv, const float *h, int ldh, const float *norm_w, float eps,
                     float *q_out, void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t,
                     const float *sin_t, int head_dim, int rot_pairs, int num_kv_heads, int block_size, int kv_dtype, int b, void *stream);
/* RmsNorm + act(W_g x) * (W_u x) -> act_out f32 [b][ld_out]; n = rows per expert, expert_sel = device pointer to the expert id (NULL: dense) */
int mrs_dec_gate_up(const mrs_dec_mat *wg, const mrs_dec_mat *wu, int n, const int32_t *expert_sel, const float *h, int ldh, const float *norm_w,
                    float eps, int activation, float *act_out, int ld_out, int b, void *stream);
/* MoE decode: gate / up of ALL top-k experts of one token in one launch (expert_sel [topk] on the device, act_out [topk][ld_out]); -3: caller loops over mrs_dec_gate_up */
int mrs_dec_gate_up_topk(const mrs_dec_mat *wg, const mrs_dec_mat *wu, int n, const int32_t *expert_sel, int topk, const float *h, const float *norm_w, float eps,
                         int activation, float *act_out, int ld_out, void *stream);
/* MoE decode, top-2: out = (out * resid_scale + w[0] W_{sel[0]} x[0]) + w[1] W_{sel[1]} x[1] in one launch (x [2][ldx]); same bits as two mrs_dec_proj launches */
int mrs_dec_proj_top2(const mrs_dec_mat *w, int n, const int32_t *expert_sel, const float *x, int ldx, float *out, float resid_scale, const float *acc_scale, void *stream);
/* (RmsNorm when norm_w) + GEMV; mode 0: out = W x; mode 1: out = out * resid_scale + s * W x with s = *acc_scale (NULL: 1) */
int mrs_dec_proj(const mrs_dec_mat *w, int n, const int32_t *expert_sel, const float *x, int ldx, const float *norm_w, float eps, float *out,
                 int ld_out, int mode, float resid_scale, const float *acc_scale, int b, void *stream);
/* lm_head of a greedy batch-1 step: out = W . RmsNorm(x) and the launch's arg-max as a packed key -> atomicMax(*amax) (u64, zero before the launch; see
 * mrs_sample_advance_embed) */
int mrs_dec_proj_argmax(const mrs_dec_mat *w, int n, const float *x, int ldx, const float *norm_w, float eps, float *out, int ld_out, void *amax, void *stream);
/* bytes of the pre-quantized activation image of b columns of k values (the layout the GEMV prologue builds in LDS) */
size_t mrs_dec_act_image_bytes(int k, int b);
/* Decode attention of the engine in ONE launch (round 3 default): split-KV waves (32-token blocks, f32 online softmax through the reference's
 * fast_exp, attention/backends/cpu/elem.rs:417-433) publish their partials; the last workgroup to arrive on the (sequence, kv head) ticket merges them
 * in the order of single_q.rs run_barrier and writes out_f32 [b][num_heads * 128] (may be NULL when an image is requested) and, for even GQA groups
 * and b <= 8, the Q8_K activation image img_out (may be NULL) for mrs_dec_proj_img.  ticket: [b * num_kv_heads] u32, zero before the first call
 * (left zero); part_o / part_m / part_l: b * num_heads * mrs_decode_attention_max_splits(max_context_len) x {128, 1, 1} floats.
 * Returns 1 when the image was written, 0 when only out_f32 was, -1 for shapes outside the kernel (head size 128, block 32, GQA 1 / 2 / 4 / 8). */
int mrs_dec_attention(float *out_f32, void *img_out, unsigned *ticket, float *part_o, float *part_m, float *part_l, const float *q, const void *k_cache,
                      const void *v_cache, int num_kv_heads, float scale, const uint32_t *block_tables, const uint32_t *context_lens, int block_size,
                      int max_context_len, int num_seqs, int num_heads, int head_size, int max_blocks_per_seq, int q_stride, int kv_block_stride,
                      int kv_head_stride, int kv_dtype, int sliding_window /* > 0: attend the last W positions only (Mistral), 0 = all */, void *stream);
size_t mrs_dec_proj_img_max_bytes(void); /* largest activation image mrs_dec_proj_img takes (the LDS budget) */
/* GEMV on a pre-quantized activation image: Q8_K quantization for K-quant weights (what mrs_dec_attention writes), Q8_0 for Q8_0 weights (mrs_dec_act_image with that
 * weight type); the image carries no tag -- the caller pairs image and weight */
int mrs_dec_proj_img(const mrs_dec_mat *w, int n, const void *x_img, float *out, int ld_out, int mode, float resid_scale, int b, void *stream);
/* Batched decode (b = 2..8): the activation image built ONCE per phase instead of by each of the 256 GEMV workgroups.  mrs_dec_act_image: x [b][ldx] f32
 * (-> RmsNorm when norm_w) -> img_out (16-byte aligned, mrs_dec_act_image_bytes(k, b) bytes: one image per column group that fits LDS) for weights of type
 * `weight_type` (K-quants: Q8_K quantization, Q8_0: Q8_0 -- BlockQ8K::from_float / quantize_row_q8_0 as in the GEMV prologue, same bytes); the *_img entry
 * points are mrs_dec_qkv / mrs_dec_qkv_neox (neox != 0) / mrs_dec_gate_up (dense) / mrs_dec_proj on such an image: bit-identical results. */
int mrs_dec_act_image(const float *x, int ldx, const float *norm_w, float eps, int k, int weight_type, int b, void *img_out, void *stream);
int mrs_dec_qkv_img(const mrs_dec_mat *wq, const mrs_dec_mat *wk, const mrs_dec_mat *wv, const void *x_img, float *q_out, void *k_cache, void *v_cache,
                    const int64_t *slot_mapping, const int32_t *positions, const float *cos_t, const float *sin_t, int head_dim, int rot_pairs,
                    int num_kv_heads, int block_size, int kv_dtype, int b, int neox, void *stream);
int mrs_dec_gate_up_img(const mrs_dec_mat *wg, const mrs_dec_mat *wu, int n, const void *x_img, int activation, float *act_out, int ld_out, int b, void *stream);
/* Batched decode on the matrix cores (round 6, csrc/ext_dec_mm.hip): the same launches as mrs_dec_proj_img / mrs_dec_gate_up_img / mrs_dec_qkv_img (interleaved RoPE) for
 * b = 1..8 columns, on the MFMA-order copy of the weights (mrs_gemm_qi_repack) instead of the decode-layout copy: integer dots on v_mfma_i32_32x32x32_i8, the engine's f32 order --
 * the same bits as the vector-ALU kernels.  Reference role: MMVQ's batch 1..8 from one weight pass (kernels/mmvq_gguf/mmvq_gguf.cu:724-792, gguf/fast_mmvq.rs:52).
 * x_img = mrs_dec_act_image(.., weight type, b, ..).  mrs_dec_mm_supported: the type is one the route takes and k x b fits its LDS budget. */
int mrs_dec_mm_supported(int type, int k, int b);
unsigned long long mrs_dec_mm_launch_count(void); /* launches of the route since the library was loaded */
void mrs_dec_mm_timeline(void *buf); /* experiments: [grid * 4][8] u64 s_memrealtime stamps of the following mrs_dec_mm_* launches, or NULL (off) */
int mrs_dec_mm_proj(const void *qi, int type, int n, int k, const void *x_img, float *out, int ld_out, int mode, float resid_scale, int b, void *stream);
int mrs_dec_mm_gate_up(const void *qi_gate, const void *qi_up, int type, int n, int k, const void *x_img, int activation, float *act_out, int ld_out, int b, void *stream);
int mrs_dec_mm_qkv(const void *qi_q, int type_q, int nq, const void *qi_k, int type_k, int nk, const void *qi_v, int type_v, int nv, int k, const void *x_img, float *q_out,
                   void *k_cache, void *v_cache, const int64_t *slot_mapping, const int32_t *positions, const float *cos_t, const float *sin_t, int head_dim, int rot_pairs,
                   int num_kv_heads, int block_size, int kv_dtype, int b, void *stream);
/* Fused HQQ dequant-GEMV for decode (ext_hqq_gemv.hip): out [b][ldo] = x [b][ldx] . W^T (+ bias) straight from the packed 4-bit / 8-bit HQQ tensor (group 64, axis 0),
 * b <= 8; dtype 0 = f32, 1 = f16, 2 = bf16 for x / scale / zero / bias / out.  Role: HqqLayer::forward_raw (hqq/mod.rs:1092-1100,1163-1171) without materialising
 * dequantize_w(); the dequantized values are bit-identical to dequantize_{4,8}bit_u8_kernel_*.  -1 = outside the fused kernel (keep dequantize + dense matmul). */
int mrs_hqq_gemv(int bits, int dtype, const void *wq, const void *scale, const void *zero, const void *bias, const void *x, int ldx, void *out, int ldo,
                 int N, int K, int b, void *stream);
/* quantized (or f32/f16/bf16) embedding rows -> f32; role of QuantMethod::embedding_forward (lib.rs:1561, gguf/mod.rs:436) */
int mrs_embedding(const void *table, int type, const int32_t *ids, float *out, int K, int tokens, void *stream);
/* round 6, greedy batch-1 step: the arg-max of lm_head folded into its epilogue (mrs_dec_proj_argmax: one u64 packed (value, index) maximum, zero before the launch) and
 * ONE launch that turns it into the next id, advances the device-resident decode state (mrs_sample_greedy_advance's updates) and gathers the next id's embedding row into
 * h_next (mrs_embedding's values) -- the role of sample_cuda_top1_row (mistralrs-core/src/ops.rs:2206) + the next step's embedding lookup. */
int mrs_sample_advance_embed(int32_t *next_ids, int32_t *tokens_out, int tokens_out_stride, int32_t *step_counter, int32_t *positions, uint32_t *context_lens,
                             int64_t *slot_mapping, const uint32_t *block_tables, int max_blocks, int block_size, void *scratch, const void *table, int type, float *h_next,
                             int K, void *stream);
/* f32 rows -> Q8_1 blocks: same bytes as launch_mmvq_gguf_quantize_q8_1_f32 with kx_padded = 32*stride_blocks */
int mrs_quantize_rows_q8_1(const float *x, void *y, int K, int stride_blocks, int rows, void *stream);
/* greedy top-1 (first maximum wins) + on-device advance of the decode state, so a captured decode step
 * replays with no host work (role of sample_causal_gen greedy, pipeline/mod.rs:2485 + inputs_processor slot math
 * pipeline/inputs_processor.rs:900-922). scratch: 8*b bytes, zero-initialised once. */
int mrs_sample_greedy_advance(const float *logits, int vocab, int b, int32_t *next_ids, int32_t *tokens_out,
                              int tokens_out_stride, int32_t *step_counter, int32_t *positions, uint32_t *context_lens,
                              int64_t *slot_mapping, const uint32_t *block_tables, int max_blocks, int block_size,
                              void *scratch, void *stream);

/* ---------------------------------------------------------------- mixture of experts, decode (Mixtral: SparseMoeBlock::forward,
 * mistralrs-core/src/models/mixtral.rs:280-304; router = moe_router_topk, ops.rs:259-336; expert GEMVs = the indexed kernels of
 * mistralrs-quant/kernels/indexed_moe/indexed_moe.cu:892-1150).  Experts are stacked [E][N][K/blk] packed GGUF blocks
 * (ffn_{gate,up,down}_exps); the expert id of a launch is read ON THE DEVICE (expert_sel), so a captured decode graph follows
 * the routing of each new token.  One token per launch (b = 1). */
int mrs_moe_router_topk(const float *x_normed, const float *gate_w /* f32 [E][K] */, int tokens, int n_experts, int K, int top_k,
                        int renormalize, int32_t *ids /* [tokens][top_k] */, float *weights /* [tokens][top_k] */,
                        float *logits_out /* optional [tokens][E] */, void *stream);
/* router on the un-normed hidden state: RmsNorm(h) * norm_w (bit-identical to mrs_rms_norm_f32) inside the router's workgroup; -3: K * 4 bytes > 64 KiB of LDS */
int mrs_moe_router_topk_norm(const float *h, const float *norm_w, float eps, const float *gate_w, int tokens, int n_experts, int K, int top_k, int renormalize,
                             int32_t *ids, float *weights, void *stream);
/* the same as n_experts workgroups per token + the last arriver's softmax / top-k (round 6; same ids and weights bit for bit).  scratch: mrs_moe_router_split_scratch_bytes
 * bytes, zero before the first call (it returns to zero) */
size_t mrs_moe_router_split_scratch_bytes(int tokens, int n_experts);
int mrs_moe_router_topk_norm_split(const float *h, const float *norm_w, float eps, const float *gate_w, int tokens, int n_experts, int K, int top_k, int renormalize,
                                   int32_t *ids, float *weights, void *scratch, void *stream);
int mrs_moe_decode_gate_up(const void *wg, const void *wu, size_t expert_stride_bytes, const int32_t *expert_sel, int type, int n, int K,
                           const float *h, const float *norm_w, float eps, int activation, void *y_out, int y_out_stride, void *stream);
int mrs_moe_decode_down(const void *w, size_t expert_stride_bytes, const int32_t *expert_sel, const float *topk_weight, int type, int n, int K,
                        const void *y_q8_1, int stride_col_y, float *out, void *stream);

/* ---------------------------------------------------------------- prefill GEMM (ext_gemm.hip)
 * out[m*ldo + n] (+)= sum_k bf16(x[m*ldx + k]) * bf16(dequant(W)[n][k]),  f32 accumulate on the bf16 matrix cores.
 * W: raw GGUF blocks [N][K/blk] (q4_k q5_k q6_k q8_0); x f32 [M][ldx]; role of fast_mmq::{plain,fused_qkv,fused_glu,fused_ffn}
 * (mistralrs-quant/src/gguf/fast_mmq.rs:528-635,762-821) at the raw-activation boundary (hard part 7).
 * Returns 0, -1 for an unsupported type / shape (K % 256 for K-quants, K % 64 for Q8_0, ldx % 4). */
int mrs_gemm_q_f32(const void *w, int ggml_type, int N, int K, const float *x, int ldx, float *out, int ldo, int M, int accumulate,
                   void *stream);
/* same for up to 3 weight matrices of ONE type that share x (q/k/v, gate/up: role of fast_mmq::{fused_qkv,fused_glu}) */
int mrs_gemm_q_f32_multi(int nseg, const void *const *w, const int *N, float *const *out, const int *ldo, int ggml_type, int K,
                         const float *x, int ldx, int M, int accumulate, void *stream);

/* Large-M variant (M > 128 is where it pays): bf16 activations in k-slab-major layout x[K/64][M][64] (written once per GEMM group by
 * mrs_convert_f32_bf16_slabs: the A tile of a k-step is then one contiguous block instead of 256 rows 2*K bytes apart on the same L2
 * channels), 256 x 128 x 64 tiles with 8 waves, split-K through `workspace` (f32 partials, summed in a fixed order) when the shape
 * has fewer tiles than CUs.  workspace may be NULL (no split); mrs_gemm_q_bf16_workspace_bytes(M) always suffices.
 * Same arithmetic as mrs_gemm_q_f32 except for the f32 summation order across k. */
int mrs_gemm_q_bf16_multi(int nseg, const void *const *w, const int *N, float *const *out, const int *ldo, int ggml_type, int K,
                          const void *x_slabs, int M, int accumulate, void *workspace, size_t workspace_bytes, void *stream);
size_t mrs_gemm_q_bf16_workspace_bytes(int M);
/* Grouped MoE GEMM of a prompt on the matrix cores (same dispatch tables as launch_moe_dispatch / launch_moe_grouped_gemm_<t>, mistralrs_quant.h):
 * for expert e and sorted position pos in [bounds[e], bounds[e+1]): acc = W_e . x[row], row = gather ? sorted[pos] / topk : pos;
 * route_w ? atomicAdd(out[sorted[pos] / topk], route_w[sorted[pos]] * acc) : out[pos] = acc.  x_slabs: bf16 slabs [K/64][x_rows][64]; w: [E * N][K] blocks.
 * Per expert bit-identical to mrs_gemm_q_bf16_multi on that expert's rows. */
int mrs_moe_gemm_q_bf16(const void *w, int ggml_type, int N, int K, int num_experts, const void *x_slabs, int x_rows, const int32_t *bounds,
                        const int32_t *sorted, int topk, int gather, const float *route_w, float *out, int ldo, int routes, void *stream);
/* Fused gate / up of a prompt: y = act(W_g x) * (W_u x) as bf16 slabs y[N/64][M][64] in ONE launch (role of fast_mmq::fused_glu, gguf/fast_mmq.rs:762-821);
 * same bits as two mrs_gemm_q_bf16_multi launches + mrs_glu_bf16_slabs.  N % 64 == 0; y_slabs must not alias x_slabs. */
int mrs_gemm_q_bf16_glu(const void *w_gate, const void *w_up, int ggml_type, int N, int K, const void *x_slabs, int M, int activation, void *y_slabs, void *stream);
/* kernel behind mrs_gemm_q_bf16_multi: 1 = producer / consumer wave specialisation, 0 = every wave stages and multiplies (round 1), -1 = by weight type (default: Q4_K -> 1);
 * identical results, kept selectable for A/B measurements (also MRS_GEMM_VARIANT) */
void mrs_gemm_set_variant(int variant);
/* x f32 [M][ldx] -> bf16 (round to nearest even) slabs y[K/64][M][64]; K % 64 == 0, ldx % 4 == 0 */
int mrs_convert_f32_bf16_slabs(const float *x, int ldx, int M, int K, void *y, void *stream);
/* producers that write the slabs directly: act(g) * u (role of fused_glu, utils/ops.rs:2953; activation codes of mistralrs_quant.h) and
 * RMSNorm (role of the RmsNorm before q/k/v and gate/up; arithmetic of mrs_rms_norm_f32), each followed by the bf16 rounding */
int mrs_glu_bf16_slabs(const float *g, const float *u, int ld, int M, int N, int activation, void *y, void *stream);
int mrs_rms_norm_bf16_slabs(const float *x, const float *w, int M, int K, float eps, void *y, void *stream);

/* ---------------------------------------------------------------- causal prompt attention (ext_attn_prefill.hip)
 * softmax(scale * Q K^T + causal) V on the bf16 matrix cores with K / V read straight from the paged cache (the chunk has been
 * scattered with reshape_and_cache first); role of Sdpa::run_attention in the prompt branch of PagedAttention::forward
 * (attention/mod.rs:254-372, paged_attention.rs:1413-1475).  head_size 128, block_size 32, bf16 cache; -1 otherwise. */
int mrs_prefill_attention_window_f32_bf16(const float *q, const void *key_cache, const void *value_cache, const uint32_t *block_table, float *out,
                                          int T, int start_pos, int num_heads, int num_kv_heads, int head_size, int block_size, int q_stride,
                                          int o_stride, int kv_block_stride, int kv_head_stride, float scale, int sliding_window, void *stream);
int mrs_prefill_attention_f32_bf16(const float *q, const void *key_cache, const void *value_cache, const uint32_t *block_table, float *out,
                                   int T, int start_pos, int num_heads, int num_kv_heads, int head_size, int block_size, int q_stride,
                                   int o_stride, int kv_block_stride, int kv_head_stride, float scale, void *stream);

/* ---------------------------------------------------------------- host-side model runner (host/runtime.cpp)
 * C++ mirror of mistralrs-core/src/models/llama.rs (Llama / CausalSelfAttention / Mlp / Block) on top of
 * QuantMethod objects (mistralrs-quant/src/lib.rs:1515-1688, gguf/mod.rs GgufMatMul), exposed through handles. */
typedef struct {
  int32_t hidden_size, intermediate_size, num_layers, num_heads, num_kv_heads, head_dim, vocab_size;
  int32_t rot_dim;            /* rotated dims per head (<= head_dim) */
  int32_t rope_interleaved;   /* 1: GGUF llama/mistral pairing (2i,2i+1); 0: neox halves */
  float rms_eps;
  int32_t block_size;         /* paged KV block size (tokens) */
  int32_t max_blocks_per_seq;
  int32_t max_batch;          /* decode batch capacity (<= 8) */
  int32_t max_context_len;
  int32_t use_fused;          /* 2: decode engine (ext_dec.hip: reference CPU-path arithmetic, needs mrs_llama_set_dec_tensor for every linear);
                                 1: round-1 fused kernels (Q8_1 activations, bit-identical to the drop-in launch sequence); 0: reference launch sequence */
  int32_t world_size, rank;   /* tensor parallel (1, 0 = single GPU) */
  int32_t num_experts;        /* 0 = dense FFN; > 0: Mixtral-style sparse MoE FFN in every layer (models/mixtral.rs:236-304) */
  int32_t num_experts_per_tok; /* top-k of the router (softmax over all experts -> top-k -> renormalise) */
  int32_t kv_f16;             /* decode engine only: 1 = f16 KV pages (the reference CPU path's default KV dtype, kv_cache/mod.rs:66-89), 0 = bf16 */
  int32_t sliding_window;     /* > 0: Mistral sliding-window attention -- a query attends the last W positions, itself included (GGUF
                                 <arch>.attention.sliding_window, gguf/normal_config.rs:792-796); decode engine + MFMA prefill only; 0 = full causal */
} mrs_llama_config;

typedef struct {  /* all device pointers, owned by the caller */
  int32_t *input_ids;       /* [max_batch] */
  int32_t *positions;       /* [max_batch] */
  uint32_t *context_lens;   /* [max_batch] */
  int64_t *slot_mapping;    /* [max_batch] */
  uint32_t *block_tables;   /* [max_batch, max_blocks_per_seq] */
  int32_t *tokens_out;      /* [max_batch, tokens_out_stride] generated ids */
  int32_t tokens_out_stride;
  int32_t *step_counter;    /* [1] */
  const float *cos_table;   /* [max_pos, rot_dim/2] f32 */
  const float *sin_table;
  float *logits;            /* [max_batch, vocab] */
  void *workspace;          /* scratch, >= mrs_llama_workspace_bytes() */
  size_t workspace_bytes;
} mrs_llama_buffers;

size_t mrs_llama_workspace_bytes(const mrs_llama_config *cfg);
/* ---- prompts in the decode engine's arithmetic (csrc/ext_gemm_qi.hip, round 4): the role of the reference CPU prompt path (QMatMul f32 fallback per activation
 * row, mistralrs-quant/src/gguf/mod.rs:465-478): Q8_K activation rows x Q4_K / Q5_K / Q6_K weights as EXACT integers on v_mfma_f32_32x32x16_f16, combined in the f32
 * order of the decode engine -- row t of the result equals mrs_dec_proj on row t bit for bit.
 * mrs_gemm_qi_repack: GGUF blocks [n][k / 256] -> the MFMA-order copy (load time).  mrs_qi_quantize: x f32 [T][ldx] (RmsNorm in the engine's order when norm_w;
 * x2 != NULL: rows are silu(x) * x2, xtmp = f32 scratch [T][K]) -> the GEMM's operand buffers `act` (mrs_qi_act_bytes).  mrs_gemm_qi: out[t * ldo + n] (+)= W[n] . act[t].
 * mrs_prefill_attention_exact: causal attention of T prompt tokens over the paged cache, per token exactly mrs_dec_attention's arithmetic (context_lens[t] =
 * position + 1, block_table = the sequence's row, max_context_len = the model's).  Returns 0; -1 bad arguments / type; -2 shape beyond the kernel's LDS budget. */
size_t mrs_gemm_qi_repack_bytes(int ggml_type, long long n, long long k); /* 0 = unsupported type (q4_k, q5_k, q6_k, q8_0 are taken) / shape (k % 256) */
int mrs_gemm_qi_repack(const void *gguf_blocks, int ggml_type, long long n, long long k, void *dst, void *stream);
size_t mrs_qi_act_bytes(int T, int K);
int mrs_qi_quantize(const float *x, const float *x2, int ldx, const float *norm_w, float eps, int T, int K, void *act, float *xtmp, void *stream);
int mrs_gemm_qi(const void *w_qi, int ggml_type, int N, int K, const void *act, int T, float *out, int ldo, int accumulate, void *stream);
/* the same with a workspace (mrs_gemm_qi_workspace_bytes(T, N) or more): launches with few workgroups run one workgroup per run of superblocks + a reduce -- same bits */
size_t mrs_gemm_qi_workspace_bytes(int T, int max_n);
int mrs_gemm_qi_ws(const void *w_qi, int ggml_type, int N, int K, const void *act, int T, float *out, int ldo, int accumulate, void *workspace, size_t workspace_bytes,
                   void *stream);
int mrs_prefill_attention_exact(const float *q, const void *k_cache, const void *v_cache, const uint32_t *block_table, const uint32_t *context_lens, float *out, int T,
                                int num_heads, int num_kv_heads, int head_size, int block_size, int q_stride, int kv_block_stride, int kv_head_stride, float scale,
                                int max_context_len, int kv_dtype, int sliding_window, int max_prompt_ctx /* largest context_lens[t], or 0 */, void *stream);
/* round 6 -- the other vec_dot partners and callers of the reference path (gguf/mod.rs:465-478; moe/experts/backends.rs:969-1100; distributed/layers.rs:965-975):
 * mrs_qi_quantize_for: the operand rows in the format the weight type multiplies with (q8_0 weights: Q8_0 blocks per 32 values; K-quants: Q8_K; mrs_qi_quantize = the
 * K-quant form).  mrs_gemm_qi* take q8_0 weights too: one v_mfma_i32_32x32x32_i8 per block of 32, p_b = ((float)isum_b dw_b) dx_b, terms added in block order.
 * mrs_gemm_qi_win: the grouped form for sparse-MoE prompts: `act` holds T_total operand rows, only rows [win[0], win[1]) -- device ints, e.g. launch_moe_dispatch's
 * expert_bounds + e -- are multiplied and only those rows of `out` are written; max_rows (host) bounds the window and sizes the grid.
 * mrs_qi_gather_rows: operand rows of T tokens -> R = T * top_k rows in the order of `sorted_routes` (flat route index t * top_k + slot per sorted position), and
 * inv[route] = its sorted position.  mrs_moe_fold_exact: h[t] <- fold over slots of h * (slot == 0 ? resid_scale : 1) + y_sorted[inv[t * top_k + slot]] * weights[..]
 * (the decode step's RESID epilogue per expert slot).  mrs_resid_scale_add_f32: h <- h * resid_scale + y (tensor-parallel row-parallel projections before the all-reduce). */
int mrs_qi_quantize_for(int w_ggml_type, const float *x, const float *x2, int ldx, const float *norm_w, float eps, int T, int K, void *act, float *xtmp, void *stream);
int mrs_gemm_qi_win(const void *w_qi, int ggml_type, int N, int K, const void *act, int T_total, const int *win, int max_rows, float *out, int ldo, int accumulate, void *stream);
int mrs_qi_gather_rows(int w_ggml_type, const void *act_src, int T, void *act_dst, int R, int K, const int *sorted_routes, int top_k, int *inv, void *stream);
int mrs_moe_fold_exact(float *h, float resid_scale, const float *y_sorted, const int *inv, const float *weights, int T, int d, int top_k, void *stream);
int mrs_resid_scale_add_f32(float *h, float resid_scale, const float *y, size_t n, void *stream);
int mrs_llama_set_qi_tensor(void *model, const char *name, const void *planes); /* MFMA-order copy of a linear (dense, or a stacked expert tensor) registered with mrs_llama_set_tensor */
int mrs_llama_prefill_is_exact(void *model); /* 1: mrs_llama_prefill runs in the decode engine's arithmetic (every linear has its MFMA-order copy, decode engine on; tensor-parallel shards and sparse-MoE layers included) */
/* prompt arithmetic: 1 = the decode engine's (default where mrs_llama_prefill_is_exact allows it), 0 = bf16-operand MFMA GEMMs + MFMA flash attention (faster TTFT,
 * logits within ~1e-2 of the engine's instead of identical), -1 = follow the MRS_PREFILL_EXACT environment variable */
int mrs_llama_set_prefill_mode(void *model, int exact); /* 2 = the bf16 path with the fused block-dequant kernels forced (A / B against the bf16 shadow copy) */
/* round 6: the selectable bf16 prompt path on a bf16 SHADOW COPY of the dense linears (csrc/ext_gemm_lt.hip): rows [n][k] bf16 from mrs_dequantize(..., 30) at load time,
 * registered per tensor; with one for every dense linear, prefill mode 0 runs plain bf16 x bf16 -> f32 library GEMMs (hipBLASLt) on bf16 activation rows.
 * mrs_lt_gemm_bf16: out f32 [T][ldo] (+)= x [T][K] . w [N][K]^T; first call per shape plans (not inside a stream capture). */
int mrs_llama_set_bf16_tensor(void *model, const char *name, const void *rows_bf16);
int mrs_llama_bf16_shadow_ok(void *model);
int mrs_lt_gemm_bf16(const void *w_bf16, const void *x_bf16, float *out, int ldo, int N, int K, int T, int accumulate, void *stream);
int mrs_rows_f32_to_bf16(const float *x, int ldx, int M, int K, void *y, void *stream);
int mrs_rows_glu_bf16(const float *g, const float *u, int ld, int M, int N, void *y, void *stream);
int mrs_rows_rms_norm_bf16(const float *x, const float *w, int M, int K, float eps, void *y, void *stream);
/* diagnostics: pull a byte range through the Infinity Cache (ext_prefetch.hip); one wave of K-deep v_mfma_f32_32x32x16_f16 on caller operands (the exactness
 * premise of mrs_gemm_qi, tests/test_gemm_qi.py) */
int mrs_l3_prefetch(const void *p, size_t bytes, int workgroups, void *sink, void *stream);
int mrs_mfma_f16_int_probe(const void *a, const void *b, float *out, int ksteps, void *stream);
/* decode-layout copy (mrs_dec_repack output, caller-owned) of a linear tensor already registered with mrs_llama_set_tensor */
int mrs_llama_set_dec_tensor(void *model, const char *name, const void *planes);
int mrs_llama_check_p2p(void *model); /* blocking read of THIS rank's p2p error word (the route is NOT changed).  MANDATORY host protocol: reduce the word with MAX over the tensor-parallel ranks; when the maximum is non-zero EVERY rank calls mrs_llama_set_p2p(model, NULL) in the same step, re-captures its graphs and repeats the steps since the last check -- a route dropped on one rank only pairs RCCL calls of different steps */
int mrs_llama_set_mode(void *model, int use_fused); /* switch the decode path (values of mrs_llama_config.use_fused) */
void *mrs_llama_create(const mrs_llama_config *cfg);
void mrs_llama_destroy(void *model);
/* name = GGUF tensor name ("token_embd.weight", "blk.3.attn_q.weight", "output_norm.weight", ...: the binding table of
 * mistralrs-core/src/gguf/normal_bindings.rs:40-220); shape [n_rows, n_cols]; data stays owned by the caller.
 * MoE layers: "blk.N.ffn_gate_inp.weight" (F32 router [E, hidden]) and the stacked experts "blk.N.ffn_{gate,up,down}_exps.weight"
 * given as [E * rows_per_expert, cols] (rank-3 GGUF tensors flattened over the expert axis) */
int mrs_llama_set_tensor(void *model, const char *name, const void *dev_ptr, int ggml_type, int64_t n_rows, int64_t n_cols);
int mrs_llama_set_kv_cache(void *model, int layer, void *key_cache, void *value_cache);
int mrs_llama_set_buffers(void *model, const mrs_llama_buffers *bufs);
/* one decode step for b sequences: embedding -> L x Block -> norm -> lm_head -> greedy sample + state advance */
int mrs_llama_decode_step(void *model, int b, void *stream);
/* the chained greedy step (batch 1, decode engine): like mrs_llama_decode_step, but it expects the hidden-state buffer to hold the embedding row of input_ids already
 * (left there by the previous chained step, or by mrs_llama_embed_state after the host changed input_ids) -- what Llama.capture_decode_graph captures */
int mrs_llama_decode_step_chained(void *model, int b, void *stream);
int mrs_llama_embed_state(void *model, int b, void *stream);
int mrs_llama_chained_ok(void *model, int b);
/* same graph without sampling: leaves logits [b, vocab] (parity tests read them) */
int mrs_llama_forward_logits(void *model, int b, void *stream);
/* Prefill of T prompt tokens of ONE sequence (role of the prompt branch of Llama::forward_embeds + PagedAttention::forward,
 * models/llama.rs:487-518, paged_attention/layers/paged_attention.rs:1413-1475): per layer RMSNorm -> q/k/v GEMMs (bf16 MFMA,
 * mrs_gemm_q_f32) -> RoPE -> reshape_and_cache -> causal attention over the freshly written pages -> o_proj GEMM (+residual) ->
 * RMSNorm -> gate/up GEMMs -> SiLU*up -> down GEMM (+residual); lm_head only for the last token (ctx.logits, llama.rs:514-517).
 * All pointers are device pointers owned by the caller; positions[t] = start + t; slot_mapping[t] as inputs_processor.rs:900-922;
 * block_tables: T identical rows [T][max_blocks] (every prompt token attends through the sequence's table with its own
 * context_lens[t] = positions[t] + 1). */
typedef struct {
  const int32_t *token_ids;     /* [T] */
  const int32_t *positions;     /* [T] */
  const int64_t *slot_mapping;  /* [T] */
  const uint32_t *block_tables; /* [T, max_blocks_per_seq] */
  const uint32_t *context_lens; /* [T] */
  float *logits;                /* [vocab] logits of the last prompt token */
  int32_t start_pos;            /* position of the first prompt token (= positions[0]) */
  void *workspace;              /* >= mrs_llama_prefill_workspace_bytes(cfg, T) */
  size_t workspace_bytes;
} mrs_llama_prefill_args;
size_t mrs_llama_prefill_workspace_bytes(const mrs_llama_config *cfg, int T);
int mrs_llama_prefill(void *model, const mrs_llama_prefill_args *args, int T, void *stream);
/* floating-point operations of one prefill of T tokens starting at an empty context (MFMA roofline numerator) */
double mrs_llama_prefill_flops(void *model, int T);
/* bytes of weights + KV streamed from HBM by one decode step at the given context (roofline numerator) */
double mrs_llama_decode_bytes(void *model, int b, int context_len);
/* ---------------------------------------------------------------- in-situ quantization (ext_isq.hip)
 * dense f32 (0) / f16 (1) / bf16 (30) weights [n_elements] -> GGML Q8_0 blocks (34 B per 32), the per-block rule of candle's
 * QTensor::quantize(.., Q8_0) that `generate_isq!` runs on the CPU (mistralrs-quant/src/utils/isq.rs:323-361), on the device.
 * Returns 0, -1 if n_elements % 32 != 0 or the dtype is unknown. */
int mrs_isq_quantize_q8_0(const void *src, int src_dtype, void *dst, long long n_elements, void *stream);
/* the other GGML targets of `generate_isq!`, quantized on the device with GGML's reference arithmetic (bit-identical blocks):
 * ggml_type 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0, 12 Q4_K, 13 Q5_K, 14 Q6_K (one lane per sub-block, 8 / 16 lanes per K-quant superblock).
 * Returns 0, -1 for an unknown dtype / type or n_elements not a multiple of the block size (32 / 256). */
int mrs_isq_quantize(const void *src, int src_dtype, void *dst, long long n_elements, int ggml_type, void *stream);
/* QuantMethod::dequantize_w for GGUF blocks (mistralrs-quant/src/gguf/mod.rs:430-432 -> candle QTensor::dequantize): packed [nrows][K/blk]
 * of ggml_type 2,3,6,7,8,10..14 -> dense [nrows][K] of out_dtype 0 f32 / 1 f16 / 30 bf16; w = scale*q - offset in f32 (the format spec of
 * kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu:141-278), one rounding to the output dtype.  Returns 0 / -1. */
/* importance-weighted ISQ (candle QTensor::quantize_imatrix; reference call sites gguf/mod.rs:238-252): Q4_K / Q5_K / Q6_K targets, imatrix f32 [k] on the device */
int mrs_isq_quantize_imatrix(const void *src, int src_dtype, void *dst, long long nrows, int k, int ggml_type, const float *imatrix, void *stream);
int mrs_dequantize(const void *w, int ggml_type, long long nrows, int K, void *out, int out_dtype, void *stream);
/* imatrix statistics of a layer (mistralrs-quant/src/imatrix.rs:73-135: ImatrixLayerStats::process / process_routed): accum[c] += sum over rows of
 * x[r][c]^2 in row order; routed: every (token, slot) row is added to accum[ids[t][s]] and counts[ids[t][s]] += 1.  dtype 0 f32 / 1 f16 / 30 bf16. */
int mrs_imatrix_accumulate(const void *x, int dtype, long long rows, int cols, float *accum, void *stream);
int mrs_imatrix_accumulate_routed(const void *x, int dtype, const uint32_t *ids, int n, int k, int cols, int per_slot, int num_experts, float *accum,
                                  float *counts, void *stream);

/* ---------------------------------------------------------------- paged KV cache manager (host/kv_cache_manager.cpp; host code only)
 * Block pool with prefix caching + per-request block tracking: the C++ counterpart of mistralrs-core/src/paged_attention/
 * block_pool.rs:267-557 (BlockPool), kv_cache_manager.rs:43-437 (KVCacheManager) and block_hash.rs:121-306 (chain hashes over full
 * blocks).  Same operations, argument meaning and outcomes; block ids are int64_t, -1 stands for the reference's `None`, -2 = the
 * caller's output capacity is too small / an id is out of range (nothing is changed in that case).  Not thread safe (the reference
 * calls it from the one engine thread per rank).  The tables it produces feed reshape_and_cache / paged_attention / mrs_llama_*. */
uint64_t mrs_kv_siphash(const void *data, size_t n, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds);
/* hash_block_tokens: SipHash-1-3(key 0) over parent|0, len, tokens, then the optional extra keys AdapterGeneration (32 bytes) and CacheSalt */
uint64_t mrs_kv_hash_block_tokens(int has_parent, uint64_t parent, const uint32_t *tokens, size_t n, const uint8_t *adapter_generation32,
                                  const char *cache_salt);
/* compute_new_block_hashes (n_existing = 0: compute_block_hashes): chain hashes of the full blocks after `existing`; -> count written */
size_t mrs_kv_compute_block_hashes(const uint32_t *tokens, size_t n_tokens, size_t block_size, const uint64_t *existing, size_t n_existing,
                                   const uint8_t *adapter_generation32, const char *cache_salt, uint64_t *out, size_t cap);
void *mrs_kv_manager_create(size_t num_gpu_blocks, size_t block_size, int enable_caching, const uint32_t *group_ids, size_t n_groups);
void mrs_kv_manager_destroy(void *mgr);
size_t mrs_kv_null_block_id(void *mgr);
size_t mrs_kv_block_size(void *mgr);
double mrs_kv_usage(void *mgr);
size_t mrs_kv_num_free_blocks(void *mgr);
size_t mrs_kv_num_usable_blocks(void *mgr);
size_t mrs_kv_num_gpu_blocks(void *mgr);
int mrs_kv_caching_enabled(void *mgr);
/* longest cached prefix (at most num_tokens - 1 tokens): -> number of block ids written, *num_computed_tokens = that * block_size */
int64_t mrs_kv_get_computed_blocks(void *mgr, const uint64_t *hashes, size_t n_hashes, size_t num_tokens, int64_t *block_ids, size_t cap,
                                   size_t *num_computed_tokens);
/* -> number of NEW block ids written (0 if the request already has enough), -1 = not enough free blocks */
int64_t mrs_kv_allocate_slots(void *mgr, uint64_t request_id, size_t num_tokens, const int64_t *computed_blocks, size_t n_computed,
                              int64_t *new_block_ids, size_t cap);
void mrs_kv_free(void *mgr, uint64_t request_id);
void mrs_kv_trim_request_to_num_tokens(void *mgr, uint64_t request_id, size_t num_tokens);
int mrs_kv_cache_blocks(void *mgr, uint64_t request_id, const uint64_t *hashes, size_t n_hashes, size_t num_computed_tokens);
int64_t mrs_kv_get_block_ids(void *mgr, uint64_t request_id, int64_t *out, size_t cap);
size_t mrs_kv_num_blocks_for_request(void *mgr, uint64_t request_id);
int mrs_kv_has_request(void *mgr, uint64_t request_id);
size_t mrs_kv_num_cached_blocks_for_request(void *mgr, uint64_t request_id);
int mrs_kv_reset_prefix_cache(void *mgr);
/* slots[i] = block * block_size + offset of token start_token + i (i64, what reshape_and_cache takes); -1 (_PAD_SLOT_ID) past the allocation */
int mrs_kv_get_slot_mapping(void *mgr, uint64_t request_id, size_t start_token, size_t num_tokens, int64_t *slots);
int mrs_kv_get_block_table(void *mgr, uint64_t request_id, size_t max_blocks, int32_t *table); /* zero padded */
/* the pool on its own (BlockPool's public surface); mrs_kv_manager_pool(mgr) borrows the manager's pool */
void *mrs_kv_pool_create(size_t num_gpu_blocks, int enable_caching, size_t hash_block_size);
void mrs_kv_pool_destroy(void *pool);
void *mrs_kv_manager_pool(void *mgr);
size_t mrs_kv_pool_null_block_id(void *pool);
size_t mrs_kv_pool_num_free_blocks(void *pool);
size_t mrs_kv_pool_num_gpu_blocks(void *pool);
double mrs_kv_pool_usage(void *pool);
size_t mrs_kv_pool_num_cached_blocks(void *pool);
size_t mrs_kv_pool_hash_block_size(void *pool);
int mrs_kv_pool_caching_enabled(void *pool);
int64_t mrs_kv_pool_block_ref_cnt(void *pool, int64_t block_id);
int64_t mrs_kv_pool_num_block_hashes(void *pool, int64_t block_id);
int64_t mrs_kv_pool_get_new_blocks(void *pool, size_t n, int64_t *out, size_t cap);
int mrs_kv_pool_free_blocks(void *pool, const int64_t *ordered_block_ids, size_t n);
int mrs_kv_pool_touch(void *pool, const int64_t *block_ids, size_t n);
int mrs_kv_pool_cache_full_blocks(void *pool, const int64_t *block_ids, size_t n_ids, const uint64_t *hashes, size_t n_hashes,
                                  size_t num_cached_blocks, size_t num_full_blocks, uint32_t group_id);
int64_t mrs_kv_pool_get_cached_block(void *pool, uint64_t hash, const uint32_t *group_ids, size_t n_groups, int64_t *out);
int mrs_kv_pool_reset_prefix_cache(void *pool);
void mrs_kv_pool_set_ref_cnt_for_test(void *pool, int64_t block_id, uint32_t value);

/* ---------------------------------------------------------------- RCCL over xGMI (ext_comm.hip), one process per GPU
 * replaces Comm::from_device / all_reduce(Sum) of mistralrs-quant/src/distributed/mod.rs:244-303,511-809 */
int mrs_comm_unique_id(void *out128);                            /* rank 0: ncclGetUniqueId -> 128 bytes */
void *mrs_comm_init(const void *id128, int rank, int world);     /* ncclCommInitRank on the current device; NULL on error */
int mrs_comm_all_reduce_sum_f32(void *comm, float *buf, size_t count, void *stream); /* in place, asynchronous on stream */
int mrs_comm_nranks(void *comm); /* ncclCommCount */
/* ---- one-shot all-reduce over peer-mapped mailboxes (ext_p2p.hip) for decode-sized messages: every rank writes its vector into every peer's
 * mailbox over xGMI (one hop), polls its own mailbox and sums in rank order (bit-identical on all ranks); graph-capturable; messages larger than
 * max_elems return -2 (keep RCCL for those).  Replaces ncclAllReduce at distributed/mod.rs:584-587 for [b, hidden] messages. */
size_t mrs_p2p_mailbox_bytes(int world, size_t max_elems);
void *mrs_p2p_alloc_mailbox(size_t bytes); /* zeroed FINE-GRAINED / uncached device memory (hipExtMallocWithFlags): peers write into it while the owner polls it */
void mrs_p2p_free_mailbox(void *p);
int mrs_ipc_get_handle(void *dev_ptr, void *out64);      /* hipIpcGetMemHandle */
void *mrs_ipc_open_handle(const void *in64);             /* hipIpcOpenMemHandle */
int mrs_ipc_close_handle(void *ptr);
void *mrs_p2p_create(int rank, int world, void *const *mailboxes, size_t max_elems); /* mailboxes: zeroed, mrs_p2p_mailbox_bytes each */
void mrs_p2p_destroy(void *comm);
size_t mrs_p2p_max_elems(void *comm);
int mrs_p2p_all_reduce_sum_f32(void *comm, float *buf, size_t count, void *stream);
int mrs_p2p_post(void *comm, const float *buf, size_t count, void *stream);   /* the two halves as separate launches (tests) */
int mrs_p2p_reduce(void *comm, float *buf, size_t count, void *stream);
int mrs_p2p_all_reduce_group(void *const *comms, float *const *bufs, int n, size_t count, void *stream); /* test: n ranks of one address space as one grid */
int mrs_p2p_error(void *comm);                            /* 1: a peer never posted (blocking read of the error word) */
int mrs_llama_set_p2p(void *model, void *p2p_comm);      /* row-parallel all-reduces of <= max_elems values take this route, larger ones RCCL */
void mrs_comm_destroy(void *comm);
int mrs_llama_set_comm(void *model, void *comm);                 /* required when cfg.world_size > 1 */
const char *mrs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
