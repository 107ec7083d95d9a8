/*
 * include/mistralrs_quant.h -- C ABI of libmistralrsquant.so (gfx950 / MI355X).
 *
 * Drop-in for the static library `libmistralrsquant.a` that mistralrs-quant's build.rs links
 * (mistralrs-quant/build.rs:220-228,247-249).  Every entry point below has the SAME symbol name
 * and argument list as the CUDA launcher it replaces; `stream` carries a hipStream_t where the
 * reference passes a cudaStream_t.  Rust-side declarations: mistralrs-quant/src/gguf/ffi.rs,
 * src/rotary/ffi.rs, src/utils/ffi.rs.  Reference-side binding: see INTEGRATION.md.
 *
 * Contract: the caller owns every buffer (inputs, outputs, scratch); nothing here
 * allocates or frees device memory; every call only enqueues work on `stream` (graph-capturable);
 * pointers are raw device addresses; weights are unmodified GGUF block bytes, row-major [N][K/blk].
 */
#ifndef MISTRALRS_QUANT_H
#define MISTRALRS_QUANT_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- activation -> Q8_1 (36-byte blocks: half d, half sum(x), 32 x int8), rows zero-padded to
 *      kx_padded (a multiple of 512: MATRIX_ROW_PADDING, fast_mmvq.rs:21).
 *      replaces mmvq_gguf.cu:1606-1641; caller: gguf/fast_mmvq.rs:340-383 */
void launch_mmvq_gguf_quantize_q8_1_bf16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream);
void launch_mmvq_gguf_quantize_q8_1_f16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream);
void launch_mmvq_gguf_quantize_q8_1_f32(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream);

/* the MoE paths' copies of the same quantizer (kernels/indexed_moe/indexed_moe.cu:673-808,1016-1023 ; Rust: src/gguf/ffi.rs:16-60 ;
 * callers gguf/cuda.rs:514-588,1340-1640).  launch_quantize_q8_1: f32 input, grid of `num_blocks_x` blocks of 256 columns per row */
void launch_quantize_q8_1(const float *x, void *vy, int kx, int kx_padded, int num_blocks_x, int num_rows, void *stream);
void launch_quantize_q8_1_bf16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream);
void launch_quantize_q8_1_f16(const void *x, void *vy, int kx, int kx_padded, int num_rows, void *stream);

/* ---- decode GEMV, batch 1..8:  dst[j*stride_col_dst + row] = W[row,:] . y_j
 *      <t> in q4_0 q4_1 q5_0 q5_1 q8_0 q2_k q3_k q4_k q5_k q6_k ; <d> in f32 f16 bf16
 *      replaces mmvq_gguf.cu:1322-1600 (MMVQ_LAUNCHER_PLAIN / _FUSED_GLU / _FUSED_QKV);
 *      callers: gguf/fast_mmvq.rs:299 (plain), :472 (fused_glu), :682 (fused_qkv).
 *      ncols_x = K (unpadded), stride_col_y = Q8_1 blocks per batch column (kx_padded/32).
 *      fused_glu: dst = act(gate.y) * (up.y), activation codes 0 silu 1 gelu 2 relu 3 gelu_erf 4 sigmoid
 *      fused_qkv: X_dst[j*nrows_X + row] */
#define MRS_DECL_MMVQ(t, d)                                                                                        \
  void launch_mmvq_gguf_##t##_##d##_plain(const void *vx, const void *vy, void *dst, int ncols_x, int nrows_x,     \
                                          int stride_col_y, int stride_col_dst, int b_size, void *stream);        \
  void launch_mmvq_gguf_##t##_##d##_fused_glu(const void *vx_gate, const void *vx_up, const void *vy, void *dst,   \
                                              int ncols_x, int nrows_x, int stride_col_y, int stride_col_dst,      \
                                              int b_size, int activation, void *stream);                           \
  void launch_mmvq_gguf_##t##_##d##_fused_qkv(const void *vx_q, const void *vx_k, const void *vx_v, const void *vy, \
                                              void *q_dst, void *k_dst, void *v_dst, int ncols_x, int nrows_q,     \
                                              int nrows_k, int nrows_v, int stride_col_y, int b_size, void *stream);
#define MRS_DECL_MMVQ_T(t) MRS_DECL_MMVQ(t, f32) MRS_DECL_MMVQ(t, f16) MRS_DECL_MMVQ(t, bf16)
MRS_DECL_MMVQ_T(q4_0) MRS_DECL_MMVQ_T(q4_1) MRS_DECL_MMVQ_T(q5_0) MRS_DECL_MMVQ_T(q5_1) MRS_DECL_MMVQ_T(q8_0)
MRS_DECL_MMVQ_T(q2_k) MRS_DECL_MMVQ_T(q3_k) MRS_DECL_MMVQ_T(q4_k) MRS_DECL_MMVQ_T(q5_k) MRS_DECL_MMVQ_T(q6_k)
#undef MRS_DECL_MMVQ_T
#undef MRS_DECL_MMVQ

/* ---- indexed MoE forward: out[task][row] = W[indices[task]][row] . y[input_dim1 == 1 ? task / topk : task], task = token * topk + slot.
 *      all_weights: E stacked packed matrices [n][k / blk]; all_inputs: Q8_1 rows of k_padded / 32 blocks; all_outputs f32 [batch*topk][n].
 *      replaces kernels/indexed_moe/indexed_moe.cu:806-1157 ; Rust: src/gguf/ffi.rs:100-260 ; caller gguf/cuda.rs:514-588
 *      (qmatmul_indexed_moe_forward <- GgufMatMul::gather_forward_raw, gguf/mod.rs:485-516).  q8_1 here is Q8_1 as a WEIGHT format (ffi.rs:268): per block
 *      d_w d_x <q,u> + 4 s_w s_x, the literal result of the reference's four vec_dot_q8_1_q8_1 calls per block (indexed_moe.cu:483-502). */
#define MRS_DECL_IMOE(t)                                                                                                          \
  void launch_indexed_moe_forward_##t##_q8_1(const void *all_weights, const void *all_inputs, const unsigned int *indices,         \
                                             float *all_outputs, int n, int k, int batch, int topk, int k_padded, int input_dim1, \
                                             void *stream);
MRS_DECL_IMOE(q4_0) MRS_DECL_IMOE(q4_1) MRS_DECL_IMOE(q5_0) MRS_DECL_IMOE(q5_1) MRS_DECL_IMOE(q8_0) MRS_DECL_IMOE(q8_1)
MRS_DECL_IMOE(q2k) MRS_DECL_IMOE(q3k) MRS_DECL_IMOE(q4k) MRS_DECL_IMOE(q5k) MRS_DECL_IMOE(q6k)
#undef MRS_DECL_IMOE

/* ---- fused MoE decode pair (f32 in/out, Q8_1 activations, task = token * topk + slot, e = indices[task]):
 *        gate_up:  out[task][row]   = (up_w[e][row] . y[token]) * act(gate_w[e][row] . y[token])   act_type 0 gelu_pytorch_tanh, else silu
 *        down_agg: out[token][row] += topk_weights[task] * (w[e][row] . y[task])                    f32 atomics; caller zero-fills out
 *      replaces kernels/indexed_moe/indexed_moe.cu:1336-1477,1618-1726 ; Rust: src/gguf/ffi.rs:520-890 ; caller gguf/cuda.rs:1427-1640
 *      (moe_gemv_fused_gate_up / moe_gemv_down_aggregate <- FastExpertsWeights::forward_*, moe/experts/backends.rs:969-1100) */
#define MRS_DECL_MOE_DECODE(t)                                                                                                      \
  void launch_moe_gemv_fused_gate_up_##t##_q8_1(const void *gate_weights, const void *up_weights, const void *all_inputs,            \
                                                const unsigned int *indices, float *all_outputs, int n, int k, int batch, int topk,  \
                                                int k_padded, int act_type, void *stream);                                          \
  void launch_moe_gemv_down_aggregate_##t##_q8_1(const void *all_weights, const void *all_inputs, const unsigned int *indices,       \
                                                 const float *topk_weights, float *all_outputs, int n, int k, int batch, int topk,   \
                                                 int k_padded, void *stream);
MRS_DECL_MOE_DECODE(q4_0) MRS_DECL_MOE_DECODE(q4_1) MRS_DECL_MOE_DECODE(q5_0) MRS_DECL_MOE_DECODE(q5_1) MRS_DECL_MOE_DECODE(q8_0) MRS_DECL_MOE_DECODE(q8_1)
MRS_DECL_MOE_DECODE(q2k) MRS_DECL_MOE_DECODE(q3k) MRS_DECL_MOE_DECODE(q4k) MRS_DECL_MOE_DECODE(q5k) MRS_DECL_MOE_DECODE(q6k)
#undef MRS_DECL_MOE_DECODE

/* ---- MoE prompt path: route dispatch -> grouped GEMM -> weighted reduce.
 *      launch_moe_dispatch: counting sort of topk_ids[total_assignments] by expert: expert_bounds[num_experts + 1] (exclusive prefix of the
 *        counts), sorted_token_ids[pos] = flat route index, sorted_source_ids[pos] = flat / topk (may be NULL); expert_counts /
 *        expert_cursors [num_experts] are caller scratch and end up as counts / bounds[e + 1] like the reference's.  The order inside an
 *        expert's segment is ascending flat index (the reference's atomic cursors leave it unspecified).
 *      launch_moe_grouped_gemm_<t>: for sorted position ti in [bounds[e], bounds[e+1]), flat = sorted_token_ids[ti]:
 *        y row = input_dim1 == 0 ? ti : input_dim1 == 1 ? flat / topk : flat ;  acc = W[e][row] . y ;
 *        topk_weights ? atomicAdd(out[flat / topk][row], acc * topk_weights[flat]) : out[ti][row] = acc      (f32)
 *      launch_moe_weighted_reduce_flat*: out[token][h] = sum_slot float(in[token][slot][h]) * w[token][slot], f32 accumulate, -> int status
 *        (_flat: f32 -> f32, _flat_bf16: f32 -> bf16, _f16_input: f16 -> f16, _bf16_input: bf16 -> bf16)
 *      replaces kernels/moe_grouped/moe_grouped.cu:630-1235 ; Rust: src/gguf/ffi.rs:285-500 ; callers gguf/cuda.rs:590-930,1340-1420 */
void launch_moe_dispatch(const int32_t *topk_ids, int32_t *expert_bounds, int32_t *sorted_token_ids, int32_t *sorted_source_ids,
                         int total_assignments, int num_experts, int topk, int32_t *expert_counts, int32_t *expert_cursors, void *stream);
#define MRS_DECL_MOE_GROUPED(t)                                                                                                     \
  void launch_moe_grouped_gemm_##t(const void *all_weights, const void *all_inputs, const int32_t *expert_bounds,                    \
                                   const int32_t *sorted_token_ids, const float *topk_weights, float *all_outputs, int N, int K,     \
                                   int K_padded, int num_experts, int topk, int input_dim1, void *stream);
MRS_DECL_MOE_GROUPED(q4_0) MRS_DECL_MOE_GROUPED(q4_1) MRS_DECL_MOE_GROUPED(q5_0) MRS_DECL_MOE_GROUPED(q5_1) MRS_DECL_MOE_GROUPED(q8_0) MRS_DECL_MOE_GROUPED(q8_1)
MRS_DECL_MOE_GROUPED(q2k) MRS_DECL_MOE_GROUPED(q3k) MRS_DECL_MOE_GROUPED(q4k) MRS_DECL_MOE_GROUPED(q5k) MRS_DECL_MOE_GROUPED(q6k)
#undef MRS_DECL_MOE_GROUPED
int launch_moe_weighted_reduce_flat(const void *inputs, const float *topk_weights, void *outputs, int num_tokens, int hidden, int topk,
                                    void *stream);
int launch_moe_weighted_reduce_flat_bf16(const void *inputs, const float *topk_weights, void *outputs, int num_tokens, int hidden,
                                         int topk, void *stream);
int launch_moe_weighted_reduce_flat_f16_input(const void *inputs, const float *topk_weights, void *outputs, int num_tokens, int hidden,
                                              int topk, void *stream);
int launch_moe_weighted_reduce_flat_bf16_input(const void *inputs, const float *topk_weights, void *outputs, int num_tokens, int hidden,
                                               int topk, void *stream);

/* ---- dense (unquantized) decode GEMV:  Y[b][row] = T( sum_k A[row][k] X[b][k] + (has_bias ? bias[row] : 0) ), A [M, K], X [B, K], Y [B, M],
 *      f32 accumulation, batch_size 1..8 (anything else runs batch 1, as the reference's dispatch).
 *      replaces kernels/gemv/gemv.cu:50-282 ; Rust: src/gemv/ffi.rs:12-56 ; callers src/gemv/mod.rs:250-470 */
void launch_gemv_bf16(const void *A, const void *X, const void *bias, void *Y, int M, int K, int batch_size, bool has_bias, void *stream);
void launch_gemv_f16(const void *A, const void *X, const void *bias, void *Y, int M, int K, int batch_size, bool has_bias, void *stream);
void launch_gemv_f32(const void *A, const void *X, const void *bias, void *Y, int M, int K, int batch_size, bool has_bias, void *stream);

/* ---- RoPE, in place, arithmetic in the tensor dtype.  `rot_dim` = number of rotated PAIRS (= cos/sin row
 *      length); is_neox: pairs (i, i+rot_dim) else interleaved (2i, 2i+1); dtype 0 f16, 1 bf16, 2 f32.
 *      replaces kernels/rotary/rotary.cu:122-196 ; Rust: src/rotary/ffi.rs ; caller rotary/mod.rs:851 */
void rotary_embedding(void *query, void *key, void *cos_cache, void *sin_cache, int32_t is_neox, int32_t head_size,
                      int64_t num_tokens, int32_t rot_dim, int32_t num_heads, int32_t num_kv_heads, int64_t query_stride,
                      int64_t key_stride, uint32_t dtype, int64_t stream);
void rotary_embedding_positions(void *query, void *key, void *cos_cache, void *sin_cache, void *positions,
                                int32_t is_neox, int32_t head_size, int64_t num_tokens, int32_t rot_dim, int32_t seq_len,
                                int32_t num_heads, int32_t num_kv_heads, int64_t query_stride, int64_t key_stride,
                                uint32_t dtype, int64_t stream);

/* ---- out = T(act(float a)) * b, strided rows.  replaces kernels/ops/ops.cu:946-971 ; Rust: src/utils/ffi.rs:274-306 */
void fused_glu_f16(const void *a, const void *b, void *output, uint32_t rows, uint32_t cols, uint32_t a_row_stride,
                   uint32_t b_row_stride, int activation, void *stream);
void fused_glu_bf16(const void *a, const void *b, void *output, uint32_t rows, uint32_t cols, uint32_t a_row_stride,
                    uint32_t b_row_stride, int activation, void *stream);
void fused_glu_f32(const void *a, const void *b, void *output, uint32_t rows, uint32_t cols, uint32_t a_row_stride,
                   uint32_t b_row_stride, int activation, void *stream);

/* ---- MMQ: prompt-sized quantized matmul behind fast_mmq.rs (shared_lhs / fused_qkv / fused_glu / fused_ffn, gguf/fast_mmq.rs:388-447,528-821).
 *      Step 1, activations -> block_q8_1_mmq (144 B per 128 values: 16 header bytes + 128 x int8; block index (i0/128)*ne1 + i1; header
 *      D4 = 4 x f32 d, DS4 = 4 x (half d, half sum x), D2S6 = 2 x half d (per 64) + 6 x half sum (per 16, first 96 values); kernels/mmq_gguf/
 *      mmq_gguf.cuh:64-89).  x [ne1 rows][s01 stride] of type_x (0 = f32, 1 = f16, 30 = bf16), `ids` optional row gather, ne00 = K,
 *      ne0 = padded K.  The _glu variants quantize activation(gate) * up (product formed in the input dtype).
 *      replaces kernels/mmq_gguf/mmq_quantize.cu:104-198,236-325 (kernels), :386-476 (launchers); Rust: src/gguf/ffi.rs:1313-1408 */
#define MRS_DECL_MMQ_QUANTIZE(L)                                                                                                               \
  void launch_mmq_quantize_q8_1_##L(const void *x, const int32_t *ids, void *vy, int type_x, int64_t ne00, int64_t s01, int64_t s02, int64_t s03, \
                                    int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, void *stream);                                           \
  void launch_mmq_quantize_glu_q8_1_##L##_f32(const float *gate, const float *up, const int32_t *ids, void *vy, int64_t ne00, int64_t s01,       \
                                              int64_t ne0, int64_t ne1, int activation, void *stream);                                           \
  void launch_mmq_quantize_glu_q8_1_##L(const void *gate, const void *up, const int32_t *ids, void *vy, int type_x, int64_t ne00, int64_t s01,   \
                                        int64_t ne0, int64_t ne1, int activation, void *stream);
MRS_DECL_MMQ_QUANTIZE(D4) MRS_DECL_MMQ_QUANTIZE(DS4) MRS_DECL_MMQ_QUANTIZE(D2S6)
#undef MRS_DECL_MMQ_QUANTIZE
/*      Step 2, dst[col * nrows_x + row] = W[row, :] . y_col  (type_dst 0 = f32, 1 = f16, 30 = bf16); the layout of y is fixed by <t>:
 *      DS4 for q4_0 q4_1 q5_1 q4_k q5_k, D2S6 for q2_k, D4 for q5_0 q8_0 q3_k q6_k (mmq_gguf.cuh:100-135).  stride_row_x in blocks.
 *      tmp_fixup / cc / nsm / smpbo / warp_size belong to the reference's stream-k tiling and are accepted and ignored.  _moe: y columns are
 *      the routes in expert-sorted order, expert e owns [expert_bounds[e], expert_bounds[e+1]) and channel e of x, column j lands in dst
 *      column ids_dst[j] (f32, column stride stride_col_dst).
 *      replaces kernels/mmq_gguf/mmq_instance_<t>.cu:216-259 and DEFINE_MMQ_MOE_LAUNCHER (mmq_gguf.cuh:3968-4010); Rust: ffi.rs:1410-1452,
 *      callers fast_mmq.rs:421-437 (dense), gguf/cuda.rs (grouped MoE prompt path).  MI355X note: these entry points keep fast_mmq.rs linking and
 *      numerically equivalent (integer dots on the stored Q8_1 images and partial sums), as LDS-tiled i8 MFMA kernels (csrc/mmq.hip,
 *      profiles/round3_mmq.md); the C++ runner's own prompt paths are mrs_gemm_qi (the decode engine's arithmetic, default) and the fused
 *      block-dequant -> bf16 MFMA GEMM mrs_gemm_q_* (selectable), both in include/mrs_hip_ext.h. */
#define MRS_DECL_MMQ(t)                                                                                                                        \
  void launch_mmq_gguf_##t(void *tmp_fixup, const void *x, const void *y, void *dst, int64_t ncols_x, int64_t nrows_x, int64_t ncols_y,          \
                           int64_t stride_row_x, int64_t stride_col_dst, int cc, int nsm, int64_t smpbo, int warp_size, int type_dst,            \
                           void *stream);                                                                                                        \
  void launch_mmq_gguf_##t##_moe(void *tmp_fixup, const void *x, const void *y, const int32_t *ids_dst, const int32_t *expert_bounds, void *dst, \
                                 int64_t ncols_x, int64_t nrows_x, int64_t ncols_dst, int64_t stride_row_x, int64_t stride_col_dst,              \
                                 int64_t num_experts, int64_t ncols_max, int cc, int nsm, int64_t smpbo, int warp_size, void *stream);
MRS_DECL_MMQ(q4_0) MRS_DECL_MMQ(q4_1) MRS_DECL_MMQ(q5_0) MRS_DECL_MMQ(q5_1) MRS_DECL_MMQ(q8_0)
MRS_DECL_MMQ(q2_k) MRS_DECL_MMQ(q3_k) MRS_DECL_MMQ(q4_k) MRS_DECL_MMQ(q5_k) MRS_DECL_MMQ(q6_k)
#undef MRS_DECL_MMQ

/* ---- HQQ: unpack + dequantize (axis-0 groups).  Wq [h][w] packed (u8; i32 with ten 3-bit fields for 3 bit), scale / zero [w] of the
 *      output dtype, out [P*h][w], P = 1/2/4/8/10 values per packed element, most significant first:
 *          out[(c*h + r)*w + j] = (T(q_c(r, j)) - zero[j]) * scale[j]     evaluated in T
 *      NO stream argument (default stream), as the reference.  replaces kernels/hqq/hqq.cu:26-80,94-155,171-230,278-345,399-468 ;
 *      Rust: src/hqq/ffi.rs:1-74 ; caller HqqLayer::dequantize (hqq/mod.rs:874-1090) <- forward / dequantize_w (:1092-1100,1163-1171) */
#define MRS_DECL_HQQ(name)                                                                                     \
  void dequantize_##name##_f32(const void *wq_packed, const void *scale, const void *zero, void *out, int h, int w);  \
  void dequantize_##name##_f16(const void *wq_packed, const void *scale, const void *zero, void *out, int h, int w);  \
  void dequantize_##name##_bf16(const void *wq_packed, const void *scale, const void *zero, void *out, int h, int w);
MRS_DECL_HQQ(8bit_u8_kernel) MRS_DECL_HQQ(4bit_u8_kernel) MRS_DECL_HQQ(2bit_u8_kernel) MRS_DECL_HQQ(1bit_u8_kernel) MRS_DECL_HQQ(3bit_32_kernel)
#undef MRS_DECL_HQQ
/* ---- HQQ bit packing: input [num_input_elements rows][input_width] of unpacked values, output row r packs rows r + i*step
 *      (step = rows / P), i = 0 most significant.  replaces kernels/hqq/hqq_bitpack.cu:7-206 ; Rust: src/hqq/bitpack_ffi.rs:1-40 ;
 *      caller HqqBits::bitpack_type (hqq/mod.rs:150-400) */
void launch_pack_1bit_kernel(const uint8_t *d_input, uint8_t *d_output, size_t num_input_elements, size_t input_width, void *stream);
void launch_pack_2bit_kernel(const uint8_t *d_input, uint8_t *d_output, size_t num_input_elements, size_t input_width, void *stream);
void launch_pack_3bit_kernel(const uint32_t *d_input, int32_t *d_output, size_t num_input_elements, size_t input_width, void *stream);
void launch_pack_4bit_kernel(const uint8_t *d_input, uint8_t *d_output, size_t num_input_elements, size_t input_width, void *stream);
void launch_pack_8bit_kernel(const uint8_t *d_input, uint8_t *d_output, size_t num_elements, void *stream);

/* MI355X-native addition (not in the reference ABI): tile policy of the int8-MFMA MMQ kernels, see csrc/mmq.hip */
void mrs_mmq_set_small_tiles_below(int n);

#ifdef __cplusplus
}
#endif
#endif
