/*
 * oracle/ggml_oracle.h -- TEST INFRASTRUCTURE ONLY (not shipped, not on the product path).
 *
 * CPU restatement of the arithmetic underneath mistral.rs' quantized hot path:
 * GGML block formats, activation quantizers, the GPU (Q8_1) and CPU (Q8_K / Q8_0)
 * dot-product semantics, and the glue ops (RMSNorm, RoPE, SiLU-GLU, attention).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY STATUS: "parity unpinned" for everything that lives in candle (CPU QMatMul,
 * quantizers): candle@35d7ae7 is a git dependency that is not vendored under
 * /root/reference and there is no Rust toolchain here.  PINNED against the reference's own
 * device code compiled for / executed on the host (oracle/build_ref.sh -> oracle/_ref/*.so,
 * oracle/ref_shim/: CUDA-vocabulary shim + a fiber runtime with block barriers and warp
 * shuffles): block decode, the Q8_1 quantizer, the MMVQ dot products and complete kernels,
 * GLU activations, the RMSNorm family, RoPE, paged-cache ops and paged attention v1 / v2 (f32).
 */
#ifndef GGML_ORACLE_H
#define GGML_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids -- reference: mistralrs-quant/src/gguf/archive.rs:73-160 */
enum orc_type {
  ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q4_1 = 3, ORC_Q5_0 = 6, ORC_Q5_1 = 7,
  ORC_Q8_0 = 8, ORC_Q8_1 = 9, ORC_Q2_K = 10, ORC_Q3_K = 11, ORC_Q4_K = 12,
  ORC_Q5_K = 13, ORC_Q6_K = 14, ORC_Q8_K = 15, ORC_BF16 = 30
};

int orc_block_size(int type);   /* elements per block  (0 = unknown type) */
int orc_type_size(int type);    /* bytes per block */

float    orc_fp16_to_fp32(uint16_t h);
uint16_t orc_fp32_to_fp16(float f);      /* round-to-nearest-even */
float    orc_bf16_to_fp32(uint16_t h);
uint16_t orc_fp32_to_bf16(float f);      /* round-to-nearest-even */

/* weights: blocks <-> f32 */
void orc_dequantize_row(int type, const void *blocks, float *out, int64_t k);
int  orc_quantize_row(int type, const float *x, void *blocks, int64_t k); /* 0 ok, -1 unsupported */
int  orc_quantize_row_imatrix(int type, const float *x, void *blocks, int64_t k, const float *qw); /* Q4_K / Q5_K / Q6_K with importance weights qw[k] (GGML quantize_row_*_impl) */
/* fill n_blocks of `type` with random-but-valid block bytes (finite, sane scales) */
void orc_random_blocks(int type, void *blocks, int64_t n_blocks, uint64_t seed, float d_scale);

/* activations */
/* GPU semantics (mmvq_gguf.cu:1220-1250): rows padded with zeros to kx_padded, d=amax/127,
 * q=roundf(x/d), ds = (half d, half sum(x)) */
void orc_quantize_q8_1(const float *x, void *y, int kx, int kx_padded, int rows);
void orc_quantize_q8_K(const float *x, void *y, int64_t k); /* candle BlockQ8K::from_float */
/* block_q8_1_mmq (144 B / 128 values; mmq_quantize.cu:104-198): layout 0 = D4, 1 = DS4, 2 = D2S6; block (i0/128)*ne1 + i1 */
void orc_quantize_q8_1_mmq(const float *x, const int32_t *ids, void *vy, int layout, int64_t ne00, int64_t s01, int64_t ne0, int64_t ne1);
int  orc_mmq_layout(int type); /* the layout the reference pairs with a weight type (mmq_gguf.cuh:100-135) */
/* D: GPU-MMQ semantics: integer dots against block_q8_1_mmq, stored partial sums where the layout has them; out / mag [ncols_y, N] */
void orc_matmul_q8_1_mmq(int type, const void *W, int N, int K, int64_t stride_row_x, const void *y_mmq, int64_t ncols_y, float *out, float *mag);

/* matmul oracles: W [N, K] packed row-major, X [B, K] f32, out [B, N] */
/* A: exact: dequantize W to f32, accumulate in f64 */
void orc_matmul_exact(int type, const void *W, int N, int K, const float *X, int B, float *out);
/* C: GPU-MMVQ semantics: integer dots against Q8_1 blocks (stride_col_y blocks per batch col),
 *    float combination done in f64 (order-free reference for any f32 summation order) */
void orc_matmul_q8_1(int type, const void *W, int N, int K, const void *y_q8_1,
                     int stride_col_y, int B, float *out);
/* same, also returns mag = SUM |terms| per output (what f32 accumulation error scales with) */
void orc_matmul_q8_1_ex(int type, const void *W, int N, int K, const void *y_q8_1,
                        int stride_col_y, int B, float *out, float *mag);
/* B: candle-CPU semantics: rows of X quantized to the vec_dot partner (Q8_K for K-quants,
 *    Q8_0 for Q4_0/Q5_0/Q8_0, Q8_1 for Q4_1/Q5_1), f32 accumulation in ggml generic order */
void orc_matmul_cpu(int type, const void *W, int N, int K, const float *X, int B, float *out);

/* glue ops (f32) */
void orc_rms_norm(const float *x, const float *w, float *out, int rows, int d, float eps);
/* rope: x [tokens, heads, head_dim] in place; cos/sin [max_pos, rot_dim/2]; positions [tokens] */
void orc_rope(float *x, const float *cos_t, const float *sin_t, const int32_t *positions,
              int tokens, int heads, int head_dim, int rot_dim, int neox);
float orc_glu_act(float x, int act); /* 0 silu 1 gelu(tanh) 2 relu 3 gelu_erf 4 sigmoid */
void orc_fused_glu(const float *a, const float *b, float *out, int64_t n, int act);
/* causal softmax attention, GQA. q [T, H, hd], k/v [S, KVH, hd], out [T, H, hd];
 * query t attends keys 0..(S-T+t) ; f32 inputs, f64 softmax */
void orc_attention(const float *q, const float *k, const float *v, float *out, int T, int S,
                   int H, int KVH, int hd, float scale, float softcap);

/* ---- oracle/cpu_path_oracle.c (round 3): the in-tree CPU decode path restated, and the engine's own f32 orders ---- */
float orc_fast_exp(float x);   /* attention/backends/cpu/elem.rs:417-433 */
/* single_q.rs run_barrier / compute_group_range, portable elem.rs bodies: q [H][hd], k / v [kv_len][KVH][hd], out [H][hd] */
void orc_attention_single_q_cpu(const float *q, const float *k, const float *v, float *out, int kv_len, int H, int KVH, int hd,
                                float scale, int n_kv_chunks);
void orc_rms_norm_candle(const float *x, const float *w, float *out, int rows, int d, float eps); /* x / sqrt(mean + eps) * w, f32 sums in order */
void orc_rms_norm_engine(const float *x, const float *w, float *out, int rows, int d, float eps); /* same expression, the engine's summation tree */
int64_t orc_div_by_mismatches(const float *x, const float *m, int64_t n);                                   /* dec_core2.cuh div_by vs `/` */
int64_t orc_round_trick_mismatches(float limit);                                                      /* the device round-half-away vs roundf, exhaustive */
float orc_silu_engine(float x);                                                                   /* x / (1 + fast_exp(-x)) */
void orc_fused_glu_engine(const float *a, const float *b, float *out, int64_t n);
void orc_attention_engine(const float *q, const float *k, const float *v, float *out, int kv_len, int H, int KVH, float scale, int bpw);
void orc_attention_engine_w(const float *q, const float *k, const float *v, float *out, int kv_len, int H, int KVH, float scale, int bpw, int window); /* sliding window */
/* llama_oracle.c: the reference CPU matvec with ONE f32 term per superblock, added in superblock order (a second CPU summation order) */
int orc_gemv_cpu_fast(int type, const void *W, int N, int K, const float *x, float *out);
/* cpu_path_oracle.c: the same integers and products in the decode engine's summation order (64 per-lane chains + butterfly) */
int orc_gemv_engine(int type, const void *W, int N, int K, const float *x, float *out);

void orc_set_threads(int n);
int  orc_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
