/* jukebox_hip.h -- C ABI of libjukebox_hip.so: the MI355X (gfx950) kernels behind the Jukebox
 * sampling path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless a
 * comment says "host".  No function allocates or synchronises; all work is enqueued on the
 * `stream` argument (a hipStream_t passed as void*).  Return value: 0 on success, negative
 * jb_status otherwise; jb_last_error() (host, thread-local) describes the last failure.
 *
 * The reference (openai/jukebox) has no FFI layer for this path -- its boundary is Python
 *  plus the optional apex pybind module for LayerNorm.  Each entry point
 * below cites the reference code it replaces; paths are relative to the reference tree.
 */
#ifndef JUKEBOX_HIP_H
#define JUKEBOX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { JB_F32 = 0, JB_F16 = 1,
               /* jb_pack_weight's dst_dtype only: an fp32 matrix as TWO f16 images, hi = half(w) and lo = half((w - hi) * 2^11),
                * each in f16 fragment order, hi first -- the weight operand of jb_gemm_args.w_split */
               JB_F16_SPLIT = 2 } jb_dtype;
typedef enum { JB_OK = 0, JB_ERR_ARG = -1, JB_ERR_UNSUPPORTED = -2, JB_ERR_HIP = -3 } jb_status;
typedef enum { JB_ACT_NONE = 0, JB_ACT_RELU = 1, JB_ACT_QUICK_GELU = 2 } jb_act;

/* attention patterns: jukebox/transformer/factored_attention.py:57-66 */
typedef enum {
    JB_ATTN_DENSE = 0, JB_ATTN_BLOCK = 1, JB_ATTN_TRANSPOSE_BLOCK = 2, JB_ATTN_PREV_BLOCK = 3,
    JB_ATTN_CROSS = 6, JB_ATTN_PRIME = 7
} jb_attn_func;

const char* jb_last_error(void);
int jb_version(void);

/* HIP streams in explicit priority classes (numerically lower = higher priority; the range is device dependent).
 * Used by the level pipeline: streams of different classes land on different hardware queues and really overlap. */
int jb_stream_priority_range(int* least /* host */, int* greatest /* host */);
int jb_stream_create(int priority, void** stream /* host, out */);
int jb_stream_destroy(void* stream);
/* A stream restricted to the compute units whose bit is set in cu_mask (host array of n_words x 32 bits): keeps
 * throughput work (another level's decode chain, a look-ahead prefill) off the CUs the latency-bound chain needs.
 * jb_cu_census (diagnostic) launches n_blocks 64-thread workgroups and reports out[2b] = XCC id, out[2b+1] = HW_ID
 * register of block b (device array of 2*n_blocks words) -- the mask-bit -> CU mapping is not documented. */
int jb_stream_create_cu_mask(const uint32_t* cu_mask /* host */, int n_words, void** stream /* host, out */);
int jb_cu_census(int n_blocks, uint32_t* out, void* stream);
/* Diagnostic: out[0] = shader-clock cycles, out[1] = 100-MHz ticks elapsed over a spin of spin_ticks ticks on `stream`
 * (device array of 2 int64): the clock the latency-bound decode chain really runs at. */
int jb_clock_probe(long long* out, int spin_ticks, void* stream);

/* Bytes of the MFMA-fragment-ordered weight image for a K x J matrix of `dtype`
 * (K padded to 32 (f16, f16 split) / 16 (f32), J padded to 16; the split image is two f16 images). */
int64_t jb_packed_weight_bytes(int K, int J, int dtype);

/* Re-lay a weight matrix for the MFMA kernels: src element (k, j) is read at
 * src[k * stride_k + j * stride_j] (elements of src_dtype) and converted to dst_dtype.
 * Replaces the per-call `self.w.type_as(x)` cast of Conv1D.forward (jukebox/transformer/ops.py:99);
 * strides let it take Conv1D.w (n_in, n_out), nn.Linear.weight (out, in), and one tap of an
 * nn.Conv1d (out, in, k) / nn.ConvTranspose1d (in, out, k) weight without a host-side copy. */
int jb_pack_weight(const void* src, int src_dtype, int64_t stride_k, int64_t stride_j, int K, int J,
                   void* dst, int dst_dtype, void* stream);

/* LayerNorm forward, fp32 statistics, affine.  Replaces apex `fused_layer_norm_cuda.forward_affine`
 * (apex/csrc/layer_norm_cuda.cpp:234-238, kernel cuApplyLayerNorm layer_norm_cuda_kernel.cu:279-323) as used by
 * jukebox/transformer/ops.py:14-24 (input cast to float, result cast back to the input dtype). */
int jb_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta,
                     int64_t rows, int width, float eps, void* stream);

/* Tiled MFMA GEMM over rows grouped in sequences, with optional input taps (conv1d k=3 dilated,
 * strided conv k=4, transposed conv phases), fused input ReLU, bias, activation, residual.
 *   for n < n_seq, t < t_out, j < J:
 *     acc = sum_tap sum_k pre(A[n*in_seq_stride + (t*in_stride + shift[tap])][k]) * W_tap[k][j]      (rows outside [0,t_in) read as 0)
 *     v   = act(round(acc + bias[j]));  if res: v = round(res[orow][j] + res_scale * v)
 *     out[orow][j] = v,   orow = n*out_seq_stride + t*out_stride + out_offset
 * Replaces t.addmm of Conv1D.forward for q_l > 1 (ops.py:97-101), nn.Conv1d / nn.ConvTranspose1d /
 * ResConv1DBlock (jukebox/vqvae/encdec.py:17,21,35,40,108; resnet.py:27-44) and nn.Linear x_out
 * (jukebox/prior/autoregressive.py:229,311) on channels-last rows.
 * qkv_split: column j belongs to part j / S; part 0 goes to `out`, parts 1/2 go to the k/v caches at row
 * n*cache_cap + cache_t0 + t when that is < cache_cap (FactoredAttention._append_cache, factored_attention.py:359-373). */
typedef struct jb_gemm_args {
    int dtype;                       /* of A, W (packed), out, res, kcache, vcache */
    const void* A; int64_t lda;
    const void* W; int64_t tap_stride;   /* packed image of tap i at W + i*tap_stride elements */
    const float* bias;               /* [J] or NULL */
    void* out; int64_t ldo;
    const void* res; int64_t ldr;    /* or NULL */
    int n_seq, t_in, t_out;
    int64_t in_seq_stride, out_seq_stride;
    int K, J;
    int n_taps, in_stride; int shift[4];
    int out_stride, out_offset;
    int pre_relu, act;
    float res_scale;
    int qkv_split, S;
    void* kcache; void* vcache; int cache_cap, cache_t0;
    /* fp32 problems (dtype JB_F32, K a multiple of 32, 16-byte aligned rows) on the f16 matrix cores at fp32 accuracy: W is a
     * JB_F16_SPLIT image per tap (same bytes as the fp32 image, so tap_stride counts fp32 elements as before).  The kernel
     * splits the fp32 activations the same way and evaluates w_hi*a_hi + 2^-11 * (w_hi*a_lo + w_lo*a_hi) with fp32
     * accumulation: three f16 MFMAs per k-tile at 16x the rate of the exact-fp32 instruction; the dropped term is
     * 2^-22 relative, below the rounding of the fp32 accumulation itself.  Inputs must be inside the f16 range (|x| <= 65504): a launch that
     * sees one outside raises a sticky device flag, jb_gemm_split_overflow.  Weights have no such limit: the caller packs
     * s * W for a power of two s that brings max |W| into [128, 256) -- small weights then keep 2^-22 relative precision down
     * to 2^-22 of the largest one, instead of going subnormal below 6e-5 -- and passes 1 / s as w_split_unscale (0 = 1): the
     * accumulators are multiplied by it before bias, activation and residual (exact: a power of two). */
    int w_split;
    float w_split_unscale;
    /* w_split problems: scratch of at least 4 * n_seq * t_in * K bytes (16-byte aligned), or NULL.  With it the activations are
     * split ONCE, by a pass of their own (input ReLU and range check there), into f16 hi / lo images that the GEMM loads as
     * finished operands -- the same operands, bit for bit, as the split inside the GEMM (which every column block of 64
     * repeats and which bounds that kernel); without it the split stays inside the GEMM.  The scratch may be reused as soon
     * as the launch is enqueued on the same stream. */
    void* a_split; int64_t a_split_bytes;
} jb_gemm_args;
int jb_gemm(const jb_gemm_args* args /* host */, void* stream);
/* 1 if a w_split launch since the last reset was given an activation outside the half range (|x| > 65504; under pre_relu only
 * x > 65504, what the ReLU clips cannot overflow) or a NaN (judged before the ReLU) -- its output is then not the convolution
 * --, 0 if not, < 0 on error.  The flag is a word in host-coherent memory: the call reads it without touching the device, so
 * it sees launches that have FINISHED (synchronise the stream first for the launches just enqueued); reset != 0 clears it.
 * The sampler looks after every window and at the end of a job (jukebox_amd/sample.py), the tests after every case. */
int jb_gemm_split_overflow(int reset);
/* Flat problems (one tap, unit strides) of at least `min_rows` output rows use the LDS-staged 256x128-tile kernel
 * (default 1024; < 0: never). */
void jb_tune_gemm_lds(int min_rows);
/* fp16 problems of one tap at unit strides (rows = (sequence, position), any pitch between sequences) with at least
 * `min_rows` output rows use the LDS-DMA 128x128-tile kernel (default 256; < 0: never).  Bit-identical to the other kernels. */
void jb_tune_gemm_glds(int min_rows);
/* Of those, the problems with K >= 128 and at least `min_tiles` 256x256 output tiles use the 8-phase kernel (one
 * 8-wave workgroup per compute unit, two LDS stages of four half-tiles, the two wave rows half a phase apart; default 512;
 * < 0: never).  Bit-identical to the other kernels.  Replaces the same Conv1D.forward at q > 1
 * (jukebox/transformer/ops.py:97-101). */
void jb_tune_gemm_8phase(int min_tiles);
/* w_split problems with a scratch buffer (jb_gemm_args.a_split): 1 = split the activations in a pass of their own (default),
 * 0 = inside the GEMM.  Bit-identical. */
void jb_tune_gemm_presplit(int on);
/* fp16 decode projections over 129 .. 160 k-tiles (5b_lyrics, K = 4800; <= 16 rows): 1 = 8-wave workgroups that walk their
 * k-tiles through two register stages (two workgroups per compute unit, every column tile resident at once; default), 0 = the
 * 16-wave kernels.  Another summation order: results agree to rounding, not bit for bit. */
void jb_tune_gemv_long(int on);

/* Weight-streaming skinny GEMM for the decode step (n_rows <= 64): out = act(LN?(x) @ W + b) (+ res),
 * one workgroup per 16 output columns, waves split K.  With ln_gamma != NULL the LayerNorm of
 * ResAttnBlock (ln_0 / ln_1, jukebox/transformer/transformer.py:62-66) is fused into the operand load.
 * qkv_split as above with the cache row taken from *t_dev (device int), so the launch is replayable in a hipGraph. */
typedef struct jb_gemv_args {
    int dtype;
    const void* x; int64_t ldx; int n_rows;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const void* W; const float* bias; int K, J;
    void* out; int64_t ldo;
    const void* res; int64_t ldr;
    int act;
    int qkv_split, S;
    void* kcache; void* vcache; int cache_cap; const int* t_dev;
    /* Folded LayerNorm (ln_gamma == ln_beta == NULL, ln_fold_c1 != NULL): LN(x)·W + b is evaluated as
     *     rstd[n] * (x[n]·W' - mean[n] * c1[j]) + b'[j],   W' = diag(gamma)·W,  c1 = column sums of W' as stored,
     *     b' = beta·W + b,
     * so the projection multiplies the RAW rows (no normalised copy is staged) and the fp32 row statistics are formed
     * from the same operand fragments while the weight stream is in flight.  The caller passes W = packed W',
     * bias = b', ln_fold_c1 = c1 (J floats).  Only where jb_gemv_ln_fold_supported() says so. */
    const float* ln_fold_c1;
    /* Optional second output of the plain projection (no LayerNorm): out2[n][j] = float(out[n][j]) + add2[n*add2_n_stride
     * + t*add2_t_stride + j], t = *t_dev (add2 may be NULL).  The decode step's last mlp.c_proj uses it to hand the
     * logits head `x.float() + cond` (add_cond_after_transformer, jukebox/prior/autoregressive.py:226-227) without a
     * separate launch. */
    float* out2; int64_t ldo2; const float* add2; int64_t add2_n_stride, add2_t_stride;
    /* Optional operand from the key-split decode attention (x == NULL, fp16, n_rows <= 32): row n of the input is the
     * log-sum-exp merge of the n_parts partial softmax states written by jb_attn_decode_split,
     *     x[n][k] = sum_s w_s * x_parts[(n*n_parts + s)*K + k],  w_s from x_ml[((n*n_head + k/d_head)*n_parts + s)*2 + {0,1}]
     * rounded to half once (the attention output of factored_attention.py:107-108).  attn.c_proj of the decode step. */
    const void* x_parts; const float* x_ml; int n_parts, n_head, d_head;
    /* optional v' column group of a q/k/v split: J = 2*S + (vcache ? S : 0) + wide, columns [J - wide, J) are written to
     * rows of `wide` elements of vcache_wide (same row numbering as kcache).  Single-head models project the value
     * through attn.c_proj once, when it is cached: v' = LN(x)·(Wv·Wp) -- see jb_attn_decode_wide; the decode step of a
     * wide-value layer passes vcache = NULL (q | k | v'). */
    void* vcache_wide; int wide;
} jb_gemv_args;
int jb_gemv(const jb_gemv_args* args /* host */, void* stream);
/* 1 if jb_gemv accepts ln_fold_c1 for this problem (whole k-tiles, the rows' operand fragments fit in registers). */
int jb_gemv_ln_fold_supported(int dtype, int K, int J, int n_rows);

/* Single-query cached attention for the decode step: one workgroup per (sample, head); the key
 * set is derived on the device from *t_dev and the pattern, softmax in fp32.
 * Replaces FactoredAttention.forward(sample=True) with q_l == 1: factored_qkv/prime_qkv cache slicing +
 * block/transpose/prev/prime/dense pattern functions + _attn (factored_attention.py:82-108,123-193,220-271).
 * q: [n][n_head*d_head] rows of ldq; caches [n][cache_cap][S]; out [n][S]. */
int jb_attn_decode(int dtype, int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache,
                   int cache_cap, void* out, int64_t ldo, int n_batch, int n_head, int d_head,
                   int block_ctx, const int* t_dev, int max_len, void* stream);

/* Wide-value form of jb_attn_decode for single-head fp16 layers (d_head = n_state a multiple of 32): vcache_w rows hold
 * v' = v·Wp (width elements), so   x_out[n] = res[n] + (sum_k p_k v'_k + bias)   is the residual stream after the
 * attention sub-block (ResAttnBlock: x + c_proj(attn), jukebox/transformer/transformer.py:62-66 with
 * factored_attention.py:104-108,118-121) and no attn.c_proj launch follows.  One workgroup per (sample, slice of d_head
 * output channels): width / d_head slices, each re-deriving the probabilities from the same keys.  bias = attn.c_proj's.
 * Returns JB_ERR_ARG when the shape is not supported (callers then keep jb_attn_decode + attn.c_proj). */
int jb_attn_decode_wide(int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache_w, int cache_cap,
                        const void* res, int64_t ldr, const float* bias, void* x_out, int64_t ldo, int n_batch, int d_head,
                        int width, int block_ctx, const int* t_dev, int max_len, void* stream);
int jb_attn_decode_wide_supported(int attn_func, int d_head, int width, int block_ctx, int max_len);
/* 480-channel heads (the 1b upsamplers): 1 = the lean form of the kernel -- the query row goes through LDS instead of 60
 * registers per lane, <= 168 registers: a workgroup then shares a compute unit with a waiting projection workgroup of a
 * pipelined chain, which is what lets TWO engines of a process run pipelined launches side by side (jb_engine_pipeline); 0
 * (default) = the fat form (198-216 registers per lane: one workgroup per otherwise empty compute unit, jb_engine_pipeline
 * then admits one engine; 2 % faster per step for an engine that has the GPU to itself).  Same arithmetic in the same
 * order: bit-identical.  Read when a launch is enqueued, i.e. when an engine's graphs are captured. */
void jb_tune_attn_decode_wide_lean(int on);

/* Tuning hook: workgroup size (multiple of 64, <= 1024) and key/value row pairs in flight per wave (2, 4 or 8) of
 * the generic kernel of jb_attn_decode (defaults 512 / 4; a value <= 0 keeps the current one).  kb < 0 disables the
 * fp16 MFMA fast path (QK^T on MFMA) so that the generic kernel runs for every dtype. */
void jb_tune_attn_decode(int threads, int kb);
/* Key-split decode attention (fp16; d_head = 32 x {1,2,4,8,15,16}): n_parts workgroups per (sample, head), each taking
 * every n_parts-th group of 16-key tiles and writing its own softmax state UNMERGED --
 *   parts[(n*n_parts + s)*S + h*d_head + c] = sum_k p_k v_k[c] / l_s (f16),   ml[((n*n_head + h)*n_parts + s)*2] = (m_s, l_s)
 * -- for jb_gemv's x_parts operand (attn.c_proj merges while it loads).  One CU cannot pull a whole 128-key K/V set
 * faster than the rest of the chip does everything else, so the decode step spreads each (sample, head) over CUs.
 * max_keys: upper bound of the key-set size of this layer over the whole sequence (block_ctx, blocks, seq_len, ...).
 * Same reference code as jb_attn_decode. */
int jb_attn_decode_split(int attn_func, const void* q, int64_t ldq, const void* kcache, const void* vcache, int cache_cap,
                         void* parts, float* ml, int n_batch, int n_head, int d_head, int block_ctx, const int* t_dev,
                         int max_keys, int n_parts, void* stream);
/* Recommended n_parts (1..4) for a layer; 0 = do not split: the shape is outside the split kernel's envelope, or the key
 * set is short enough (default: <= 128 keys) for one pass of jb_attn_decode's 8-wave workgroup, where a split only adds
 * the merge (measured: 2.11 vs 2.37 ms per upsampler step). */
int jb_attn_decode_split_parts(int dtype, int d_head, int max_keys);
/* Tuning hooks: most splits per (sample, head) (1..4, default 4), waves per split workgroup for short key sets (default 2);
 * smallest max_keys that is split (default 129). */
void jb_tune_attn_decode_split(int max_parts, int waves);
void jb_tune_attn_decode_split_min_keys(int min_keys);

/* Chunked-prefill attention (q_l > 1) on MFMA with LDS-staged k/v tiles and online softmax:
 * queries at positions t0 .. t0+n_q-1 against the caches (already holding those positions).
 * Replaces FactoredAttention.forward(sample=True) with q_l > 1 -- _pad_to_block_ctx, the masked
 * pattern functions and get_mask (factored_attention.py:15-28,135-193,240-249,310-323).
 * q, out: [n][n_q][S]. */
int jb_attn_prefill(int dtype, int attn_func, const void* q, const void* kcache, const void* vcache, int cache_cap,
                    void* out, int n_batch, int n_head, int d_head, int block_ctx, int t0, int n_q, void* stream);
/* 1 (default) = fp16 prefill attention with 4-wave workgroups sharing vector-staged K/V tiles (every pattern except
 * transpose, which keeps the one-wave kernel); 0 = the one-wave kernel everywhere.  Measured on MI355X (upsampler,
 * 4096 x 16 tokens): 586 -> 459 ms per primed window. */
void jb_tune_attn_prefill_v2(int enable);

/* Attention probabilities of one head for queries t0..t0+n_q-1 (softmax over each query's pattern key set), fp32 rows
 * out[n][out_row0 + i][key position < n_keys_out]; zeros where a key is not attended.  Serves the lyric alignment
 * (jukebox/align.py:44-49: z_forward(get_attn_weights={alignment_layer}) -> FactoredAttention.w, factored_attention.py:101-105). */
int jb_attn_probs(int dtype, int attn_func, const void* q, const void* kcache, int cache_cap, float* out,
                  int64_t out_n_stride, int out_row0, int n_keys_out, int n_batch, int n_head, int d_head, int head,
                  int block_ctx, int t0, int n_q, void* stream);

/* Token/start embedding + position embedding + conditioning for positions t0..t0+n_t-1
 * (t0 taken from *t_dev when t_dev != NULL).  ConditionalAutoregressive2D.get_emb,
 * jukebox/prior/autoregressive.py:177-197.  tokens [n][tok_stride] int64 holds the token of
 * position t at column t; start [n] rows of start_stride floats (y_cond, or start_token with stride 0);
 * x_cond [n][*][width] with time stride xc_t_stride (0 when broadcast).  out [n][n_t][width]. */
int jb_embed(int out_dtype, void* out, const int64_t* tokens, int64_t tok_stride, const float* x_emb,
             const float* pos_emb, const float* start, int64_t start_stride, const float* x_cond,
             int64_t xc_n_stride, int64_t xc_t_stride, int n_batch, int width, int t0, const int* t_dev, int n_t,
             void* stream);

/* xf = float(h) + x_cond[:, t] (add_cond_after_transformer, autoregressive.py:226-227,307-309).  Output row of
 * (sample n, chunk position c) is xf + n*xf_n_stride + c*width (xf_n_stride = 0 means n_t*width, i.e. packed). */
int jb_final_add(int h_dtype, const void* h, float* xf, int64_t xf_n_stride, const float* x_cond, int64_t xc_n_stride,
                 int64_t xc_t_stride, int n_batch, int width, int t0, const int* t_dev, int n_t, void* stream);

/* Temperature, top-k / nucleus filtering and categorical sampling of one token per row, written to
 * tokens[n][t] (t = *t_dev); optional copy of the raw logits to preds[n][t][bins].
 * autoregressive.py:233-235 + filter_logits (jukebox/transformer/ops.py:113-142).  Randomness is a counter-based
 * (Philox4x32-10) uniform keyed by (seed, stream_id, sample_base + n, pos_base + t): stream_id separates the levels of
 * one job, pos_base is the window's start so that the position is ABSOLUTE -- no two windows, levels or samples share a
 * draw.  top_k == 1 is argmax (lowest index on ties).  bins <= 4096. */
typedef struct jb_sample_params {
    float temp; int top_k; float top_p; int sample_base; uint64_t seed; int pos_base; int stream_id;
} jb_sample_params;
int jb_sample_logits(const float* logits, int n_batch, int bins, const jb_sample_params* params /* device */,
                     int64_t* tokens, int64_t tok_stride, const int* t_dev, float* preds, int64_t preds_n_stride,
                     void* stream);
/* The decode step's tail in one launch: jb_sample_logits, then the embedding of the NEXT position
 * x_next[n] = x_emb[token] + pos_emb[t+1] + x_cond[n][t+1] (jb_embed for position t+1; skipped when t+1 == seq_len),
 * then *t_dev = t + 1 by the workgroup that finishes last (*ticket: device counter, zero before the first call,
 * left at zero). */
int jb_sample_step(const float* logits, int n_batch, int bins, const jb_sample_params* params /* device */, int64_t* tokens,
                   int64_t tok_stride, int* t_dev, float* preds, int64_t preds_n_stride, int x_dtype, void* x_next,
                   const float* x_emb, const float* pos_emb, const float* x_cond, int64_t xc_n_stride, int64_t xc_t_stride,
                   int width, int seq_len, unsigned* ticket, void* stream);

/* Codebook gather (BottleneckBlock.dequantise/decode, jukebox/vqvae/bottleneck.py:121-123,138-147),
 * output channels-last rows [n*T][emb_width] fp32. */
int jb_vq_gather(const int64_t* codes, const float* codebook, float* out, int64_t n_codes, int emb_width, int bins,
                 void* stream);

/* Nearest-code search: argmin_j ||x||^2 - 2 x.k_j + ||k_j||^2 given xk = x @ k^T
 * (BottleneckBlock.quantise, bottleneck.py:112-119). */
int jb_vq_argmin(const float* x, const float* xk, const float* codebook, int64_t* codes, int64_t rows,
                 int emb_width, int bins, void* stream);

/* ---- decode engine: one prior's transformer bound to static buffers ------------------------------ */
typedef struct jb_layer {
    int attn_func;
    const void *w_attn, *w_proj, *w_fc, *w_proj2;          /* packed, engine dtype */
    const float *b_attn, *b_proj, *b_fc, *b_proj2;
    const float *ln0_g, *ln0_b, *ln1_g, *ln1_b;
    void *kcache, *vcache;                                 /* [n_batch][cache_cap][n_state], engine dtype */
    int cache_cap;
    /* cross-attention layers (attn_func 6, factored_attention.py:46-48,273-287): w_attn is n_in x n_state (query only);
     * c_enc_kv is given as its key and value halves, each n_in x n_state, with the (2*n_state) bias; the caches hold
     * the projected encoder states, cache_cap = encoder length. */
    const void *w_enc_k, *w_enc_v; const float* b_enc_kv;
    /* optional folded-LayerNorm images of c_attn and c_fc for the decode step (see jb_gemv_args.ln_fold_c1): packed
     * diag(gamma)·W, beta·W + b, column sums.  NULL = the decode step normalises rows in the projection kernel.
     * Prefill always uses w_attn / w_fc with an explicit LayerNorm. */
    const void *w_attn_f, *w_fc_f; const float *b_attn_f, *b_fc_f, *c1_attn, *c1_fc;
    /* optional wide-value form of a single-head self-attention layer (fp16 engines, needs the folded images): vcache_w
     * [n_batch][cache_cap][width] holds v' = v·Wp, the value already carried through attn.c_proj, so the decode step's
     * attention writes the residual stream directly (jb_attn_decode_wide) and the attn.c_proj launch disappears.
     * w_attn_fw / b_attn_fw / c1_attn_w: folded decode image [Wq | Wk | Wv·Wp] (width x (2 n_state + width)).
     * Prefill keeps its own attention over kcache / vcache and attn.c_proj; it additionally fills vcache_w with
     * v·Wp (one GEMM per sample over the v rows it has just cached, weights w_proj).  NULL = off. */
    const void* w_attn_fw; const float *b_attn_fw, *c1_attn_w; void* vcache_w;
} jb_layer;

typedef struct jb_engine_cfg {
    int dtype, n_batch, width, n_state, n_head, n_mlp, n_layers;
    int seq_len, block_ctx, bins;
    float ln_eps;
    const float *x_emb, *pos_emb;
    const float* x_out_packed;                             /* fp32 packed (K = width, J = bins) */
    const float* start; int64_t start_stride;
    const float* x_cond; int64_t xc_n_stride, xc_t_stride;
    int add_cond_after;
    const void* encoder_kv; int enc_len;                   /* [n][enc_len][width], engine dtype (cross-attention models) */
    float* hidden_out; int64_t hidden_n_stride;            /* optional: final hidden states of prefilled positions, fp32 [n][seq_len][width] */
    /* decode-step work buffers */
    void *x_a, *x_b, *q, *att, *mlp;                        /* engine dtype: [n][W],[n][W],[n][S],[n][S],[n][M] */
    float *xf, *logits;                                    /* [n][W], [n][bins] */
    /* key-split decode attention (fp16 engines, see jb_attn_decode_split): att_parts [n][4][S] engine dtype,
     * att_ml [n][n_head][4][2] fp32; NULL = one workgroup per (sample, head) (jb_attn_decode) */
    void* att_parts; float* att_ml;
    unsigned* ticket;                                      /* device counter for jb_sample_step, zero-initialised */
    /* prefill work buffers for chunks of <= chunk_cap positions */
    int chunk_cap;
    void *c_xa, *c_xb, *c_h, *c_q, *c_att, *c_mlp;          /* [n*chunk_cap][W|W|W|S|S|M] */
    float* c_xf;                                           /* [n*chunk_cap][W] (only when preds != NULL) */
    int64_t* tokens; int64_t tok_stride;
    int* t_dev;
    float* preds; int64_t preds_n_stride;                  /* optional [n][seq_len][bins] */
    const jb_sample_params* sample_params;                 /* device */
    /* optional recording of one layer/head's attention probabilities during prefill (alignment): */
    int rec_layer, rec_head, rec_keys;                     /* rec_layer < 0: off */
    float* rec_out; int64_t rec_n_stride;                  /* [n][seq_len][rec_keys] */
    /* row pitch of `att` in elements (0 = n_state).  With n_state not a whole number of k-tiles (5b_lyrics: 1200 = 37.5 x 32)
     * a pitch rounded up to the k-tile, the padding zeroed once by the caller, lets the decode step's attn.c_proj run its
     * branch-free path (the packed weight image is zero-padded to whole k-tiles anyway). */
    int att_ld;
    /* optional: (18 * launches_per_step + 1) * 32 zero-initialised words for software-pipelined launches
     * (jb_engine_pipeline): per launch slot a completion count (32 words apart), then per slot nine ticket counters and
     * eight shard flags (32 words apart each); the last 32-word group starts with an error word (slot + 1 of a launch whose wait for its producer timed out; 0 = none; never reset by
     * the library). */
    unsigned* pipe_words;
    /* rows the decode-step buffers x_a, x_b and mlp hold (0 = n_batch).  With 16 rows, the pipelined launches of a single-head
     * engine hand their activation blocks over in the order the consumer's MFMA operands want them -- [k-tile][lane][8 channels],
     * always 16 rows -- instead of [row][channel] (jb_tune_pipeline). */
    int act_rows;
} jb_engine_cfg;

/* Transformer.forward(sample=True) + the token loop of ConditionalAutoregressive2D.sample/primed_sample
 * (jukebox/transformer/transformer.py:169-192, jukebox/prior/autoregressive.py:222-236,289-347). */
int jb_engine_create(const jb_engine_cfg* cfg /* host */, const jb_layer* layers /* host array */, void** handle);
int jb_engine_destroy(void* handle);
/* Cross-attention models: project cfg.encoder_kv through every cross layer's c_enc_kv into that layer's k/v cache
 * (decode_qkv at sample_t == 0, factored_attention.py:273-280).  Call once per window before prefill / decode. */
int jb_engine_set_encoder_kv(void* handle, void* stream);
/* Prefill positions t0..t0+n_t-1 (tokens already in cfg.tokens): fills the k/v caches, leaves *t_dev = t0+n_t. */
int jb_engine_prefill(void* handle, int t0, int n_t, void* stream);
/* Run n_steps decode steps starting at position t0 (sets *t_dev = t0 and embeds position t0 first).  One step =
 * L x [c_attn | attention | attn.c_proj | mlp.c_fc | mlp.c_proj] | logits | sample + embed(t+1) + counter: 5 L + 2 launches
 * (wide-value layers have no attn.c_proj launch: 4 per layer; jb_engine_launches_per_step reports the count).
 * use_graph = 1 captures one step into a hipGraph on first use and replays it (as pipelined launches while those are
 * switched on: jb_engine_pipeline); use_graph = 0 is always the eager plain chain on `stream`; use_graph = 2 (diagnostics,
 * pipelined launches switched on) enqueues the pipelined launches without the graph executor; use_graph = 3 replays the
 * plain chain's graph even while pipelined launches are switched on (the sampler's in-situ comparison of the two forms).
 * A pipelined decode is host-synchronous and must not be called on a stream that is being captured. */
int jb_engine_decode(void* handle, int t0, int n_steps, int use_graph, void* stream);
/* Software-pipelined launches of the decode step (graph replay only): the launches of a step go to streams of the engine's
 * own (each on a hardware queue of its own) -- TWO, alternating by launch, for multi-head engines; THREE by kind of launch for
 * single-head engines on wide-value layers (0: c_attn, c_fc, the logits head; 1: mlp.c_proj, the sampler; 2: the attention,
 * dispatched four phases ahead of its flags on compute units reserved by mask: jb_tune_pipeline) --, so a
 * launch is dispatched -- and requests its weight stream, whose addresses never depend on activations, the attention its K / v'
 * rows of earlier positions -- while its producer still runs; what it reads from its producer it reads after polling the
 * producer's completion words, through write-through stores / L1-bypassing loads.  Same kernels' arithmetic in the same
 * order: tokens and logits are bit-identical to the plain chain.  A pipelined jb_engine_decode is HOST-SYNCHRONOUS: it
 * drains the caller's stream, runs the steps on its streams and returns when they are done (no queue of the process holds a
 * waiting packet meanwhile).  enable != 0 returns JB_ERR_UNSUPPORTED unless every launch of this engine's step has a
 * pipelined form (cfg.pipe_words given, fp16, <= 16 samples; single head: every layer a wide-value layer of one 480- or
 * 256-channel head, width and n_mlp of 32..64 k-tiles -- the 1b upsamplers, small_prior; multi-head: folded LayerNorm, heads of 150 / 256 / 512
 * channels), and while ANOTHER engine of the process has them on: a waiting launch occupies compute units, and the waiters
 * of two engines can keep each other's producers from being placed (enable = 0 or jb_engine_destroy releases the right).
 * The streams must feed different hardware queues; the first pipelined decode makes them, checks that pairwise with a
 * two-kernel handshake and keeps the plain chain otherwise (jb_engine_pipelined then reports 0).  enable = 0 RELEASES them --
 * streams, hardware queues, graphs --, so that nothing outlives the phase that uses it (more hardware queues in the process,
 * even idle ones, slow every plain launch chain next to them; the reference's loop leaves nothing behind either:
 * jukebox/sample.py:90-121); the next enable makes new ones (milliseconds).  enable = 2 is enable = 1 with fresh streams at
 * the next decode.  The engine must be idle in every case.  Replaces the same reference code as jb_engine_decode. */
int jb_engine_pipeline(void* handle, int enable);
/* The hand-off form of the decode step, read when an engine is created: operand_order = 1 (default) -- the activation blocks
 * between the launches (x_a, x_b, mlp) in MFMA operand order ([k-tile][lane][8 channels] of 16 rows: a consumer wave's fetch of
 * a k-tile is one contiguous KiB, a producer's 16 x 16 tile two runs of 256 bytes, 16-byte stores) where the engine can (the
 * shapes jb_engine_pipeline accepts for a single head, cfg.act_rows >= 16): such an engine runs the SAME kernels as a plain
 * chain (the kernel boundary is the hand-shake) and as software-pipelined launches (completion words).  0 -- [row][channel]
 * and, for the plain chain, the plain kernels (16 half lines per wave request).  Measured on the upsampler step at 16 samples,
 * pipelined launches: 1.407 against 1.522 ms.  Same arithmetic in every form: tokens and logits are bit-identical.  The reference
 * has no counterpart (its hand-off between two layers is a tensor in HBM: jukebox/transformer/transformer.py:62-66,82-86).
 * Bit 1 of the argument (value | 2), read when an engine's streams for pipelined launches are made: keep single-head engines
 * on the TWO-stream form of the pipelined step (launches alternate by slot).  Default: three streams by kind of launch -- the
 * attention launches on a stream of their own, dispatched four phases ahead of their flags, on compute units reserved by mask
 * (DESIGN.md section 5): 1.370 against 1.405 ms per upsampler step at position 4096, 1.388 against 1.487 at 7900, bit-identical. */
void jb_tune_pipeline(int operand_order);
/* 1 while the engine's decode steps run as pipelined launches, else 0 (also after a fallback to the plain chain). */
int jb_engine_pipelined(void* handle);
/* 1 while the engine holds a pair of streams for pipelined launches (from its first pipelined decode until
 * jb_engine_pipeline(handle, 0) / jb_engine_destroy), else 0. */
int jb_engine_pipeline_resident(void* handle);
/* Measurement aid: n_steps passes over all layers launching only the LayerNorm-fused projections (attn.c_attn and
 * mlp.c_fc -- the dominant kernel of the decode step) with their real arguments, back to back on `stream`, bracketed
 * by one HIP event pair.  Synchronises.  out[0] = average microseconds per launch, out[1] = launches timed,
 * out[2] = average algorithmic bytes per launch. */
int jb_engine_probe_projection(void* handle, int t0, int n_steps, void* stream, double* out /* host, 3 doubles */);
/* Number of kernel launches in one decode step (for launch-overhead accounting). */
int jb_engine_launches_per_step(void* handle);
/* Algorithmic HBM bytes of one decode step at position t (weights once per step + k/v rows read
 * and written + logits head + activation rows). */
double jb_engine_step_bytes(void* handle, int t);

#ifdef __cplusplus
}
#endif
#endif
