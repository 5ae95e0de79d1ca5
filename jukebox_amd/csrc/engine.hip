// Decode engine: one prior's transformer bound to static device buffers.
//
// Drives Transformer.forward(sample=True) (jukebox/transformer/transformer.py:169-192) and the token loops of
// ConditionalAutoregressive2D.sample / primed_sample (jukebox/prior/autoregressive.py:222-236,289-347):
//   decode step  = L x [LN0+c_attn(+k/v append) | attention (key-split) | c_proj(+merge)+res | LN1+c_fc+gelu | c_proj+res]
//                  -> logits -> sample + embed(t+1) + (t += 1)            (5 L + 2 launches, one hipGraph);
//                  the last c_proj also writes `x.float() + cond` for the logits head, the sampler embeds the next position;
//                  wide-value layers (single head, jb_layer.vcache_w) have no attn.c_proj launch: 4 launches per layer
//   prefill      = the same per layer on a chunk of positions with the tiled GEMM and the MFMA attention.
// The position t lives in device memory (*t_dev) so that the captured graph is replayable for every step.
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"

// the engine of the process that runs pipelined launches (jb_engine_pipeline): one at a time
static std::mutex g_pipe_mutex;
static void* g_pipe_owner = nullptr;
static bool pipe_own(void* e) {                       // caller holds g_pipe_mutex
    if (g_pipe_owner && g_pipe_owner != e) return false;
    g_pipe_owner = e;
    return true;
}
static void pipe_disown(const void* e) {              // caller holds g_pipe_mutex
    if (g_pipe_owner == e) g_pipe_owner = nullptr;
}

struct JbEngine {
    jb_engine_cfg cfg;
    std::vector<jb_layer> layers;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t capture_stream = nullptr;   // the legacy default stream cannot be captured: record on a private one
    // wide-value layers append k and v' (not v) in the decode step: until the next window's prefill starts again at
    // position 0, the v rows of decoded positions are stale and a prefill that would attend to them is refused
    bool v_rows_stale = false;
    // software-pipelined launches (jb_engine_pipeline): the launches of a step alternate between two streams of the engine's
    // own, as two single-stream graphs that are replayed side by side
    bool pipelined = false;
    bool operand_order = false;             // activation blocks of the decode step in operand order (decided at creation: the
                                            // captured graphs and the embedding of a call's first position must agree)
    hipStream_t pstream[3] = {nullptr, nullptr, nullptr};          // the engine's own pair / triple (setup_pipeline_streams)
    hipGraph_t pgraph[3] = {nullptr, nullptr, nullptr};
    hipGraphExec_t pexec[3] = {nullptr, nullptr, nullptr};
    int n_pstreams = 2;                     // 3: the attention launches on a stream of their own (pipe_streams)
};

// Hand-off form of the decode step, read when an engine is CREATED (jb_tune_pipeline): activation blocks
// between the launches in MFMA operand order (common.h: JB_FRAG_*; single-head engines whose decode buffers hold 16 rows,
// jb_engine_cfg.act_rows) or as [row][channel].
static int g_pipe_frag = 1;
// bit 1 (value 2): the two-stream form of the pipelined step for engines that would run the three-stream form (read when the
// streams are made: A/B measurements, tests)
static int g_pipe_two_streams = 0;
extern "C" void jb_tune_pipeline(int operand_order) { g_pipe_frag = (operand_order & 1) ? 1 : 0; g_pipe_two_streams = (operand_order >> 1) & 1; }

__global__ void set_int_kernel(int* p, int v) { *p = v; }
// [row][channel] -> operand order (common.h: jb_frag_el), n rows of `width` halves; rows n..15 of dst are left alone
__global__ void to_operand_order_kernel(const f16* __restrict__ src, f16* __restrict__ dst, int n, int width) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 8;        // 8 consecutive channels of one row: one 16-byte piece
    if (i >= n * width) return;
    const int row = i / width, col = i - row * width;
    *reinterpret_cast<f16x8*>(dst + jb_frag_el(row, col)) = *reinterpret_cast<const f16x8*>(src + (int64_t)row * width + col);
}
__global__ void inc_int_kernel(int* p) { *p += 1; }

static bool decide_operand_order(const JbEngine* e);

extern "C" int jb_engine_create(const jb_engine_cfg* cfg, const jb_layer* layers, void** handle) {
    JB_REQUIRE(cfg && layers && handle, "null pointer");
    JB_REQUIRE(cfg->dtype == JB_F32 || cfg->dtype == JB_F16, "bad dtype");
    JB_REQUIRE(cfg->n_batch >= 1 && cfg->n_batch <= 64, "n_batch must be 1..64");
    JB_REQUIRE(cfg->width > 0 && cfg->n_state > 0 && cfg->n_head > 0 && cfg->n_mlp > 0 && cfg->n_layers > 0, "bad dims");
    JB_REQUIRE(cfg->n_state % cfg->n_head == 0, "n_state must divide by n_head");
    JB_REQUIRE(cfg->seq_len > 0 && cfg->bins > 0, "bad seq_len / bins");
    JB_REQUIRE(cfg->x_emb && cfg->pos_emb && cfg->start, "missing embedding tables");
    JB_REQUIRE(cfg->x_a && cfg->x_b && cfg->q && cfg->att && cfg->mlp && cfg->xf && cfg->logits, "missing decode buffers");
    JB_REQUIRE(cfg->tokens && cfg->t_dev && cfg->sample_params, "missing token / counter / sampler buffers");
    for (int l = 0; l < cfg->n_layers; ++l) {
        const jb_layer& L = layers[l];
        JB_REQUIRE(L.attn_func == 0 || L.attn_func == 1 || L.attn_func == 2 || L.attn_func == 3 || L.attn_func == 6 ||
                       L.attn_func == 7, "unsupported attn_func");
        JB_REQUIRE(L.attn_func == 0 || L.attn_func == 6 || L.attn_func == 7 || cfg->block_ctx > 0, "block_ctx required");
        JB_REQUIRE(L.attn_func != 6 || (L.w_enc_k && L.w_enc_v && L.b_enc_kv && cfg->encoder_kv && cfg->enc_len > 0 &&
                                        L.cache_cap == cfg->enc_len), "cross-attention layer needs c_enc_kv and encoder_kv");
        JB_REQUIRE(L.w_attn && L.w_proj && L.w_fc && L.w_proj2 && L.b_attn && L.b_proj && L.b_fc && L.b_proj2 &&
                       L.ln0_g && L.ln0_b && L.ln1_g && L.ln1_b && L.kcache && L.vcache && L.cache_cap > 0,
                   "incomplete layer descriptor");
        const int j_attn = L.attn_func == 6 ? cfg->n_state : 3 * cfg->n_state;
        JB_REQUIRE((!L.w_attn_f && !L.b_attn_f && !L.c1_attn) ||
                       (L.w_attn_f && L.b_attn_f && L.c1_attn &&
                        jb_gemv_ln_fold_supported(cfg->dtype, cfg->width, j_attn, cfg->n_batch)),
                   "folded LayerNorm image of c_attn is incomplete or unsupported for this shape");
        JB_REQUIRE((!L.w_fc_f && !L.b_fc_f && !L.c1_fc) ||
                       (L.w_fc_f && L.b_fc_f && L.c1_fc &&
                        jb_gemv_ln_fold_supported(cfg->dtype, cfg->width, cfg->n_mlp, cfg->n_batch)),
                   "folded LayerNorm image of mlp.c_fc is incomplete or unsupported for this shape");
        const bool any_wide = L.w_attn_fw || L.b_attn_fw || L.c1_attn_w || L.vcache_w;
        JB_REQUIRE(!any_wide ||
                       (L.w_attn_fw && L.b_attn_fw && L.c1_attn_w && L.vcache_w &&
                        cfg->dtype == JB_F16 && cfg->n_head == 1 &&
                        jb_attn_decode_wide_supported(L.attn_func, cfg->n_state, cfg->width, cfg->block_ctx, cfg->seq_len) &&
                        jb_gemv_ln_fold_supported(cfg->dtype, cfg->width, 2 * cfg->n_state + cfg->width, cfg->n_batch)),
                   "wide-value layer: all four fields, fp16, one head, shapes accepted by jb_attn_decode_wide / the folded c_attn");
    }
    JB_REQUIRE((cfg->att_parts == nullptr) == (cfg->att_ml == nullptr), "att_parts and att_ml come together");
    JB_REQUIRE(cfg->bins <= 0 || !cfg->x_out_packed || cfg->ticket, "ticket counter missing");
    JB_REQUIRE(cfg->width % 4 == 0, "width must be a multiple of 4");
    JB_REQUIRE(cfg->att_ld == 0 || cfg->att_ld >= cfg->n_state, "att_ld must be 0 or >= n_state");
    JB_REQUIRE(cfg->act_rows == 0 || cfg->act_rows >= cfg->n_batch, "act_rows must be 0 or >= n_batch");
    JB_REQUIRE(!cfg->rec_out || (cfg->rec_layer >= 0 && cfg->rec_layer < cfg->n_layers && cfg->rec_keys > 0 &&
                                 cfg->rec_head >= 0 && cfg->rec_head < cfg->n_head), "bad attention recording request");
    JbEngine* e = new JbEngine();
    e->cfg = *cfg;
    e->layers.assign(layers, layers + cfg->n_layers);
    e->operand_order = decide_operand_order(e);
    *handle = e;
    return JB_OK;
}

// The pair of streams of the pipelined launches and its two parity graphs go: a pair that merely EXISTS -- two more hardware
// queues in the process -- slows every plain launch chain that runs next to it (profiles/r04_pipe_in_job.log; BENCH_r04: the
// upper levels of every job after a process's first ran 3.8x / 1.8x slower), so nothing of it may outlive the phase that
// uses it.  The pair is idle whenever no jb_engine_decode is in progress (a pipelined decode is host-synchronous).
static void release_pipeline(JbEngine* e) {
    for (int k = 0; k < 3; ++k) {
        if (e->pstream[k]) (void)hipStreamSynchronize(e->pstream[k]);
        if (e->pexec[k]) (void)hipGraphExecDestroy(e->pexec[k]);
        if (e->pgraph[k]) (void)hipGraphDestroy(e->pgraph[k]);
        e->pexec[k] = nullptr; e->pgraph[k] = nullptr;
    }
    for (int k = 0; k < 3; ++k) {
        if (e->pstream[k]) (void)hipStreamDestroy(e->pstream[k]);
        e->pstream[k] = nullptr;
    }
}

extern "C" int jb_engine_destroy(void* handle) {
    if (!handle) return JB_OK;
    JbEngine* e = (JbEngine*)handle;
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    release_pipeline(e);
    {
        std::lock_guard<std::mutex> lock(g_pipe_mutex);
        pipe_disown(e);
    }
    if (e->capture_stream) (void)hipStreamDestroy(e->capture_stream);
    delete e;
    return JB_OK;
}

static bool layer_wide(const jb_engine_cfg& c, const jb_layer& L);

extern "C" int jb_engine_launches_per_step(void* handle) {
    if (!handle) return 0;
    const JbEngine* e = (const JbEngine*)handle;
    int n = 5 * e->cfg.n_layers + 2;
    for (const jb_layer& L : e->layers) n -= layer_wide(e->cfg, L) ? 1 : 0;
    return n;
}

// Completion slots of a pipelined step: its launches alternate between two streams whose graphs are replayed step after step, so
// a step must hold an EVEN number of them -- multi-head engines (5 L + 2 launches) end in one pad launch that only waits and publishes.
static int pipe_slots(const JbEngine* e) { const int n = jb_engine_launches_per_step((void*)e); return n + (n & 1); }
__global__ void pipe_pad_kernel(JbPipe pipe) {
    const unsigned own = jb_pipe_own(pipe);
    jb_pipe_wait(pipe, own);
    jb_pipe_publish(pipe, own);
}

#define JB_TRY(call)                \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != JB_OK) return rc__; \
    } while (0)

// LayerNorm + projection of the decode step (which = 0: ln_0 + attn.c_attn, 1: ln_1 + mlp.c_fc): the folded image when the
// layer carries one (jb_gemv_args.ln_fold_c1), else gamma / beta for the in-kernel normalisation.
static void fill_ln_proj(jb_gemv_args& g, const jb_engine_cfg& c, const jb_layer& L, int which) {
    g.dtype = c.dtype; g.n_rows = c.n_batch; g.ldx = c.width; g.K = c.width; g.ln_eps = c.ln_eps;
    const void* wf = which == 0 ? L.w_attn_f : L.w_fc_f;
    if (wf) {
        g.W = wf; g.bias = which == 0 ? L.b_attn_f : L.b_fc_f; g.ln_fold_c1 = which == 0 ? L.c1_attn : L.c1_fc;
    } else {
        g.W = which == 0 ? L.w_attn : L.w_fc; g.bias = which == 0 ? L.b_attn : L.b_fc;
        g.ln_gamma = which == 0 ? L.ln0_g : L.ln1_g; g.ln_beta = which == 0 ? L.ln0_b : L.ln1_b;
    }
}

// Largest key set a layer's query can see over the whole sequence (decides the key split of its decode attention).
static int layer_max_keys(const jb_engine_cfg& c, const jb_layer& L) {
    switch (L.attn_func) {
        case JB_ATTN_BLOCK: case JB_ATTN_PREV_BLOCK: return c.block_ctx;
        case JB_ATTN_TRANSPOSE_BLOCK: return (c.seq_len + c.block_ctx - 1) / c.block_ctx;
        case JB_ATTN_PRIME: case JB_ATTN_CROSS: return L.cache_cap;
        default: return c.seq_len;
    }
}
// Splits per (sample, head) of a layer's decode attention; 0 = one workgroup per (sample, head) (jb_attn_decode).
static int layer_split_parts(const jb_engine_cfg& c, const jb_layer& L) {
    if (!c.att_parts || c.n_batch > 32) return 0;
    return jb_attn_decode_split_parts(c.dtype, c.n_state / c.n_head, layer_max_keys(c, L));
}

// Wide-value layer (jb_layer.vcache_w): the decode attention writes the residual stream, no attn.c_proj launch.  Layers
// whose key sets are long enough for the key split keep the split (+ merging c_proj) form.
static bool layer_wide(const jb_engine_cfg& c, const jb_layer& L) {
    return L.vcache_w && L.w_attn_fw && layer_split_parts(c, L) == 0;
}

// ln_0 + attn.c_attn of the decode step with the k / v (or k / v') rows appended at *t_dev.
static void fill_c_attn(jb_gemv_args& g, const jb_engine_cfg& c, const jb_layer& L) {
    const int S = c.n_state;
    g = {};
    fill_ln_proj(g, c, L, 0);
    g.x = c.x_a; g.out = c.q; g.ldo = S; g.act = JB_ACT_NONE;
    if (L.attn_func == JB_ATTN_CROSS) { g.J = S; return; }        // query only; k/v come from the encoder (set_encoder_kv)
    g.qkv_split = 1; g.S = S; g.kcache = L.kcache; g.cache_cap = L.cache_cap; g.t_dev = c.t_dev;
    if (layer_wide(c, L)) {
        g.W = L.w_attn_fw; g.bias = L.b_attn_fw; g.ln_fold_c1 = L.c1_attn_w; g.ln_gamma = g.ln_beta = nullptr;
        g.J = 2 * S + c.width; g.vcache = nullptr; g.vcache_wide = L.vcache_w; g.wide = c.width;
    } else {
        g.J = 3 * S; g.vcache = L.vcache;
    }
}

static bool pipe_operand_order(const JbEngine* e);
// The embedding of position t0 into x_a, in the layout the step's first projection fetches: rows, or -- engines on operand-order
// blocks -- re-laid through x_b, which is free until the first attention launch writes it.
static int enqueue_embed(JbEngine* e, int t0, hipStream_t s) {
    const jb_engine_cfg& c = e->cfg;
    const bool frag = pipe_operand_order(e);
    JB_TRY(jb_embed(c.dtype, frag ? c.x_b : c.x_a, c.tokens, c.tok_stride, c.x_emb, c.pos_emb, c.start, c.start_stride, c.x_cond,
                    c.xc_n_stride, c.xc_t_stride, c.n_batch, c.width, t0, nullptr, 1, s));
    if (frag) {
        const int pieces = c.n_batch * c.width / 8;
        to_operand_order_kernel<<<(pieces + 255) / 256, 256, 0, s>>>((const f16*)c.x_b, (f16*)c.x_a, c.n_batch, c.width);
        JB_CHECK_LAUNCH();
    }
    return JB_OK;
}

static bool pipeline_eligible_multi_head(const JbEngine* e);
static bool pipeline_eligible(const JbEngine* e);
// Whether this engine's decode step hands its activation blocks over in operand order (common.h): the single-head wide-value
// form whose every launch has a pipelined kernel form (pipeline_eligible: fp16, <= 16 samples, one 480-channel head, whole
// 32-channel k-tiles in width and MLP), decode buffers of 16 rows, and the switch (jb_tune_pipeline).  Such an engine runs the
// SAME kernels in both launch forms: software-pipelined launches synchronise through their completion words, the plain chain
// launches them with JB_PIPE_NO_SYNC -- the kernel boundary is the hand-shake -- and keeps the operand-order blocks and the
// 16-byte stores (round 6: the plain chain's fetch of the 61-KB block had the same 16-half-lines-per-request pattern).
static bool pipe_operand_order(const JbEngine* e) { return e->operand_order; }
static bool decide_operand_order(const JbEngine* e) {
    return g_pipe_frag && e->cfg.act_rows >= 16 && !pipeline_eligible_multi_head(e) && pipeline_eligible(e);
}

// One decode step at position *t_dev (x_a already holds that position's embedding); everything position-dependent is
// read on the device.  Leaves the next position's embedding in x_a and *t_dev advanced.
// parity < 0: every launch on `s` (the plain chain).  parity 0 / 1: software-pipelined step -- only the even / odd launches
// are enqueued (on `s`), each with its completion slot; the other half goes to the other stream by a second call.
// Three-stream form (e->n_pstreams == 3: single-head engines on wide-value layers, pipe_streams): the launches go to streams BY
// KIND -- 0: c_attn, c_fc, the logits head; 1: mlp.c_proj, the sampler; 2: the attention -- and `parity` names the stream.  An
// attention launch is then dispatched when the PREVIOUS layer's attention has finished, four phases ahead of its flags instead
// of one: its K / v' rows of earlier positions (up to 245 KB per workgroup through one compute unit's memory pipe: the decode
// step grew from 1.33 ms at position 64 to 1.49 ms at 7900 with the key count of the strided layers,
// profiles/r06c20_step_vs_position.log) have landed when the query arrives.  What the kernels assumed of a two-stream step is
// restated there: residual rows (two launches old) are read behind the wait, write-through; the attention stream's first launch
// of a step waits for the previous step's last slot before it reads the position (JB_PIPE_PRE_WAIT).
enum { K_C_ATTN = 0, K_ATTENTION, K_ATTN_PROJ, K_C_FC, K_MLP_PROJ, K_LOGITS, K_SAMPLE, K_PAD };
static int enqueue_step(JbEngine* e, hipStream_t s, int parity = -1) {
    const jb_engine_cfg& c = e->cfg;
    const int N = c.n_batch, W = c.width, S = c.n_state, M = c.n_mlp, H = c.n_head, d = S / H;
    const int n_slots = pipe_slots(e);
    int slot = 0;
    // completion protocol (common.h): 1 -- a flag word per ticket shard, polled by eight lanes -- for engines of >= 8 samples
    // (every launch of their step has >= 8 workgroups, so every shard has a member); 0 -- the two-level ticket with one flag --
    // for smaller batches (measured on the 16-sample upsampler step: 1.541 vs 1.601 ms, profiles/r04_bench_engine_up_proto{1,0}.log)
    JbPipe pp{c.pipe_words, c.pipe_words ? c.pipe_words + (size_t)n_slots * JB_PIPE_PAD : nullptr,
              c.pipe_words ? c.pipe_words + jb_pipe_words(n_slots) - JB_PIPE_PAD : nullptr, 0, 0,
              (getenv("JB_PIPE_TIMEOUT_MS") ? atoll(getenv("JB_PIPE_TIMEOUT_MS")) : 2000ll) * 100000ll,
              (c.pipe_words && getenv("JB_PIPE_DEBUG")) ? reinterpret_cast<long long*>(c.pipe_words + jb_pipe_words(n_slots)) : nullptr,
              c.n_batch >= 8 ? 1 : 0, 0, n_slots - 1};
    const bool frag = pipe_operand_order(e);
    const bool by_kind = parity >= 0 && e->n_pstreams == 3;
    static const int stream_of_kind[] = {0, 2, 1, 0, 1, 0, 1, 0};
    bool attention_seen = false;
    if (parity < 0) pp.dbg = nullptr;          // (stamps are per completion slot)
    // the pipeline slot of the next launch -- for the plain chain NULL (the plain kernels), or, for an engine on operand-order
    // blocks, the pipelined kernel form without its hand-shake (JB_PIPE_NO_SYNC); `mine`: this call enqueues it; `layout`: which
    // of the launch's activation blocks (JB_FRAG_X operand, JB_FRAG_OUT output, JB_FRAG_RES residual) are in operand order
    bool mine = true;
    auto next = [&](int kind, int layout = 0) -> const JbPipe* {
        mine = parity < 0 || (by_kind ? stream_of_kind[kind] : (slot & 1)) == parity;
        pp.slot = parity < 0 ? JB_PIPE_NO_SYNC : slot;
        pp.prev = slot == 0 ? n_slots - 1 : slot - 1;
        pp.frag = frag ? layout : 0;
        if (by_kind && kind == K_ATTENTION && !attention_seen) { pp.frag |= JB_PIPE_PRE_WAIT; attention_seen = true; }
        ++slot;
        return (parity < 0 && !frag) ? nullptr : &pp;
    };
    for (int l = 0; l < c.n_layers; ++l) {
        const jb_layer& L = e->layers[l];
        jb_gemv_args g;
        // a7/a8/a9: ln_0 + c_attn, k/v appended at *t_dev
        fill_c_attn(g, c, L);
        { const JbPipe* pipe = next(K_C_ATTN, JB_FRAG_X); if (mine) JB_TRY(jb_gemv_impl(&g, pipe, s)); }      // (q and the cache rows stay [row][channel])
        // attention, then attn.c_proj + residual: x_b = x_a + a
        const int parts = layer_split_parts(c, L);
        g = {};
        const int att_ld = c.att_ld ? c.att_ld : S;         // padded pitch (zeros): K covers the padding, the image has zero rows there
        g.dtype = c.dtype; g.ldx = S; g.n_rows = N; g.W = L.w_proj; g.bias = L.b_proj; g.K = S; g.J = W;
        g.out = c.x_b; g.ldo = W; g.res = c.x_a; g.ldr = W;
        if (layer_wide(c, L)) {
            const JbPipe* pipe = next(K_ATTENTION, JB_FRAG_RES | JB_FRAG_OUT);
            if (mine) JB_TRY(jb_attn_decode_wide_impl(L.attn_func, c.q, S, L.kcache, L.vcache_w, L.cache_cap, c.x_a, W, L.b_proj, c.x_b,
                                                      W, N, S, W, c.block_ctx, c.t_dev, c.seq_len, pipe, s));
        } else if (parts > 0) {
            if (parity >= 0) {
                jb_set_error("jb_engine: the key-split attention has no pipelined form");
                return JB_ERR_UNSUPPORTED;
            }
            (void)next(K_ATTENTION);
            JB_TRY(jb_attn_decode_split(L.attn_func, c.q, S, L.kcache, L.vcache, L.cache_cap, c.att_parts, c.att_ml, N, H, d,
                                        c.block_ctx, c.t_dev, layer_max_keys(c, L), parts, s));
            g.x_parts = c.att_parts; g.x_ml = c.att_ml; g.n_parts = parts; g.n_head = H; g.d_head = d;
        } else {
            // multi-head layers: attention into c.att (pitch att_ld, pad columns stay zero), then attn.c_proj + residual
            const JbPipe* pipe = next(K_ATTENTION);
            if (mine) JB_TRY(jb_attn_decode_impl(c.dtype, L.attn_func, c.q, S, L.kcache, L.vcache, L.cache_cap, c.att, att_ld, N, H, d,
                                                 c.block_ctx, c.t_dev, c.seq_len, pipe, s));
            g.x = c.att; g.ldx = att_ld;
            const int KT = c.dtype == JB_F16 ? 32 : 16;
            if (att_ld % KT == 0 && att_ld - S < KT) g.K = att_ld;
        }
        if (!layer_wide(c, L)) { const JbPipe* pipe = next(K_ATTN_PROJ); if (mine) JB_TRY(jb_gemv_impl(&g, pipe, s)); }
        // ln_1 + mlp.c_fc + quick_gelu
        g = {};
        fill_ln_proj(g, c, L, 1);
        g.x = c.x_b; g.J = M; g.out = c.mlp; g.ldo = M; g.act = JB_ACT_QUICK_GELU;
        { const JbPipe* pipe = next(K_C_FC, JB_FRAG_X | JB_FRAG_OUT); if (mine) JB_TRY(jb_gemv_impl(&g, pipe, s)); }
        // mlp.c_proj + residual: x_a = x_b + m   (h = x + a + m, transformer.py:82-83); the last layer also hands the
        // logits head xf = float(x_a) (+ cond[t], autoregressive.py:226-227)
        g = {};
        g.dtype = c.dtype; g.x = c.mlp; g.ldx = M; g.n_rows = N; g.W = L.w_proj2; g.bias = L.b_proj2; g.K = M; g.J = W;
        g.out = c.x_a; g.ldo = W; g.res = c.x_b; g.ldr = W;
        if (l + 1 == c.n_layers) {
            g.out2 = c.xf; g.ldo2 = W; g.t_dev = c.t_dev;
            if (c.add_cond_after && c.x_cond) { g.add2 = c.x_cond; g.add2_n_stride = c.xc_n_stride; g.add2_t_stride = c.xc_t_stride; }
        }
        { const JbPipe* pipe = next(K_MLP_PROJ, JB_FRAG_X | JB_FRAG_RES | JB_FRAG_OUT); if (mine) JB_TRY(jb_gemv_impl(&g, pipe, s)); }   // (xf stays fp32 [row][channel])
    }
    jb_gemv_args g = {};
    g.dtype = JB_F32; g.x = c.xf; g.ldx = W; g.n_rows = N; g.W = c.x_out_packed; g.K = W; g.J = c.bins;
    g.out = c.logits; g.ldo = c.bins;
    { const JbPipe* pipe = next(K_LOGITS); if (mine) JB_TRY(jb_gemv_impl(&g, pipe, s)); }
    {
        const JbPipe* pipe = next(K_SAMPLE, JB_FRAG_OUT);
        if (mine) JB_TRY(jb_sample_step_impl(c.logits, N, c.bins, c.sample_params, c.tokens, c.tok_stride, c.t_dev, c.preds,
                                             c.preds_n_stride, c.dtype, c.x_a, c.x_emb, c.pos_emb, c.x_cond, c.xc_n_stride,
                                             c.xc_t_stride, W, c.seq_len, c.ticket, pipe, s));
    }
    if (slot < n_slots) {                      // an odd number of launches: the pad launch makes the step's slots even (pipe_slots)
        const JbPipe* pipe = next(K_PAD);
        if (pipe && mine) {
            pipe_pad_kernel<<<8, 64, 0, s>>>(*pipe);
            JB_CHECK_LAUNCH();
        }
    }
    return JB_OK;
}

// Software-pipelined launches of the decode step.  Available where every launch of the step has a
// pipelined form: fp16, <= 16 samples, wide-value layers throughout (one head of 480 channels), widths of 33..64 k-tiles.
// ... or (multi-head engines: 5b_lyrics' top prior) five launches per layer with the MFMA decode attention: fp16, <= 16 samples, folded
// LayerNorm, no key-split layer, heads of 150 (ragged) / 256 / 512 channels, projections of 33..64 or 129..160 k-tiles.
static bool pipeline_eligible_multi_head(const JbEngine* e) {
    const jb_engine_cfg& c = e->cfg;
    if (!c.pipe_words || c.dtype != JB_F16 || c.n_batch > 16 || !c.x_out_packed || c.width % 32 || c.n_mlp % 32) return false;
    const int att_ld = c.att_ld ? c.att_ld : c.n_state, d = c.n_state / c.n_head;
    auto lnf_ok = [](int nkt) { return (nkt >= 33 && nkt <= 64) || (nkt >= 129 && nkt <= 160); };
    auto proj_ok = [](int nkt) { return nkt >= 1; };              // 4 / 8 waves (any length) or 16 waves (129..160 k-tiles), fp16
    const int nkt_logits = c.width / 16;                          // fp32 head: no 16-wave pipelined form
    if (!lnf_ok(c.width / 32) || att_ld % 32 || !proj_ok(att_ld / 32) || att_ld - c.n_state >= 32 || !proj_ok(c.n_mlp / 32) ||
        (nkt_logits > 128 && nkt_logits <= 160)) return false;
    if (!jb_attn_decode_pipe_supported(c.dtype, d, c.n_state, att_ld, c.n_state)) return false;
    for (const jb_layer& L : e->layers)
        if (layer_wide(c, L) || layer_split_parts(c, L) != 0 || !L.w_attn_f || !L.w_fc_f) return false;
    return true;
}

static bool pipeline_eligible(const JbEngine* e) {
    const jb_engine_cfg& c = e->cfg;
    if (pipeline_eligible_multi_head(e)) return true;
    if (!c.pipe_words || c.dtype != JB_F16 || c.n_batch > 16 || c.n_head != 1 || (c.n_state != 480 && c.n_state != 256) || !c.x_out_packed) return false;
    if (c.width % 32 || c.n_mlp % 32 || c.width / 32 < 32 || c.width / 32 > 64 || c.n_mlp / 32 < 32 || c.n_mlp / 32 > 64) return false;
    for (const jb_layer& L : e->layers)
        if (!layer_wide(c, L) || !L.w_fc_f || layer_max_keys(c, L) > 128) return false;     // one 16-key tile per attention wave
    return true;
}

extern "C" int jb_engine_pipeline(void* handle, int enable) {
    JB_REQUIRE(handle, "null engine");
    JbEngine* e = (JbEngine*)handle;
    if (enable && !pipeline_eligible(e)) JB_UNSUPPORTED("this engine's decode step has launches without a pipelined form (needs pipe_words, "
                                                        "fp16, <= 16 samples, and either wide-value layers of one 480- or 256-channel head on "
                                                        "32..64 k-tiles or folded-LayerNorm multi-head layers without key splits)");
    // ONE pipelined engine per process.  A waiting launch holds up to 180 workgroup slots (8 waves at <= 96 registers per lane:
    // two such workgroups fill a compute unit) while it spins, and the producer it waits for must still find room -- the
    // wide-value attention workgroup needs an otherwise EMPTY compute unit, and the waiters of two engines can leave none (round
    // 4: every slot timed out).  Round 5 admitted two engines on a lean form of that kernel; side by side they ran at 2.02 ms
    // per step against 2.09 for two plain chains and the job did not get faster (HISTORY.md sections 4.5 / 4.6): removed in round 6.
    JB_REQUIRE(enable >= 0 && enable <= 2, "enable must be 0, 1 or 2");
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    if (enable) {
        if (!pipe_own(e)) JB_UNSUPPORTED("another engine of this process runs pipelined launches (one at a time: switch it off or "
                                         "destroy it first)");
        if (enable == 2) release_pipeline(e);   // a fresh pair of streams and fresh graphs at the next decode
    } else {
        // switched off = gone: streams, hardware queues and graphs are released now and made again (milliseconds) the next
        // time the launches are switched on
        release_pipeline(e);
        pipe_disown(e);
    }
    e->pipelined = enable != 0;
    return JB_OK;
}

extern "C" int jb_engine_pipelined(void* handle) { return handle && ((JbEngine*)handle)->pipelined ? 1 : 0; }

// 1 while the engine holds a pair of streams for pipelined launches (made at the first pipelined decode, released by
// jb_engine_pipeline(handle, 0)), else 0.
extern "C" int jb_engine_pipeline_resident(void* handle) { return handle && ((JbEngine*)handle)->pstream[0] ? 1 : 0; }

// Pipelined launches wait on each other ACROSS streams, and HIP multiplexes streams onto a few in-order hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default): a waiting launch that sits in the same hardware queue AHEAD of its producer -- or, with
// two pipelined engines, ahead of the other engine's producer while that engine's waiter sits ahead of ours -- never ends.
// Every stream that carries pipelined launches is therefore checked, once, against every other such stream of the process
// with a two-kernel handshake: a kernel on stream a spins (bounded: 1.5 s) for a flag that a kernel on stream b sets; if
// both streams feed one in-order queue the setter cannot start and the spin times out.  Engines whose streams cannot be
// told apart keep the plain launch chain.
__global__ void pipe_probe_wait_kernel(unsigned* flag, unsigned* result) {
    const long long t0 = wall_clock64();
    while (jb_ld_word(flag) == 0u) {
        if (wall_clock64() - t0 > 150000000ll) { *result = 0u; return; }      // 1.5 s: longer than a busy stream's backlog
        __builtin_amdgcn_s_sleep(32);
    }
    *result = 1u;
}
__global__ void pipe_probe_set_kernel(unsigned* flag) { jb_st_word(flag, 1u); }

// 1: launches on a and b run side by side, 0: they share an in-order queue, < 0: error.  `a` must be the caller's own (or a
// fresh) stream: it is synchronised; `b` may be another engine's busy stream: the setter just queues behind its backlog.
static int streams_overlap(hipStream_t a, hipStream_t b, unsigned* scratch /* device, 2 words */) {
    if (a == b) return 0;
    JB_HIP(hipMemsetAsync(scratch, 0, 2 * sizeof(unsigned), a));
    JB_HIP(hipStreamSynchronize(a));
    pipe_probe_wait_kernel<<<1, 1, 0, a>>>(scratch, scratch + 1);
    pipe_probe_set_kernel<<<1, 1, 0, b>>>(scratch);
    JB_CHECK_LAUNCH();
    unsigned r = 0;
    JB_HIP(hipMemcpyAsync(&r, scratch + 1, sizeof(unsigned), hipMemcpyDeviceToHost, a));      // never the legacy stream
    JB_HIP(hipStreamSynchronize(a));
    return r == 1u ? 1 : 0;
}

// The two streams of a pipelined engine are the engine's own.  A waiting launch blocks the in-order hardware queue it sits
// in, so these streams must not share a queue with each other (deadlock) nor with anybody else (everything behind a waiter
// crawls: measured, the 3-level job went from 80 to 128 s when the side stream was an ordinary pooled stream).  HIP gives
// no handle on the stream -> queue mapping, except that a stream created with a compute-unit mask gets a hardware queue
// with that mask: two masks that differ (each all compute units but one) are two queues of their own.  The
// pair is still verified with the handshake above.  The caller's stream never waits on them: a pipelined decode drains the
// caller's stream, runs on the pair and returns when the pair is done (decode_pipelined).
// 3 where the step is four launches per layer on a single head (wide-value layers throughout: the 1b upsamplers), else 2.
static int pipe_streams(const JbEngine* e) {
    return (!g_pipe_two_streams && !pipeline_eligible_multi_head(e) && pipeline_eligible(e) && !getenv("JB_PIPE_TWO_STREAMS")) ? 3 : 2;
}
static int setup_pipeline_streams(JbEngine* e) {       // caller holds g_pipe_mutex
    unsigned* scratch = e->cfg.pipe_words + jb_pipe_words(pipe_slots(e)) - JB_PIPE_PAD + 8;
    // every pair of the process gets two masks nobody else has (each leaves out ONE compute unit): should the runtime key
    // hardware queues by mask, two engines' pairs still never meet in one queue
    static int g_pairs = 0;
    const int pair = g_pairs++;
    int dev = 0;
    JB_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    JB_HIP(hipGetDeviceProperties(&prop, dev));
    const int n_cu = prop.multiProcessorCount;                   // 256 on MI355X; one mask bit per compute unit
    JB_REQUIRE(n_cu >= 8 && n_cu <= 1024, "unexpected compute-unit count");
    const int usable = n_cu;
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32);
    e->n_pstreams = pipe_streams(e);
    // Three streams: the attention launches own the LAST 72 of 256 compute units (mask bit i is compute unit i / 8 of XCD i mod 8:
    // nine per XCD), the projections' two streams the first 184.  An attention workgroup (8 waves at ~200 registers) needs an
    // EMPTY compute unit, and with the attention dispatched four phases ahead the two parked projection launches (120 + 120
    // workgroups, one per unit) left it 16: the first 16 workgroups ran, the rest were never placed and the parked launches
    // waited for them -- every wait of the step timed out (profiles/r06c22_three_streams_first_run.log).  180 projection
    // workgroups still get a unit each.
    const int att_lo = e->n_pstreams == 3 ? usable - (usable * 72 + 255) / 256 : 0;
    for (int k = 0; k < e->n_pstreams; ++k) {
        std::fill(mask.begin(), mask.end(), 0u);
        const int lo = (e->n_pstreams == 3 && k == 2) ? att_lo : 0, hi = (e->n_pstreams == 3 && k < 2) ? att_lo : usable;
        for (int b = lo; b < hi; ++b) mask[b >> 5] |= 1u << (b & 31);
        // the unit a mask leaves out (masks differ: should the runtime key hardware queues by mask).  Two streams: another one for
        // every pair of the process.  Three streams: FIXED ones -- workgroup b runs on XCD b mod 8, the 180 workgroups of the wide
        // c_attn are 23 on each of XCDs 0..3 and 22 on XCDs 4..7, and the projections' region has 23 units per XCD: a unit taken
        // from XCD 0..3 makes one workgroup of every c_attn wait for a slot (with the masks of the THIRD triple of a process the
        // step went from 1.37 to 1.83 ms: profiles/r06c24_recreate.log), so the projections give up units of XCDs 7 and 6
        const int bit = e->n_pstreams == 3 ? (k == 0 ? att_lo - 1 : (k == 1 ? att_lo - 2 : usable - 3)) : hi - 1 - (2 * pair + k) % (hi - lo);
        mask[bit >> 5] &= ~(1u << (bit & 31));
        JB_HIP(hipExtStreamCreateWithCUMask(&e->pstream[k], (uint32_t)mask.size(), mask.data()));
    }
    int ov = 1;
    for (int a = 0; a < e->n_pstreams && ov == 1; ++a)
        for (int b = a + 1; b < e->n_pstreams && ov == 1; ++b) ov = streams_overlap(e->pstream[a], e->pstream[b], scratch);
    if (ov != 1) {
        for (int k = 0; k < 3; ++k) { if (e->pstream[k]) (void)hipStreamDestroy(e->pstream[k]); e->pstream[k] = nullptr; }
        if (ov < 0) return ov;
        JB_UNSUPPORTED("the streams of the pipelined launches share a hardware queue");
    }
    return JB_OK;
}

// The engine's pair of streams and its two parity graphs (once).
static int prepare_pipeline(JbEngine* e) {
    if (e->pstream[0] && e->pexec[0] && e->pexec[1] && (e->n_pstreams < 3 || e->pexec[2])) return JB_OK;
    // one engine at a time: the handshake synchronises, and another thread's synchronous calls must not fall into this
    // thread's capture
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    if (!e->pstream[0]) JB_TRY(setup_pipeline_streams(e));
    if (!e->capture_stream) JB_HIP(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
    for (int k = 0; k < e->n_pstreams; ++k) {
        if (e->pexec[k]) continue;
        JB_HIP(hipStreamBeginCapture(e->capture_stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_step(e, e->capture_stream, k);
        hipGraph_t gph = nullptr;
        const hipError_t ce = hipStreamEndCapture(e->capture_stream, &gph);
        if (rc != JB_OK) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (ce != hipSuccess) { jb_set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ce)); return JB_ERR_HIP; }
        if (e->pgraph[k]) (void)hipGraphDestroy(e->pgraph[k]);
        e->pgraph[k] = gph;
        JB_HIP(hipGraphInstantiate(&e->pexec[k], gph, nullptr, nullptr, 0));
    }
    return JB_OK;
}

// n_steps pipelined decode steps from the state jb_engine_decode has prepared on `s`.  HOST-SYNCHRONOUS: the caller's
// stream is drained first and the call returns when the pair has finished, so no queue of the process ever holds a packet
// that WAITS (an event wait is a barrier packet; one that stays pending for the seconds a window takes -- the join of the
// round-3 form sat in the caller's queue for the whole call -- cost the pair's launches 18 us dispatch stalls for the first
// ~130 steps of every call whenever its hardware queue shared a pipe with one of the pair's: profiles/r04_pipe_in_job.log,
// cases B / E against A / D).  The calling thread belongs to this level anyway.
static int decode_pipelined(JbEngine* e, int n_steps, hipStream_t s, bool use_graph = true) {
    const int n_slots = pipe_slots(e);
    JB_TRY(prepare_pipeline(e));
    // completion counts and tickets start from zero in every call
    JB_HIP(hipMemsetAsync(e->cfg.pipe_words, 0, (jb_pipe_words(n_slots) - JB_PIPE_PAD) * sizeof(unsigned), s));
    JB_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n_steps; ++i)
        for (int k = 0; k < e->n_pstreams; ++k) {
            if (use_graph) JB_HIP(hipGraphLaunch(e->pexec[k], e->pstream[k]));
            else JB_TRY(enqueue_step(e, e->pstream[k], k));        // diagnostics: the same launches without the graph executor
        }
    for (int k = 0; k < e->n_pstreams; ++k) JB_HIP(hipStreamSynchronize(e->pstream[k]));
    return JB_OK;
}

extern "C" int jb_engine_decode(void* handle, int t0, int n_steps, int use_graph, void* stream) {
    JB_REQUIRE(handle, "null engine");
    JbEngine* e = (JbEngine*)handle;
    JB_REQUIRE(t0 >= 0 && n_steps >= 0 && t0 + n_steps <= e->cfg.seq_len, "step range outside the sequence");
    JB_REQUIRE(e->cfg.x_out_packed && e->cfg.bins > 0, "this engine has no logits head (only_encode)");
    hipStream_t s = (hipStream_t)stream;
    set_int_kernel<<<1, 1, 0, s>>>(e->cfg.t_dev, t0);
    JB_CHECK_LAUNCH();
    if (n_steps == 0) return JB_OK;
    for (const jb_layer& L : e->layers) e->v_rows_stale = e->v_rows_stale || layer_wide(e->cfg, L);
    JB_TRY(enqueue_embed(e, t0, s));         // later positions are embedded by the sampler of the step before
    if (use_graph == 2) {        // diagnostics: the pipelined launches without the graph executor
        JB_REQUIRE(e->pipelined, "use_graph = 2 asks for pipelined launches: switch them on first (jb_engine_pipeline)");
        return decode_pipelined(e, n_steps, s, false);
    }
    if (!use_graph) {            // the eager plain chain, whatever jb_engine_pipeline says
        for (int i = 0; i < n_steps; ++i) JB_TRY(enqueue_step(e, s));
        return JB_OK;
    }
    if (e->pipelined && use_graph != 3) {
        const int rc = decode_pipelined(e, n_steps, s);
        if (rc != JB_ERR_UNSUPPORTED || e->pstream[0]) return rc;
        // no hardware queue of its own for the side stream: the plain chain from here on, and another engine may have them
        e->pipelined = false;
        std::lock_guard<std::mutex> lock(g_pipe_mutex);
        pipe_disown(e);
    }
    if (!e->graph_exec) {
        // One eager step first, so that every kernel's dynamic-LDS attribute is configured outside of capture;
        // it computes position t0, which the first replay recomputes identically (the sampler's random stream is
        // keyed by position), so the counter and the embedding of t0 are simply restored.  Then one step is recorded
        // on a private stream -- recording executes nothing -- and replayed on the caller's stream.
        JB_TRY(enqueue_step(e, s));
        set_int_kernel<<<1, 1, 0, s>>>(e->cfg.t_dev, t0);
        JB_CHECK_LAUNCH();
        JB_TRY(enqueue_embed(e, t0, s));
        if (!e->capture_stream) JB_HIP(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
        hipStream_t cs = e->capture_stream;
        JB_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_step(e, cs);
        hipGraph_t gph = nullptr;
        hipError_t ce = hipStreamEndCapture(cs, &gph);
        if (rc != JB_OK) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (ce != hipSuccess) { jb_set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ce)); return JB_ERR_HIP; }
        e->graph = gph;
        JB_HIP(hipGraphInstantiate(&e->graph_exec, e->graph, nullptr, nullptr, 0));
    }
    for (int i = 0; i < n_steps; ++i) JB_HIP(hipGraphLaunch(e->graph_exec, s));
    return JB_OK;
}

// decode_qkv at sample_t == 0 (factored_attention.py:273-280): key / value = c_enc_kv(encoder_kv), once per window.
extern "C" int jb_engine_set_encoder_kv(void* handle, void* stream) {
    JB_REQUIRE(handle, "null engine");
    JbEngine* e = (JbEngine*)handle;
    const jb_engine_cfg& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int N = c.n_batch, W = c.width, S = c.n_state;
    for (int l = 0; l < c.n_layers; ++l) {
        const jb_layer& L = e->layers[l];
        if (L.attn_func != JB_ATTN_CROSS) continue;
        for (int part = 0; part < 2; ++part) {
            jb_gemm_args g = {};
            g.dtype = c.dtype; g.A = c.encoder_kv; g.lda = W; g.W = part == 0 ? L.w_enc_k : L.w_enc_v;
            g.bias = L.b_enc_kv + part * S; g.out = part == 0 ? L.kcache : L.vcache; g.ldo = S;
            g.n_seq = N; g.t_in = c.enc_len; g.t_out = c.enc_len; g.in_seq_stride = c.enc_len; g.out_seq_stride = c.enc_len;
            g.K = W; g.J = S; g.n_taps = 1; g.in_stride = 1; g.out_stride = 1; g.res_scale = 1.0f;
            JB_TRY(jb_gemm(&g, s));
        }
    }
    return JB_OK;
}

// Chunked prefill of positions t0 .. t0+n_t-1 (primed_sample, autoregressive.py:284-318), in sub-chunks of
// at most chunk_cap positions; outputs are discarded unless preds is set (get_preds).
extern "C" int jb_engine_prefill(void* handle, int t0, int n_t, void* stream) {
    JB_REQUIRE(handle, "null engine");
    JbEngine* e = (JbEngine*)handle;
    const jb_engine_cfg& c = e->cfg;
    JB_REQUIRE(t0 >= 0 && n_t > 0 && t0 + n_t <= c.seq_len, "chunk outside the sequence");
    JB_REQUIRE(c.chunk_cap > 0 && c.c_xa && c.c_xb && c.c_h && c.c_q && c.c_att && c.c_mlp, "prefill buffers missing");
    JB_REQUIRE(!c.preds || c.c_xf, "c_xf required when preds is set");
    if (t0 == 0) e->v_rows_stale = false;
    JB_REQUIRE(!e->v_rows_stale, "prefill at t0 > 0 after decode steps: wide-value layers did not append v rows (prefill the "
                                 "window from position 0, or build the engine without wide-value layers)");
    hipStream_t s = (hipStream_t)stream;
    const int N = c.n_batch, W = c.width, S = c.n_state, M = c.n_mlp, H = c.n_head, d = S / H;
    for (int off = 0; off < n_t; off += c.chunk_cap) {
        const int C = (n_t - off < c.chunk_cap) ? n_t - off : c.chunk_cap;
        const int p0 = t0 + off;
        const int64_t rows = (int64_t)N * C;
        JB_TRY(jb_embed(c.dtype, c.c_xa, c.tokens, c.tok_stride, c.x_emb, c.pos_emb, c.start, c.start_stride, c.x_cond,
                        c.xc_n_stride, c.xc_t_stride, N, W, p0, nullptr, C, s));
        auto base = [&](jb_gemm_args& g, const void* A, int64_t lda, int K, const void* Wp, const float* b, int J,
                        void* out, int64_t ldo) {
            g = {};
            g.dtype = c.dtype; g.A = A; g.lda = lda; g.W = Wp; g.bias = b; g.out = out; g.ldo = ldo;
            g.n_seq = N; g.t_in = C; g.t_out = C; g.in_seq_stride = C; g.out_seq_stride = C; g.K = K; g.J = J;
            g.n_taps = 1; g.in_stride = 1; g.out_stride = 1; g.res_scale = 1.0f;
        };
        for (int l = 0; l < c.n_layers; ++l) {
            const jb_layer& L = e->layers[l];
            jb_gemm_args g;
            JB_TRY(jb_layernorm_fwd(c.c_xa, c.dtype, c.c_h, c.dtype, L.ln0_g, L.ln0_b, rows, W, c.ln_eps, s));
            if (L.attn_func == JB_ATTN_CROSS) {
                base(g, c.c_h, W, W, L.w_attn, L.b_attn, S, c.c_q, S);
            } else {
                base(g, c.c_h, W, W, L.w_attn, L.b_attn, 3 * S, c.c_q, S);
                g.qkv_split = 1; g.S = S; g.kcache = L.kcache; g.vcache = L.vcache; g.cache_cap = L.cache_cap; g.cache_t0 = p0;
            }
            JB_TRY(jb_gemm(&g, s));
            if (L.vcache_w && p0 < L.cache_cap) {
                // wide-value layer: v' = v·Wp for the decode steps, from the v rows just cached: 4x fewer FLOPs than carrying
                // Wv·Wp as extra c_attn columns (measured: +104 ms per 4096 x 16-token window that way).
                const int Cn = (C < L.cache_cap - p0) ? C : L.cache_cap - p0;
                // one launch for all samples: rows (sample, position), a sample's rows cache_cap apart in both caches
                jb_gemm_args gv;
                base(gv, (const f16*)L.vcache + (int64_t)p0 * S, S, S, L.w_proj, nullptr, W, (f16*)L.vcache_w + (int64_t)p0 * W, W);
                gv.t_in = gv.t_out = Cn; gv.in_seq_stride = gv.out_seq_stride = L.cache_cap;
                JB_TRY(jb_gemm(&gv, s));
            }
            JB_TRY(jb_attn_prefill(c.dtype, L.attn_func, c.c_q, L.kcache, L.vcache, L.cache_cap, c.c_att, N, H, d,
                                   c.block_ctx, p0, C, s));
            if (c.rec_out && l == c.rec_layer)
                JB_TRY(jb_attn_probs(c.dtype, L.attn_func, c.c_q, L.kcache, L.cache_cap, c.rec_out, c.rec_n_stride, p0,
                                     c.rec_keys, N, H, d, c.rec_head, c.block_ctx, p0, C, s));
            base(g, c.c_att, S, S, L.w_proj, L.b_proj, W, c.c_xb, W);
            g.res = c.c_xa; g.ldr = W;
            JB_TRY(jb_gemm(&g, s));
            JB_TRY(jb_layernorm_fwd(c.c_xb, c.dtype, c.c_h, c.dtype, L.ln1_g, L.ln1_b, rows, W, c.ln_eps, s));
            base(g, c.c_h, W, W, L.w_fc, L.b_fc, M, c.c_mlp, M);
            g.act = JB_ACT_QUICK_GELU;
            JB_TRY(jb_gemm(&g, s));
            base(g, c.c_mlp, M, M, L.w_proj2, L.b_proj2, W, c.c_xa, W);
            g.res = c.c_xb; g.ldr = W;
            JB_TRY(jb_gemm(&g, s));
        }
        if (c.hidden_out)      // only_encode models: the activations are the output (autoregressive.py:155-157)
            JB_TRY(jb_final_add(c.dtype, c.c_xa, c.hidden_out + (int64_t)p0 * W, c.hidden_n_stride,
                                c.add_cond_after ? c.x_cond : nullptr, c.xc_n_stride, c.xc_t_stride, N, W, p0, nullptr, C, s));
        if (c.preds) {
            JB_REQUIRE(c.x_out_packed, "preds requested from an engine without a logits head");
            JB_TRY(jb_final_add(c.dtype, c.c_xa, c.c_xf, 0, c.add_cond_after ? c.x_cond : nullptr, c.xc_n_stride,
                                c.xc_t_stride, N, W, p0, nullptr, C, s));
            jb_gemm_args g;
            base(g, c.c_xf, W, W, c.x_out_packed, nullptr, c.bins, c.preds, c.bins);
            g.dtype = JB_F32;
            g.out_seq_stride = c.preds_n_stride / c.bins; g.out_offset = p0;
            JB_TRY(jb_gemm(&g, s));
        }
    }
    set_int_kernel<<<1, 1, 0, s>>>(c.t_dev, t0 + n_t);
    JB_CHECK_LAUNCH();
    return JB_OK;
}

// In-situ timing of the dominant kernel: n_steps passes over all layers launching ONLY the LayerNorm-fused
// projections (attn.c_attn, mlp.c_fc) with their real arguments -- 2*L launches per pass, every launch streaming a
// different, cold weight matrix as in the real step -- back to back on `stream`, bracketed by one HIP event pair.
// Synchronises the stream.  out[0] = average microseconds per launch, out[1] = launches timed, out[2] = average
// algorithmic bytes per launch (weights once + activation rows in + rows out).
extern "C" int jb_engine_probe_projection(void* handle, int t0, int n_steps, void* stream, double* out) {
    JB_REQUIRE(handle && out, "null pointer");
    JbEngine* e = (JbEngine*)handle;
    JB_REQUIRE(t0 >= 0 && n_steps > 0 && t0 < e->cfg.seq_len, "position outside the sequence");
    hipStream_t s = (hipStream_t)stream;
    const jb_engine_cfg& c = e->cfg;
    const int N = c.n_batch, W = c.width, S = c.n_state, M = c.n_mlp;
    set_int_kernel<<<1, 1, 0, s>>>(c.t_dev, t0);
    JB_CHECK_LAUNCH();
    hipEvent_t e0, e1;
    JB_HIP(hipEventCreate(&e0));
    JB_HIP(hipEventCreate(&e1));
    auto burst = [&](int reps) -> int {
        for (int i = 0; i < reps; ++i)
            for (int l = 0; l < c.n_layers; ++l) {
                const jb_layer& L = e->layers[l];
                // (the kernel form the engine's own steps launch: operand-order blocks where it has them)
                JbPipe pp{c.pipe_words, nullptr, nullptr, JB_PIPE_NO_SYNC, -1, 0, nullptr, 0, 0};      // (runs: a valid word, read and ignored)
                const bool frag = pipe_operand_order(e);
                jb_gemv_args g;
                fill_c_attn(g, c, L);
                pp.frag = JB_FRAG_X;
                JB_TRY(jb_gemv_impl(&g, frag ? &pp : nullptr, s));
                g = {};
                fill_ln_proj(g, c, L, 1);
                g.x = c.x_b; g.J = M; g.out = c.mlp; g.ldo = M; g.act = JB_ACT_QUICK_GELU;
                pp.frag = JB_FRAG_X | JB_FRAG_OUT;
                JB_TRY(jb_gemv_impl(&g, frag ? &pp : nullptr, s));
            }
        return JB_OK;
    };
    JB_TRY(burst(1));                                   // warm-up pass (instruction caches, LDS attributes)
    JB_HIP(hipEventRecord(e0, s));
    JB_TRY(burst(n_steps));
    JB_HIP(hipEventRecord(e1, s));
    JB_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    JB_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double launches = 2.0 * c.n_layers * n_steps;
    const double esz = c.dtype == JB_F16 ? 2.0 : 4.0;
    // ALGORITHMIC bytes: a wide-value layer's c_attn streams W x (2S + W) weights, but the work it stands for is the
    // reference's c_attn (W x 3S) plus the attn.c_proj it absorbed (S x W) -- only those are counted
    double b_attn = 0.0;
    for (const jb_layer& L : e->layers) {
        const double J = L.attn_func == JB_ATTN_CROSS ? S : (layer_wide(c, L) ? 4.0 * S : 3.0 * S);
        b_attn += ((double)W * J * esz + (double)N * W * esz + (double)N * J * esz) / c.n_layers;
    }
    const double b_fc = (double)W * M * esz + (double)N * W * esz + (double)N * M * esz;
    out[0] = (double)ms * 1e3 / launches;
    out[1] = launches;
    out[2] = 0.5 * (b_attn + b_fc);
    return JB_OK;
}

// Algorithmic HBM bytes of one decode step at position t: every weight matrix once, the k/v rows
// each layer's pattern reads plus the row it appends, the fp32 logits head, and the activation rows between kernels.
extern "C" double jb_engine_step_bytes(void* handle, int t) {
    if (!handle) return 0.0;
    const JbEngine* e = (const JbEngine*)handle;
    const jb_engine_cfg& c = e->cfg;
    const double esz = c.dtype == JB_F16 ? 2.0 : 4.0;
    const double N = c.n_batch, W = c.width, S = c.n_state, M = c.n_mlp;
    double bytes = 0.0;
    for (int l = 0; l < c.n_layers; ++l) {
        const jb_layer& L = e->layers[l];
        const int bc = c.block_ctx;
        double keys = 0;
        switch (L.attn_func) {
            case JB_ATTN_DENSE: keys = t + 1; break;
            case JB_ATTN_BLOCK: keys = t % bc + 1; break;
            case JB_ATTN_TRANSPOSE_BLOCK: keys = t / bc + 1; break;
            case JB_ATTN_PREV_BLOCK: keys = t >= bc ? bc : 0; break;
            case JB_ATTN_PRIME: keys = t + 1 < L.cache_cap ? t + 1 : L.cache_cap; break;
            case JB_ATTN_CROSS: keys = L.cache_cap; break;
        }
        const double w_attn = L.attn_func == JB_ATTN_CROSS ? W * S : 3.0 * W * S;
        bytes += (w_attn + S * W + W * M + M * W) * esz;                 // weights, once per step
        bytes += keys * S * 2.0 * esz * N;                               // k/v rows read
        if (L.attn_func != JB_ATTN_CROSS) bytes += S * 2.0 * esz * N;    // k/v row appended
    }
    bytes += (double)c.bins * W * 4.0;                                   // logits head (fp32)
    return bytes;
}
