// libmasr_hip.so -- engine: weight store, workspace, orchestration of the Conformer hot path and
// the C ABI declared in include/masr_hip.h.  Host-side only; every kernel lives in the sibling
// .hip files and is launched on the caller's stream.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/masr_hip.h"
#include "common.h"

using namespace masr;
namespace masr {
int lm_device_view(masr_lm* lm, LmView* out);      // lm_scorer.cpp: the LM table in the memory of the current device
bool lm_word_based(const masr_lm* lm);
}
extern "C" const char* masr_lm_last_error(void);

static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return 1;
}
namespace masr {
int engine_fail(const std::string& m) { return fail(m); }      // pool.hip reports through the same masr_last_error()
}
#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" +     \
                        std::to_string(__LINE__) + ")");                                                \
    } while (0)
#define CHK(expr)              \
    do {                       \
        int _r = (expr);       \
        if (_r) return _r;     \
    } while (0)

// errors noted by the kernel launchers (a refused dynamic-LDS opt-in, ...) since the last check on this thread
static thread_local std::string g_launch_err;
namespace masr {
void note_launch_error(const char* what, hipError_t e) {
    if (g_launch_err.empty()) g_launch_err = std::string(what) + ": " + hipGetErrorString(e);
}
void ensure_dynamic_lds(const void* fn, size_t bytes, LdsAttr& st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (bytes <= st.granted[dev]) return;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        note_launch_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
        return;
    }
    st.granted[dev] = bytes;
}
}  // namespace masr
// after a group of launches: a launcher-side error first, then the runtime's sticky launch error
#define LAUNCHCHK()                                     \
    do {                                                \
        if (!g_launch_err.empty()) {                    \
            std::string _m = g_launch_err;              \
            g_launch_err.clear();                       \
            (void)hipGetLastError();                    \
            return fail(_m);                            \
        }                                               \
        HIPCHK(hipGetLastError());                      \
    } while (0)
// every entry point that touches the device first makes the engine's GPU current on the calling thread (an engine on
// device != 0 may be driven from a worker thread whose current device is still 0)
#define ENTER(e) HIPCHK(hipSetDevice((e)->cfg.device_id))

static int g_no_ffn_tail = 0;     // masr_debug_set key 8: 1 = the QKV projection as its own launch after the first FFN (A/B)
// row blocks below which the FFN splits d_ff across workgroups (masr_debug_set key 13): up to 191 row blocks the split (256 /
// rowblocks ways) + its reduction beat one full-d_ff workgroup per row block on a quarter to three quarters of the CUs
// (128 streams: chunk call 5.34 -> 3.16 ms, 256 streams 6.52 -> 4.99 ms; tools/chunk_step_ab.py)
static int g_ffn_split_blocks = 192;
static int g_no_ffn_head = 0;     // masr_debug_set key 9: 1 = depthwise conv and pointwise_conv2 as their own launches before the second FFN (A/B)
// masr_debug_set key 30: 1 = few rows: [depthwise conv -> LN -> SiLU -> pointwise_conv2 + residual] as the head stage of the d_ff-split
// FFN launch (every slice repeats it on its row block's rows; slice 0 publishes them).  Built in round 4 because round 3 priced
// it at -3.4 us per layer; MEASURED (tools/chunk_lat.py, MASR_AB=30:0,30:1,30:0,30:1, one process, one box): 16 streams 1.177 /
// 1.162 ms per chunk call without it, 1.199 / 1.170 ms with it; 128 streams 3.011 / 3.017 vs 3.037 / 3.042 ms -- the 128
// dependent MFMAs + the window loads it adds to EVERY slice's critical path cost what the removed 11 us launch (whose columns
// spread over 32 workgroups) cost.  Off by default; identical frame decisions either way.
static int g_split_head = 0;
static int g_efficient_fused = 1;   // masr_debug_set key 31: 0 = Efficient-Conformer layers keep separate out-proj / pw1 / dwconv / pw2 launches (A/B)
// masr_debug_set key 34: 1 = offline Conformer layers run attention AND the [out-proj -> LN -> pw1 -> GLU] chain as ONE launch
// (attention.hip attn_chain_kernel: 32 queries x all four heads per workgroup, context rows in LDS).  Built in round 4 (verdict
// item 6, priced at -0.13 ms per step in round 3), bit-identical to the two launches -- and MEASURED no faster: 58.3 us per launch
// against 25.9 + 32.7 us (rocprofv3, one trace), 6.428 vs 6.412 ms per 32 x 10 s pass (tools/attn_chain_ab.py): a workgroup that
// owns 32 queries of all four heads stages four heads' K' / V tiles per 64 MFMAs per wave where attention_kernel's 128 queries of
// one head stage one -- the saved prologue / epilogue / att round trip is paid back in staging.  Off by default.
static int g_attn_chain = 0;
// masr_debug_set key 35: 1 = one-chunk d_ff slices of few rows (<= 8 row blocks) run ffn_coop.hip -- all eight waves on both products
// (GEMM 1 as 16 x 16 x 4 tiles without a K split, GEMM 2 as 32 x 32 x 2), every weight fragment from packed copies, all loads in
// flight before the LayerNorm -- instead of the producer / consumer kernel, whose two roles run one after the other when a
// workgroup owns ONE chunk.  Built in round 4 on the estimate of 2 x 1.7 us of matrix pipe saved per launch; MEASURED slower:
// 16 streams 1.167 / 1.195 ms per chunk call against 1.135 / 1.109 ms (tools/chunk_lat.py MASR_AB=35:0,35:1,35:0,35:1; with the
// weights read from the row-major matrices: 1.33 ms -- 16 / 32 cache lines per load instruction).  The launch is bound by its
// dependent memory round trips (rows written by another XCD, LayerNorm, LDS exchange, partial store), not by the 256 MFMAs.  Off.
static int g_ffn_coop = 0;
static int g_few_rows_path = 1;   // masr_debug_set key 29: 0 = offline Conformer layers of few row blocks keep the row-block chain kernel (A/B)
// masr_debug_set key 36: row blocks from which a full-context Squeezeformer layer runs as attention + the two fused stage kernels of
// sqz_layer.hip (0 = never: the twelve separate launches, kept for A/B and the bit-identity test).  128 since round 6: the half-rate
// layers of BASELINE configs[2]'s second pass (144 row blocks, 100 of them valid) are 0.4 ms per call faster fused, on one lane
// and on two (17.5 -> 17.0 / 19.2 -> 18.8 ms); passes of ~120 half-rate row blocks (32 x 10 s) stay on the d_ff-split launches
// (fused from 96: 21.2 against 18.9 ms per call of three such passes, round 5)
static int g_sqz_fused_blocks = 128;
static int g_no_chain = 0;   // masr_debug_set key 5: 1 = separate out-projection and pointwise_conv1 kernels (A/B)

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes) return 0;
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        bytes = 0;
        size_t want = n + n / 8 + 256;
        HIPCHK(hipMalloc(&p, want));
        // zero once: kernels that skip the padded row blocks of a batch (masr_debug_set key 38) leave those rows as they are, and
        // what is there must at least be finite -- a later 0 * stale product must not see the NaN patterns of fresh memory
        HIPCHK(hipMemset(p, 0, want));
        // (the fill runs on the NULL stream, which a non-blocking stream does not wait for: without this a lane's first launches
        // could write the buffer BEFORE it is cleared -- the first two-lane call lost its second pass that way)
        HIPCHK(hipStreamSynchronize(nullptr));
        bytes = want;
        return 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Pinned host staging for the per-call descriptor tables (AttSeq rows, cache pointers) of the streaming chunk step: the
// copies to the device are truly asynchronous, so the chunk step does not stall the host on the work queued before it.
// One call owns the area at a time: `begin` waits for the previous call's copies (normally long done).
struct PinnedStage {
    char* p = nullptr;
    size_t bytes = 0, used = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    int begin(size_t need) {
        if (pending) {
            HIPCHK(hipEventSynchronize(ev));
            pending = false;
        }
        if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        if (need > bytes) {
            if (p) HIPCHK(hipHostFree(p));
            p = nullptr;
            bytes = 0;
            const size_t want = need + need / 4 + 4096;
            HIPCHK(hipHostMalloc((void**)&p, want, hipHostMallocDefault));
            bytes = want;
        }
        used = 0;
        return 0;
    }
    // copy `n` bytes of `src` through the staging area to `dst` (device) on stream `s`
    int push(void* dst, const void* src, size_t n, hipStream_t s) {
        if (n == 0) return 0;
        const size_t at = (used + 15) & ~(size_t)15;
        if (at + n > bytes) return fail("descriptor staging area overflow");
        memcpy(p + at, src, n);
        used = at + n;
        HIPCHK(hipMemcpyAsync(dst, p + at, n, hipMemcpyHostToDevice, s));
        return 0;
    }
    int end(hipStream_t s) {
        HIPCHK(hipEventRecord(ev, s));
        pending = true;
        return 0;
    }
    void release() {
        if (pending) (void)hipEventSynchronize(ev);
        if (p) (void)hipHostFree(p);
        if (ev) (void)hipEventDestroy(ev);
        p = nullptr;
        ev = nullptr;
        bytes = 0;
        pending = false;
    }
};

struct LayerW {
    float *ln_ffm_w, *ln_ffm_b, *ffm_w1, *ffm_b1, *ffm_w2, *ffm_b2;
    float *ln_mha_w, *ln_mha_b, *wqkv, *bqkv, *wo, *bo, *pos_u, *pos_v, *wpos, *ptab;
    float *ln_conv_w, *ln_conv_b, *pw1_w, *pw1_b, *dw_w, *dw_b, *cln_w, *cln_b, *pw2_w, *pw2_b, *gconst;
    float *chain_w, *chain_b;     // [Wo; W_pw1] [768,256] and [bo; b_pw1] for the fused out-proj -> LN -> pw1 -> GLU kernel
    float *ln_ff_w, *ln_ff_b, *ff_w1, *ff_b1, *ff_w2, *ff_b2;
    float *ln_fin_w, *ln_fin_b;
};

struct SqLayerW {   // Squeezeformer block (post-LN, adaptive scale/bias, BatchNorm conv module)
    float *att_s, *att_b, *wqkv, *bqkv, *wo, *bo, *pos_u, *pos_v, *wpos, *ptab, *ln1_w, *ln1_b;
    float *f1_s, *f1_b, *f1_w1, *f1_b1, *f1_w2, *f1_b2, *ln2_w, *ln2_b;
    float *cv_s, *cv_b, *pw1_w, *pw1_b, *dw_w, *dw_b, *bn_scale, *bn_shift, *pw2_w, *pw2_b, *ln3_w, *ln3_b, *gconst;
    float *f2_s, *f2_b, *f2_w1, *f2_b1, *f2_w2, *f2_b2, *ln4_w, *ln4_b;
};

struct Ds2LayerW {  // DeepSpeech2 RNN layer: input projection of both directions stacked, recurrent weights, LayerNorm
    float *wih, *bih, *whh, *ln_w, *ln_b;
    int kin;
};

struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
};

struct Stream {
    bool open = false;
    int offset = 0;
    int offset_r = 0;   // frames emitted at the reduced rate (Squeezeformer blocks between time reduction and recovery)
    int cap = 0;
    int history = -1;   // required_cache_size of forward_chunk: < 0 keep every key, >= 0 attend over at most that many cached keys
    int cache_t1 = 0;   // cached keys the next chunk attends over, in input-rate frames (= att_cache.size(2) of the reference)
    int first_half = 0; // half-rate layers (Squeezeformer between reduction and recovery, Efficient-Conformer behind the stride
                        // layer): index of their first kept cache entry (advances by next_cache_start // 2 per step)
    DevBuf att;  // [L][cap][2*d]  (k | v per row)
    DevBuf cnn;  // [L][kernel-1][d]   (DeepSpeech2: LSTM state [L][2 (h, c)][rnn_size])
    DevBuf cnn2; // Conformer / Squeezeformer: second half of the double-buffered cnn cache (a chunk step reads `cnn`, writes
                 // `cnn2`, then the two are swapped -- lets one kernel read the history and write the new cache without ordering)
};

struct GBeam {        // device-resident streaming CTC prefix beam search (masr_gbeam_*)
    bool open = false, started = false;
    int beam = 0, blank = 0, cap = 0;
    DevBuf pool, state;
    masr_lm* lm = nullptr;       // external scorer bound with masr_gbeam_set_lm (not owned)
    float alpha = 0.f, beta = 0.f;
};

enum ProfKind { PROF_NONE = 0, PROF_GEMM = 1, PROF_FFN1 = 2, PROF_CONV2 = 3, PROF_ATT = 4, PROF_FBANK = 5,
                PROF_FFN_TAIL = 6, PROF_FFN_HEAD = 7 };     // 2 = the plain fused FFN kernel; 6 = its variant with the QKV tail stage, 7 = with the conv-module head stage (own kernel names)

}  // namespace

// Everything a forward call WRITES on the device (and the pinned descriptor staging of the chunk step): one set per LANE.
// masr_select_lane parks the active set and brings another one in, so two offline passes can be in flight on two streams of one
// engine (one set of weights) -- launches that share a lane must be ordered by their stream, as before.
struct EngineWs {
    DevBuf gx, rnn_out, hstate, cstate, ds2_lens;                               // DeepSpeech2 workspaces
    PinnedStage stage;
    DevBuf qplanes, attp, cnnptrs, ffpart;                                      // planar q|k|v and attention output, [B][Tpad][256]
    DevBuf fb_scratch;
    DevBuf x1, x2, x, ln, hid, qkv, att, lnpad, glu, dwo, logits, feats, enc, idx, maxp, attseq, gain, nframes, lens, xsave,
        xred, xh;      // xh: rows updated by the head stage of a d_ff-split FFN launch (few rows)
    void release_all() {
        for (DevBuf* b : {&gx, &rnn_out, &hstate, &cstate, &ds2_lens, &qplanes, &attp, &cnnptrs, &ffpart, &fb_scratch, &x1, &x2, &x,
                          &ln, &hid, &qkv, &att, &lnpad, &glu, &dwo, &logits, &feats, &enc, &idx, &maxp, &attseq, &gain, &nframes,
                          &lens, &xsave, &xred, &xh})
            b->release();
        stage.release();
    }
};

// hipMemset runs on the NULL stream, which a non-blocking stream (every torch side stream, the library's own) does not wait for:
// the host waits for the fill, so that whatever is launched next on any stream sees the cleared memory
static int clear_sync(void* p, int value, size_t n) {
    HIPCHK(hipMemset(p, value, n));
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}

struct masr_engine : EngineWs {
    masr_config cfg;
    EngineWs parked[MASR_LANES];     // the workspace sets of the lanes that are not active (parked[lane] is unused)
    int lane = 0;
    hipEvent_t pack_ev = nullptr;    // recorded behind the last call that built a packed weight copy on first use ...
    void* pack_stream = nullptr;     // ... on this stream: calls on OTHER streams wait for it before they read the copies
    bool pack_pending = false;
    bool finalized = false;
    std::map<std::string, HostTensor> host;
    std::vector<void*> owned;   // device weight allocations
    // weights
    float *cmvn_mean = nullptr, *cmvn_istd = nullptr, *conv1_w = nullptr, *conv1_b = nullptr, *conv2_w = nullptr,
          *conv2_b = nullptr, *embed_w = nullptr, *embed_b = nullptr, *after_w = nullptr, *after_b = nullptr,
          *ctc_w = nullptr, *ctc_b = nullptr, *pe = nullptr;
    std::vector<LayerW> layers;
    std::vector<SqLayerW> sq_layers;
    std::vector<Ds2LayerW> ds2_layers;
    std::vector<GBeam> gbeams;
    DevBuf beam_pool, beam_state;
    // whole-utterance prefix searches launched on OTHER streams than the first one get workspaces of their own: two searches may
    // run next to each other (predict_batch: the passes' searches on two side streams), launches on one stream are ordered anyway
    std::map<void*, std::pair<DevBuf, DevBuf>> beam_ws;
    // masr_mean_square runs on whatever stream prepares the NEXT pass (the facade's side stream, the contract step's copy stream)
    // while the feature launch of the current pass reads e->gain on the compute stream: its chunk sums and its unused gain slots
    // live in a scratch of their own, one per calling stream (launches on one stream are ordered anyway)
    std::map<void*, DevBuf> ms_ws;
    int skip_padding = 7;            // masr_debug_set key 38: the offline Squeezeformer launches skip the all-padding row blocks of a batch
    void* beam_first_stream = nullptr;
    bool beam_first_set = false;
    std::map<const float*, std::pair<DevBuf, DevBuf>> ffn_packed;  // fp32 FFN weights in fragment order (ffn_pc.hip VAR == 2), per W1 pointer
    std::map<const float*, std::pair<DevBuf, DevBuf>> ffn_dual_packed;  // the same in the two-chain order of ffn_dual.hip (and its QKV tail weights)
    std::map<const float*, std::pair<DevBuf, DevBuf>> x3_packed;   // exploratory split-bf16 FFN: packed weights per FFN (W1 pointer)
    long long* beam_prof = nullptr;                                             // debug: phase cycle counters (masr_debug_set key 2)                                               // GPU beam search scratch                               // DeepSpeech2 workspaces
    float *preln_w = nullptr, *preln_b = nullptr, *tr_dw_w = nullptr, *tr_dw_b = nullptr, *tr_pw_w = nullptr,
          *tr_pw_b = nullptr, *rec_w = nullptr, *rec_b = nullptr;
    int reduce_idx = -1, recover_idx = -1;
    int stride_idx = -1, n_group_layers = 0, group_size = 3;   // Efficient-Conformer (model_kind 2)
    bool conv_bn = false;            // Conformer with cnn_module_norm: batch_norm (cfg.reserved[0] = 1): LayerW::cln_w / cln_b hold the folded scale / shift
    // fbank tables
    float *window = nullptr, *melwt = nullptr, *tw512 = nullptr, *twr4 = nullptr;
    FbankTables fbank_tables() const { return FbankTables{window, melwt, tw512, twr4, mel_lo}; }
    int* mel_lo = nullptr;
    // mfcc / linear tables (built on first use)
    float *dct = nullptr, *lifter = nullptr;
    int dct_ceps = 0;
    double *lin_win = nullptr, *lin_tw = nullptr;
    double lin_scale = 0.0;
    // streams
    std::vector<Stream> streams;
    // profiling
    int prof_kind = 0;
    int prof_stride = 1, prof_seen = 0;                     // every prof_stride-th matching launch is timed (masr_debug_set key 16)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    double prof_flops = 0.0;
};

namespace {

template <class T>
int upload(masr_engine* e, const std::vector<T>& v, T** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
    HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    e->owned.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return 0;
}

int get(masr_engine* e, const std::string& name, std::vector<int64_t> shape, const HostTensor** out) {
    auto it = e->host.find(name);
    if (it == e->host.end()) return fail("missing tensor: " + name);
    int64_t want = 1, have = 1;
    for (auto s : shape) want *= s;
    for (auto s : it->second.shape) have *= s;
    if (want != have)
        return fail("tensor " + name + ": expected " + std::to_string(want) + " elements, got " + std::to_string(have));
    *out = &it->second;
    return 0;
}

int up(masr_engine* e, const std::string& name, std::vector<int64_t> shape, float** out) {
    const HostTensor* t;
    CHK(get(e, name, shape, &t));
    return upload(e, t->v, out);
}

// ---- profiling helpers ------------------------------------------------------------------------------
struct ProfScope {
    masr_engine* e;
    hipStream_t s;
    bool on;
    ProfScope(masr_engine* e_, hipStream_t s_, int kind, double flops) : e(e_), s(s_) {
        on = e->prof_kind != 0 && (e->prof_kind == kind || (e->prof_kind == PROF_GEMM && (kind == PROF_FFN1 || kind == PROF_CONV2 || kind == PROF_FFN_TAIL || kind == PROF_FFN_HEAD)));
        if (on && e->prof_stride > 1) on = (e->prof_seen++ % e->prof_stride) == 0;     // a sample of the launches: two event
        if (!on) return;                                                                // records cost ~6 us of stream time
        if (e->prof_used == e->prof_events.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            e->prof_events.push_back({a, b});
        }
        hipEventRecord(e->prof_events[e->prof_used].first, s);
        e->prof_flops += flops;
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(e->prof_events[e->prof_used].second, s);
        e->prof_used++;
    }
};

// Packed weight copies are built on first use by whatever call needs them first, on that call's stream.  With two lanes a call
// on ANOTHER stream may find the copy in the map while the packing launch is still queued: the call that packed records an event
// behind itself, and calls on other streams wait for that event until it has completed.
struct CallGuard {
    masr_engine* e;
    hipStream_t s;
    size_t n0;
    static size_t packed_count(const masr_engine* e) { return e->ffn_packed.size() + e->ffn_dual_packed.size() + e->x3_packed.size(); }
    CallGuard(masr_engine* e_, hipStream_t s_) : e(e_), s(s_), n0(packed_count(e_)) {
        if (!e->pack_pending) return;
        if (hipEventQuery(e->pack_ev) == hipSuccess) {
            e->pack_pending = false;
        } else {
            (void)hipGetLastError();          // (hipErrorNotReady is not an error of ours)
            if ((void*)s != e->pack_stream) (void)hipStreamWaitEvent(s, e->pack_ev, 0);
        }
    }
    ~CallGuard() {
        if (packed_count(e) == n0) return;
        if (!e->pack_ev && hipEventCreateWithFlags(&e->pack_ev, hipEventDisableTiming) != hipSuccess) return;
        // (a pending event of another stream: this call waited for it at its start, so the new record covers it)
        if (hipEventRecord(e->pack_ev, s) == hipSuccess) {
            e->pack_stream = (void*)s;
            e->pack_pending = true;
        }
    }
};

static int g_ffn_dual = 0;         // masr_debug_set key 24: 0 = the full FFN launches run ffn_pc.hip (one accumulator chain per wave) instead of ffn_dual.hip (A/B)
static int g_ffn_packed = 2;       // masr_debug_set key 23: 0 = the full FFN launches stream their weights through the wave-private LDS slabs (A/B)
// masr_debug_set key 20 -- EXPLORATORY precision mode, never the contract path: the big offline GEMMs (conv2, embed projection,
// the two FFN GEMMs, unfused) run as split-bf16 products on the bf16 matrix pipe (gemm_bf16x3.hip)
static int g_bf16x3 = 0;

void gemm(masr_engine* e, hipStream_t s, const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
          int M, int N, int K, int act, float alpha, const float* R, int ldr, int kind = PROF_GEMM,
          const int* lens = nullptr, int mask_tp = 0) {
    GemmArgs a{};
    a.A = A; a.W = W; a.bias = bias; a.C = C; a.R = R; a.lens = lens;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr;
    a.act = act; a.alpha = alpha; a.mask_tp = mask_tp;
    ProfScope ps(e, s, kind, 2.0 * M * (double)N * K);
    if (g_bf16x3 && launch_gemm_bf16x3(a, A_PLAIN, s)) return;
    launch_gemm(a, A_PLAIN, EPI_STD, s);
}

static int g_rowgemm_packed = 1;   // masr_debug_set key 25: 0 = the offline out-proj + pw1 chain and the CTC head stream their weights through LDS slabs (A/B)
// fragment-ordered copy of a [N, 256] weight matrix (rows padded to a multiple of 256 with zeros), built on first use and kept
const float* packed_rows_of(masr_engine* e, const float* W, int N, hipStream_t s) {
    auto it = e->ffn_packed.find(W);
    if (it == e->ffn_packed.end()) {
        const int Np = (N + 255) / 256 * 256;
        std::pair<DevBuf, DevBuf> pk;
        if (pk.first.ensure((size_t)Np * 256 * sizeof(float))) return nullptr;
        launch_pack_rows_pc(W, pk.first.as<float>(), Np, s, N);
        it = e->ffn_packed.emplace(W, pk).first;
    }
    return it->second.first.as<float>();
}

// fragment-order copies of an FFN's two weight matrices (ffn_pc.hip layout), cached per W1 pointer
int packed_ffn_of(masr_engine* e, const float* w1, const float* w2, hipStream_t s, const float** p1, const float** p2) {
    const int d = e->cfg.d_model;
    auto it = e->ffn_packed.find(w1);
    if (it == e->ffn_packed.end()) {
        std::pair<DevBuf, DevBuf> pk;
        CHK(pk.first.ensure((size_t)e->cfg.d_ff * d * sizeof(float)));
        CHK(pk.second.ensure((size_t)e->cfg.d_ff * d * sizeof(float)));
        launch_pack_ffn_pc(w1, w2, pk.first.as<float>(), pk.second.as<float>(), e->cfg.d_ff, s);
        it = e->ffn_packed.emplace(w1, pk).first;
    }
    *p1 = it->second.first.as<float>();
    *p2 = it->second.second.as<float>();
    return 0;
}
// the fused Squeezeformer stages cover these sizes (launch_sqz_stage's own checks, sqz_layer.hip): anything else takes the separate launches
static bool sqz_stage_supported(const masr_engine* e) {
    const int K = e->cfg.cnn_kernel;
    return e->cfg.d_model == 256 && e->cfg.d_ff % 128 == 0 && e->cfg.d_ff >= 256 && (K == 31 || K == 15);
}

void rowgemm(masr_engine* e, hipStream_t s, int pro, int epi, const float* A, int lda, const float* lnw,
             const float* lnb, const float* W, const float* bias, float* C, int ldc, int M, int N, const float* R,
             int ldr, float alpha, const int* lens, int mask_tp, int seq_t, int pad, int* out_idx, float* out_maxp,
             int kind = PROF_GEMM, int mstride = 4, int out_seq_t = 0, int out_pad_l = 0, int out_pad_tot = 0,
             int plane_cols = 0, long plane_stride = 0, int a_seq_t = 0, int a_seq_stride = 0, const AttSeq* kv_seqs = nullptr,
             int kv_tq = 0) {
    RowGemmArgs a{};
    a.A = A; a.lda = lda; a.lnw = lnw; a.lnb = lnb; a.W = W; a.bias = bias; a.C = C; a.ldc = ldc; a.M = M; a.N = N;
    a.R = R; a.ldr = ldr; a.alpha = alpha; a.lens = lens; a.mask_tp = mask_tp; a.seq_t = seq_t; a.pad = pad;
    a.out_idx = out_idx; a.out_maxp = out_maxp; a.eps = 1e-5f; a.mstride = mstride;
    a.out_seq_t = out_seq_t; a.out_pad_l = out_pad_l; a.out_pad_tot = out_pad_tot;
    a.plane_cols = plane_cols; a.plane_stride = plane_stride; a.a_seq_t = a_seq_t; a.a_seq_stride = a_seq_stride;
    a.kv_seqs = kv_seqs; a.kv_tq = kv_tq;
    // full row-block launches (the K-split kernel of few row blocks reads W itself) take the packed copy of their weights
    if (g_rowgemm_packed && M >= 112 * 32 && pro != RG_PRO_HIST && pro != RG_PRO_DWCONV && (epi == RG_EPI_CTC || N % 256 == 0))
        a.Wp = packed_rows_of(e, W, N, s);
    ProfScope ps(e, s, kind, 2.0 * M * (double)N * 256);
    launch_rowgemm(a, pro, epi, s);
}

}  // namespace

namespace {
// fbank tables (float32 arithmetic mirrors torchaudio's get_mel_banks / povey window); they do not
// depend on the model, so a weight-less engine can already run the feature front-end.
int build_fbank_tables(masr_engine* e) {
    {
        std::vector<float> win(400), tw512(2 * 257), melw((size_t)80 * 257, 0.f);
        std::vector<int> lo(80), hi(80);
        for (int i = 0; i < 400; ++i) {
            const double hann = 0.5 - 0.5 * cos(2.0 * M_PI * i / 399.0);
            win[i] = (float)pow(hann, 0.85);
        }
        for (int k = 0; k <= 256; ++k) {
            tw512[2 * k] = (float)cos(-2.0 * M_PI * k / 512.0);
            tw512[2 * k + 1] = (float)sin(-2.0 * M_PI * k / 512.0);
        }
        const double mel_low = 1127.0 * log(1.0 + 20.0 / 700.0), mel_high = 1127.0 * log(1.0 + 8000.0 / 700.0);
        const float delta = (float)((mel_high - mel_low) / 81.0), mlow = (float)mel_low;
        for (int m = 0; m < 80; ++m) {
            const float left = mlow + (float)m * delta, center = mlow + ((float)m + 1.0f) * delta,
                        right = mlow + ((float)m + 2.0f) * delta;
            lo[m] = 257;
            hi[m] = 0;
            for (int k = 0; k < 256; ++k) {
                const float freq = 31.25f * (float)k;
                const float mel = 1127.0f * logf(1.0f + freq / 700.0f);
                const float upv = (mel - left) / (center - left), down = (right - mel) / (right - center);
                const float wv = fmaxf(0.f, fminf(upv, down));
                melw[(size_t)m * 257 + k] = wv;
                if (wv > 0.f) {
                    lo[m] = std::min(lo[m], k);
                    hi[m] = std::max(hi[m], k + 1);
                }
            }
            if (lo[m] > hi[m]) lo[m] = hi[m] = 0;
        }
        // transposed weights [tap][filter] for the register-FFT kernel (tap i of filter m = bin lo[m] + i, zero past hi[m])
        constexpr int TAPS = 16;
        std::vector<float> melwt((size_t)TAPS * 80, 0.f);
        for (int m = 0; m < 80; ++m) {
            if (hi[m] - lo[m] > TAPS) return fail("fbank tables: a mel filter is wider than 16 bins");
            for (int i = 0; i < hi[m] - lo[m]; ++i) melwt[(size_t)i * 80 + m] = melw[(size_t)m * 257 + lo[m] + i];
        }
        // twiddles of the Stockham radix-4 passes Ns = 4, 16, 64 of the 256-point FFT, one row per lane j:
        // [pass][r - 1][j] = exp(-2 pi i r (j mod Ns) / (4 Ns)), r = 1..3
        std::vector<float> twr4((size_t)3 * 3 * 64 * 2);
        for (int p = 0; p < 3; ++p) {
            const int Ns = p == 0 ? 4 : p == 1 ? 16 : 64;
            for (int r = 1; r <= 3; ++r)
                for (int j = 0; j < 64; ++j) {
                    const double ang = -2.0 * M_PI * (double)(r * (j % Ns)) / (double)(4 * Ns);
                    twr4[(((size_t)p * 3 + (r - 1)) * 64 + j) * 2] = (float)cos(ang);
                    twr4[(((size_t)p * 3 + (r - 1)) * 64 + j) * 2 + 1] = (float)sin(ang);
                }
        }
        CHK(upload(e, win, &e->window));
        CHK(upload(e, tw512, &e->tw512));
        CHK(upload(e, melwt, &e->melwt));
        CHK(upload(e, twr4, &e->twr4));
        CHK(upload(e, lo, &e->mel_lo));

    }
    return 0;
}
}  // namespace

// ---- side streams (masr_side_stream) ------------------------------------------------------------------------------------------
// The library owns the streams its callers run beside the main stream: two for the prefix searches of consecutive passes, one for
// the per-pass preparation, one for copies.  ONE SET PER DEVICE, created with the FIRST engine on that device -- a fixed, early place
// in the process's history -- and shared by every engine there; non-blocking, default priority.  Rounds 3-5 borrowed
// torch.cuda.Stream()s created wherever a predictor first needed one: the HIP runtime deals streams onto a small pool of hardware
// queues (GPU_MAX_HW_QUEUES, default 4) at creation, and in a process that had already created its share a search stream landed on
// the main stream's queue and ran BEHIND the next encoder pass (BASELINE configs[2]: 63 vs 46 ms per call by process history,
// repaired then by raising GPU_MAX_HW_QUEUES from masr_amd/__init__.py).  Creating them early (the first half of round 6) only moved
// the dependence: the runtime deals queues round-robin in creation order, whatever was created BEFORE the first engine shifts the
// deal -- after torch.distributed had initialised RCCL (one stream) search stream 0 and the second lane's stream shared the
// NULL stream's queue (configs[2] sharpened 30.0 instead of 23.2 ms per call), and in a cold process the preparation stream did
// (every upload of a pass issued beside a running encoder started 4 - 6 ms late).  So the set is CHOSEN BY PROBING: candidate
// streams are created until three hardware queues other than the NULL stream's have been seen (two streams share a queue when an
// empty kernel on one waits for a spinning kernel on the other; ~0.3 ms per probe, at most 16 candidates, once per device):
//   queue B: search stream 0 and the second lane (never used together)     queue C: search stream 1
//   queue D: preparation and copies (short work only)                      the NULL stream's queue: nobody
// With fewer queues than that (GPU_MAX_HW_QUEUES < 4) the roles double up in that order.  MASR_SIDE_DEBUG=1 prints the choice.
// Measured and rejected (round 6, configs[2] sharpened head, passes of 32, ms per call): highest priority 28.9, lowest 28.8,
// default 24.0 -- a queue of another priority class has a hardware-queue pool of its own (never aliased), but every launch beside
// it pays for it (the same call WITHOUT its search kernels: 22.3 against 19.7 ms).  MASR_SIDE_PRIORITY=high|low for the A/B.
// hipExtStreamCreateWithCUMask (a dedicated queue and a CU reservation) can only create streams that synchronise with the NULL
// stream, which is torch's default stream: every search would serialise with the encoder it is meant to run beside.
struct SideStreams {
    hipStream_t s[MASR_SIDE_STREAMS] = {};
    bool made = false;
};
static std::mutex g_side_mu;
static std::map<int, SideStreams> g_side;
static int side_streams_of(int dev, SideStreams** out) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStreams& ss = g_side[dev];
    if (!ss.made) {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char* pe = getenv("MASR_SIDE_PRIORITY");          // A/B only
        const int prio = (pe && pe[0] == 'h') ? greatest : (pe && pe[0] == 'l') ? least : 0;
        if (pe) fprintf(stderr, "masr side streams: priority %d (device range: least %d .. greatest %d)\n", prio, least, greatest);
        // true when an empty kernel on `b` waits for a kernel that spins on `a`: the two streams share a hardware queue
        auto aliased = [](hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb, bool* out) -> int {
            launch_queue_spin(30000, a);                       // 300 us
            HIPCHK(hipEventRecord(ea, a));
            launch_queue_nop(b);
            HIPCHK(hipEventRecord(eb, b));
            HIPCHK(hipEventSynchronize(eb));
            *out = hipEventQuery(ea) == hipSuccess;            // the spin was over before the empty kernel got through
            (void)hipGetLastError();
            HIPCHK(hipEventSynchronize(ea));
            return 0;
        };
        hipEvent_t ea = nullptr, eb = nullptr;
        HIPCHK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
        HIPCHK(hipDeviceSynchronize());
        std::vector<hipStream_t> cls;                          // one stream per hardware queue seen so far (not the NULL stream's)
        std::vector<hipStream_t> spare;                        // candidates on a queue already represented (kept: destroying a
        const bool dbg = getenv("MASR_SIDE_DEBUG") != nullptr; // stream would hand its slot of the deal to the next one created)
        for (int c = 0; c < 16 && cls.size() < 3; ++c) {
            hipStream_t st = nullptr;
            HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
            bool same = false;
            if (aliased(nullptr, st, ea, eb, &same)) return 1;
            for (size_t k = 0; k < cls.size() && !same; ++k)
                if (aliased(cls[k], st, ea, eb, &same)) return 1;
            (same ? spare : cls).push_back(st);
        }
        (void)hipEventDestroy(ea);
        (void)hipEventDestroy(eb);
        if (cls.empty()) {                                     // one hardware queue in all: everything shares it
            hipStream_t st = nullptr;
            if (!spare.empty()) st = spare.back();
            else HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
            cls.push_back(st);
        }
        hipStream_t qB = cls[0], qC = cls[cls.size() > 1 ? 1 : 0], qD = cls[cls.size() > 2 ? 2 : cls.size() - 1];
        // roles that share a queue still get streams of their own where a spare on that queue exists (ordering stays per role)
        ss.s[0] = qB;
        ss.s[1] = qC;
        ss.s[2] = qD;
        ss.s[3] = qD;
        ss.s[4] = qB;
        if (dbg)
            fprintf(stderr, "masr side streams (device %d): %zu hardware queues besides the NULL stream's after %zu candidates; search 0 / "
                    "lane 1 = %p, search 1 = %p, preparation / copy = %p\n", dev, cls.size(), cls.size() + spare.size(), (void*)qB,
                    (void*)qC, (void*)qD);
        ss.made = true;
    }
    *out = &ss;
    return 0;
}

extern "C" {

const char* masr_last_error(void) { return g_err.c_str(); }
int masr_version(void) { return 1; }

int masr_create(const masr_config* cfg, masr_engine** out) {
    if (!cfg || !out) return fail("null argument");
    if (cfg->model_kind < 0 || cfg->model_kind > 3)
        return fail("model_kind must be 0 (conformer), 1 (squeezeformer, non-streaming), 2 (efficient_conformer) or 3 (deepspeech2)");
    // n_mels = input feature size of the model: 80 (fbank), n_mfcc (mfcc) or 161 (linear) -- audio_featurizer.py:141-154
    if (cfg->n_mels < 7 || cfg->n_mels > 512) return fail("input feature size (n_mels) must be in [7, 512]");
    if (cfg->model_kind == 3) {
        if (cfg->d_model != 1024) return fail("deepspeech2: the LSTM step kernel is specialised for rnn_size=1024");
        if (cfg->num_blocks <= 0) return fail("deepspeech2: num_rnn_layers must be positive");
    } else if (cfg->d_model != 256 || cfg->heads != 4) {
        return fail("kernels are specialised for d_model=256, heads=4");
    } else if (cfg->d_ff % 128) {
        return fail("unsupported d_ff");
    }
    if (cfg->model_kind == 3) {
    } else if (cfg->model_kind == 0 || cfg->model_kind == 2) {
        if (cfg->cnn_kernel != 15) return fail("conformer: cnn_module_kernel must be 15");
    } else {
        if (cfg->cnn_kernel != 31) return fail("squeezeformer: cnn_module_kernel must be 31");
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("no HIP device");
    HIPCHK(hipSetDevice(cfg->device_id));
    {
        SideStreams* ss = nullptr;                   // (created with the first engine of the device: a fixed place in the process's history)
        if (side_streams_of(cfg->device_id, &ss)) return 1;
    }
    masr_engine* e = new masr_engine();
    e->cfg = *cfg;
    if (e->cfg.max_pos <= 0) e->cfg.max_pos = 5000;
    e->reduce_idx = cfg->model_kind == 1 ? cfg->reserved[0] : -1;
    e->conv_bn = cfg->model_kind == 0 && cfg->reserved[0] == 1;
    e->recover_idx = cfg->model_kind == 1 ? cfg->reserved[1] : -1;
    if (cfg->model_kind == 2) {
        e->stride_idx = cfg->reserved[0];
        e->n_group_layers = cfg->reserved[1];
        e->group_size = cfg->reserved[2];
        if (e->group_size != 3) { delete e; return fail("efficient_conformer: group_size must be 3"); }
    }
    if (build_fbank_tables(e)) {
        masr_destroy(e);
        return 1;
    }
    *out = e;
    return 0;
}

void masr_destroy(masr_engine* e) {
    if (!e) return;
    for (void* p : e->owned) (void)hipFree(p);
    e->release_all();
    for (auto& w : e->parked) w.release_all();
    e->beam_pool.release();
    e->beam_state.release();
    if (e->pack_ev) (void)hipEventDestroy(e->pack_ev);
    for (auto& s : e->streams) {
        s.att.release();
        s.cnn.release();
        s.cnn2.release();
    }
    for (auto& g : e->gbeams) {
        g.pool.release();
        g.state.release();
    }
    for (auto& kv : e->beam_ws) {
        kv.second.first.release();
        kv.second.second.release();
    }
    for (auto& kv : e->ms_ws) kv.second.release();
    for (auto& kv : e->x3_packed) {
        kv.second.first.release();
        kv.second.second.release();
    }
    for (auto& kv : e->ffn_packed) {
        kv.second.first.release();
        kv.second.second.release();
    }
    for (auto& kv : e->ffn_dual_packed) {
        kv.second.first.release();
        kv.second.second.release();
    }
    for (auto& ev : e->prof_events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    delete e;
}

// The fragment-ordered weight copies (ffn_packed / ffn_dual_packed / x3_packed) are built on first use and keyed by the device
// pointer of the weights they were packed from.  A reload (masr_load_tensor on a finalized engine, then masr_finalize) re-uploads
// into the SAME device buffers when the sizes are unchanged, so the keys would still match while the copies hold the old
// values: every reload drops them (after the device has drained: launches in flight may still read them).
static void drop_packed_weights(masr_engine* e) {
    if (e->ffn_packed.empty() && e->ffn_dual_packed.empty() && e->x3_packed.empty()) return;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (auto* m : {&e->ffn_packed, &e->ffn_dual_packed, &e->x3_packed}) {
        for (auto& kv : *m) {
            kv.second.first.release();
            kv.second.second.release();
        }
        m->clear();
    }
}

int masr_load_tensor(masr_engine* e, const char* name, const float* host, const int64_t* shape, int32_t ndim) {
    if (!e || !name || !host) return fail("null argument");
    std::string n(name);
    if (n.rfind("encoder.", 0) != 0 && n.rfind("ctc.", 0) != 0 && n.rfind("decoder.ctc_lo.", 0) != 0 && n != "__pos_table__")
        return 0;
    drop_packed_weights(e);
    HostTensor t;
    int64_t cnt = 1;
    for (int i = 0; i < ndim; ++i) {
        t.shape.push_back(shape[i]);
        cnt *= shape[i];
    }
    t.v.assign(host, host + cnt);
    e->host[n] = std::move(t);
    e->finalized = false;
    return 0;
}

static int finalize_squeezeformer(masr_engine* e, hipStream_t s);
static int finalize_ds2(masr_engine* e);
static int enc_dim(const masr_engine* e) {      // width of the encoder output rows
    return e->cfg.model_kind == 3 ? e->cfg.d_model * (e->cfg.causal ? 1 : 2) : e->cfg.d_model;
}
static int layer_kernel(const masr_engine* e, int i) {          // efficient conformer: kernel // stride after the stride layer
    return (e->cfg.model_kind == 2 && e->stride_idx >= 0 && i > e->stride_idx) ? e->cfg.cnn_kernel / 2 : e->cfg.cnn_kernel;
}
static bool layer_grouped(const masr_engine* e, int i) { return e->cfg.model_kind == 2 && i < e->n_group_layers; }

int masr_finalize(masr_engine* e, void* stream) {
    if (!e) return fail("null engine");
    if (e->cfg.model_kind == 1) return finalize_squeezeformer(e, (hipStream_t)stream);
    if (e->cfg.model_kind == 3) return finalize_ds2(e);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(e->cfg.device_id));
    const int d = e->cfg.d_model, dff = e->cfg.d_ff, L = e->cfg.num_blocks, V = e->cfg.vocab_size, F = e->cfg.n_mels;
    const int F1 = (F - 1) / 2, F2 = (F1 - 1) / 2, K = e->cfg.cnn_kernel, H = e->cfg.heads, dk = d / H;
    const HostTensor* t;
    CHK(up(e, "encoder.global_cmvn.mean", {F}, &e->cmvn_mean));
    CHK(up(e, "encoder.global_cmvn.istd", {F}, &e->cmvn_istd));
    {   // conv1 [d,1,3,3] -> [9][d]
        CHK(get(e, "encoder.embed.conv.0.weight", {d, 1, 3, 3}, &t));
        std::vector<float> w(9 * d);
        for (int c = 0; c < d; ++c)
            for (int k = 0; k < 9; ++k) w[k * d + c] = t->v[c * 9 + k];
        CHK(upload(e, w, &e->conv1_w));
        CHK(up(e, "encoder.embed.conv.0.bias", {d}, &e->conv1_b));
    }
    {   // conv2 [co,ci,3,3] -> [co][ci / 32][kh*3+kw][ci % 32]: the implicit GEMM walks the nine window positions of one
        // 32-channel block in consecutive K slabs, so the overlapping input columns of neighbouring positions (kw = 2 of one
        // output, kw = 0 of the next) are re-read a slab or two later -- out of L1 / L2 instead of HBM
        CHK(get(e, "encoder.embed.conv.2.weight", {d, d, 3, 3}, &t));
        std::vector<float> w((size_t)d * 9 * d);
        for (int co = 0; co < d; ++co)
            for (int ci = 0; ci < d; ++ci)
                for (int k = 0; k < 9; ++k)
                    w[(size_t)co * 9 * d + (size_t)(ci / 32) * 9 * 32 + k * 32 + ci % 32] = t->v[((size_t)co * d + ci) * 9 + k];
        CHK(upload(e, w, &e->conv2_w));
        CHK(up(e, "encoder.embed.conv.2.bias", {d}, &e->conv2_b));
    }
    {   // embed.out.0 [d][c*F2+f] -> [d][f*d + c]   (subsampling.py:110 flattens channel-major)
        CHK(get(e, "encoder.embed.out.0.weight", {d, (int64_t)d * F2}, &t));
        std::vector<float> w((size_t)d * d * F2);
        for (int o = 0; o < d; ++o)
            for (int c = 0; c < d; ++c)
                for (int f = 0; f < F2; ++f) w[(size_t)o * d * F2 + f * d + c] = t->v[(size_t)o * d * F2 + c * F2 + f];
        CHK(upload(e, w, &e->embed_w));
        CHK(up(e, "encoder.embed.out.0.bias", {d}, &e->embed_b));
    }
    {   // positional table (conformer/embedding.py:31-37)
        auto it = e->host.find("__pos_table__");
        if (it != e->host.end()) {
            if ((int64_t)it->second.v.size() != (int64_t)e->cfg.max_pos * d) return fail("__pos_table__ has wrong size");
            CHK(upload(e, it->second.v, &e->pe));
        } else {
            std::vector<float> pe((size_t)e->cfg.max_pos * d);
            for (int i = 0; i < d; i += 2) {
                const float div = expf((float)i * (float)(-(log(10000.0) / d)));
                for (int p = 0; p < e->cfg.max_pos; ++p) {
                    pe[(size_t)p * d + i] = sinf((float)p * div);
                    pe[(size_t)p * d + i + 1] = cosf((float)p * div);
                }
            }
            CHK(upload(e, pe, &e->pe));
        }
    }
    e->layers.assign(L, LayerW{});
    for (int i = 0; i < L; ++i) {
        LayerW& w = e->layers[i];
        const std::string p = "encoder.encoders." + std::to_string(i) + ".";
        auto ln = [&](const std::string& n, float** ww, float** bb) -> int {
            CHK(up(e, p + n + ".weight", {d}, ww));
            return up(e, p + n + ".bias", {d}, bb);
        };
        CHK(ln("norm_ff_macaron", &w.ln_ffm_w, &w.ln_ffm_b));
        CHK(ln("norm_mha", &w.ln_mha_w, &w.ln_mha_b));
        CHK(ln("norm_conv", &w.ln_conv_w, &w.ln_conv_b));
        CHK(ln("norm_ff", &w.ln_ff_w, &w.ln_ff_b));
        CHK(ln("norm_final", &w.ln_fin_w, &w.ln_fin_b));
        if (e->conv_bn) {
            // cnn_module_norm: batch_norm (conformer/convolution.py:60-67): eval-mode BatchNorm1d folded into y = x * scale + shift
            const HostTensor *tw, *tb, *tm, *tv;
            CHK(get(e, p + "conv_module.norm.weight", {d}, &tw));
            CHK(get(e, p + "conv_module.norm.bias", {d}, &tb));
            CHK(get(e, p + "conv_module.norm.running_mean", {d}, &tm));
            CHK(get(e, p + "conv_module.norm.running_var", {d}, &tv));
            std::vector<float> sc(d), sh(d);
            for (int c = 0; c < d; ++c) {
                sc[c] = tw->v[c] / sqrtf(tv->v[c] + 1e-5f);
                sh[c] = tb->v[c] - tm->v[c] * sc[c];
            }
            CHK(upload(e, sc, &w.cln_w));
            CHK(upload(e, sh, &w.cln_b));
        } else {
            CHK(ln("conv_module.norm", &w.cln_w, &w.cln_b));
        }
        CHK(up(e, p + "feed_forward_macaron.w_1.weight", {dff, d}, &w.ffm_w1));
        CHK(up(e, p + "feed_forward_macaron.w_1.bias", {dff}, &w.ffm_b1));
        CHK(up(e, p + "feed_forward_macaron.w_2.weight", {d, dff}, &w.ffm_w2));
        CHK(up(e, p + "feed_forward_macaron.w_2.bias", {d}, &w.ffm_b2));
        CHK(up(e, p + "feed_forward.w_1.weight", {dff, d}, &w.ff_w1));
        CHK(up(e, p + "feed_forward.w_1.bias", {dff}, &w.ff_b1));
        CHK(up(e, p + "feed_forward.w_2.weight", {d, dff}, &w.ff_w2));
        CHK(up(e, p + "feed_forward.w_2.bias", {d}, &w.ff_b2));
        {   // fused QKV projection [3d, d]
            std::vector<float> wq((size_t)3 * d * d), bq(3 * d);
            const char* nm[3] = {"linear_q", "linear_k", "linear_v"};
            for (int j = 0; j < 3; ++j) {
                CHK(get(e, p + "self_attn." + nm[j] + ".weight", {d, d}, &t));
                memcpy(&wq[(size_t)j * d * d], t->v.data(), sizeof(float) * d * d);
                CHK(get(e, p + "self_attn." + nm[j] + ".bias", {d}, &t));
                memcpy(&bq[j * d], t->v.data(), sizeof(float) * d);
            }
            CHK(upload(e, wq, &w.wqkv));
            CHK(upload(e, bq, &w.bqkv));
        }
        CHK(up(e, p + "self_attn.linear_out.weight", {d, d}, &w.wo));
        CHK(up(e, p + "self_attn.linear_out.bias", {d}, &w.bo));
        CHK(up(e, p + "self_attn.linear_pos.weight", {d, d}, &w.wpos));
        const int gk = layer_grouped(e, i) ? dk * e->group_size : dk;
        CHK(up(e, p + "self_attn.pos_bias_u", {H, gk}, &w.pos_u));
        CHK(up(e, p + "self_attn.pos_bias_v", {H, gk}, &w.pos_v));
        CHK(up(e, p + "conv_module.pointwise_conv1.weight", {2 * d, d, 1}, &w.pw1_w));   // rows: value c, gate d + c
        CHK(up(e, p + "conv_module.pointwise_conv1.bias", {2 * d}, &w.pw1_b));
        {   // [Wo; W_pw1] and [bo; b_pw1]: one continuous weight stream for the fused kernel of the offline path
            const HostTensor *a, *b, *c2, *d2;
            CHK(get(e, p + "self_attn.linear_out.weight", {d, d}, &a));
            CHK(get(e, p + "self_attn.linear_out.bias", {d}, &b));
            CHK(get(e, p + "conv_module.pointwise_conv1.weight", {2 * d, d, 1}, &c2));
            CHK(get(e, p + "conv_module.pointwise_conv1.bias", {2 * d}, &d2));
            std::vector<float> cw(a->v), cb(b->v);
            cw.insert(cw.end(), c2->v.begin(), c2->v.end());
            cb.insert(cb.end(), d2->v.begin(), d2->v.end());
            CHK(upload(e, cw, &w.chain_w));
            CHK(upload(e, cb, &w.chain_b));
        }
        {   // glu(bias): what the zero left-padding of the causal conv turns into behind pointwise_conv1 + GLU
            std::vector<float> z(d, 0.f);
            CHK(upload(e, z, &w.gconst));
            launch_glu_const(w.pw1_b, w.gconst, (hipStream_t)stream);
        }
        {   // depthwise [d,1,K_i] -> [K_i][d]
            const int K = layer_kernel(e, i);
            CHK(get(e, p + "conv_module.depthwise_conv.weight", {d, 1, K}, &t));
            std::vector<float> wd((size_t)K * d);
            for (int c = 0; c < d; ++c)
                for (int j = 0; j < K; ++j) wd[(size_t)j * d + c] = t->v[(size_t)c * K + j];
            CHK(upload(e, wd, &w.dw_w));
            CHK(up(e, p + "conv_module.depthwise_conv.bias", {d}, &w.dw_b));
        }
        CHK(up(e, p + "conv_module.pointwise_conv2.weight", {d, d, 1}, &w.pw2_w));
        CHK(up(e, p + "conv_module.pointwise_conv2.bias", {d}, &w.pw2_b));
        {   // positional keys p_j = W_pos * PE(j) for every position, once (attention.py:229-231)
            void* pt = nullptr;
            HIPCHK(hipMalloc(&pt, (size_t)e->cfg.max_pos * d * sizeof(float)));
            e->owned.push_back(pt);
            w.ptab = (float*)pt;
            gemm(e, s, e->pe, d, w.wpos, nullptr, w.ptab, d, e->cfg.max_pos, d, d, ACT_NONE, 1.f, nullptr, 0, PROF_NONE);
        }
    }
    CHK(up(e, "encoder.after_norm.weight", {d}, &e->after_w));
    CHK(up(e, "encoder.after_norm.bias", {d}, &e->after_b));
    CHK(up(e, "ctc.ctc_lo.weight", {V, d}, &e->ctc_w));
    CHK(up(e, "ctc.ctc_lo.bias", {V}, &e->ctc_b));

    HIPCHK(hipStreamSynchronize(s));
    e->host.clear();
    e->finalized = true;
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Encoder orchestration
// ------------------------------------------------------------------------------------------------
namespace {

struct EncodeCtx {
    int nseq;        // sequences (utterances or streams)
    int Tq;          // encoder frames per sequence in this call
    const int* lens; // device feature lengths for pad masking, or nullptr (streaming)
};

// post_*: the LayerNorm that follows the block (y <- LayerNorm(x), y may be x); fused into the split-mode reduction of small M
// tail / tail_done: a row-local stage to run on the finished rows inside the same kernel (fused QKV projection, ffn_pc.hip TAIL);
// *tail_done = true when the kernel did it, otherwise the caller launches it
int ffn(masr_engine* e, hipStream_t s, int M, const float* lnw, const float* lnb, const float* w1, const float* b1,
        const float* w2, const float* b2, float scale = 0.5f, int affine = 0, const float* post_w = nullptr,
        const float* post_b = nullptr, float* post_y = nullptr, const FfnTail* tail = nullptr, bool* tail_done = nullptr,
        const FfnHead* head = nullptr, bool* head_done = nullptr) {
    const int d = e->cfg.d_model, dff = e->cfg.d_ff;
    // few rows (streaming chunk steps): split d_ff across workgroups so that >= ~128 CUs work on the block
    int nsplit = 1;
    const int rowblocks = (M + 31) / 32;
    const FfnPostLn post{post_w, post_b, post_y, 1e-5f};
    if (rowblocks < g_ffn_split_blocks) {
        nsplit = std::min(dff / 128, std::max(1, (rowblocks < 64 ? 128 : 256) / rowblocks));
        CHK(e->ffpart.ensure((size_t)nsplit * M * d * sizeof(float)));
    }
    // exploratory, bit 2 of key 20: the fused split-bf16 FFN (ffn_x3.hip).  (An unfused version -- LayerNorm, then two split-bf16
    // GEMMs with the hidden tensor in HBM -- measured 44 + 82 + 6 us against the 144 us of the fused exact-fp32 kernel at
    // B = 32 x 10 s: the 65 MB round trip of the hidden tensor ate what the bf16 pipe saved.)
    const bool x3 = (g_bf16x3 & 2) && nsplit == 1 && !affine && d == 256 && dff % 128 == 0;
    const bool want_tail = tail && nsplit == 1 && !g_no_ffn_tail && !x3;
    // (few rows: the head stage rides on the d_ff-split launch, every slice repeating it on the row block's rows -- key 30)
    const bool split_head = head && head->glu && nsplit > 1 && g_split_head && !affine && !x3 && head->ktaps == 15 && d == 256 &&
                            g_ffn_packed >= 2;
    const bool want_head = head && head->glu && ((nsplit == 1 && !want_tail && !g_no_ffn_head && !affine && !x3 &&
                                                  (head->ktaps == 15 || head->ktaps == 7)) || split_head);
    if (head_done) *head_done = want_head;
    if (head && head->glu && !want_head) {
        // the rest of the conv module as its own two launches: depthwise conv + LayerNorm + SiLU, pointwise_conv2 + mask + residual
        CHK(e->dwo.ensure((size_t)M * d * sizeof(float)));
        launch_dwconv_ln_silu(head->glu, head->dw_w, head->dw_b, head->lnw, head->lnb, e->dwo.as<float>(), M / head->seq_t,
                              head->seq_t, head->ktaps, 1e-5f, s, head->gconst);
        float* x = e->x.as<float>();
        rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->dwo.as<float>(), d, nullptr, nullptr, head->W, head->bias, x, d, M, d, x, d,
                1.f, head->lens, head->lens ? head->seq_t : 0, 0, 0, nullptr, nullptr, PROF_GEMM, head->mstride);
    }
    if (tail && tail->pre_lnw && !want_tail)          // the deferred LayerNorm of the previous layer, as its own launch
        launch_layernorm(e->x.as<float>(), tail->pre_lnw, tail->pre_lnb, e->x.as<float>(), M, 1e-5f, 0, 0, nullptr, s);
    if (x3) {
        // packed (hi, lo) weights of this FFN, built on first use and kept (keyed by the W1 pointer): + 4 MB per FFN
        auto it = e->x3_packed.find(w1);
        if (it == e->x3_packed.end()) {
            std::pair<DevBuf, DevBuf> pk;
            CHK(pk.first.ensure(ffn_x3_packed_elems(dff) * sizeof(unsigned short)));
            CHK(pk.second.ensure(ffn_x3_packed_elems(dff) * sizeof(unsigned short)));
            launch_pack_ffn_x3(w1, w2, pk.first.as<unsigned short>(), pk.second.as<unsigned short>(), dff, s);
            it = e->x3_packed.emplace(w1, pk).first;
        }
        float* x = e->x.as<float>();
        {
            ProfScope ps(e, s, PROF_FFN1, 4.0 * M * (double)dff * d);
            if (!launch_ffn_x3(x, lnw, lnb, it->second.first.as<unsigned short>(), b1, it->second.second.as<unsigned short>(), b2, M,
                               dff, 1e-5f, scale, s))
                return fail("ffn(): the split-bf16 FFN kernel rejected the sizes");
        }
        if (tail_done) *tail_done = false;
        if (post_y) launch_layernorm(x, post_w, post_b, post_y, M, 1e-5f, 0, 0, nullptr, s);
        return 0;
    }
    // few rows, one chunk of 128 hidden units per workgroup: the kernel in which all eight waves work on both products (key 35)
    if (g_ffn_coop && nsplit > 1 && nsplit == dff / 128 && !want_head && !x3 && d == 256) {
        // packed copies: ffn_pc.hip's (W2 is shared with it) + the 16 x 16 x 4 fragment order of W1 (its own map: keyed by W1)
        auto it = e->ffn_packed.find(w1);
        if (it == e->ffn_packed.end()) {
            std::pair<DevBuf, DevBuf> pk;
            CHK(pk.first.ensure((size_t)dff * d * sizeof(float)));
            CHK(pk.second.ensure((size_t)dff * d * sizeof(float)));
            launch_pack_ffn_pc(w1, w2, pk.first.as<float>(), pk.second.as<float>(), dff, s);
            it = e->ffn_packed.emplace(w1, pk).first;
        }
        auto ic = e->ffn_dual_packed.find(w1 + 1);            // (+ 1: a key of its own next to ffn_dual.hip's entries for the same W1)
        if (ic == e->ffn_dual_packed.end()) {
            std::pair<DevBuf, DevBuf> pk;
            CHK(pk.first.ensure((size_t)dff * d * sizeof(float)));
            launch_pack_ffn_coop_w1(w1, pk.first.as<float>(), dff, s);
            ic = e->ffn_dual_packed.emplace(w1 + 1, pk).first;
        }
        ProfScope psc(e, s, PROF_FFN1, 4.0 * M * (double)dff * d);
        launch_ffn_coop(e->x.as<float>(), lnw, lnb, ic->second.first.as<float>(), b1, it->second.second.as<float>(), M, dff, 1e-5f,
                        affine, e->ffpart.as<float>(), s);
        launch_ffn_reduce(e->x.as<float>(), e->ffpart.as<float>(), b2, M, nsplit, scale, s, post_y ? &post : nullptr);
        if (tail_done) *tail_done = false;
        return 0;
    }
    ProfScope ps(e, s, want_tail ? PROF_FFN_TAIL : want_head ? PROF_FFN_HEAD : PROF_FFN1,
                 4.0 * M * (double)dff * d + (want_tail ? 2.0 * M * (double)tail->N * d : 0.0) + (want_head ? 2.0 * M * (double)d * d : 0.0));
    // full launches stream PACKED weight copies straight into registers (ffn_pc.hip VAR == 2; built on first use, + 4 MB per FFN)
    const float *kw1 = w1, *kw2 = w2;
    const bool packed = g_ffn_packed && (nsplit == 1 || g_ffn_packed >= 2) && d == 256;      // (key 23 = 2: the d_ff-split launches of small M too)
    if (packed) {
        auto it = e->ffn_packed.find(w1);
        if (it == e->ffn_packed.end()) {
            std::pair<DevBuf, DevBuf> pk;
            CHK(pk.first.ensure((size_t)dff * d * sizeof(float)));
            CHK(pk.second.ensure((size_t)dff * d * sizeof(float)));
            launch_pack_ffn_pc(w1, w2, pk.first.as<float>(), pk.second.as<float>(), dff, s);
            it = e->ffn_packed.emplace(w1, pk).first;
        }
        kw1 = it->second.first.as<float>();
        kw2 = it->second.second.as<float>();
    }
    // ... and so do the row-local stages that ride on the launch (QKV tail, pointwise_conv2 head)
    FfnTail ptail{};
    FfnHead phead{};
    auto packed_rows = [&](const float* W, int N, const float** out) -> int {
        auto it = e->ffn_packed.find(W);
        if (it == e->ffn_packed.end()) {
            std::pair<DevBuf, DevBuf> pk;
            CHK(pk.first.ensure((size_t)N * d * sizeof(float)));
            launch_pack_rows_pc(W, pk.first.as<float>(), N, s);
            it = e->ffn_packed.emplace(W, pk).first;
        }
        *out = it->second.first.as<float>();
        return 0;
    };
    // two accumulator chains per wave (ffn_dual.hip): same arithmetic in the same order, its own packing order
    if (packed && nsplit == 1 && g_ffn_dual && dff % 256 == 0 && dff >= 512 && !(want_tail && tail->N != 768) && !(affine && (want_tail || want_head))) {
        auto it = e->ffn_dual_packed.find(w1);
        if (it == e->ffn_dual_packed.end()) {
            std::pair<DevBuf, DevBuf> pk;
            CHK(pk.first.ensure((size_t)dff * d * sizeof(float)));
            CHK(pk.second.ensure((size_t)dff * d * sizeof(float)));
            launch_pack_ffn_dual(w1, w2, pk.first.as<float>(), pk.second.as<float>(), dff, s);
            it = e->ffn_dual_packed.emplace(w1, pk).first;
        }
        if (want_tail) {
            ptail = *tail;
            auto tw = e->ffn_dual_packed.find(tail->W);
            if (tw == e->ffn_dual_packed.end()) {
                std::pair<DevBuf, DevBuf> pk;
                CHK(pk.first.ensure((size_t)768 * d * sizeof(float)));
                launch_pack_rows_dual(tail->W, pk.first.as<float>(), s);
                tw = e->ffn_dual_packed.emplace(tail->W, pk).first;
            }
            ptail.W = tw->second.first.as<float>();
        }
        if (want_head) {
            phead = *head;
            CHK(packed_rows(head->W, d, &phead.W));
        }
        const int done = launch_ffn_dual(e->x.as<float>(), lnw, lnb, it->second.first.as<float>(), b1, it->second.second.as<float>(),
                                         b2, M, dff, 1e-5f, scale, affine, s, want_tail ? &ptail : nullptr, want_head ? &phead : nullptr);
        if (done < 0) return fail("ffn(): the two-chain FFN kernel rejected the launch");
        if (want_head && done != 4) return fail("ffn(): head stage was not launched");
        if (tail_done) *tail_done = done == 2;
        if (post_y) launch_layernorm(e->x.as<float>(), post_w, post_b, post_y, M, 1e-5f, 0, 0, nullptr, s);
        return 0;
    }
    if (packed && want_tail && tail->N % 256 == 0) {
        ptail = *tail;
        CHK(packed_rows(tail->W, tail->N, &ptail.W));
        tail = &ptail;
    }
    if (packed && want_head) {
        phead = *head;
        CHK(packed_rows(head->W, d, &phead.W));
        if (split_head) {
            CHK(e->xh.ensure((size_t)M * d * sizeof(float)));
            phead.xout = e->xh.as<float>();
        }
        head = &phead;
    }
    const int done = launch_ffn_fused(e->x.as<float>(), lnw, lnb, kw1, b1, kw2, b2, M, dff, 1e-5f, scale, affine,
                                      nsplit > 1 ? e->ffpart.as<float>() : nullptr, nsplit, s, post_y ? &post : nullptr,
                                      want_tail ? tail : nullptr, want_head ? head : nullptr, packed);
    if (done < 0) return fail("ffn(): launch rejected");
    if (want_head && !(done & 4)) return fail("ffn(): head stage was not launched");
    if (tail_done) *tail_done = done == 2;
    if (post_y && !(done & 1)) launch_layernorm(e->x.as<float>(), post_w, post_b, post_y, M, 1e-5f, 0, 0, nullptr, s);
    return 0;
}

static int g_hot_weights = 0;      // masr_debug_set key 19 (timing experiment only): every chunk-step layer runs on layer 0's weights
static int g_embed_split = 1;      // masr_debug_set key 15: 0 = the offline embed projection never splits K
// feats [nseq, T, 80] -> x [nseq*Tq, d] (embed incl. x*sqrt(d))
int embed(masr_engine* e, hipStream_t s, const float* feats, int nseq, int T, int* Tq_out, const int* skip_lens = nullptr) {
    const int d = e->cfg.d_model, F = e->cfg.n_mels, F1 = (F - 1) / 2, F2 = (F1 - 1) / 2;
    const int T1 = (T - 1) / 2, Tq = (T1 - 1) / 2;
    if (T < 7 || Tq <= 0) return fail("input too short for Conv2dSubsampling4 (need >= 7 frames)");
    const int M = nseq * Tq;
    CHK(e->x1.ensure((size_t)nseq * T1 * F1 * d * sizeof(float)));
    CHK(e->x2.ensure((size_t)M * F2 * d * sizeof(float)));
    CHK(e->x.ensure((size_t)M * d * sizeof(float)));
    launch_conv1(feats, e->cmvn_mean, e->cmvn_istd, e->conv1_w, e->conv1_b, e->x1.as<float>(), nseq, T, F, d, s);
    {
        GemmArgs a{};
        a.A = e->x1.as<float>(); a.W = e->conv2_w; a.bias = e->conv2_b; a.C = e->x2.as<float>();
        a.M = M * F2; a.N = d; a.K = 9 * d; a.ldc = d; a.act = ACT_RELU; a.alpha = 1.f;
        a.T1 = T1; a.F1 = F1; a.T2 = Tq; a.F2 = F2; a.Cc = d;
        if (skip_lens) { a.lens = skip_lens; a.skip_rps = Tq * F2; a.skip_div = F2; }      // tiles of padded frames only: not computed
        ProfScope ps(e, s, PROF_CONV2, 2.0 * a.M * (double)a.N * a.K);
        const int tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
        if (g_bf16x3 && tiles >= 640 && launch_gemm_bf16x3(a, A_CONV2, s)) {
            // exploratory split-bf16 mode
        } else if (tiles < 640) {
            // streaming chunk steps: ~1 workgroup of 4 waves per CU leaves the load -> LDS -> MFMA chain of every 32-wide K slab
            // exposed (2.5 us per slab, 72 slabs); split K so that ~4-5 workgroups per CU overlap each other's latencies
            const int nsplit = std::min(8, std::max(2, 1280 / tiles));
            CHK(e->ffpart.ensure((size_t)nsplit * a.M * a.N * sizeof(float)));
            launch_gemm_splitk(a, e->ffpart.as<float>(), nsplit, s, A_CONV2);
        } else {
            launch_gemm(a, A_CONV2, EPI_STD, s);
        }
    }
    {   // Conformer: (W.x + b) * sqrt(d)  (embedding.py:97);  Squeezeformer: W.(x * sqrt(d)) + b  (subsampling.py:72-75)
        GemmArgs a{};
        a.A = e->x2.as<float>(); a.lda = F2 * d; a.W = e->embed_w; a.bias = e->embed_b; a.C = e->x.as<float>(); a.ldc = d;
        a.M = M; a.N = d; a.K = F2 * d; a.act = ACT_NONE; a.alpha = sqrtf((float)d);
        a.bias_after_alpha = e->cfg.model_kind == 1 ? 1 : 0;
        if (skip_lens) { a.lens = skip_lens; a.skip_rps = Tq; a.skip_div = 1; }
        ProfScope ps(e, s, PROF_GEMM, 2.0 * M * (double)d * F2 * d);
        const int tiles = ((M + 63) / 64) * ((d + 63) / 64);
        const long wide = (long)((M + 63) / 64) * ((d + 127) / 128);      // 64x128 tiles of the unsplit launch
        const long t128 = (long)((M + 127) / 128) * ((d + 127) / 128);
        if (g_bf16x3 && tiles >= 128 && launch_gemm_bf16x3(a, A_PLAIN, s)) {
            // exploratory split-bf16 mode: 5x less matrix-pipe time, no K split needed
        } else if (g_embed_split && tiles >= 128 && t128 >= 100 && t128 <= 128) {
            // B = 32 x 10 s: 124 tiles of 128x128 -- four K quarters on 8-wave workgroups = 496 workgroups, two per CU, four
            // waves per SIMD: 163 + 11 us (GEMM + reduction) against 180 + 9 us for two K halves on 64x128 tiles and 203 us unsplit
            CHK(e->ffpart.ensure((size_t)4 * M * d * sizeof(float)));
            launch_gemm_splitk(a, e->ffpart.as<float>(), 4, s);
        } else if (g_embed_split && tiles >= 128 && wide >= 200 && wide <= 320) {
            // about one 4-wave workgroup per CU: two K halves put two waves on every SIMD
            CHK(e->ffpart.ensure((size_t)2 * M * d * sizeof(float)));
            launch_gemm_splitk(a, e->ffpart.as<float>(), 2, s);
        } else if (tiles < 128) {              // few rows, K = 4864: split K so that ~256 workgroups share the weight stream
            const int nsplit = std::min(F2 * d / 32, std::max(2, 256 / tiles));
            CHK(e->ffpart.ensure((size_t)nsplit * M * d * sizeof(float)));
            launch_gemm_splitk(a, e->ffpart.as<float>(), nsplit, s);
        } else {
            launch_gemm(a, A_PLAIN, EPI_STD, s);
        }
    }
    *Tq_out = Tq;
    return 0;
}

int ensure_layer_ws(masr_engine* e, int nseq, int Tq) {
    const int d = e->cfg.d_model, pad = e->cfg.cnn_kernel - 1;
    const size_t M = (size_t)nseq * Tq;
    CHK(e->ln.ensure(M * d * sizeof(float)));
    CHK(e->qkv.ensure(M * 3 * d * sizeof(float)));
    CHK(e->att.ensure(M * d * sizeof(float)));
    CHK(e->lnpad.ensure((size_t)nseq * (Tq + pad) * d * sizeof(float)));
    CHK(e->glu.ensure((size_t)nseq * (Tq + pad) * d * sizeof(float)));
    CHK(e->dwo.ensure(M * d * sizeof(float)));
    return 0;
}

// conv module on x (in place residual).
//  offline  (hist == false): LayerNorm + zero history rows + pad masking are fused into the pointwise_conv1
//                            GEMM's A-tile prologue (rowgemm PRO_LN_PAD); lnpad is not used.
//  streaming (hist == true): lnpad rows [0,pad) of every sequence already hold the cnn cache; LayerNorm writes
//                            the new rows behind them (needed for the next cache) and the GEMM reads lnpad.
int conv_module(masr_engine* e, hipStream_t s, const LayerW& w, const EncodeCtx& c, bool hist, int K = 0, int mstride = 4,
                bool pw1_done = false, bool pw2_later = false) {
    if (K <= 0) K = e->cfg.cnn_kernel;
    const int d = e->cfg.d_model, pad = K - 1;
    const int M = c.nseq * c.Tq, Mp = c.nseq * (c.Tq + pad);
    float* x = e->x.as<float>();
    if (pw1_done) {
        // glu buffer already filled by mhsa_out_pw1 (fused out-projection -> LayerNorm -> pointwise_conv1 -> GLU)
    } else if (hist) {
        launch_layernorm(x, w.ln_conv_w, w.ln_conv_b, e->lnpad.as<float>(), M, 1e-5f, c.Tq, pad, c.lens, s);
        rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_GLU, e->lnpad.as<float>(), d, nullptr, nullptr, w.pw1_w, w.pw1_b,
                e->glu.as<float>(), d, Mp, 2 * d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
    } else {
        // only the real rows go through the GEMM (M = nseq*Tq: 248 workgroups at B=32 x 10 s, one per CU); they land in the
        // padded layout, whose history rows are the constant glu(bias) that the depthwise kernel substitutes itself
        // (non-causal build, streaming: False: the depthwise Conv1d pads the GLU output with (K-1)/2 zero rows on both sides --
        //  those rows of the glu buffer are zeroed once per call by the caller)
        rowgemm(e, s, RG_PRO_LN_PAD, RG_EPI_GLU, x, d, w.ln_conv_w, w.ln_conv_b, w.pw1_w, w.pw1_b, e->glu.as<float>(), d,
                M, 2 * d, nullptr, 0, 1.f, c.lens, 0, c.Tq, 0, nullptr, nullptr, PROF_GEMM, mstride, c.Tq,
                e->cfg.causal ? pad : pad / 2, pad);
    }
    if (pw2_later) return 0;       // the rest of the module is the head stage of the FFN launch that follows
    const float* gconst = (hist || !e->cfg.causal) ? nullptr : w.gconst;
    {
        // few row blocks: depthwise conv + LayerNorm + SiLU as the prologue of the K-split pointwise_conv2 launch (the chunk
        // steps' kernel; the unmaterialised history rows of the offline causal build are substituted there too)
        RowGemmArgs a{};
        a.A = e->glu.as<float>(); a.lda = d; a.lnw = w.cln_w; a.lnb = w.cln_b; a.dw_w = w.dw_w; a.dw_b = w.dw_b; a.gconst = gconst;
        a.W = w.pw2_w; a.bias = w.pw2_b; a.C = x; a.ldc = d; a.R = x; a.ldr = d; a.M = M; a.N = d; a.alpha = 1.f; a.eps = 1e-5f;
        a.mstride = mstride; a.seq_t = c.Tq; a.pad = pad; a.lens = c.lens; a.mask_tp = c.lens ? c.Tq : 0;
        ProfScope ps(e, s, PROF_GEMM, 2.0 * M * (double)d * d);
        if (!e->conv_bn && g_few_rows_path && launch_rowgemm(a, RG_PRO_DWCONV, RG_EPI_RESID, s)) return 0;
    }
    if (e->conv_bn)        // BatchNorm build: the depthwise kernel's BN variant (the Squeezeformer's), then pointwise_conv2
        launch_dwconv_bn_silu(e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, e->dwo.as<float>(), c.nseq, c.Tq, K, s, gconst);
    else
        launch_dwconv_ln_silu(e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, e->dwo.as<float>(), c.nseq, c.Tq, K,
                              1e-5f, s, gconst);
    rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->dwo.as<float>(), d, nullptr, nullptr, w.pw2_w, w.pw2_b, x, d, M, d, x, d,
            1.f, c.lens, c.lens ? c.Tq : 0, 0, 0, nullptr, nullptr, PROF_GEMM, mstride);
    return 0;
}

// Streaming conv module of the Conformer family (LayerNorm conv norm), two launches on the small-M kernel:
//   [cnn cache | LayerNorm(new rows)] -> pointwise_conv1 + GLU        (+ the new cache)          (rowgemm_small PRO_HIST)
//   causal depthwise conv + LayerNorm + SiLU -> pointwise_conv2 + residual                       (rowgemm_small PRO_DWCONV)
// (convolution.py:98-131).  Where the small-M kernel does not apply (>= 2048 rows, frames per chunk not a multiple of 4) the
// same steps run as separate launches: conv_hist, pointwise_conv1, dwconv_ln_silu, pointwise_conv2.
int conv_module_stream(masr_engine* e, hipStream_t s, const LayerW& w, int n, int Tq, float* const* cache_rd,
                       float* const* cache_wr, int K, bool pw2_later = false) {
    const int d = e->cfg.d_model, pad = K - 1, M = n * Tq, Mp = n * (Tq + pad);
    float* x = e->x.as<float>();
    {
        RowGemmArgs a{};
        a.A = x; a.lda = d; a.lnw = w.ln_conv_w; a.lnb = w.ln_conv_b; a.W = w.pw1_w; a.bias = w.pw1_b;
        a.C = e->glu.as<float>(); a.ldc = d; a.M = Mp; a.N = 2 * d; a.alpha = 1.f; a.eps = 1e-5f; a.mstride = 4;
        a.seq_t = Tq; a.pad = pad; a.cache_rd = cache_rd; a.cache_wr = cache_wr;
        ProfScope ps(e, s, PROF_GEMM, 2.0 * Mp * (double)(2 * d) * d);
        if (!launch_rowgemm(a, RG_PRO_HIST, RG_EPI_GLU, s)) {
            launch_conv_hist(x, w.ln_conv_w, w.ln_conv_b, cache_rd, cache_wr, e->lnpad.as<float>(), n, Tq, pad, 0, 1e-5f, s);
            a.A = e->lnpad.as<float>();
            launch_rowgemm(a, RG_PRO_PLAIN, RG_EPI_GLU, s);
        }
    }
    if (pw2_later) return 0;       // depthwise conv ... pointwise_conv2: head stage of the FFN launch that follows
    {
        RowGemmArgs a{};
        a.A = e->glu.as<float>(); a.lda = d; a.lnw = w.cln_w; a.lnb = w.cln_b; a.dw_w = w.dw_w; a.dw_b = w.dw_b;
        a.W = w.pw2_w; a.bias = w.pw2_b; a.C = x; a.ldc = d; a.R = x; a.ldr = d; a.M = M; a.N = d; a.alpha = 1.f;
        a.eps = 1e-5f; a.mstride = 4; a.seq_t = Tq; a.pad = pad;
        ProfScope ps(e, s, PROF_GEMM, 2.0 * M * (double)d * d);
        if (!launch_rowgemm(a, RG_PRO_DWCONV, RG_EPI_RESID, s)) {
            launch_dwconv_ln_silu(e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, e->dwo.as<float>(), n, Tq, K, 1e-5f, s,
                                  nullptr);
            a.A = e->dwo.as<float>();
            launch_rowgemm(a, RG_PRO_PLAIN, RG_EPI_RESID, s);
        }
    }
    return 0;
}

// multi-head self-attention block on x (in place residual); seqs describe where K/V live
// kv_seqs (streaming): the k | v columns are appended to the streams' caches by the projection itself
void mhsa(masr_engine* e, hipStream_t s, const LayerW& w, int M, const AttSeq* kv_seqs = nullptr, int kv_tq = 0) {
    const int d = e->cfg.d_model;
    rowgemm(e, s, RG_PRO_LN, RG_EPI_STORE, e->x.as<float>(), d, w.ln_mha_w, w.ln_mha_b, w.wqkv, w.bqkv, e->qkv.as<float>(),
            3 * d, M, 3 * d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr, PROF_GEMM, 4, 0, 0, 0, 0, 0, 0, 0, kv_seqs, kv_tq);
}
void mhsa_out(masr_engine* e, hipStream_t s, const LayerW& w, int M) {
    const int d = e->cfg.d_model;
    float* x = e->x.as<float>();
    rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->att.as<float>(), d, nullptr, nullptr, w.wo, w.bo, x, d, M, d, x, d, 1.f,
            nullptr, 0, 0, 0, nullptr, nullptr);
}

// offline conformer layer: attention out-projection + residual and the conv module's LayerNorm + pointwise_conv1 + GLU in ONE
// kernel (rowgemm EPI_CHAIN): the updated rows go to x (global) and, without leaving the CU, through the second GEMM
// K: the layer's depthwise kernel size (Efficient Conformer: 15 before, 7 behind the stride layer); att / a_seq_t / a_seq_stride: the
// attention output when it lives in a per-sequence padded buffer (grouped attention's planes)
void mhsa_out_pw1(masr_engine* e, hipStream_t s, const LayerW& w, const EncodeCtx& c, int mstride = 4, int K = 0,
                  const float* att = nullptr, int a_seq_t = 0, int a_seq_stride = 0) {
    const int d = e->cfg.d_model, pad = (K > 0 ? K : e->cfg.cnn_kernel) - 1, M = c.nseq * c.Tq;
    float* x = e->x.as<float>();
    RowGemmArgs a{};
    a.A = att ? att : e->att.as<float>(); a.lda = d; a.a_seq_t = a_seq_t; a.a_seq_stride = a_seq_stride; a.lnw = w.ln_conv_w; a.lnb = w.ln_conv_b; a.W = w.chain_w; a.bias = w.chain_b;
    a.C = e->glu.as<float>(); a.ldc = d; a.M = M; a.N = 3 * d; a.R = x; a.R2 = x; a.ldr = d; a.alpha = 1.f;
    a.lens = c.lens; a.seq_t = c.Tq; a.mstride = mstride; a.eps = 1e-5f;
    a.out_seq_t = c.Tq; a.out_pad_l = e->cfg.causal ? pad : pad / 2; a.out_pad_tot = pad;
    if (g_rowgemm_packed) a.Wp = packed_rows_of(e, w.chain_w, 3 * d, s);
    ProfScope ps(e, s, PROF_GEMM, 2.0 * M * (double)(3 * d) * d);
    launch_rowgemm(a, RG_PRO_PLAIN, RG_EPI_CHAIN, s);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Squeezeformer (non-streaming build): weights + full-context forward
// reference: masr/model_utils/squeezeformer/{encoder,attention,convolution,positionwise,subsampling,time_reduction}.py
// ------------------------------------------------------------------------------------------------
static int upload_conv_frontend(masr_engine* e, const std::string& c1, const std::string& c2, const std::string& proj) {
    const int d = e->cfg.d_model, F = e->cfg.n_mels, F1 = (F - 1) / 2, F2 = (F1 - 1) / 2;
    const HostTensor* t;
    CHK(up(e, "encoder.global_cmvn.mean", {F}, &e->cmvn_mean));
    CHK(up(e, "encoder.global_cmvn.istd", {F}, &e->cmvn_istd));
    CHK(get(e, c1 + ".weight", {d, 1, 3, 3}, &t));
    {
        std::vector<float> w(9 * d);
        for (int c = 0; c < d; ++c)
            for (int k = 0; k < 9; ++k) w[k * d + c] = t->v[c * 9 + k];
        CHK(upload(e, w, &e->conv1_w));
    }
    CHK(up(e, c1 + ".bias", {d}, &e->conv1_b));
    CHK(get(e, c2 + ".weight", {d, d, 3, 3}, &t));
    {
        std::vector<float> w((size_t)d * 9 * d);
        for (int co = 0; co < d; ++co)
            for (int ci = 0; ci < d; ++ci)
                for (int k = 0; k < 9; ++k)          // [co][ci / 32][kh*3+kw][ci % 32], see masr_finalize
                    w[(size_t)co * 9 * d + (size_t)(ci / 32) * 9 * 32 + k * 32 + ci % 32] = t->v[((size_t)co * d + ci) * 9 + k];
        CHK(upload(e, w, &e->conv2_w));
    }
    CHK(up(e, c2 + ".bias", {d}, &e->conv2_b));
    CHK(get(e, proj + ".weight", {d, (int64_t)d * F2}, &t));
    {
        std::vector<float> w((size_t)d * d * F2);
        for (int o = 0; o < d; ++o)
            for (int c = 0; c < d; ++c)
                for (int f = 0; f < F2; ++f) w[(size_t)o * d * F2 + f * d + c] = t->v[(size_t)o * d * F2 + c * F2 + f];
        CHK(upload(e, w, &e->embed_w));
    }
    CHK(up(e, proj + ".bias", {d}, &e->embed_b));
    return 0;
}

static int upload_pos_table(masr_engine* e) {
    const int d = e->cfg.d_model;
    auto it = e->host.find("__pos_table__");
    if (it != e->host.end()) {
        if ((int64_t)it->second.v.size() != (int64_t)e->cfg.max_pos * d) return fail("__pos_table__ has wrong size");
        return upload(e, it->second.v, &e->pe);
    }
    std::vector<float> pe((size_t)e->cfg.max_pos * d);
    for (int i = 0; i < d; i += 2) {
        const float div = expf((float)i * (float)(-(log(10000.0) / d)));
        for (int p = 0; p < e->cfg.max_pos; ++p) {
            pe[(size_t)p * d + i] = sinf((float)p * div);
            pe[(size_t)p * d + i + 1] = cosf((float)p * div);
        }
    }
    return upload(e, pe, &e->pe);
}

static int finalize_squeezeformer(masr_engine* e, hipStream_t s) {
    HIPCHK(hipSetDevice(e->cfg.device_id));
    const int d = e->cfg.d_model, dff = e->cfg.d_ff, L = e->cfg.num_blocks, V = e->cfg.vocab_size, K = e->cfg.cnn_kernel,
              H = e->cfg.heads, dk = d / H;
    const HostTensor* t;
    CHK(upload_conv_frontend(e, "encoder.embed.pw_conv", "encoder.embed.dw_conv", "encoder.embed.input_proj.0"));
    CHK(upload_pos_table(e));
    CHK(up(e, "encoder.preln.weight", {d}, &e->preln_w));
    CHK(up(e, "encoder.preln.bias", {d}, &e->preln_b));
    e->sq_layers.assign(L, SqLayerW{});
    for (int i = 0; i < L; ++i) {
        SqLayerW& w = e->sq_layers[i];
        const std::string p = "encoder.encoders." + std::to_string(i) + ".";
        auto vec = [&](const std::string& n, float** out) -> int { return up(e, p + n, {d}, out); };
        CHK(vec("self_attn.ada_scale", &w.att_s));
        CHK(vec("self_attn.ada_bias", &w.att_b));
        {
            std::vector<float> wq((size_t)3 * d * d), bq(3 * d);
            const char* nm[3] = {"linear_q", "linear_k", "linear_v"};
            for (int j = 0; j < 3; ++j) {
                CHK(get(e, p + "self_attn." + nm[j] + ".weight", {d, d}, &t));
                memcpy(&wq[(size_t)j * d * d], t->v.data(), sizeof(float) * d * d);
                CHK(get(e, p + "self_attn." + nm[j] + ".bias", {d}, &t));
                memcpy(&bq[j * d], t->v.data(), sizeof(float) * d);
            }
            CHK(upload(e, wq, &w.wqkv));
            CHK(upload(e, bq, &w.bqkv));
        }
        CHK(up(e, p + "self_attn.linear_out.weight", {d, d}, &w.wo));
        CHK(up(e, p + "self_attn.linear_out.bias", {d}, &w.bo));
        CHK(up(e, p + "self_attn.linear_pos.weight", {d, d}, &w.wpos));
        CHK(up(e, p + "self_attn.pos_bias_u", {H, dk}, &w.pos_u));
        CHK(up(e, p + "self_attn.pos_bias_v", {H, dk}, &w.pos_v));
        {
            void* pt = nullptr;
            HIPCHK(hipMalloc(&pt, (size_t)e->cfg.max_pos * d * sizeof(float)));
            e->owned.push_back(pt);
            w.ptab = (float*)pt;
            gemm(e, s, e->pe, d, w.wpos, nullptr, w.ptab, d, e->cfg.max_pos, d, d, ACT_NONE, 1.f, nullptr, 0, PROF_NONE);
        }
        CHK(vec("layer_norm1.weight", &w.ln1_w)); CHK(vec("layer_norm1.bias", &w.ln1_b));
        CHK(vec("layer_norm2.weight", &w.ln2_w)); CHK(vec("layer_norm2.bias", &w.ln2_b));
        CHK(vec("layer_norm3.weight", &w.ln3_w)); CHK(vec("layer_norm3.bias", &w.ln3_b));
        CHK(vec("layer_norm4.weight", &w.ln4_w)); CHK(vec("layer_norm4.bias", &w.ln4_b));
        CHK(vec("ffn1.ada_scale", &w.f1_s)); CHK(vec("ffn1.ada_bias", &w.f1_b));
        CHK(up(e, p + "ffn1.w_1.weight", {dff, d}, &w.f1_w1)); CHK(up(e, p + "ffn1.w_1.bias", {dff}, &w.f1_b1));
        CHK(up(e, p + "ffn1.w_2.weight", {d, dff}, &w.f1_w2)); CHK(up(e, p + "ffn1.w_2.bias", {d}, &w.f1_b2));
        CHK(vec("ffn2.ada_scale", &w.f2_s)); CHK(vec("ffn2.ada_bias", &w.f2_b));
        CHK(up(e, p + "ffn2.w_1.weight", {dff, d}, &w.f2_w1)); CHK(up(e, p + "ffn2.w_1.bias", {dff}, &w.f2_b1));
        CHK(up(e, p + "ffn2.w_2.weight", {d, dff}, &w.f2_w2)); CHK(up(e, p + "ffn2.w_2.bias", {d}, &w.f2_b2));
        CHK(vec("conv_module.ada_scale", &w.cv_s)); CHK(vec("conv_module.ada_bias", &w.cv_b));
        CHK(up(e, p + "conv_module.pointwise_conv1.weight", {2 * d, d, 1}, &w.pw1_w));
        CHK(up(e, p + "conv_module.pointwise_conv1.bias", {2 * d}, &w.pw1_b));
        {   // streaming-trained build (causal conv): the K-1 zero frames padded in front of pointwise_conv1 become glu(bias)
            std::vector<float> z(d, 0.f);
            CHK(upload(e, z, &w.gconst));
            launch_glu_const(w.pw1_b, w.gconst, s);
        }
        {
            CHK(get(e, p + "conv_module.depthwise_conv.weight", {d, 1, K}, &t));
            std::vector<float> wd((size_t)K * d);
            for (int c = 0; c < d; ++c)
                for (int j = 0; j < K; ++j) wd[(size_t)j * d + c] = t->v[(size_t)c * K + j];
            CHK(upload(e, wd, &w.dw_w));
            CHK(up(e, p + "conv_module.depthwise_conv.bias", {d}, &w.dw_b));
        }
        {   // eval-mode BatchNorm1d folded: y = x * scale + shift, scale = w / sqrt(var + eps), shift = b - mean * scale
            const HostTensor *tw, *tb, *tm, *tv;
            CHK(get(e, p + "conv_module.norm.weight", {d}, &tw));
            CHK(get(e, p + "conv_module.norm.bias", {d}, &tb));
            CHK(get(e, p + "conv_module.norm.running_mean", {d}, &tm));
            CHK(get(e, p + "conv_module.norm.running_var", {d}, &tv));
            std::vector<float> sc(d), sh(d);
            for (int c = 0; c < d; ++c) {
                sc[c] = tw->v[c] / sqrtf(tv->v[c] + 1e-5f);
                sh[c] = tb->v[c] - tm->v[c] * sc[c];
            }
            CHK(upload(e, sc, &w.bn_scale));
            CHK(upload(e, sh, &w.bn_shift));
        }
        CHK(up(e, p + "conv_module.pointwise_conv2.weight", {d, d, 1}, &w.pw2_w));
        CHK(up(e, p + "conv_module.pointwise_conv2.bias", {d}, &w.pw2_b));
    }
    {
        // TimeReductionLayer1D: depthwise k = 5, stride 2, padding 3; TimeReductionLayerStream (streaming-trained build,
        // squeezeformer/model.py:37-41): k = 1, stride 2, no padding == the k = 5 kernel with only tap 3 (t = 2j) non-zero
        const int tk = e->cfg.causal ? 1 : 5;
        CHK(get(e, "encoder.time_reduction_layer.dw_conv.weight", {d, 1, tk}, &t));
        std::vector<float> wd((size_t)5 * d, 0.f);
        for (int c = 0; c < d; ++c)
            for (int j = 0; j < tk; ++j) wd[(size_t)(tk == 1 ? 3 : j) * d + c] = t->v[(size_t)c * tk + j];
        CHK(upload(e, wd, &e->tr_dw_w));
        CHK(up(e, "encoder.time_reduction_layer.dw_conv.bias", {d}, &e->tr_dw_b));
        CHK(up(e, "encoder.time_reduction_layer.pw_conv.weight", {d, d, 1}, &e->tr_pw_w));
        CHK(up(e, "encoder.time_reduction_layer.pw_conv.bias", {d}, &e->tr_pw_b));
        CHK(up(e, "encoder.time_recover_layer.weight", {d, d}, &e->rec_w));
        CHK(up(e, "encoder.time_recover_layer.bias", {d}, &e->rec_b));
    }
    CHK(up(e, "ctc.ctc_lo.weight", {V, d}, &e->ctc_w));
    CHK(up(e, "ctc.ctc_lo.bias", {V}, &e->ctc_b));
    if (sqz_stage_supported(e)) {
        // the fused stages' fragment-order weight copies are made HERE, once, behind this call's synchronisation: a forward pass on
        // any stream then only reads the caches (they used to be filled by whichever stream ran the first pass, with nothing
        // ordering a second stream's reads behind that pack kernel)
        for (int i = 0; i < L; ++i) {
            const SqLayerW& w = e->sq_layers[i];
            const float *p1, *p2;
            CHK(packed_ffn_of(e, w.f1_w1, w.f1_w2, s, &p1, &p2));
            CHK(packed_ffn_of(e, w.f2_w1, w.f2_w2, s, &p1, &p2));
            if (!packed_rows_of(e, w.wo, d, s) || !packed_rows_of(e, w.pw1_w, 2 * d, s) || !packed_rows_of(e, w.pw2_w, d, s) ||
                !packed_rows_of(e, w.wqkv, 3 * d, s))
                return fail("squeezeformer: packing the layer's weights failed");
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    e->host.clear();
    e->finalized = true;
    return 0;
}

// SqueezeformerEncoder.forward, streaming = False (squeezeformer/encoder.py:168-216)
static int encode_full_squeezeformer(masr_engine* e, hipStream_t s, const float* feats, const int* lens, int B, int T,
                                     float* enc_out, int chunk) {
    const int d = e->cfg.d_model, H = e->cfg.heads, K = e->cfg.cnn_kernel, half = (K - 1) / 2;
    int T0 = 0;
    // Row blocks / tiles that hold padded frames only are not computed (the valid frames' results do not depend on them: pad masks,
    // klen; reference encoder.py:168-216 computes and masks them): conv2, the input projection, attention (queries and keys stop at the
    // valid length) and the fused stage kernels -- a length-sorted batch padded to its longest utterance costs its VALID frames, so
    // one pass can take utterances of any lengths.  Off (masr_debug_set key 38 = 0) when the caller decodes the padded frames too.
    const bool skip = e->skip_padding != 0 && lens != nullptr;
    const int skm = skip ? e->skip_padding : 0;        // (bits, for A/B: 1 stage kernels, 2 conv2 + input projection, 4 attention)
    CHK(embed(e, s, feats, B, T, &T0, (skm & 2) ? lens : nullptr));
    if (T0 >= e->cfg.max_pos) return fail("sequence longer than max_pos");
    CHK(ensure_layer_ws(e, B, T0));
    CHK(e->attseq.ensure(sizeof(AttSeq) * B));
    CHK(e->xsave.ensure((size_t)B * T0 * d * sizeof(float)));
    CHK(e->xred.ensure((size_t)B * ((T0 + 1) / 2) * d * sizeof(float)));
    float* x = e->x.as<float>();
    launch_layernorm(x, e->preln_w, e->preln_b, x, B * T0, 1e-5f, 0, 0, nullptr, s);
    int Tq = T0, mstride = 4, pstride = 1;
    const bool causal = e->cfg.causal != 0;   // streaming-trained build: K-1 history rows in front (constant glu(bias))
    const int pad_l = causal ? K - 1 : half;
    auto new_resolution = [&]() -> int {     // zero the symmetric pad rows of the GLU buffer, rebuild the descriptors
        if (!causal) HIPCHK(hipMemsetAsync(e->glu.p, 0, (size_t)B * (Tq + 2 * half) * d * sizeof(float), s));
        launch_attseq_full(e->attseq.as<AttSeq>(), e->qkv.as<float>(), e->att.as<float>(), lens, B, Tq, mstride, s, (skm & 4) ? 1 : 0);
        return 0;
    };
    CHK(new_resolution());
    const int L = e->cfg.num_blocks;
    bool qkv_ready = false;       // the previous layer's fused end stage has already written this layer's q | k | v
    for (int i = 0; i < L; ++i) {
        const SqLayerW& w = e->sq_layers[i];
        if (i == e->reduce_idx) {
            // TimeReductionLayer1D (time_reduction.py:53-76): keep x for the recovery, halve the frame rate
            HIPCHK(hipMemcpyAsync(e->xsave.p, x, (size_t)B * Tq * d * sizeof(float), hipMemcpyDeviceToDevice, s));
            launch_time_reduce_dw(x, e->tr_dw_w, e->tr_dw_b, lens, e->xred.as<float>(), B, Tq, mstride, s);
            const int Lr = (Tq + 1) / 2;
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_STORE, e->xred.as<float>(), d, nullptr, nullptr, e->tr_pw_w, e->tr_pw_b, x, d,
                    B * Lr, d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
            Tq = Lr; mstride = 8; pstride = 2;
            CHK(new_resolution());
        }
        if (i == e->recover_idx && e->reduce_idx >= 0) {
            // x = saved + Linear(repeat_interleave(x, 2))[:, :T0]   (encoder.py:199-205)
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_STORE, x, d, nullptr, nullptr, e->rec_w, e->rec_b, e->xred.as<float>(), d,
                    B * Tq, d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
            launch_recover_add(e->xsave.as<float>(), e->xred.as<float>(), x, B, T0, Tq, s);
            Tq = T0; mstride = 4; pstride = 1;
            CHK(new_resolution());
        }
        const int M = B * Tq;
        // Fused layer (sqz_layer.hip): attention + two row-block kernels.  Taken when the row blocks fill the chip (below that the
        // d_ff-split FFN and the K-split projections of the unfused sequence are the faster launches); bit-identical either way.
        const bool fused = g_sqz_fused_blocks > 0 && (M + 31) / 32 >= g_sqz_fused_blocks && sqz_stage_supported(e);
        // x = LN1(x + MHSA(ada(x)))
        if (!qkv_ready)
            rowgemm(e, s, RG_PRO_AFFINE, RG_EPI_STORE, x, d, w.att_s, w.att_b, w.wqkv, w.bqkv, e->qkv.as<float>(), 3 * d, M,
                    3 * d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
        qkv_ready = false;
        {
            ProfScope ps(e, s, PROF_ATT, 6.0 * d * (double)Tq * Tq * B);
            // (chunk > 0: decoding_chunk_size of the streaming-trained build; the kernel thins the mask by the layer's rate)
            launch_attention(e->attseq.as<AttSeq>(), B, Tq, H, 3 * d, 3 * d, w.ptab, w.pos_u, w.pos_v, chunk, pstride, s);
        }
        if (fused) {
            // the next layer's QKV projection rides on this layer's last launch unless the frame rate changes in between
            const bool next_same = i + 1 < L && i + 1 != e->reduce_idx && !(i + 1 == e->recover_idx && e->reduce_idx >= 0);
            SqzStageArgs a{};
            a.x = x; a.out = x; a.att = e->att.as<float>();
            a.head_w = packed_rows_of(e, w.wo, d, s); a.head_b = w.bo;
            a.ln_a_w = w.ln1_w; a.ln_a_b = w.ln1_b; a.ln_b_w = w.ln2_w; a.ln_b_b = w.ln2_b;
            a.ffn_s = w.f1_s; a.ffn_b = w.f1_b; a.b1 = w.f1_b1; a.b2 = w.f1_b2;
            CHK(packed_ffn_of(e, w.f1_w1, w.f1_w2, s, &a.w1, &a.w2));
            a.tail_w = packed_rows_of(e, w.pw1_w, 2 * d, s); a.tail_b = w.pw1_b; a.tail_s = w.cv_s; a.tail_sb = w.cv_b; a.tail_n = 2 * d;
            a.glu_out = e->glu.as<float>(); a.glu_pad_l = pad_l; a.glu_pad_tot = 2 * half;
            a.lens = lens; a.M = M; a.dff = e->cfg.d_ff; a.seq_t = Tq; a.mstride = mstride; a.ktaps = K; a.eps = 1e-5f; a.skip_pad = skm & 1;
            if (!a.head_w || !a.tail_w) return fail("squeezeformer: packing the layer's weights failed");
            {
                ProfScope ps(e, s, PROF_FFN_TAIL, 4.0 * M * (double)e->cfg.d_ff * d + 2.0 * M * (double)(3 * d) * d);
                if (!launch_sqz_stage(a, 0, s)) return fail("squeezeformer: the fused mid stage rejected the sizes");
            }
            SqzStageArgs b{};
            b.x = x; b.out = i == L - 1 ? enc_out : x; b.glu = e->glu.as<float>();
            b.head_w = packed_rows_of(e, w.pw2_w, d, s); b.head_b = w.pw2_b;
            b.dw_w = w.dw_w; b.dw_b = w.dw_b; b.bn_scale = w.bn_scale; b.bn_shift = w.bn_shift; b.gconst = causal ? w.gconst : nullptr;
            b.gpad = causal ? nullptr : w.gconst; b.glu_pad_l = pad_l;
            b.ln_a_w = w.ln3_w; b.ln_a_b = w.ln3_b; b.ln_b_w = w.ln4_w; b.ln_b_b = w.ln4_b;
            b.ffn_s = w.f2_s; b.ffn_b = w.f2_b; b.b1 = w.f2_b1; b.b2 = w.f2_b2;
            CHK(packed_ffn_of(e, w.f2_w1, w.f2_w2, s, &b.w1, &b.w2));
            if (next_same) {
                const SqLayerW& nx = e->sq_layers[i + 1];
                b.tail_w = packed_rows_of(e, nx.wqkv, 3 * d, s); b.tail_b = nx.bqkv; b.tail_s = nx.att_s; b.tail_sb = nx.att_b;
                b.tail_n = 3 * d; b.tail_out = e->qkv.as<float>();
                if (!b.tail_w) return fail("squeezeformer: packing the layer's weights failed");
            }
            b.lens = lens; b.M = M; b.dff = e->cfg.d_ff; b.seq_t = Tq; b.mstride = mstride; b.ktaps = K; b.eps = 1e-5f; b.skip_pad = skm & 1;
            if (!b.head_w) return fail("squeezeformer: packing the layer's weights failed");
            {
                ProfScope ps(e, s, PROF_FFN_HEAD, 4.0 * M * (double)e->cfg.d_ff * d + 2.0 * M * (double)d * d +
                                                      (next_same ? 2.0 * M * (double)(3 * d) * d : 0.0));
                if (!launch_sqz_stage(b, 1, s)) return fail("squeezeformer: the fused end stage rejected the sizes");
            }
            qkv_ready = next_same;
            continue;
        }
        rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->att.as<float>(), d, nullptr, nullptr, w.wo, w.bo, x, d, M, d, x, d, 1.f,
                nullptr, 0, 0, 0, nullptr, nullptr);
        launch_layernorm(x, w.ln1_w, w.ln1_b, x, M, 1e-5f, 0, 0, nullptr, s);
        // x = LN2(x + FFN1(ada(x)))   (the post-LayerNorm rides on the d_ff-split reduction where the block runs split: few row
        // blocks, i.e. exactly the launches that take this branch by default; otherwise ffn() launches it)
        CHK(ffn(e, s, M, w.f1_s, w.f1_b, w.f1_w1, w.f1_b1, w.f1_w2, w.f1_b2, 1.0f, 1, w.ln2_w, w.ln2_b, x));
        // x = LN3(x + Conv(ada(x)))   symmetric depthwise conv: (K-1)/2 zero rows on both sides of the GLU output
        rowgemm(e, s, RG_PRO_AFFINE, RG_EPI_GLU, x, d, w.cv_s, w.cv_b, w.pw1_w, w.pw1_b, e->glu.as<float>(), d, M, 2 * d,
                nullptr, 0, 1.f, lens, 0, Tq, 0, nullptr, nullptr, PROF_GEMM, mstride, Tq, pad_l, 2 * half);
        launch_dwconv_bn_silu(e->glu.as<float>(), w.dw_w, w.dw_b, w.bn_scale, w.bn_shift, e->dwo.as<float>(), B, Tq, K, s,
                              causal ? w.gconst : nullptr);
        rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->dwo.as<float>(), d, nullptr, nullptr, w.pw2_w, w.pw2_b, x, d, M, d, x,
                d, 1.f, lens, Tq, 0, 0, nullptr, nullptr, PROF_GEMM, mstride);
        launch_layernorm(x, w.ln3_w, w.ln3_b, x, M, 1e-5f, 0, 0, nullptr, s);
        // x = LN4(x + FFN2(ada(x)))
        CHK(ffn(e, s, M, w.f2_s, w.f2_b, w.f2_w1, w.f2_b1, w.f2_w2, w.f2_b2, 1.0f, 1, w.ln4_w, w.ln4_b, i == L - 1 ? enc_out : x));
    }
    LAUNCHCHK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Efficient Conformer, full-context forward (efficient_conformer/encoder.py:213-265): Conformer layers with
// grouped attention in the first n_group_layers blocks, a stride-2 conv block at stride_idx (AvgPool residual),
// half frame rate + kernel 7 afterwards.  Weights share the Conformer key names (masr_finalize).
// ------------------------------------------------------------------------------------------------
static int encode_full_efficient(masr_engine* e, hipStream_t s, const float* feats, const int* lens, int B, int T,
                                 float* enc_out, int chunk) {
    const int d = e->cfg.d_model, H = e->cfg.heads, G = e->group_size;
    const bool causal = e->cfg.causal != 0;
    int T0 = 0;
    CHK(embed(e, s, feats, B, T, &T0));
    if (T0 >= e->cfg.max_pos) return fail("sequence longer than max_pos");
    CHK(ensure_layer_ws(e, B, T0));
    const int Tg = (T0 + G - 1) / G, Tpad = Tg * G;
    const size_t plane = (size_t)B * Tpad * d;
    CHK(e->attseq.ensure(sizeof(AttSeq) * 2 * B));
    CHK(e->qplanes.ensure(3 * plane * sizeof(float)));
    CHK(e->attp.ensure(plane * sizeof(float)));
    CHK(e->xsave.ensure((size_t)B * ((T0 + 1) / 2) * d * sizeof(float)));
    float* x = e->x.as<float>();
    float* qp = e->qplanes.as<float>();
    AttSeq* seq_g = e->attseq.as<AttSeq>();
    AttSeq* seq_r = seq_g + B;
    HIPCHK(hipMemsetAsync(qp, 0, 3 * plane * sizeof(float), s));     // time padding rows of q/k/v stay zero (pad4group)
    int Tq = T0, mstride = 4, pstride = 1;
    launch_attseq_grouped(seq_g, qp, qp + plane, qp + 2 * plane, e->attp.as<float>(), lens, B, Tg, G, mstride, s);
    launch_attseq_full(seq_r, e->qkv.as<float>(), e->att.as<float>(), lens, B, Tq, mstride, s);
    // streaming: False build (symmetric conv): the (K - 1) / 2 zero rows on both sides of every sequence in the GLU buffer are
    // cleared once per frame rate / kernel size (the layers only ever write the real rows)
    if (!causal) HIPCHK(hipMemsetAsync(e->glu.p, 0, (size_t)B * (Tq + e->cfg.cnn_kernel - 1) * d * sizeof(float), s));
    const int L = e->cfg.num_blocks;
    for (int i = 0; i < L; ++i) {
        const LayerW& w = e->layers[i];
        int M = B * Tq;
        const int Ki = layer_kernel(e, i);
        // enough row blocks for the full (non d_ff-split) FFN launch: the Conformer's fused launches (key 31 = 0: separate ones)
        const bool fused = g_efficient_fused && !g_no_chain && !g_no_ffn_head && i != e->stride_idx &&
                           (M + 31) / 32 >= g_ffn_split_blocks && (Ki == 15 || Ki == 7) && d == 256;
        const EncodeCtx ctx0{B, Tq, lens};
        // regular layers: LayerNorm + fused QKV projection ride on the first FFN kernel (tail stage), like the Conformer
        const FfnTail tail{w.ln_mha_w, w.ln_mha_b, w.wqkv, w.bqkv, e->qkv.as<float>(), 3 * d, 3 * d, nullptr, nullptr};
        bool qkv_done = false;
        // grouped layers: the same tail stage writes q | k | v PLANAR into the time-padded buffers of the grouped attention
        // (round 4; ffn_pc.hip only -- the two-chain kernel keeps the separate projection)
        const FfnTail gtail{w.ln_mha_w, w.ln_mha_b, w.wqkv, w.bqkv, qp, 3 * d, d, nullptr, nullptr, (long)plane, Tq, Tpad - Tq};
        const bool planar_tail = layer_grouped(e, i) && g_efficient_fused && !g_ffn_dual;
        CHK(ffn(e, s, M, w.ln_ffm_w, w.ln_ffm_b, w.ffm_w1, w.ffm_b1, w.ffm_w2, w.ffm_b2, 0.5f, 0, nullptr, nullptr, nullptr,
                layer_grouped(e, i) ? (planar_tail ? &gtail : nullptr) : &tail, &qkv_done));
        if (layer_grouped(e, i)) {
            if (Tq != T0) return fail("grouped attention after the stride layer is not supported");
            // q | k | v -> planar, time-padded buffers; attention over T/3 positions with d_k' = 192
            if (!qkv_done)
                rowgemm(e, s, RG_PRO_LN, RG_EPI_STORE, x, d, w.ln_mha_w, w.ln_mha_b, w.wqkv, w.bqkv, qp, d, M, 3 * d, nullptr,
                        0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr, PROF_GEMM, mstride, Tq, 0, Tpad - Tq, d, (long)plane);
            {
                ProfScope ps(e, s, PROF_ATT, 6.0 * d * (double)Tq * Tq * B / G);
                launch_attention_grouped(seq_g, B, Tg, H, G, w.ptab, Tq, w.pos_u, w.pos_v, s, chunk);
            }
            if (fused) mhsa_out_pw1(e, s, w, ctx0, mstride, Ki, e->attp.as<float>(), Tq, Tpad);
            else
                rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->attp.as<float>(), d, nullptr, nullptr, w.wo, w.bo, x, d, M, d, x, d,
                        1.f, nullptr, 0, 0, 0, nullptr, nullptr, PROF_GEMM, mstride, 0, 0, 0, 0, 0, Tq, Tpad);
        } else {
            if (!qkv_done) mhsa(e, s, w, M);
            {
                ProfScope ps(e, s, PROF_ATT, 6.0 * d * (double)Tq * Tq * B);
                launch_attention(seq_r, B, Tq, H, 3 * d, 3 * d, w.ptab, w.pos_u, w.pos_v, chunk, pstride, s);
            }
            if (fused) mhsa_out_pw1(e, s, w, ctx0, mstride, Ki);
            else mhsa_out(e, s, w, M);
        }
        EncodeCtx ctx{B, Tq, lens};
        if (fused) {
            // like the offline Conformer layer: [out-proj + residual -> LN -> pw1 -> GLU] was one kernel; the rest of the conv
            // module is the head stage of the second FFN launch (15 taps before, 7 behind the stride layer)
            const FfnHead head{e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, causal ? w.gconst : nullptr,
                               w.pw2_w, w.pw2_b, lens, Tq, Ki, mstride, nullptr};
            CHK(ffn(e, s, M, w.ln_ff_w, w.ln_ff_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2, 0.5f, 0, nullptr, nullptr, nullptr, nullptr,
                    nullptr, &head));
            launch_layernorm(x, w.ln_fin_w, w.ln_fin_b, x, M, 1e-5f, 0, 0, nullptr, s);
            continue;
        }
        if (i == e->stride_idx) {
            // StrideConformerEncoderLayer (encoder.py:454-545): x = AvgPool(x) + conv_module_stride2(LN(x))
            const int K = layer_kernel(e, i), pad = K - 1, T2 = (Tq + 1) / 2;
            if (causal)      // the K - 1 history rows go through pointwise_conv1 + GLU like the reference's left padding
                rowgemm(e, s, RG_PRO_LN_PAD, RG_EPI_GLU, x, d, w.ln_conv_w, w.ln_conv_b, w.pw1_w, w.pw1_b, e->glu.as<float>(), d,
                        B * (Tq + pad), 2 * d, nullptr, 0, 1.f, lens, 0, Tq, pad, nullptr, nullptr, PROF_GEMM, mstride);
            else             // symmetric: the real rows land between (K - 1) / 2 zero rows on either side (same stride-2 window sum)
                rowgemm(e, s, RG_PRO_LN_PAD, RG_EPI_GLU, x, d, w.ln_conv_w, w.ln_conv_b, w.pw1_w, w.pw1_b, e->glu.as<float>(), d,
                        M, 2 * d, nullptr, 0, 1.f, lens, 0, Tq, 0, nullptr, nullptr, PROF_GEMM, mstride, Tq, pad / 2, pad);
            launch_dwconv_stride2_ln_silu(e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, e->dwo.as<float>(), B, Tq, K,
                                          1e-5f, s);
            launch_avgpool2(x, e->xsave.as<float>(), B, Tq, s);
            Tq = T2; mstride *= 2; pstride *= 2; M = B * Tq;
            if (!causal)     // half rate, kernel 7 from here on: a new padded layout
                HIPCHK(hipMemsetAsync(e->glu.p, 0, (size_t)B * (Tq + layer_kernel(e, i + 1) - 1) * d * sizeof(float), s));
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->dwo.as<float>(), d, nullptr, nullptr, w.pw2_w, w.pw2_b, x, d, M, d,
                    e->xsave.as<float>(), d, 1.f, lens, Tq, 0, 0, nullptr, nullptr, PROF_GEMM, mstride);
            launch_attseq_full(seq_r, e->qkv.as<float>(), e->att.as<float>(), lens, B, Tq, mstride, s);
        } else {
            CHK(conv_module(e, s, w, ctx, false, layer_kernel(e, i), mstride));
        }
        CHK(ffn(e, s, M, w.ln_ff_w, w.ln_ff_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2));
        launch_layernorm(x, w.ln_fin_w, w.ln_fin_b, x, M, 1e-5f, 0, 0, nullptr, s);
    }
    launch_layernorm(x, e->after_w, e->after_b, enc_out, B * Tq, 1e-5f, 0, 0, nullptr, s);
    LAUNCHCHK();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// DeepSpeech2 (model_kind 3): weights + forward (full utterances and stateful chunks)
// reference: masr/model_utils/deepspeech2/{model.py:64-108, encoder.py:36-129, conv.py:5-22}
// ------------------------------------------------------------------------------------------------
static int finalize_ds2(masr_engine* e) {
    HIPCHK(hipSetDevice(e->cfg.device_id));
    const int H = e->cfg.d_model, L = e->cfg.num_blocks, ndir = e->cfg.causal ? 1 : 2, V = e->cfg.vocab_size;
    const int F = e->cfg.n_mels, F1 = (F - 1) / 2, F2 = (F1 - 1) / 2, C = 32, D = H * ndir;
    const HostTensor* t;
    CHK(up(e, "encoder.global_cmvn.mean", {F}, &e->cmvn_mean));
    CHK(up(e, "encoder.global_cmvn.istd", {F}, &e->cmvn_istd));
    CHK(get(e, "encoder.conv.conv.0.weight", {C, 1, 3, 3}, &t));
    {
        std::vector<float> w(9 * C);
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < 9; ++k) w[k * C + c] = t->v[c * 9 + k];
        CHK(upload(e, w, &e->conv1_w));
    }
    CHK(up(e, "encoder.conv.conv.0.bias", {C}, &e->conv1_b));
    CHK(get(e, "encoder.conv.conv.2.weight", {C, C, 3, 3}, &t));
    {
        std::vector<float> w((size_t)C * 9 * C);
        for (int co = 0; co < C; ++co)
            for (int ci = 0; ci < C; ++ci)
                for (int k = 0; k < 9; ++k) w[(size_t)co * 9 * C + k * C + ci] = t->v[((size_t)co * C + ci) * 9 + k];
        CHK(upload(e, w, &e->conv2_w));
    }
    CHK(up(e, "encoder.conv.conv.2.bias", {C}, &e->conv2_b));
    e->ds2_layers.assign(L, Ds2LayerW{});
    for (int i = 0; i < L; ++i) {
        Ds2LayerW& w = e->ds2_layers[i];
        const std::string p = "encoder.rnns." + std::to_string(i) + ".";
        const int kin = i == 0 ? C * F2 : D;
        w.kin = kin;
        std::vector<float> wih((size_t)ndir * 4 * H * kin), bih((size_t)ndir * 4 * H), whh((size_t)ndir * 4 * H * H);
        for (int dir = 0; dir < ndir; ++dir) {
            const std::string suf = dir ? "_reverse" : "";
            const HostTensor *a, *b, *c, *d;
            CHK(get(e, p + "rnn.weight_ih_l0" + suf, {4 * H, kin}, &a));
            CHK(get(e, p + "rnn.weight_hh_l0" + suf, {4 * H, H}, &b));
            CHK(get(e, p + "rnn.bias_ih_l0" + suf, {4 * H}, &c));
            CHK(get(e, p + "rnn.bias_hh_l0" + suf, {4 * H}, &d));
            float* dst = wih.data() + (size_t)dir * 4 * H * kin;
            if (i == 0) {   // conv output is channels-last here: column f*32 + c  <-  reference column c*F2 + f (conv.py:20)
                for (int r = 0; r < 4 * H; ++r)
                    for (int c2 = 0; c2 < C; ++c2)
                        for (int f = 0; f < F2; ++f) dst[(size_t)r * kin + f * C + c2] = a->v[(size_t)r * kin + c2 * F2 + f];
            } else {
                std::copy(a->v.begin(), a->v.end(), dst);
            }
            std::copy(b->v.begin(), b->v.end(), whh.begin() + (size_t)dir * 4 * H * H);
            for (int r = 0; r < 4 * H; ++r) bih[(size_t)dir * 4 * H + r] = c->v[r] + d->v[r];
        }
        CHK(upload(e, wih, &w.wih));
        CHK(upload(e, bih, &w.bih));
        CHK(upload(e, whh, &w.whh));
        CHK(up(e, p + "layer_norm.weight", {D}, &w.ln_w));
        CHK(up(e, p + "layer_norm.bias", {D}, &w.ln_b));
    }
    CHK(up(e, "decoder.ctc_lo.weight", {V, D}, &e->ctc_w));
    CHK(up(e, "decoder.ctc_lo.bias", {V}, &e->ctc_b));
    e->host.clear();
    e->finalized = true;
    return 0;
}

// feats [nseq, T, 80] -> enc_out [nseq*Tq, D].  lens: device feature lengths (full utterances) or nullptr (chunks:
// every row is valid, inference_predictor.py:70).  st: per-sequence streams carrying (h, c) of every layer, or nullptr.
static int ds2_forward(masr_engine* e, hipStream_t s, const float* feats, const int* lens, int nseq, int T, float* enc_out,
                       Stream** st, int* Tq_out) {
    const int H = e->cfg.d_model, L = e->cfg.num_blocks, ndir = e->cfg.causal ? 1 : 2, D = H * ndir, C = 32;
    const int F = e->cfg.n_mels, F1 = (F - 1) / 2, F2 = (F1 - 1) / 2;
    const int T1 = (T - 1) / 2, Tq = (T1 - 1) / 2;
    if (T < 7 || Tq <= 0) return fail("input too short for Conv2dSubsampling4Pure (need >= 7 frames)");
    if (st && ndir != 1) return fail("deepspeech2: stateful chunks need the uni-directional (streaming) model");
    const int M = nseq * Tq;
    CHK(e->x1.ensure((size_t)nseq * T1 * F1 * C * sizeof(float)));
    CHK(e->x2.ensure((size_t)M * F2 * C * sizeof(float)));
    CHK(e->gx.ensure((size_t)M * ndir * 4 * H * sizeof(float)));
    CHK(e->rnn_out.ensure((size_t)M * D * sizeof(float)));
    CHK(e->ln.ensure((size_t)M * D * sizeof(float)));
    CHK(e->hstate.ensure((size_t)2 * ndir * nseq * H * sizeof(float)));
    CHK(e->cstate.ensure((size_t)ndir * nseq * H * sizeof(float)));
    const int* xl = nullptr;
    if (lens) {
        CHK(e->ds2_lens.ensure(sizeof(int) * nseq));
        launch_ds2_lens(lens, nseq, Tq, e->ds2_lens.as<int>(), s);
        xl = e->ds2_lens.as<int>();
    }
    launch_conv1(feats, e->cmvn_mean, e->cmvn_istd, e->conv1_w, e->conv1_b, e->x1.as<float>(), nseq, T, F, C, s);
    {
        GemmArgs a{};
        a.A = e->x1.as<float>(); a.W = e->conv2_w; a.bias = e->conv2_b; a.C = e->x2.as<float>();
        a.M = M * F2; a.N = C; a.K = 9 * C; a.ldc = C; a.act = ACT_RELU; a.alpha = 1.f;
        a.T1 = T1; a.F1 = F1; a.T2 = Tq; a.F2 = F2; a.Cc = C;
        ProfScope ps(e, s, PROF_CONV2, 2.0 * a.M * (double)a.N * a.K);
        launch_gemm(a, A_CONV2, EPI_STD, s);
    }
    const float* in = e->x2.as<float>();
    const size_t hsz = (size_t)ndir * nseq * H;
    float* hbuf = e->hstate.as<float>();
    float* cbuf = e->cstate.as<float>();
    for (int l = 0; l < L; ++l) {
        const Ds2LayerW& w = e->ds2_layers[l];
        gemm(e, s, in, w.kin, w.wih, w.bih, e->gx.as<float>(), ndir * 4 * H, M, ndir * 4 * H, w.kin, ACT_NONE, 1.f, nullptr, 0);
        if (st) {
            for (int i = 0; i < nseq; ++i) {
                const float* hc = st[i]->cnn.as<float>() + (size_t)l * 2 * H;
                HIPCHK(hipMemcpyAsync(hbuf + (size_t)i * H, hc, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(cbuf + (size_t)i * H, hc + H, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
            }
        } else {
            HIPCHK(hipMemsetAsync(hbuf, 0, sizeof(float) * hsz, s));
            HIPCHK(hipMemsetAsync(cbuf, 0, sizeof(float) * hsz, s));
        }
        // (the whole sequence of a layer as one cooperative launch with W_hh resident in registers and a barrier in global memory
        //  per step was built and measured in round 6: 10.0 against 6.0 ms at B = 1, 26.4 against 25.7 ms at B = 32 --
        //  tools/studies/lstm_seq_study.hip)
        for (int step = 0; step < Tq; ++step)
            launch_lstm_step(e->gx.as<float>(), w.whh, hbuf + (size_t)(step & 1) * hsz, hbuf + (size_t)((step + 1) & 1) * hsz,
                             cbuf, e->rnn_out.as<float>(), xl, nseq, Tq, H, step, ndir, s);
        if (st) {
            const float* hfin = hbuf + (size_t)(Tq & 1) * hsz;
            for (int i = 0; i < nseq; ++i) {
                float* hc = st[i]->cnn.as<float>() + (size_t)l * 2 * H;
                HIPCHK(hipMemcpyAsync(hc, hfin + (size_t)i * H, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(hc + H, cbuf + (size_t)i * H, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
            }
        }
        float* out = l == L - 1 ? enc_out : e->ln.as<float>();
        launch_layernorm_generic(e->rnn_out.as<float>(), w.ln_w, w.ln_b, out, M, D, 1e-5f, s);
        in = out;
    }
    LAUNCHCHK();
    if (Tq_out) *Tq_out = Tq;
    return 0;
}

extern "C" {

int masr_encode_full(masr_engine* e, const float* feats_dev, const int32_t* feat_lens_dev, int32_t B, int32_t T,
                     int32_t decoding_chunk_size, float* enc_out_dev, void* stream) {
    if (!e || !e->finalized) return fail("engine not finalized");
    ENTER(e);
    CallGuard call_guard(e, (hipStream_t)stream);
    if (B <= 0) return fail("empty batch");
    hipStream_t s = (hipStream_t)stream;
    if (e->cfg.model_kind == 3) return ds2_forward(e, s, feats_dev, feat_lens_dev, B, T, enc_out_dev, nullptr, nullptr);
    // decoding_chunk_size > 0 limits the attention to chunks in the streaming-trained builds only (use_dynamic_chunk follows
    // `streaming`, model.py of every family; utils/mask.py:117-143 ignores it otherwise)
    const int chunk = (decoding_chunk_size > 0 && e->cfg.causal) ? decoding_chunk_size : 0;
    if (e->cfg.model_kind == 1) return encode_full_squeezeformer(e, s, feats_dev, feat_lens_dev, B, T, enc_out_dev, chunk);
    if (e->cfg.model_kind == 2) return encode_full_efficient(e, s, feats_dev, feat_lens_dev, B, T, enc_out_dev, chunk);
    const int d = e->cfg.d_model, H = e->cfg.heads, pad = e->cfg.cnn_kernel - 1;
    int Tq = 0;
    CHK(embed(e, s, feats_dev, B, T, &Tq));
    if (Tq >= e->cfg.max_pos) return fail("sequence longer than max_pos");   // embedding.py:48-50 assert
    const int M = B * Tq;
    CHK(ensure_layer_ws(e, B, Tq));
    CHK(e->attseq.ensure(sizeof(AttSeq) * B));
    float* x = e->x.as<float>();
    EncodeCtx ctx{B, Tq, feat_lens_dev};
    launch_attseq_full(e->attseq.as<AttSeq>(), e->qkv.as<float>(), e->att.as<float>(), feat_lens_dev, B, Tq, 4, s);
    if (!e->cfg.causal)     // symmetric conv: the (K-1)/2 pad rows on both sides of every sequence stay zero for all layers
        HIPCHK(hipMemsetAsync(e->glu.p, 0, (size_t)B * (Tq + pad) * d * sizeof(float), s));
    const LayerW* prev = nullptr;                 // layer whose norm_final is still pending (it rides on the next FFN launch)
    const bool few_rows = g_few_rows_path && (M + 31) / 32 < std::min(rowgemm_small_blocks(), g_ffn_split_blocks);
    for (const LayerW& w : e->layers) {
        // first macaron FFN with the attention block's LayerNorm + fused QKV projection as its tail stage (full kernel only)
        const FfnTail tail{w.ln_mha_w, w.ln_mha_b, w.wqkv, w.bqkv, e->qkv.as<float>(), 3 * d, 3 * d,
                           prev ? prev->ln_fin_w : nullptr, prev ? prev->ln_fin_b : nullptr};
        bool qkv_done = false;
        CHK(ffn(e, s, M, w.ln_ffm_w, w.ln_ffm_b, w.ffm_w1, w.ffm_b1, w.ffm_w2, w.ffm_b2, 0.5f, 0, nullptr, nullptr, nullptr, &tail,
                &qkv_done));
        if (!qkv_done) mhsa(e, s, w, M);
        // attention and the chain kernel behind it as ONE launch (32 queries x all four heads per workgroup; key 34 = 0: two launches)
        const bool fuse_ac = g_attn_chain && !few_rows && !g_no_chain && !e->conv_bn && H == 4 && d == 256 && g_rowgemm_packed;
        if (fuse_ac) {
            AttnChainArgs a{};
            a.seqs = e->attseq.as<AttSeq>(); a.nseq = B; a.q_stride = 3 * d; a.kv_stride = 3 * d;
            a.chunk_size = (decoding_chunk_size > 0 && e->cfg.causal) ? decoding_chunk_size : 0; a.pos_stride = 1;
            a.ptab = w.ptab; a.bias_u = w.pos_u; a.bias_v = w.pos_v;
            a.Wp = packed_rows_of(e, w.chain_w, 3 * d, s); a.bias = w.chain_b; a.lnw = w.ln_conv_w; a.lnb = w.ln_conv_b;
            a.R = x; a.R2 = x; a.C = e->glu.as<float>(); a.lens = feat_lens_dev; a.seq_t = Tq; a.mstride = 4;
            a.out_pad_l = e->cfg.causal ? pad : pad / 2; a.out_pad_tot = pad; a.eps = 1e-5f;
            if (!a.Wp) return fail("packed chain weights: allocation failed");
            ProfScope ps(e, s, PROF_ATT, 6.0 * d * (double)Tq * Tq * B + 2.0 * M * (double)(3 * d) * d);
            launch_attn_chain(a, Tq, s);
        } else {
            ProfScope ps(e, s, PROF_ATT, 6.0 * d * (double)Tq * Tq * B);
            launch_attention(e->attseq.as<AttSeq>(), B, Tq, H, 3 * d, 3 * d, w.ptab, w.pos_u, w.pos_v,
                             (decoding_chunk_size > 0 && e->cfg.causal) ? decoding_chunk_size : 0, 1, s);   // use_dynamic_chunk only in the streaming build
        }
        if (few_rows) {
            // few row blocks (one utterance, a handful of short ones): latency-cut kernels of the chunk steps -- the row-block
            // chain kernel would put [out-proj -> LN -> pw1] of a row block on ONE workgroup (28.6 us at 7 row blocks against
            // 7.4 + 7.7 us for the two K-split launches whose columns spread over the chip); norm_final rides on the split
            // FFN's reduction
            mhsa_out(e, s, w, M);
            const bool fuse = g_split_head && e->cfg.cnn_kernel == 15 && g_ffn_packed >= 2 && !g_no_ffn_head && !e->conv_bn;
            CHK(conv_module(e, s, w, ctx, false, 0, 4, false, fuse));
            const FfnHead head{e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, e->cfg.causal ? w.gconst : nullptr,
                               w.pw2_w, w.pw2_b, feat_lens_dev, Tq, e->cfg.cnn_kernel, 4, nullptr};
            CHK(ffn(e, s, M, w.ln_ff_w, w.ln_ff_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2, 0.5f, 0, w.ln_fin_w, w.ln_fin_b, x, nullptr,
                    nullptr, fuse ? &head : nullptr));
            continue;                              // (prev stays null: nothing deferred)
        } else if (g_no_chain || e->conv_bn) {      // (BatchNorm build: the fused head stage carries the LayerNorm variant only)
            mhsa_out(e, s, w, M);
            CHK(conv_module(e, s, w, ctx, false));
            CHK(ffn(e, s, M, w.ln_ff_w, w.ln_ff_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2));
        } else {
            // out-projection + residual + LayerNorm + pointwise_conv1 + GLU in one kernel; the rest of the conv module
            // (depthwise conv, LayerNorm, SiLU, pointwise_conv2, residual) is the head stage of the second FFN kernel
            if (!fuse_ac) mhsa_out_pw1(e, s, w, ctx);
            const FfnHead head{e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, e->cfg.causal ? w.gconst : nullptr,
                               w.pw2_w, w.pw2_b, feat_lens_dev, Tq, e->cfg.cnn_kernel, 4, nullptr};
            CHK(ffn(e, s, M, w.ln_ff_w, w.ln_ff_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2, 0.5f, 0, nullptr, nullptr, nullptr, nullptr,
                    nullptr, &head));
        }
        prev = &w;                                 // norm_final deferred to the next layer's first FFN launch
    }
    if (prev) launch_layernorm(x, prev->ln_fin_w, prev->ln_fin_b, x, M, 1e-5f, 0, 0, nullptr, s);
    launch_layernorm(x, e->after_w, e->after_b, enc_out_dev, M, 1e-5f, 0, 0, nullptr, s);
    LAUNCHCHK();
    return 0;
}

static int g_ctc_fused_blocks = 160;     // masr_debug_set key 27: row blocks from which the fused CTC head (rowgemm EPI_CTC) runs
static int ctc_head(masr_engine* e, const float* enc_dev, int M, float* probs_dev, int write_probs, int32_t* argmax_dev,
                    float* maxprob_dev, hipStream_t s) {
    const int d = enc_dim(e), V = e->cfg.vocab_size;
    float* logits = probs_dev;
    if (!logits) {
        CHK(e->logits.ensure((size_t)M * V * sizeof(float)));
        logits = e->logits.as<float>();
    }
    gemm(e, s, enc_dev, d, e->ctc_w, e->ctc_b, logits, V, M, V, d, ACT_NONE, 1.f, nullptr, 0);
    launch_softmax_argmax(logits, M, V, V, write_probs, argmax_dev, maxprob_dev, s);
    LAUNCHCHK();
    return 0;
}

int masr_ctc_probs(masr_engine* e, const float* enc_dev, int32_t M, float* probs_dev, int32_t* argmax_dev,
                   float* maxprob_dev, void* stream) {
    if (!e || !e->finalized) return fail("engine not finalized");
    ENTER(e);
    CallGuard call_guard(e, (hipStream_t)stream);
    if (!probs_dev) return fail("probs_dev is null");
    if (e->cfg.vocab_size > 16384) return fail("vocab_size > 16384 is not supported by the softmax / pruning kernels (one 256-thread workgroup holds a row in registers)");
    return ctc_head(e, enc_dev, M, probs_dev, 1, argmax_dev, maxprob_dev, (hipStream_t)stream);
}

int masr_ctc_greedy_frames(masr_engine* e, const float* enc_dev, int32_t M, int32_t* argmax_dev, float* maxprob_dev,
                           void* stream) {
    if (!e || !e->finalized) return fail("engine not finalized");
    ENTER(e);
    CallGuard call_guard(e, (hipStream_t)stream);
    if (e->cfg.model_kind == 3)      // K = 1024 / 2048 rows: generic GEMM + softmax statistics (logits stay in a workspace)
        return ctc_head(e, enc_dev, M, nullptr, 0, argmax_dev, maxprob_dev, (hipStream_t)stream);
    // few row blocks (one utterance: 7; the Efficient Conformer's half-rate output at 32 x 10 s: 124): the fused head gives a
    // workgroup 32 rows x the WHOLE vocabulary (165 us whether 7 or 248 row blocks run); below g_ctc_fused_blocks the logits go
    // through the tiled GEMM (vocabulary spread over the CUs) into a workspace and the softmax statistics are a second launch
    if ((M + 31) / 32 < g_ctc_fused_blocks) return ctc_head(e, enc_dev, M, nullptr, 0, argmax_dev, maxprob_dev, (hipStream_t)stream);
    // fused: logits GEMM + online softmax statistics + argmax, nothing but (idx, prob) leaves the chip
    rowgemm(e, (hipStream_t)stream, RG_PRO_PLAIN, RG_EPI_CTC, enc_dev, e->cfg.d_model, nullptr, nullptr, e->ctc_w,
            e->ctc_b, nullptr, 0, M, e->cfg.vocab_size, nullptr, 0, 1.f, nullptr, 0, 0, 0, argmax_dev, maxprob_dev);
    LAUNCHCHK();
    return 0;
}

int masr_ctc_collapse(masr_engine* e, const int32_t* argmax_dev, const float* maxprob_dev, const int32_t* n_frames_dev,
                      int32_t B, int32_t Tp, int32_t blank, int32_t* tokens_dev, int32_t* n_tokens_dev,
                      float* score_dev, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    launch_ctc_collapse(argmax_dev, maxprob_dev, n_frames_dev, B, Tp, blank, tokens_dev, n_tokens_dev, score_dev,
                        (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

int masr_argmax_rows(masr_engine* e, const float* probs_dev, int32_t M, int32_t V, int32_t* argmax_dev,
                     float* maxprob_dev, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    if (V > 16384) return fail("V > 16384 not supported (one 256-thread workgroup holds a row in registers)");
    launch_argmax_rows(probs_dev, M, V, argmax_dev, maxprob_dev, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

int masr_ctc_topk(masr_engine* e, const float* probs_dev, int32_t M, int32_t V, int32_t top_n, float cutoff_prob,
                  int32_t* idx_dev, float* logp_dev, int32_t* count_dev, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    if (V > 16384) return fail("V > 16384 not supported (one 256-thread workgroup holds a row in registers)");
    if (top_n <= 0) return fail("top_n must be positive");
    launch_topk_prune(probs_dev, M, V, top_n, cutoff_prob, idx_dev, logp_dev, count_dev, -1, nullptr, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

int masr_ctc_topk_blank(masr_engine* e, const float* probs_dev, int32_t M, int32_t V, int32_t top_n, float cutoff_prob,
                        int32_t blank, int32_t* idx_dev, float* logp_dev, int32_t* count_dev, float* blank_logp_dev,
                        void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    if (V > 16384) return fail("V > 16384 not supported (one 256-thread workgroup holds a row in registers)");
    if (top_n <= 0) return fail("top_n must be positive");
    if (blank < 0 || blank >= V || !blank_logp_dev) return fail("masr_ctc_topk_blank: blank id out of range or null output");
    launch_topk_prune(probs_dev, M, V, top_n, cutoff_prob, idx_dev, logp_dev, count_dev, blank, blank_logp_dev,
                      (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

static int g_beam_narrow = 1;      // masr_debug_set key 37: 0 = every frame of the GPU prefix search on the wide (1024-thread) step (A/B, tests)
static int g_beam_lm_cache = 1;    // masr_debug_set key 32: 0 = the GPU prefix search probes the scorer once per (prefix, candidate) pair (A/B)
static int bind_lm(BeamGpuArgs& a, masr_lm* lm, float alpha, float beta) {
    a.use_lm = 0;
    a.lm_cache = g_beam_lm_cache;
    a.narrow = g_beam_narrow;
    a.alpha = alpha;
    a.beta = beta;
    if (!lm) return 0;
    if (lm_device_view(lm, &a.lm)) return fail(masr_lm_last_error());
    a.use_lm = 1;
    return 0;
}

int masr_beam_search_gpu(masr_engine* e, const int32_t* idx_dev, const float* logp_dev, const int32_t* count_dev,
                         const int32_t* frames_dev, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size, int32_t blank,
                         int32_t* tokens_dev, int32_t max_len, int32_t* len_dev, float* score_dev, void* stream) {
    return masr_beam_search_gpu_lm(e, idx_dev, logp_dev, count_dev, frames_dev, B, T_stride, K, beam_size, blank, nullptr, 0.f,
                                   0.f, nullptr, tokens_dev, max_len, len_dev, score_dev, stream);
}

int masr_beam_search_gpu_lm(masr_engine* e, const int32_t* idx_dev, const float* logp_dev, const int32_t* count_dev,
                            const int32_t* frames_dev, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size, int32_t blank,
                            masr_lm* lm, float alpha, float beta, const float* blank_logp_dev, int32_t* tokens_dev,
                            int32_t max_len, int32_t* len_dev, float* score_dev, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    if (B <= 0 || T_stride <= 0) return fail("empty batch");
    if (e->cfg.vocab_size > 16384 && e->finalized) return fail("vocab_size > 16384 not supported");
    BeamGpuArgs a{};
    a.cidx = idx_dev; a.clp = logp_dev; a.ccount = count_dev; a.frames = frames_dev;
    a.blank_lp = lm ? blank_logp_dev : nullptr;
    a.T_stride = T_stride; a.K = K; a.beam = beam_size; a.blank = blank; a.max_len = max_len;
    a.pool_cap = T_stride * beam_size + 1;
    if (lm && lm_word_based(lm))
        return fail("word-based language models are searched on host threads (masr_beam_search_batch_lm): the spelling "
                    "dictionary is not in the GPU kernel");
    if (K > 64 || beam_size > 512 || beam_size < 1 || beam_gpu_lds_bytes(beam_size, K, lm != nullptr) > 160 * 1024)
        return fail("beam search on the GPU needs cutoff_top_n <= 64, beam_size <= 512 and beam_size*cutoff_top_n*4 B + tables "
                    "within 160 KB of LDS; use masr_beam_search_batch (host threads) beyond that");
    DevBuf *pool = &e->beam_pool, *state = &e->beam_state;
    if (!e->beam_first_set) {
        e->beam_first_set = true;
        e->beam_first_stream = stream;
    } else if (stream != e->beam_first_stream) {
        auto& ws = e->beam_ws[stream];
        pool = &ws.first;
        state = &ws.second;
    }
    CHK(pool->ensure((size_t)B * a.pool_cap * 2 * sizeof(int)));
    CHK(state->ensure((size_t)B * beam_state_bytes(beam_size)));
    a.pool_parent = pool->as<int>();
    a.pool_ch = a.pool_parent + (size_t)B * a.pool_cap;
    beam_state_carve(state->p, B, beam_size, &a.state_h, &a.state_i, &a.state_f);
    CHK(bind_lm(a, lm, alpha, beta));
    a.init = 1;
    a.prof = e->beam_prof;
    a.tokens = tokens_dev; a.len = len_dev; a.score = score_dev;
    if (launch_beam_search(a, B, (hipStream_t)stream)) return fail("beam search launch rejected the sizes");
    LAUNCHCHK();
    return 0;
}

// ---- streaming beam search on the device (BeamSearchDecoder.decode_chunk / reset_decoder) -----------------------------
static void gbeam_args(GBeam& g, BeamGpuArgs& a) {
    a.pool_cap = g.cap;
    a.pool_parent = g.pool.as<int>();
    a.pool_ch = a.pool_parent + g.cap;
    beam_state_carve(g.state.p, 1, g.beam, &a.state_h, &a.state_i, &a.state_f);
}

int masr_gbeam_open(masr_engine* e, int32_t beam_size, int32_t blank, int32_t max_frames, int32_t* handle) {
    if (!e || !handle) return fail("null argument");
    ENTER(e);
    if (beam_size < 1 || beam_size > 512) return fail("beam_size must be in [1, 512]");
    if (max_frames <= 0) max_frames = 5000;
    int id = -1;
    for (size_t i = 0; i < e->gbeams.size(); ++i)
        if (!e->gbeams[i].open) { id = (int)i; break; }
    if (id < 0) {
        e->gbeams.emplace_back();
        id = (int)e->gbeams.size() - 1;
    }
    GBeam& g = e->gbeams[id];
    g.beam = beam_size;
    g.blank = blank;
    g.cap = max_frames * beam_size + 1;
    CHK(g.pool.ensure((size_t)g.cap * 2 * sizeof(int)));
    CHK(g.state.ensure(beam_state_bytes(beam_size)));
    g.lm = nullptr;
    g.alpha = g.beta = 0.f;
    g.open = true;
    g.started = false;
    *handle = id;
    return 0;
}

static int gbeam_of(masr_engine* e, int id, GBeam** out) {
    if (!e) return fail("null engine");
    if (id < 0 || id >= (int)e->gbeams.size() || !e->gbeams[id].open) return fail("bad beam handle");
    *out = &e->gbeams[id];
    return 0;
}

int masr_gbeam_set_lm(masr_engine* e, int32_t handle, masr_lm* lm, float alpha, float beta) {
    GBeam* g;
    CHK(gbeam_of(e, handle, &g));
    if (g->started) return fail("the language model of a stream can only change between utterances (after masr_gbeam_reset)");
    if (lm && lm_word_based(lm)) return fail("word-based language models are searched on host threads (masr_beam_*)");
    g->lm = lm;
    g->alpha = alpha;
    g->beta = beta;
    return 0;
}

int masr_gbeam_reset(masr_engine* e, int32_t handle) {
    GBeam* g;
    CHK(gbeam_of(e, handle, &g));
    g->started = false;
    return 0;
}

int masr_gbeam_close(masr_engine* e, int32_t handle) {
    GBeam* g;
    CHK(gbeam_of(e, handle, &g));
    ENTER(e);
    HIPCHK(hipDeviceSynchronize());
    g->pool.release();
    g->state.release();
    g->open = false;
    return 0;
}

int masr_gbeam_advance(masr_engine* e, int32_t handle, const int32_t* idx_dev, const float* logp_dev,
                       const int32_t* count_dev, int32_t T, int32_t K, int32_t* tokens_dev, int32_t max_len,
                       int32_t* len_dev, float* score_dev, void* stream) {
    return masr_gbeam_advance_lm(e, handle, idx_dev, logp_dev, count_dev, nullptr, T, K, tokens_dev, max_len, len_dev, score_dev,
                                 stream);
}

int masr_gbeam_advance_lm(masr_engine* e, int32_t handle, const int32_t* idx_dev, const float* logp_dev,
                          const int32_t* count_dev, const float* blank_logp_dev, int32_t T, int32_t K, int32_t* tokens_dev,
                          int32_t max_len, int32_t* len_dev, float* score_dev, void* stream) {
    GBeam* g;
    CHK(gbeam_of(e, handle, &g));
    ENTER(e);
    if (T < 0 || K > 64 || beam_gpu_lds_bytes(g->beam, K, g->lm != nullptr) > 160 * 1024)
        return fail("unsupported chunk / cutoff_top_n");
    BeamGpuArgs a{};
    a.cidx = idx_dev; a.clp = logp_dev; a.ccount = count_dev; a.frames = nullptr;
    a.blank_lp = g->lm ? blank_logp_dev : nullptr;
    a.T_stride = T; a.K = K; a.beam = g->beam; a.blank = g->blank; a.max_len = max_len;
    gbeam_args(*g, a);
    CHK(bind_lm(a, g->lm, g->alpha, g->beta));
    a.init = g->started ? 0 : 1;
    a.prof = nullptr;
    a.tokens = tokens_dev; a.len = len_dev; a.score = score_dev;
    // (the kernel stops allocating trie nodes at the pool capacity; max_frames bounds the utterance length)
    if (launch_beam_search(a, 1, (hipStream_t)stream)) return fail("beam search launch rejected the sizes");
    LAUNCHCHK();
    g->started = true;
    return 0;
}

int masr_fbank_batch(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev,
                     int32_t B, int32_t n_max, int32_t use_db_normalization, float target_db, float* feats_dev,
                     int32_t* n_frames_dev, int16_t* norm_pcm_dev, float* gain_dev, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    if (sample_format != 0 && sample_format != 1) return fail("sample_format must be 0 (int16) or 1 (float32)");
    hipStream_t s = (hipStream_t)stream;
    const int T_max = n_max >= 400 ? 1 + (n_max - 400) / 160 : 0;
    CHK(e->gain.ensure(sizeof(float) * fbank_gain_scratch_floats(B)));
    float* gain = e->gain.as<float>();
    if (use_db_normalization == 2) {      // gains evaluated by the caller (masr_mean_square + the host's numpy, audio.py:287-304)
        if (!gain_dev) return fail("use_db_normalization = 2 needs the gains in gain_dev");
        HIPCHK(hipMemcpyAsync(gain, gain_dev, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    }
    {
        ProfScope ps(e, s, PROF_FBANK, 0.0);
        launch_fbank(samples_dev, sample_format, n_samples_dev, B, n_max, use_db_normalization, target_db, e->fbank_tables(),
                     feats_dev, T_max, gain, norm_pcm_dev, s);
    }
    if (gain_dev && use_db_normalization == 1)
        HIPCHK(hipMemcpyAsync(gain_dev, gain, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (n_frames_dev) launch_frame_counts(n_samples_dev, B, n_frames_dev, nullptr, 0, s);
    LAUNCHCHK();
    return 0;
}

int masr_mean_square(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                     int32_t n_max, float* mean_square_dev, void* stream) {
    if (!e || !mean_square_dev) return fail("null argument");
    ENTER(e);
    if (sample_format != 0 && sample_format != 1) return fail("sample_format must be 0 (int16) or 1 (float32)");
    DevBuf& ws = e->ms_ws[stream];
    CHK(ws.ensure(sizeof(float) * fbank_gain_scratch_floats(B)));
    launch_mean_square(samples_dev, sample_format, n_samples_dev, B, n_max, ws.as<float>(), mean_square_dev,
                       (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

// kaldi.mfcc(num_mel_bins=80, num_ceps=n_ceps) on top of the fbank front-end (audio_featurizer.py:98-117).  The DCT / lifter
// tables follow torchaudio's float32 construction (functional.create_dct(norm='ortho') with column 0 = sqrt(1/80), transposed;
// lifter 1 + 0.5 * 22 * sin(pi * i / 22)).
int masr_mfcc_batch(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                    int32_t n_max, int32_t use_db_normalization, float target_db, int32_t n_ceps, float* mfcc_dev,
                    int32_t* n_frames_dev, float* gain_dev, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    if (n_ceps <= 0 || n_ceps > 80) return fail("n_ceps must be in [1, 80] (num_ceps <= num_mel_bins)");
    hipStream_t s = (hipStream_t)stream;
    if (e->dct_ceps != n_ceps) {
        std::vector<float> dct((size_t)80 * n_ceps), lif(n_ceps);
        const float step = (float)(M_PI / 80.0);
        for (int m = 0; m < 80; ++m)
            for (int c = 0; c < n_ceps; ++c) {
                float v = cosf((step * ((float)m + 0.5f)) * (float)c);
                if (c == 0) v *= (float)(1.0 / sqrt(2.0));
                v *= (float)sqrt(2.0 / 80.0);
                if (c == 0) v = (float)sqrt(1.0 / 80.0);
                dct[(size_t)m * n_ceps + c] = v;
            }
        for (int c = 0; c < n_ceps; ++c) lif[c] = 1.0f + 11.0f * sinf(((float)M_PI * (float)c) / 22.0f);
        HIPCHK(hipStreamSynchronize(s));
        CHK(upload(e, dct, &e->dct));           // (a changed n_ceps leaves the old table to the engine's owned list)
        CHK(upload(e, lif, &e->lifter));
        e->dct_ceps = n_ceps;
    }
    const int T_max = n_max >= 400 ? 1 + (n_max - 400) / 160 : 0;
    CHK(e->fb_scratch.ensure((size_t)B * std::max(T_max, 1) * 80 * sizeof(float)));
    CHK(masr_fbank_batch(e, samples_dev, sample_format, n_samples_dev, B, n_max, use_db_normalization, target_db,
                         e->fb_scratch.as<float>(), n_frames_dev, nullptr, gain_dev, stream));
    launch_mfcc(e->fb_scratch.as<float>(), (long)B * T_max, n_ceps, e->dct, e->lifter, mfcc_dev, s);
    LAUNCHCHK();
    return 0;
}

// AudioFeaturizer._compute_linear (audio_featurizer.py:73-95) for a padded batch: float32 (or int16 / 2^15) samples, optional
// dB normalisation (the same gain as the fbank path), [B, T, 161] log power spectra with T = (n_max - 320) / 160 + 1.
int masr_linear_batch(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                      int32_t n_max, int32_t use_db_normalization, float target_db, float* feats_dev, int32_t* n_frames_dev,
                      float* gain_dev, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    if (sample_format != 0 && sample_format != 1) return fail("sample_format must be 0 (int16) or 1 (float32)");
    hipStream_t s = (hipStream_t)stream;
    if (!e->lin_win) {
        std::vector<double> win(320), tw(640);
        double sumsq = 0.0;
        for (int i = 0; i < 320; ++i) {                 // np.hanning(320): 0.5 + 0.5 * cos(pi * (1 - M + 2 i) / (M - 1))
            win[i] = 0.5 + 0.5 * cos(M_PI * (double)(1 - 320 + 2 * i) / 319.0);
            sumsq += win[i] * win[i];
            tw[2 * i] = cos(2.0 * M_PI * i / 320.0);
            tw[2 * i + 1] = -sin(2.0 * M_PI * i / 320.0);
        }
        CHK(upload(e, win, &e->lin_win));
        CHK(upload(e, tw, &e->lin_tw));
        e->lin_scale = sumsq * 16000.0;
    }
    const int T_max = n_max >= 320 ? (n_max - 320) / 160 + 1 : 0;
    CHK(e->gain.ensure(sizeof(float) * fbank_gain_scratch_floats(B)));
    float* gain = e->gain.as<float>();
    if (use_db_normalization == 2) {
        if (!gain_dev) return fail("use_db_normalization = 2 needs the gains in gain_dev");
        HIPCHK(hipMemcpyAsync(gain, gain_dev, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    } else if (use_db_normalization)          // T_max = 0: only the RMS -> gain kernels of the fbank front-end run
        launch_fbank(samples_dev, sample_format, n_samples_dev, B, n_max, 1, target_db, e->fbank_tables(), nullptr, 0, gain, nullptr, s);
    launch_linear_spec(samples_dev, sample_format, n_samples_dev, B, n_max, use_db_normalization, gain, e->lin_win, e->lin_tw,
                       e->lin_scale, feats_dev, T_max, s);
    if (gain_dev && use_db_normalization == 1)
        HIPCHK(hipMemcpyAsync(gain_dev, gain, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (n_frames_dev) launch_linear_frame_counts(n_samples_dev, B, n_frames_dev, s);
    LAUNCHCHK();
    return 0;
}

// the offline hot path of one padded batch: samples -> features -> encoder -> CTC greedy -> collapse, either into three
// arrays (masr_transcribe_batch) or into packed rows (masr_transcribe_rows)
static int transcribe_impl(masr_engine* e, const void* samples_dev, int32_t fmt, const int32_t* n_samples_dev, int32_t B,
                           int32_t n_max, int32_t use_db_normalization, float target_db, const float* gain_dev,
                           int32_t decode_all_frames, int32_t* tokens_dev, int32_t* n_tokens_dev, float* score_dev,
                           int32_t* rows_dev, void* stream) {
    if (!e || !e->finalized) return fail("engine not finalized");
    ENTER(e);
    CallGuard call_guard(e, (hipStream_t)stream);
    if (n_max < 400) return fail("n_max < 400 samples: no frame");
    hipStream_t s = (hipStream_t)stream;
    const int d = enc_dim(e), F = e->cfg.n_mels;
    const int T = 1 + (n_max - 400) / 160, T1 = (T - 1) / 2, Tsub = (T1 - 1) / 2;
    const bool halved = e->cfg.model_kind == 2 && e->stride_idx >= 0;      // efficient conformer: one more stride-2 stage
    const int Tq = halved ? (Tsub + 1) / 2 : Tsub;
    if (Tsub <= 0) return fail("utterances too short");
    CHK(e->feats.ensure((size_t)B * T * F * sizeof(float)));
    CHK(e->nframes.ensure(sizeof(int) * 2 * B));
    CHK(e->enc.ensure((size_t)B * Tq * d * sizeof(float)));
    CHK(e->idx.ensure(sizeof(int) * B * Tq));
    CHK(e->maxp.ensure(sizeof(float) * B * Tq));
    int* nfr = e->nframes.as<int>();
    int* nenc = nfr + B;
    // (mode 2 reads the caller's gains: masr_fbank_batch copies them into its own scratch, the caller's array is not written)
    CHK(masr_fbank_batch(e, samples_dev, fmt, n_samples_dev, B, n_max, use_db_normalization, target_db,
                         e->feats.as<float>(), nullptr, nullptr, use_db_normalization == 2 ? const_cast<float*>(gain_dev) : nullptr,
                         stream));
    launch_frame_counts(n_samples_dev, B, nfr, nenc, halved ? 1 : 0, s);
    {
        // decode_all_frames: the padded frames are decoded too (the reference's batch evaluation, trainer.py:340) -- then they
        // must be computed, whatever masr_debug_set key 38 says
        const int keep = e->skip_padding;
        if (decode_all_frames) e->skip_padding = 0;
        const int rc = masr_encode_full(e, e->feats.as<float>(), nfr, B, T, -1, e->enc.as<float>(), stream);
        e->skip_padding = keep;
        if (rc) return rc;
    }
    CHK(masr_ctc_greedy_frames(e, e->enc.as<float>(), B * Tq, e->idx.as<int>(), e->maxp.as<float>(), stream));
    if (rows_dev) {
        launch_ctc_collapse_rows(e->idx.as<int>(), e->maxp.as<float>(), decode_all_frames ? nullptr : nenc, B, Tq, 0, rows_dev, s);
        LAUNCHCHK();
        return 0;
    }
    CHK(masr_ctc_collapse(e, e->idx.as<int>(), e->maxp.as<float>(), decode_all_frames ? nullptr : nenc, B, Tq, 0,
                          tokens_dev, n_tokens_dev, score_dev, stream));
    return 0;
}

int masr_transcribe_batch(masr_engine* e, const int16_t* pcm_dev, const int32_t* n_samples_dev, int32_t B,
                          int32_t n_max, int32_t use_db_normalization, float target_db, int32_t decode_all_frames,
                          int32_t* tokens_dev, int32_t* n_tokens_dev, float* score_dev, void* stream) {
    if (use_db_normalization == 2) return fail("masr_transcribe_batch: supplied gains go through masr_transcribe_rows");
    return transcribe_impl(e, pcm_dev, 0, n_samples_dev, B, n_max, use_db_normalization, target_db, nullptr, decode_all_frames,
                           tokens_dev, n_tokens_dev, score_dev, nullptr, stream);
}

int masr_transcribe_rows(masr_engine* e, const void* samples_dev, int32_t sample_format, const int32_t* n_samples_dev, int32_t B,
                         int32_t n_max, int32_t use_db_normalization, float target_db, const float* gain_dev,
                         int32_t decode_all_frames, int32_t* rows_dev, void* stream) {
    if (!rows_dev) return fail("null argument");
    if (sample_format != 0 && sample_format != 1) return fail("sample_format: 0 = int16 PCM, 1 = float32");
    if (use_db_normalization == 2 && !gain_dev) return fail("use_db_normalization = 2 needs the gains in gain_dev");
    return transcribe_impl(e, samples_dev, sample_format, n_samples_dev, B, n_max, use_db_normalization, target_db, gain_dev,
                           decode_all_frames, nullptr, nullptr, nullptr, rows_dev, stream);
}

// ---- streaming -----------------------------------------------------------------------------------------

int masr_stream_open(masr_engine* e, int32_t max_frames_out, int32_t* stream_id) {
    if (!e || !e->finalized) return fail("engine not finalized");
    ENTER(e);
    if (e->cfg.model_kind == 3) {
        if (!e->cfg.causal) return fail("deepspeech2: streaming needs the uni-directional model");
    } else if (!e->cfg.causal) {
        return fail("chunked streaming needs the streaming-trained (causal conv) build");
    }
    if (e->conv_bn) return fail("cnn_module_norm=batch_norm: only the full-context forward is implemented (the chunk-step kernels carry the LayerNorm variant)");
    if (max_frames_out <= 0 || max_frames_out > e->cfg.max_pos) max_frames_out = e->cfg.max_pos;
    int id = -1;
    for (size_t i = 0; i < e->streams.size(); ++i)
        if (!e->streams[i].open) { id = (int)i; break; }
    if (id < 0) {
        e->streams.emplace_back();
        id = (int)e->streams.size() - 1;
    }
    Stream& st = e->streams[id];
    const int d = e->cfg.d_model, L = e->cfg.num_blocks, pad = e->cfg.cnn_kernel - 1;
    st.cap = max_frames_out;
    if (e->cfg.model_kind == 3) {      // LSTM state (h, c) per layer; no attention cache, no frame limit
        st.cap = 1 << 30;
        CHK(st.cnn.ensure((size_t)L * 2 * d * sizeof(float)));
        CHK(clear_sync(st.cnn.p, 0, (size_t)L * 2 * d * sizeof(float)));
        st.offset = 0;
        st.open = true;
        *stream_id = id;
        return 0;
    }
    CHK(st.att.ensure((size_t)L * st.cap * 2 * d * sizeof(float)));
    CHK(st.cnn.ensure((size_t)L * pad * d * sizeof(float)));
    CHK(clear_sync(st.cnn.p, 0, (size_t)L * pad * d * sizeof(float)));
    CHK(st.cnn2.ensure((size_t)L * pad * d * sizeof(float)));
    CHK(clear_sync(st.cnn2.p, 0, (size_t)L * pad * d * sizeof(float)));
    if (e->cfg.model_kind == 2)     // planar caches of the grouped layers: rows behind the last key must read as zero
        CHK(clear_sync(st.att.p, 0, (size_t)L * st.cap * 2 * d * sizeof(float)));
    st.offset = 0;
    st.offset_r = 0;
    st.history = -1;
    st.cache_t1 = 0;
    st.first_half = 0;
    st.open = true;
    *stream_id = id;
    return 0;
}

static int stream_of(masr_engine* e, int id, Stream** out) {
    if (!e) return fail("null engine");
    if (id < 0 || id >= (int)e->streams.size() || !e->streams[id].open) return fail("bad stream id");
    *out = &e->streams[id];
    return 0;
}

int masr_stream_reset(masr_engine* e, int32_t stream_id) {
    Stream* st;
    CHK(stream_of(e, stream_id, &st));
    ENTER(e);
    const int d = e->cfg.d_model, L = e->cfg.num_blocks, pad = e->cfg.cnn_kernel - 1;
    HIPCHK(hipDeviceSynchronize());
    CHK(clear_sync(st->cnn.p, 0, (e->cfg.model_kind == 3 ? (size_t)L * 2 * d : (size_t)L * pad * d) * sizeof(float)));
    if (e->cfg.model_kind == 2) CHK(clear_sync(st->att.p, 0, (size_t)L * st->cap * 2 * d * sizeof(float)));
    st->offset = 0;
    st->offset_r = 0;
    st->cache_t1 = 0;
    st->first_half = 0;
    return 0;
}

int masr_stream_close(masr_engine* e, int32_t stream_id) {
    Stream* st;
    CHK(stream_of(e, stream_id, &st));
    ENTER(e);
    HIPCHK(hipDeviceSynchronize());
    st->att.release();
    st->cnn.release();
    st->cnn2.release();
    st->open = false;
    return 0;
}

int masr_stream_set_history(masr_engine* e, int32_t stream_id, int32_t required_cache_size) {
    Stream* st;
    CHK(stream_of(e, stream_id, &st));
    if (required_cache_size >= 0 && e->cfg.model_kind == 3)
        return fail("required_cache_size: DeepSpeech2 streams carry an LSTM state, not an attention cache");
    st->history = required_cache_size;
    return 0;
}

int masr_encoder_frames(masr_engine* e, int32_t feature_frames, int32_t* encoder_frames) {
    if (!e || !encoder_frames) return fail("null argument");
    const int T1 = (feature_frames - 1) / 2, Tsub = (T1 - 1) / 2;
    const bool halved = e->cfg.model_kind == 2 && e->stride_idx >= 0;
    *encoder_frames = feature_frames < 7 ? 0 : (halved ? (Tsub + 1) / 2 : Tsub);
    return 0;
}

int masr_engine_info(masr_engine* e, int32_t* device_id, int32_t* n_mels, int32_t* vocab_size) {
    if (!e) return fail("null engine");
    if (device_id) *device_id = e->cfg.device_id;
    if (n_mels) *n_mels = e->cfg.n_mels;
    if (vocab_size) *vocab_size = e->cfg.vocab_size;
    return 0;
}

int masr_stream_offset(masr_engine* e, int32_t stream_id, int32_t* offset) {
    Stream* st;
    CHK(stream_of(e, stream_id, &st));
    *offset = st->offset;
    return 0;
}

int masr_stream_room(masr_engine* e, int32_t stream_id, int32_t* frames_left) {
    if (!frames_left) return fail("null argument");
    Stream* st;
    CHK(stream_of(e, stream_id, &st));
    // (the Efficient-Conformer's grouped layers pad a chunk to a multiple of the group size before the room check of
    //  masr_encode_chunk: that slack is taken off HERE, for that family only, so that a caller compares plain frame counts)
    const int slack = e->cfg.model_kind == 2 ? std::max(0, e->group_size) : 0;
    *frames_left = std::max(0, st->cap - st->offset - slack);
    return 0;
}

int masr_stream_cache_len(masr_engine* e, int32_t stream_id, int32_t* cache_len) {
    Stream* st;
    CHK(stream_of(e, stream_id, &st));
    *cache_len = st->cache_t1;
    return 0;
}

}  // extern "C"

// forward_chunk's cache bookkeeping (conformer/encoder.py:390-410, squeezeformer/encoder.py:288-297,338-347,
// efficient_conformer/encoder.py:316-336,365-381), as a window into append-only caches: a chunk of T0 input-rate frames attends
// over the last cache_t1 cached frames + itself; afterwards the cache is cut at next_cache_start.
struct ChunkWindow {
    int cache_t1, first_full;    // cached frames attended over; absolute index of the first of them (= positional index of key 0)
    int next_cache_t1, ncs;      // after the step
};
static ChunkWindow chunk_window(const Stream& st, int T0) {
    ChunkWindow w;
    w.cache_t1 = st.cache_t1;
    w.first_full = st.offset - st.cache_t1;
    const int key_size = st.cache_t1 + T0;
    w.ncs = st.history < 0 ? 0 : st.history == 0 ? key_size : std::max(key_size - st.history, 0);
    w.next_cache_t1 = key_size - w.ncs;
    return w;
}
// half-rate layers: the reference stores their cache repeat-interleaved at the input rate and reads every second entry back, so a
// step sees `want` entries starting at first_half; they must be exactly the entries appended since (true for chunks of an even
// number of frames; the reference itself fails or drops the newest entry otherwise)
static int half_rate_window(const Stream& st, int want, int* first, int* count) {
    const int avail = st.offset_r - st.first_half;
    if (want != avail)
        return fail("bounded attention history: the half-rate cache window is not contiguous with the new frames (odd chunk lengths "
                    "with required_cache_size >= 0 are not supported)");
    *first = st.first_half;
    *count = avail;
    return 0;
}

// Squeezeformer chunk step (streaming-trained build), n streams in lock-step: SqueezeformerEncoder.forward_chunk
// (squeezeformer/encoder.py:240-362) with required_cache_size < 0.  Every layer keeps its key/value cache at its OWN frame
// rate (the reference stores the half-rate layers repeat-interleaved and reads every second entry back, :338-347), the cnn
// cache holds the last K-1 adaptive-scaled input rows of the conv module.  TimeReductionLayerStream (k = 1, stride 2) and the
// recovery are chunk-local.
static int encode_chunk_squeezeformer(masr_engine* e, hipStream_t s, std::vector<Stream*>& st, const float* feats, int Tc,
                                      float* probs_dev, int32_t* argmax_dev, float* maxprob_dev) {
    const int n = (int)st.size();
    const int d = e->cfg.d_model, H = e->cfg.heads, L = e->cfg.num_blocks, K = e->cfg.cnn_kernel, pad = K - 1;
    int T0 = 0;
    CHK(embed(e, s, feats, n, Tc, &T0));
    const int Tr = (T0 + 1) / 2;
    for (int i = 0; i < n; ++i)
        if (st[i]->offset + T0 > st[i]->cap) return fail("stream exceeds its max_frames_out / max_pos");
    CHK(ensure_layer_ws(e, n, T0));
    CHK(e->xsave.ensure((size_t)n * T0 * d * sizeof(float)));
    CHK(e->xred.ensure((size_t)n * Tr * d * sizeof(float)));
    CHK(e->attseq.ensure(sizeof(AttSeq) * (size_t)n * L));
    auto reduced = [&](int l) { return e->reduce_idx >= 0 && l >= e->reduce_idx && l < e->recover_idx; };
    std::vector<AttSeq> hs((size_t)n * L);
    std::vector<float*> hp((size_t)n * L * 2);    // cnn cache bases: current (read) | next (written)
    std::vector<ChunkWindow> win(n);
    std::vector<int> hfirst(n), hcount(n);
    for (int i = 0; i < n; ++i) {
        win[i] = chunk_window(*st[i], T0);
        // a half-rate layer reads pos_emb[::2] rows minus its queries: ceil((cache_t1 + T0) / 2) - Tr cached entries (encoder.py:338-342)
        CHK(half_rate_window(*st[i], (win[i].cache_t1 + T0 + 1) / 2 - Tr, &hfirst[i], &hcount[i]));
    }
    for (int l = 0; l < L; ++l) {
        const int Tl = reduced(l) ? Tr : T0;
        for (int i = 0; i < n; ++i) {
            AttSeq& a = hs[(size_t)l * n + i];
            const int off = reduced(l) ? st[i]->offset_r : st[i]->offset;
            const int first = reduced(l) ? hfirst[i] : win[i].first_full;       // first cache row attended over
            const int ncached = reduced(l) ? hcount[i] : win[i].cache_t1;
            float* cache = st[i]->att.as<float>() + (size_t)l * st[i]->cap * 2 * d;
            a.q = e->qkv.as<float>() + (size_t)i * Tl * 3 * d;
            a.k = cache + (size_t)first * 2 * d;
            a.v = a.k + d;
            a.out = e->att.as<float>() + (size_t)i * Tl * d;
            a.nq = Tl;
            a.nk = ncached + Tl;           // (the new rows are appended at k + (nk - nq) rows = cache row `off`)
            a.klen = a.nk;
            a.pos0 = win[i].first_full;    // key j of a half-rate layer sits at position first_full + 2 j (pos_emb[:, ::2])
            a.q_abs0 = off;
            a.pad_ = 0;
            hp[(size_t)l * n + i] = st[i]->cnn.as<float>() + (size_t)l * pad * d;
            hp[(size_t)(L + l) * n + i] = st[i]->cnn2.as<float>() + (size_t)l * pad * d;
        }
    }
    CHK(e->cnnptrs.ensure(sizeof(float*) * hp.size()));
    CHK(e->stage.begin(sizeof(AttSeq) * hs.size() + sizeof(float*) * hp.size() + 64));
    CHK(e->stage.push(e->attseq.p, hs.data(), sizeof(AttSeq) * hs.size(), s));
    CHK(e->stage.push(e->cnnptrs.p, hp.data(), sizeof(float*) * hp.size(), s));
    CHK(e->stage.end(s));
    float* x = e->x.as<float>();
    launch_layernorm(x, e->preln_w, e->preln_b, x, n * T0, 1e-5f, 0, 0, nullptr, s);
    CHK(e->enc.ensure((size_t)n * T0 * d * sizeof(float)));
    int Tq = T0;
    for (int l = 0; l < L; ++l) {
        const SqLayerW& w = e->sq_layers[l];
        if (l == e->reduce_idx) {
            HIPCHK(hipMemcpyAsync(e->xsave.p, x, (size_t)n * T0 * d * sizeof(float), hipMemcpyDeviceToDevice, s));
            launch_time_reduce_dw(x, e->tr_dw_w, e->tr_dw_b, nullptr, e->xred.as<float>(), n, T0, 4, s);
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_STORE, e->xred.as<float>(), d, nullptr, nullptr, e->tr_pw_w, e->tr_pw_b, x, d,
                    n * Tr, d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
            Tq = Tr;
        }
        if (l == e->recover_idx && e->reduce_idx >= 0) {
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_STORE, x, d, nullptr, nullptr, e->rec_w, e->rec_b, e->xred.as<float>(), d,
                    n * Tq, d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
            launch_recover_add(e->xsave.as<float>(), e->xred.as<float>(), x, n, T0, Tq, s);
            Tq = T0;
        }
        const int M = n * Tq;
        const AttSeq* seqs = e->attseq.as<AttSeq>() + (size_t)l * n;
        rowgemm(e, s, RG_PRO_AFFINE, RG_EPI_STORE, x, d, w.att_s, w.att_b, w.wqkv, w.bqkv, e->qkv.as<float>(), 3 * d, M,
                3 * d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr, PROF_GEMM, 4, 0, 0, 0, 0, 0, 0, 0, seqs, Tq);
        launch_attention(seqs, n, Tq, H, 3 * d, 2 * d, w.ptab, w.pos_u, w.pos_v, 0, reduced(l) ? 2 : 1, s);
        rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->att.as<float>(), d, nullptr, nullptr, w.wo, w.bo, x, d, M, d, x, d, 1.f,
                nullptr, 0, 0, 0, nullptr, nullptr);
        launch_layernorm(x, w.ln1_w, w.ln1_b, x, M, 1e-5f, 0, 0, nullptr, s);
        CHK(ffn(e, s, M, w.f1_s, w.f1_b, w.f1_w1, w.f1_b1, w.f1_w2, w.f1_b2, 1.0f, 1, w.ln2_w, w.ln2_b, x));
        // conv module: [cnn cache | ada(x)] -> pointwise_conv1 + GLU -> causal depthwise + BatchNorm + SiLU -> pointwise_conv2
        float* const* cptr = e->cnnptrs.as<float*>() + (size_t)l * n;
        {   // [cnn cache | ada_scale * x + ada_bias] -> pointwise_conv1 + GLU (+ the new cache) in one launch where the small-M
            // kernel applies, else history / affine rows first
            RowGemmArgs a{};
            a.A = x; a.lda = d; a.lnw = w.cv_s; a.lnb = w.cv_b; a.W = w.pw1_w; a.bias = w.pw1_b; a.C = e->glu.as<float>();
            a.ldc = d; a.M = n * (Tq + pad); a.N = 2 * d; a.alpha = 1.f; a.eps = 1e-5f; a.mstride = 4; a.seq_t = Tq; a.pad = pad;
            a.cache_rd = cptr; a.cache_wr = cptr + (size_t)L * n; a.hist_affine = 1;
            ProfScope ps(e, s, PROF_GEMM, 2.0 * a.M * (double)(2 * d) * d);
            if (!launch_rowgemm(a, RG_PRO_HIST, RG_EPI_GLU, s)) {
                launch_conv_hist(x, w.cv_s, w.cv_b, cptr, cptr + (size_t)L * n, e->lnpad.as<float>(), n, Tq, pad, 1, 1e-5f, s);
                a.A = e->lnpad.as<float>();
                launch_rowgemm(a, RG_PRO_PLAIN, RG_EPI_GLU, s);
            }
        }
        launch_dwconv_bn_silu(e->glu.as<float>(), w.dw_w, w.dw_b, w.bn_scale, w.bn_shift, e->dwo.as<float>(), n, Tq, K, s);
        rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->dwo.as<float>(), d, nullptr, nullptr, w.pw2_w, w.pw2_b, x, d, M, d, x,
                d, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
        launch_layernorm(x, w.ln3_w, w.ln3_b, x, M, 1e-5f, 0, 0, nullptr, s);
        CHK(ffn(e, s, M, w.f2_s, w.f2_b, w.f2_w1, w.f2_b1, w.f2_w2, w.f2_b2, 1.0f, 1, w.ln4_w, w.ln4_b,
                l == L - 1 ? e->enc.as<float>() : x));
    }
    if (e->cfg.vocab_size > 16384) return fail("vocab_size > 16384 is not supported by the softmax / pruning kernels (one 256-thread workgroup holds a row in registers)");
    CHK(ctc_head(e, e->enc.as<float>(), n * T0, probs_dev, probs_dev ? 1 : 0, argmax_dev, maxprob_dev, s));
    for (int i = 0; i < n; ++i) {
        st[i]->offset += T0;
        st[i]->offset_r += Tr;
        st[i]->cache_t1 = win[i].next_cache_t1;
        st[i]->first_half += win[i].ncs / 2;
        std::swap(st[i]->cnn, st[i]->cnn2);
    }
    return 0;
}

// Efficient-Conformer chunk step, n streams in lock-step: EfficientConformerEncoder.forward_chunk
// (efficient_conformer/encoder.py:267-392) with required_cache_size < 0.  Grouped layers keep PLANAR key / value caches at the
// input frame rate (the flat regrouping runs over cache + chunk), layers behind the stride layer keep interleaved k|v rows
// at half the rate (the reference stores them repeat-interleaved, :370); cnn caches hold each layer's own K-1 rows.
static int encode_chunk_efficient(masr_engine* e, hipStream_t s, std::vector<Stream*>& st, const float* feats, int Tc,
                                  float* probs_dev, int32_t* argmax_dev, float* maxprob_dev) {
    const int n = (int)st.size();
    const int d = e->cfg.d_model, H = e->cfg.heads, L = e->cfg.num_blocks, G = e->group_size;
    int T0 = 0;
    CHK(embed(e, s, feats, n, Tc, &T0));
    const int T2 = (T0 + 1) / 2, Tg = (T0 + G - 1) / G, Tpad = Tg * G;
    for (int i = 0; i < n; ++i)
        if (st[i]->offset + T0 + G > st[i]->cap) return fail("stream exceeds its max_frames_out / max_pos");
    CHK(ensure_layer_ws(e, n, T0));
    const size_t plane = (size_t)n * Tpad * d;
    CHK(e->qplanes.ensure(3 * plane * sizeof(float)));
    CHK(e->attp.ensure(plane * sizeof(float)));
    CHK(e->xsave.ensure((size_t)n * T2 * d * sizeof(float)));
    CHK(e->attseq.ensure(sizeof(AttSeq) * (size_t)n * L));
    float* qp = e->qplanes.as<float>();
    HIPCHK(hipMemsetAsync(qp, 0, 3 * plane * sizeof(float), s));      // time padding rows of the new q / k / v stay zero
    auto rate = [&](int l) { return l > e->stride_idx ? 2 : 1; };
    const int padmax = e->cfg.cnn_kernel - 1;
    std::vector<AttSeq> hs((size_t)n * L);
    std::vector<PlaneCopy> pcs((size_t)n * e->n_group_layers);
    std::vector<float*> hp((size_t)n * L * 2);   // cnn cache bases: current (read) | next (written)
    std::vector<ChunkWindow> win(n);
    std::vector<int> hfirst(n), hcount(n);
    for (int i = 0; i < n; ++i) {
        win[i] = chunk_window(*st[i], T0);
        // layers behind the stride layer read att_cache[:, :, ::2]: ceil(cache_t1 / 2) entries (encoder.py:353)
        CHK(half_rate_window(*st[i], (win[i].cache_t1 + 1) / 2, &hfirst[i], &hcount[i]));
    }
    for (int l = 0; l < L; ++l) {
        if (layer_grouped(e, l) && rate(l) != 1) return fail("grouped attention after the stride layer is not supported");
        for (int i = 0; i < n; ++i) {
            AttSeq& a = hs[(size_t)l * n + i];
            float* cache = st[i]->att.as<float>() + (size_t)l * st[i]->cap * 2 * d;
            if (layer_grouped(e, l)) {
                // the flat regrouping [T, 256] -> [T / 3, 4, 192] runs over (kept cache + chunk): it starts at the first kept row
                float* kpl = cache;
                float* vpl = cache + (size_t)st[i]->cap * d;
                a.q = qp + (size_t)i * Tpad * d;
                a.k = kpl + (size_t)win[i].first_full * d;
                a.v = vpl + (size_t)win[i].first_full * d;
                a.out = e->attp.as<float>() + (size_t)i * Tpad * d;
                a.nq = Tg;
                a.nk = (win[i].cache_t1 + T0 + G - 1) / G;
                a.klen = a.nk;
                a.pos0 = win[i].first_full;
                a.q_abs0 = 0;
                a.pad_ = win[i].cache_t1 + T0;                 // true number of keys (rows of P behind it read as zero)
                PlaneCopy& c = pcs[(size_t)l * n + i];
                c.src_k = qp + plane + (size_t)i * Tpad * d;
                c.src_v = qp + 2 * plane + (size_t)i * Tpad * d;
                c.dst_k = kpl + (size_t)st[i]->offset * d;
                c.dst_v = vpl + (size_t)st[i]->offset * d;
            } else {
                const int Tl = rate(l) == 2 ? T2 : T0;
                const int off = rate(l) == 2 ? st[i]->offset_r : st[i]->offset;
                const int first = rate(l) == 2 ? hfirst[i] : win[i].first_full;
                const int ncached = rate(l) == 2 ? hcount[i] : win[i].cache_t1;
                a.q = e->qkv.as<float>() + (size_t)i * Tl * 3 * d;
                a.k = cache + (size_t)first * 2 * d;
                a.v = a.k + d;
                a.out = e->att.as<float>() + (size_t)i * Tl * d;
                a.nq = Tl;
                a.nk = ncached + Tl;
                a.klen = a.nk;
                a.pos0 = win[i].first_full;
                a.q_abs0 = off;
                a.pad_ = 0;
            }
            hp[(size_t)l * n + i] = st[i]->cnn.as<float>() + (size_t)l * padmax * d;
            hp[(size_t)(L + l) * n + i] = st[i]->cnn2.as<float>() + (size_t)l * padmax * d;     // written half of the double buffer
        }
    }
    CHK(e->cnnptrs.ensure(sizeof(float*) * hp.size() + sizeof(PlaneCopy) * pcs.size()));
    PlaneCopy* pc_dev = reinterpret_cast<PlaneCopy*>(e->cnnptrs.as<float*>() + hp.size());
    CHK(e->stage.begin(sizeof(AttSeq) * hs.size() + sizeof(float*) * hp.size() + sizeof(PlaneCopy) * pcs.size() + 64));
    CHK(e->stage.push(e->attseq.p, hs.data(), sizeof(AttSeq) * hs.size(), s));
    CHK(e->stage.push(e->cnnptrs.p, hp.data(), sizeof(float*) * hp.size(), s));
    CHK(e->stage.push(pc_dev, pcs.data(), sizeof(PlaneCopy) * pcs.size(), s));
    CHK(e->stage.end(s));
    float* x = e->x.as<float>();
    for (int l = 0; l < L; ++l) {
        const LayerW& w = e->layers[l];
        const int Tq = rate(l) == 2 ? T2 : T0;
        int M = n * Tq;
        const AttSeq* seqs = e->attseq.as<AttSeq>() + (size_t)l * n;
        CHK(ffn(e, s, M, w.ln_ffm_w, w.ln_ffm_b, w.ffm_w1, w.ffm_b1, w.ffm_w2, w.ffm_b2));
        if (layer_grouped(e, l)) {
            rowgemm(e, s, RG_PRO_LN, RG_EPI_STORE, x, d, w.ln_mha_w, w.ln_mha_b, w.wqkv, w.bqkv, qp, d, M, 3 * d, nullptr, 0,
                    1.f, nullptr, 0, 0, 0, nullptr, nullptr, PROF_GEMM, 4, Tq, 0, Tpad - Tq, d, (long)plane);
            launch_kv_append_planar(pc_dev + (size_t)l * n, n, Tq, s);
            launch_attention_grouped(seqs, n, Tg, H, G, w.ptab, 0, w.pos_u, w.pos_v, s);
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->attp.as<float>(), d, nullptr, nullptr, w.wo, w.bo, x, d, M, d, x, d,
                    1.f, nullptr, 0, 0, 0, nullptr, nullptr, PROF_GEMM, 4, 0, 0, 0, 0, 0, Tq, Tpad);
        } else {
            mhsa(e, s, w, M, seqs, Tq);                 // k | v rows go straight to the streams' caches
            launch_attention(seqs, n, Tq, H, 3 * d, 2 * d, w.ptab, w.pos_u, w.pos_v, 0, rate(l), s);
            mhsa_out(e, s, w, M);
        }
        const int K = layer_kernel(e, l), pad = K - 1;
        float* const* cptr = e->cnnptrs.as<float*>() + (size_t)l * n;
        if (l == e->stride_idx) {
            launch_cnn_cache_move(cptr, e->lnpad.as<float>(), n, Tq, pad, 0, s);
            // StrideConformerEncoderLayer (encoder.py:454-545): x = AvgPool(x) + conv_module_stride2([cache | LN(x)])
            launch_layernorm(x, w.ln_conv_w, w.ln_conv_b, e->lnpad.as<float>(), M, 1e-5f, Tq, pad, nullptr, s);
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_GLU, e->lnpad.as<float>(), d, nullptr, nullptr, w.pw1_w, w.pw1_b,
                    e->glu.as<float>(), d, n * (Tq + pad), 2 * d, nullptr, 0, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
            launch_cnn_cache_move(cptr + (size_t)L * n, e->lnpad.as<float>(), n, Tq, pad, 1, s);
            launch_dwconv_stride2_ln_silu(e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, e->dwo.as<float>(), n, Tq, K,
                                          1e-5f, s);
            launch_avgpool2(x, e->xsave.as<float>(), n, Tq, s);
            M = n * T2;
            rowgemm(e, s, RG_PRO_PLAIN, RG_EPI_RESID, e->dwo.as<float>(), d, nullptr, nullptr, w.pw2_w, w.pw2_b, x, d, M, d,
                    e->xsave.as<float>(), d, 1.f, nullptr, 0, 0, 0, nullptr, nullptr);
        } else {
            CHK(conv_module_stream(e, s, w, n, Tq, cptr, cptr + (size_t)L * n, K));
        }
        CHK(ffn(e, s, M, w.ln_ff_w, w.ln_ff_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2, 0.5f, 0, w.ln_fin_w, w.ln_fin_b, x));
    }
    CHK(e->enc.ensure((size_t)n * T2 * d * sizeof(float)));
    launch_layernorm(x, e->after_w, e->after_b, e->enc.as<float>(), n * T2, 1e-5f, 0, 0, nullptr, s);
    if (e->cfg.vocab_size > 16384) return fail("vocab_size > 16384 is not supported by the softmax / pruning kernels (one 256-thread workgroup holds a row in registers)");
    CHK(ctc_head(e, e->enc.as<float>(), n * T2, probs_dev, probs_dev ? 1 : 0, argmax_dev, maxprob_dev, s));
    for (int i = 0; i < n; ++i) {
        st[i]->offset += T0;
        st[i]->offset_r += T2;
        st[i]->cache_t1 = win[i].next_cache_t1;
        st[i]->first_half += win[i].ncs / 2;
        std::swap(st[i]->cnn, st[i]->cnn2);
    }
    return 0;
}

extern "C" {

int masr_encode_chunk(masr_engine* e, const int32_t* stream_ids, int32_t n, const float* feats_dev, int32_t Tc,
                      float* probs_dev, int32_t* argmax_dev, float* maxprob_dev, void* stream) {
    if (!e || !e->finalized) return fail("engine not finalized");
    ENTER(e);
    CallGuard call_guard(e, (hipStream_t)stream);
    if (n <= 0) return fail("no streams");
    hipStream_t s = (hipStream_t)stream;
    const int d = e->cfg.d_model, H = e->cfg.heads, L = e->cfg.num_blocks, pad = e->cfg.cnn_kernel - 1;
    std::vector<Stream*> st(n);
    for (int i = 0; i < n; ++i) CHK(stream_of(e, stream_ids[i], &st[i]));
    int Tq = 0;
    if (e->cfg.model_kind == 1 || e->cfg.model_kind == 2) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (st[i] == st[j]) return fail("duplicate stream id in one call");
        return e->cfg.model_kind == 1 ? encode_chunk_squeezeformer(e, s, st, feats_dev, Tc, probs_dev, argmax_dev, maxprob_dev)
                                      : encode_chunk_efficient(e, s, st, feats_dev, Tc, probs_dev, argmax_dev, maxprob_dev);
    }
    if (e->cfg.model_kind == 3) {       // deepspeech2/model.py:79-108: chunk conv + LSTM stack with carried (h, c)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (st[i] == st[j]) return fail("duplicate stream id in one call");
        const int Tq3 = ((Tc - 1) / 2 - 1) / 2;
        if (Tq3 <= 0) return fail("chunk too short");
        CHK(e->enc.ensure((size_t)n * Tq3 * enc_dim(e) * sizeof(float)));
        CHK(ds2_forward(e, s, feats_dev, nullptr, n, Tc, e->enc.as<float>(), st.data(), &Tq));
        if (e->cfg.vocab_size > 16384) return fail("vocab_size > 16384 is not supported by the softmax / pruning kernels (one 256-thread workgroup holds a row in registers)");
        CHK(ctc_head(e, e->enc.as<float>(), n * Tq, probs_dev, probs_dev ? 1 : 0, argmax_dev, maxprob_dev, s));
        for (int i = 0; i < n; ++i) st[i]->offset += Tq;
        return 0;
    }
    CHK(embed(e, s, feats_dev, n, Tc, &Tq));
    for (int i = 0; i < n; ++i)
        if (st[i]->offset + Tq > st[i]->cap) return fail("stream exceeds its max_frames_out / max_pos");
    const int M = n * Tq;
    CHK(ensure_layer_ws(e, n, Tq));
    CHK(e->attseq.ensure(sizeof(AttSeq) * (size_t)n * L));
    // descriptors for all layers in one H2D copy
    std::vector<AttSeq> hs((size_t)n * L);
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < n; ++i) {
            AttSeq& a = hs[(size_t)l * n + i];
            float* cache = st[i]->att.as<float>() + (size_t)l * st[i]->cap * 2 * d;
            // keys the chunk attends over: the last cache_t1 cached rows + its own (encoder.py:390-395,397-410: the reference
            // trims its cache tensor to required_cache_size after every step; here the rows stay where they were appended and
            // the window moves).  The positional index of the first key is offset - cache_t1.
            const int cache_t1 = st[i]->cache_t1;
            const int first = st[i]->offset - cache_t1;
            a.q = e->qkv.as<float>() + (size_t)i * Tq * 3 * d;
            a.k = cache + (size_t)first * 2 * d;
            a.v = a.k + d;
            a.out = e->att.as<float>() + (size_t)i * Tq * d;
            a.nq = Tq;
            a.nk = cache_t1 + Tq;          // (the new rows are appended at k + (nk - nq) rows = cache row `offset`)
            a.klen = a.nk;
            a.pos0 = first;
            a.q_abs0 = st[i]->offset;
            a.pad_ = 0;
        }
    std::vector<float*> hp((size_t)n * L * 2);    // cnn cache base of (layer, stream): current (read) | next (written)
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < n; ++i) {
            hp[(size_t)l * n + i] = st[i]->cnn.as<float>() + (size_t)l * pad * d;
            hp[(size_t)(L + l) * n + i] = st[i]->cnn2.as<float>() + (size_t)l * pad * d;
        }
    CHK(e->cnnptrs.ensure(sizeof(float*) * hp.size()));
    CHK(e->stage.begin(sizeof(AttSeq) * hs.size() + sizeof(float*) * hp.size() + 64));
    CHK(e->stage.push(e->attseq.p, hs.data(), sizeof(AttSeq) * hs.size(), s));
    CHK(e->stage.push(e->cnnptrs.p, hp.data(), sizeof(float*) * hp.size(), s));
    CHK(e->stage.end(s));
    float* x = e->x.as<float>();
    EncodeCtx ctx{n, Tq, nullptr};
    for (int l = 0; l < L; ++l) {
        const LayerW& w = e->layers[g_hot_weights ? 0 : l];
        CHK(ffn(e, s, M, w.ln_ffm_w, w.ln_ffm_b, w.ffm_w1, w.ffm_b1, w.ffm_w2, w.ffm_b2));
        mhsa(e, s, w, M, e->attseq.as<AttSeq>() + (size_t)l * n, Tq);     // q -> qkv buffer, k|v rows -> the streams' caches
        launch_attention(e->attseq.as<AttSeq>() + (size_t)l * n, n, Tq, H, 3 * d, 2 * d, w.ptab, w.pos_u, w.pos_v, 0, 1, s);
        mhsa_out(e, s, w, M);
        float* const* cptr = e->cnnptrs.as<float*>() + (size_t)l * n;
        // [depthwise conv -> LN -> SiLU -> pointwise_conv2 + residual] rides on the second FFN launch as its head stage (on the
        // d_ff-split launch of few streams every slice repeats it on the row block's rows; one launch less on the step's chain)
        const bool fuse = g_split_head && e->cfg.cnn_kernel == 15 && g_ffn_packed >= 2 && !g_no_ffn_head;
        CHK(conv_module_stream(e, s, w, n, Tq, cptr, cptr + (size_t)L * n, e->cfg.cnn_kernel, fuse));
        const FfnHead head{e->glu.as<float>(), w.dw_w, w.dw_b, w.cln_w, w.cln_b, nullptr, w.pw2_w, w.pw2_b, nullptr, Tq,
                           e->cfg.cnn_kernel, 4, nullptr};
        CHK(ffn(e, s, M, w.ln_ff_w, w.ln_ff_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2, 0.5f, 0, w.ln_fin_w, w.ln_fin_b, x, nullptr,
                nullptr, fuse ? &head : nullptr));
    }
    CHK(e->enc.ensure((size_t)M * d * sizeof(float)));
    launch_layernorm(x, e->after_w, e->after_b, e->enc.as<float>(), M, 1e-5f, 0, 0, nullptr, s);
    if (e->cfg.vocab_size > 16384) return fail("vocab_size > 16384 is not supported by the softmax / pruning kernels (one 256-thread workgroup holds a row in registers)");
    CHK(ctc_head(e, e->enc.as<float>(), M, probs_dev, probs_dev ? 1 : 0, argmax_dev, maxprob_dev, s));
    for (int i = 0; i < n; ++i) {
        st[i]->cache_t1 = chunk_window(*st[i], Tq).next_cache_t1;
        st[i]->offset += Tq;
        std::swap(st[i]->cnn, st[i]->cnn2);
    }
    return 0;
}

int masr_stream_export_cache(masr_engine* e, int32_t stream_id, float* att_dev, float* cnn_dev, void* stream) {
    Stream* st;
    CHK(stream_of(e, stream_id, &st));
    ENTER(e);
    hipStream_t s = (hipStream_t)stream;
    const int d = e->cfg.d_model, L = e->cfg.num_blocks, pad = e->cfg.cnn_kernel - 1, H = e->cfg.heads;
    if (e->cfg.model_kind == 3) {       // att_dev <- h [L][rnn_size], cnn_dev <- c [L][rnn_size]
        for (int l = 0; l < L; ++l) {
            const float* hc = st->cnn.as<float>() + (size_t)l * 2 * d;
            if (att_dev) HIPCHK(hipMemcpyAsync(att_dev + (size_t)l * d, hc, sizeof(float) * d, hipMemcpyDeviceToDevice, s));
            if (cnn_dev) HIPCHK(hipMemcpyAsync(cnn_dev + (size_t)l * d, hc + d, sizeof(float) * d, hipMemcpyDeviceToDevice, s));
        }
        return 0;
    }
    // what the reference's (trimmed) cache tensor holds: the last t = cache_t1 input-rate frames [L, H, t, 2 dk]
    const int t = st->cache_t1;
    if (att_dev && t > 0) {
        const int first = st->offset - t;
        if (e->cfg.model_kind == 2) {     // planar grouped layers; half-rate layers repeat-interleaved, the LAST t entries (:370,380)
            for (int l = 0; l < L; ++l) {
                const float* base = st->att.as<float>() + (size_t)l * st->cap * 2 * d;
                float* o = att_dev + (size_t)l * t * 2 * d;
                if (layer_grouped(e, l))
                    launch_export_att_planar(base + (size_t)first * d, base + (size_t)st->cap * d + (size_t)first * d, o, H, t, d / H, s);
                else if (l > e->stride_idx)
                    launch_export_att(base, o, 1, H, st->cap, t, d / H, s, 2, 2 * st->offset_r - t);
                else
                    launch_export_att(base + (size_t)first * 2 * d, o, 1, H, st->cap, t, d / H, s);
            }
        } else if (e->cfg.model_kind == 1) {     // half-rate layers: every kept entry twice, the FIRST t of them (encoder.py:345-351)
            for (int l = 0; l < L; ++l) {
                const float* base = st->att.as<float>() + (size_t)l * st->cap * 2 * d;
                if (l >= e->reduce_idx && l < e->recover_idx)
                    launch_export_att(base, att_dev + (size_t)l * t * 2 * d, 1, H, st->cap, t, d / H, s, 2, 2 * st->first_half);
                else
                    launch_export_att(base + (size_t)first * 2 * d, att_dev + (size_t)l * t * 2 * d, 1, H, st->cap, t, d / H, s);
            }
        } else {
            launch_export_att(st->att.as<float>() + (size_t)first * 2 * d, att_dev, L, H, st->cap, t, d / H, s);
        }
    }
    if (cnn_dev && e->cfg.model_kind == 2) {       // each layer's own K-1 rows, left-padded with zeros to cnn_module_kernel - 1
        HIPCHK(hipMemsetAsync(cnn_dev, 0, (size_t)L * pad * d * sizeof(float), s));
        for (int l = 0; l < L; ++l)
            launch_export_cnn_layer(st->cnn.as<float>() + (size_t)l * pad * d, cnn_dev + (size_t)l * pad * d, layer_kernel(e, l) - 1,
                                    pad, d, s);
    } else if (cnn_dev) {
        launch_export_cnn(st->cnn.as<float>(), cnn_dev, L, pad, d, s);
    }
    LAUNCHCHK();
    return 0;
}

// ---- single ops -------------------------------------------------------------------------------------------

int masr_op_layernorm(masr_engine* e, const float* x_dev, const float* w_dev, const float* b_dev, float* y_dev,
                      int32_t M, float eps, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    launch_layernorm(x_dev, w_dev, b_dev, y_dev, M, eps, 0, 0, nullptr, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

int masr_op_gemm(masr_engine* e, const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev,
                 float* c_dev, int32_t M, int32_t N, int32_t K, int32_t act, float alpha, void* stream) {
    if (!e) return fail("null engine");
    ENTER(e);
    CallGuard call_guard(e, (hipStream_t)stream);
    if (K % 32) return fail("K must be a multiple of 32");
    gemm(e, (hipStream_t)stream, a_dev, K, w_dev, bias_dev, c_dev, N, M, N, K, act, alpha, res_dev, N);
    LAUNCHCHK();
    return 0;
}

int masr_select_lane(masr_engine* e, int32_t lane) {
    if (!e) return fail("null engine");
    if (lane < 0 || lane >= MASR_LANES) return fail("masr_select_lane: lane must be in [0, " + std::to_string(MASR_LANES) + ")");
    if (lane == e->lane) return 0;
    EngineWs& active = *e;                   // park the active set, bring the lane's own in (pointers and sizes only)
    e->parked[e->lane] = active;
    active = e->parked[lane];
    e->parked[lane] = EngineWs();
    e->lane = lane;
    return 0;
}

int masr_side_stream(masr_engine* e, int32_t kind, void** stream_out) {
    if (!e || !stream_out) return fail("null argument");
    if (kind < 0 || kind >= MASR_SIDE_STREAMS)
        return fail("masr_side_stream: kind must be 0 / 1 (prefix search), 2 (preparation), 3 (copy) or 4 (second encoder lane)");
    SideStreams* ss = nullptr;
    if (side_streams_of(e->cfg.device_id, &ss)) return 1;
    *stream_out = (void*)ss->s[kind];
    return 0;
}

int masr_debug_set(masr_engine* e, int32_t key, int32_t value) {
    if (!e) return fail("null engine");
#if !MASR_EXPERIMENTS
    if (value != 0 && (key == 20 || key == 21 || key == 22 || key == 24 || key == 30 || key == 34 || key == 35))
        return fail("masr_debug_set: this key selects an experimental kernel that is not in this build (MASR_BUILD_EXPERIMENTS=1)");
#endif
    if (key == 1) set_ffn_variant(value);
    else if (key == 5) g_no_chain = value;
    else if (key == 6) set_rowgemm_small(value);
    else if (key == 7) set_attention_fewq(value);
    else if (key == 14) set_attention_fold(value);
    else if (key == 15) g_embed_split = value;
    else if (key == 19) g_hot_weights = value;
    else if (key == 20) g_bf16x3 = value;
    else if (key == 21) set_gemm_bf16x3_waves(value);
    else if (key == 22) set_ffn_x3_rotation(value);
    else if (key == 23) g_ffn_packed = value;
    else if (key == 24) g_ffn_dual = value;
    else if (key == 25) g_rowgemm_packed = value;
    else if (key == 26) set_attention_grouped_fold(value);
    else if (key == 27) g_ctc_fused_blocks = value;
    else if (key == 28) set_attention_fewq_wgs(value);
    else if (key == 29) g_few_rows_path = value;
    else if (key == 30) g_split_head = value;
    else if (key == 31) g_efficient_fused = value;
    else if (key == 32) g_beam_lm_cache = value;
    else if (key == 33) set_conv2_mid_fill(value);
    else if (key == 34) g_attn_chain = value;
    else if (key == 35) g_ffn_coop = value;
    else if (key == 36) g_sqz_fused_blocks = value;
    else if (key == 37) g_beam_narrow = value;
    else if (key == 38) e->skip_padding = value;
    else if (key == 17) set_gemm_waves(value);
    else if (key == 18) set_conv1_nt(value);
    else if (key == 16) { e->prof_stride = value > 1 ? value : 1; e->prof_seen = 0; }
    else if (key == 8) g_no_ffn_tail = value;
    else if (key == 9) g_no_ffn_head = value;
    else if (key == 12) set_rowgemm_small_blocks(value);
    else if (key == 13) g_ffn_split_blocks = value;
    else if (key == 2) {            // beam search phase profile of workgroup 0: value 1 = on, 0 = print + off
        if (value) {
            if (!e->beam_prof) {
                void* p = nullptr;
                HIPCHK(hipMalloc(&p, 16 * sizeof(long long)));
                e->owned.push_back(p);
                e->beam_prof = (long long*)p;
            }
        } else if (e->beam_prof) {
            long long h[16];
            HIPCHK(hipMemcpy(h, e->beam_prof, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "beam phases (cycles, last workgroup): setup+hash %lld  extensions %lld  prefixes+count %lld  select %lld  compact %lld\n",
                    h[0], h[1], h[2], h[3], h[4]);
            fprintf(stderr, "  narrow step (%lld frames): tables+candidates %lld  hash+contexts %lld  children+scorer table %lld  extensions %lld  "
                    "prefixes+select %lld  scan %lld  survivors %lld  new prefixes %lld\n", h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13]);
            if (h[15] > 0) fprintf(stderr, "  wide step (%lld frames): %.1f distinct scorer contexts per frame\n", h[15], (double)h[14] / (double)h[15]);
            e->beam_prof = nullptr;
        }
    }
    else return fail("unknown debug key");
    return 0;
}

int masr_profile_select(masr_engine* e, int32_t kind) {
    if (!e) return fail("null engine");
    e->prof_kind = kind;
    return 0;
}

int masr_profile_read(masr_engine* e, double* total_ms, int64_t* launches, double* flops, int32_t reset) {
    if (!e) return fail("null engine");
    ENTER(e);
    double tot = 0.0;
    for (size_t i = 0; i < e->prof_used; ++i) {
        HIPCHK(hipEventSynchronize(e->prof_events[i].second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->prof_events[i].first, e->prof_events[i].second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)e->prof_used;
    if (flops) *flops = e->prof_flops;
    if (reset) {
        e->prof_used = 0;
        e->prof_flops = 0.0;
    }
    return 0;
}

}  // extern "C"
