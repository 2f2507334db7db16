// libmasr_hip.so -- serving pool: any number of concurrent predict_stream sessions on ONE engine, one C call per step.
//
// What the reference keeps per stream in python (masr/predict.py:237-343: samples not yet framed, feature frames not yet
// consumed by a decoding window, the greedy decoder's history; infer_server.py:103-156 holds one such predictor per websocket)
// is kept here per session; the device work of all sessions that were fed since the last step is done together:
//   pending PCM (int16 wire format or float32) -> x / 2^15 -> [carried-over | new] samples per session in pinned memory -> ONE
//   upload -> mean squares (numpy's summation order) -> gains by the caller's evaluator (the reference's scalar numpy expressions;
//   NULL: libm) -> ONE ragged feature launch -> the new frames appended to the sessions' rows of a device-resident frame pool
//   -> the 67-frame decoding windows of all sessions gathered from it and advanced in lock-step through masr_encode_chunk ->
//   (argmax, max prob) frames appended to device-resident histories -> ONE collapse launch over the histories of the sessions
//   that advanced -> ONE copy back of the packed rows [tokens | count | score bits].
// Round 3 did this framing in python (masr_amd/serving.py: 1.4 of the 4.4 ms of a 128-stream step); the python class keeps the
// sessions' handles and builds the text.  Greedy sessions only (beam-search sessions keep their per-session python path).
// Host-side only: the kernels are the engine's (masr_hip.h entry points) plus two copy / collapse launches in elementwise.hip.
#include <string.h>

#include <chrono>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/masr_hip.h"
#include "common.h"

using namespace masr;
namespace masr {
int engine_fail(const std::string& m);
}

#define PFAIL(msg) return masr::engine_fail(msg)
#define PHIP(expr)                                                                                                        \
    do {                                                                                                                  \
        hipError_t _e = (expr);                                                                                           \
        if (_e != hipSuccess) return masr::engine_fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (pool.hip)"); \
    } while (0)
#define PCHK(expr)         \
    do {                   \
        int _r = (expr);   \
        if (_r) return _r; \
    } while (0)

namespace {

// chunked decoding geometry of the reference facade (predict.py:283-290): 16 encoder frames per chunk, subsampling 4, context 7
constexpr int kDecodingChunk = 16, kSubsampling = 4, kContext = 7;
constexpr int kWindow = (kDecodingChunk - 1) * kSubsampling + kContext;      // 67 feature frames per window
constexpr int kStride = kSubsampling * kDecodingChunk;                       // 64
constexpr int kOverlap = kContext - kSubsampling;                            // 3 frames carried over
constexpr int64_t kMaxFeedSamples = (int64_t)1 << 28;                        // per feed (4.6 h at 16 kHz): anything above is a caller bug

struct Dev {                     // grow-only device buffer; `keep` bytes of the old contents survive a growth
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n, size_t keep, hipStream_t s) {
        if (n <= bytes) return 0;
        void* q = nullptr;
        const size_t want = n + n / 4 + 256;
        PHIP(hipMalloc(&q, want));
        PHIP(hipMemsetAsync(q, 0, want, s));
        if (p && keep) PHIP(hipMemcpyAsync(q, p, std::min(keep, bytes), hipMemcpyDeviceToDevice, s));
        if (p) {
            PHIP(hipStreamSynchronize(s));
            PHIP(hipFree(p));
        }
        p = q;
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

struct Pinned {                  // pinned host area, bump-allocated per step; copies out of it are asynchronous
    char* p = nullptr;
    size_t bytes = 0, used = 0;
    std::vector<char*> retired;
    int reset() {
        for (char* r : retired) (void)hipHostFree(r);
        retired.clear();
        used = 0;
        return 0;
    }
    int take(size_t n, void** out) {
        const size_t at = (used + 63) & ~(size_t)63;
        if (at + n > bytes) {                      // copies in flight keep the old area alive until the next reset()
            if (p) retired.push_back(p);
            const size_t want = std::max(2 * bytes, 2 * n + 4096);
            void* q = nullptr;
            PHIP(hipHostMalloc(&q, want, hipHostMallocDefault));
            p = (char*)q;
            bytes = want;
            used = n;
            *out = p;
            return 0;
        }
        used = at + n;
        *out = p + at;
        return 0;
    }
    void release() {
        reset();
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct Session {
    int sid = -1;                 // engine stream id == pool handle
    int row = 0;                  // row of the frame pool and of the histories
    std::vector<float> remained;  // samples not yet turned into frames (already normalised: re-normalised on every call, audio.py:304)
    int f0 = 0, nf = 0;           // live feature frames: f0 .. f0 + nf of the session's row
    int frames = 0;               // encoder frames decoded so far
};

}  // namespace

struct masr_pool {
    masr_engine* e = nullptr;
    int device = 0;
    int method = 0, n_mfcc = 40, use_db = 1, F = 80, min_samples = 400, max_frames_out = 0;
    float target_db = -20.f;
    std::map<int, Session> sessions;
    std::vector<int> free_rows;
    int n_rows = 0;
    // device-resident state
    Dev feat;                     // [rows_cap][feat_cap][F]
    int rows_cap = 0, feat_cap = 512;
    Dev hist_idx, hist_mp;        // [hist_rows][hist_cap]
    int hist_rows = 0, hist_cap = 0;
    // per-step work buffers
    Dev samples, lens, ms, gains, feats, win, idx, mp, segs, meta, rows_out;
    Pinned pin;
    hipEvent_t ev = nullptr;      // the previous step's copies out of `pin`
    bool pending = false;
    // results of the last step (pinned; valid until the next step)
    std::vector<int32_t> handles, state;
    // host time per phase, accumulated (masr_pool_profile): assemble | upload + mean squares + wait | gains | features + frames
    // | windows | collapse + copy back + wait; and the number of steps
    double prof[6] = {0, 0, 0, 0, 0, 0};
    long prof_steps = 0;
};

namespace {

int feature_frames(const masr_pool* p, long n) {
    return n >= p->min_samples ? (int)((n - p->min_samples) / 160 + 1) : 0;
}

// make the histories at least [rows][frames] and the frame pool at least [rows][feat_cap] (contents kept)
int grow_state(masr_pool* p, int rows, int frames, hipStream_t s) {
    if (rows > p->hist_rows || frames > p->hist_cap) {
        const int r1 = rows <= p->hist_rows ? p->hist_rows : std::max({rows, 2 * p->hist_rows, 16});
        const int f1 = frames <= p->hist_cap ? p->hist_cap : std::max({frames, 2 * p->hist_cap, 256});
        for (Dev* d : {&p->hist_idx, &p->hist_mp}) {
            Dev nd;
            PCHK(nd.ensure((size_t)r1 * f1 * 4, 0, s));
            if (d->p && p->hist_rows && p->hist_cap)
                PHIP(hipMemcpy2DAsync(nd.p, (size_t)f1 * 4, d->p, (size_t)p->hist_cap * 4, (size_t)p->hist_cap * 4, p->hist_rows,
                                      hipMemcpyDeviceToDevice, s));
            PHIP(hipStreamSynchronize(s));
            d->release();
            *d = nd;
        }
        p->hist_rows = r1;
        p->hist_cap = f1;
    }
    if (rows > p->rows_cap) {
        const int r1 = std::max({rows, 2 * p->rows_cap, 16});
        PCHK(p->feat.ensure((size_t)r1 * p->feat_cap * p->F * 4, (size_t)p->rows_cap * p->feat_cap * p->F * 4, s));
        p->rows_cap = r1;
    }
    return 0;
}

// wider rows of the frame pool (a whole utterance fed in one call)
int widen_feat(masr_pool* p, int cap, hipStream_t s) {
    Dev nd;
    const size_t row_old = (size_t)p->feat_cap * p->F * 4, row_new = (size_t)cap * p->F * 4;
    PCHK(nd.ensure((size_t)p->rows_cap * row_new, 0, s));
    if (p->feat.p && p->rows_cap)
        PHIP(hipMemcpy2DAsync(nd.p, row_new, p->feat.p, row_old, row_old, p->rows_cap, hipMemcpyDeviceToDevice, s));
    PHIP(hipStreamSynchronize(s));
    p->feat.release();
    p->feat = nd;
    p->feat_cap = cap;
    return 0;
}

int upload(masr_pool* p, Dev& dst, const void* host_in_pin, size_t n, hipStream_t s) {
    PCHK(dst.ensure(n, 0, s));
    PHIP(hipMemcpyAsync(dst.p, host_in_pin, n, hipMemcpyHostToDevice, s));
    return 0;
}

}  // namespace

extern "C" {

int masr_pool_create(masr_engine* e, int32_t feature_method, int32_t n_mfcc, int32_t use_db_normalization, float target_db,
                     int32_t max_frames_out, masr_pool** out) {
    if (!e || !out) PFAIL("null argument");
    if (feature_method < 0 || feature_method > 2) PFAIL("feature_method: 0 = fbank, 1 = mfcc, 2 = linear");
    masr_pool* p = new masr_pool();
    p->e = e;
    int32_t dev = 0, n_mels = 80;
    if (masr_engine_info(e, &dev, &n_mels, nullptr)) {
        delete p;
        return 1;
    }
    p->device = dev;
    p->method = feature_method;
    p->n_mfcc = n_mfcc;
    p->use_db = use_db_normalization ? 1 : 0;
    p->target_db = target_db;
    p->max_frames_out = max_frames_out;
    p->F = feature_method == 2 ? 161 : feature_method == 1 ? n_mfcc : 80;
    p->min_samples = feature_method == 2 ? 320 : 400;
    *out = p;
    return 0;
}

void masr_pool_destroy(masr_pool* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : p->sessions) (void)masr_stream_close(p->e, kv.second.sid);
    for (Dev* d : {&p->feat, &p->hist_idx, &p->hist_mp, &p->samples, &p->lens, &p->ms, &p->gains, &p->feats, &p->win, &p->idx,
                   &p->mp, &p->segs, &p->meta, &p->rows_out})
        d->release();
    p->pin.release();
    if (p->ev) (void)hipEventDestroy(p->ev);
    delete p;
}

int masr_pool_open(masr_pool* p, int32_t* handle) {
    if (!p || !handle) PFAIL("null argument");
    int32_t sid = -1;
    PCHK(masr_stream_open(p->e, p->max_frames_out, &sid));
    Session s;
    s.sid = sid;
    if (!p->free_rows.empty()) {
        s.row = p->free_rows.back();
        p->free_rows.pop_back();
    } else {
        s.row = p->n_rows++;
    }
    p->sessions[sid] = std::move(s);
    *handle = sid;
    return 0;
}

int masr_pool_close(masr_pool* p, int32_t handle) {
    if (!p) PFAIL("null pool");
    auto it = p->sessions.find(handle);
    if (it == p->sessions.end()) PFAIL("masr_pool_close: unknown handle");
    PCHK(masr_stream_close(p->e, handle));
    p->free_rows.push_back(it->second.row);
    p->sessions.erase(it);
    return 0;
}

int masr_pool_reset(masr_pool* p, int32_t handle) {
    if (!p) PFAIL("null pool");
    auto it = p->sessions.find(handle);
    if (it == p->sessions.end()) PFAIL("masr_pool_reset: unknown handle");
    PCHK(masr_stream_reset(p->e, handle));
    Session& s = it->second;
    s.remained.clear();
    s.f0 = s.nf = s.frames = 0;
    return 0;
}

int masr_pool_step(masr_pool* p, int32_t n_feeds, const int32_t* feed_handle, const void* const* feed_samples,
                   const int64_t* feed_n, const int32_t* feed_format, const int32_t* feed_is_end, masr_gain_fn gain_fn,
                   void* gain_user, int32_t* n_sessions, const int32_t** handles_out, const int32_t** state_out,
                   const int32_t** rows_host, int32_t* row_width, int32_t** rows_dev, void* stream) {
    if (!p || !n_sessions) PFAIL("null argument");
    PHIP(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    *n_sessions = 0;
    if (row_width) *row_width = 0;
    if (rows_host) *rows_host = nullptr;
    if (rows_dev) *rows_dev = nullptr;
    p->handles.clear();
    p->state.clear();
    if (n_feeds <= 0) return 0;
    // ---- validate EVERYTHING before any session state is touched: a bad feed fails the call with nothing committed ----------
    if (!feed_handle || !feed_samples || !feed_n || !feed_format || !feed_is_end) PFAIL("masr_pool_step: null feed arrays");
    for (int k = 0; k < n_feeds; ++k) {
        if (p->sessions.find(feed_handle[k]) == p->sessions.end()) PFAIL("masr_pool_step: unknown handle");
        if (feed_format[k] != 0 && feed_format[k] != 1) PFAIL("masr_pool_step: sample format 0 = int16 PCM, 1 = float32");
        if (feed_n[k] < 0 || feed_n[k] > kMaxFeedSamples) PFAIL("masr_pool_step: feed_n out of range (0 .. 2^28 samples)");
        if (feed_n[k] > 0 && !feed_samples[k]) PFAIL("masr_pool_step: null sample pointer with feed_n > 0");
    }
    // ---- sessions of this step, in the order they were first fed ------------------------------------------------------
    std::vector<Session*> sess;
    std::vector<int> end_flag;
    std::map<int, int> pos;
    std::vector<std::vector<int>> feeds_of;
    for (int k = 0; k < n_feeds; ++k) {
        auto it = p->sessions.find(feed_handle[k]);
        auto q = pos.find(feed_handle[k]);
        if (q == pos.end()) {
            q = pos.emplace(feed_handle[k], (int)sess.size()).first;
            sess.push_back(&it->second);
            end_flag.push_back(0);
            feeds_of.emplace_back();
        }
        end_flag[q->second] |= feed_is_end[k] ? 1 : 0;
        feeds_of[q->second].push_back(k);
    }
    // a session whose stream cannot take the frames this step would emit (masr_encode_chunk would refuse the whole lock-step
    // call: "stream exceeds its max_frames_out / max_pos") is LEFT OUT, untouched, and reported with state -1: one full
    // session does not fail the step of the others
    std::vector<int32_t> refused;
    {
        std::vector<Session*> keep_s;
        std::vector<int> keep_e;
        std::vector<std::vector<int>> keep_f;
        for (size_t i = 0; i < sess.size(); ++i) {
            long tot = (long)sess[i]->remained.size();
            for (int k : feeds_of[i]) tot += feed_n[k];
            const int nfr = sess[i]->nf + feature_frames(p, tot);
            long emit = 0;
            if (!((nfr < kWindow && !end_flag[i]) || nfr < kContext)) {
                const int left = end_flag[i] ? kContext : kWindow;
                for (int cur = 0; cur <= nfr - left; cur += kStride) {
                    const int len = std::min(cur + kWindow, nfr) - cur;
                    emit += ((len - 1) / 2 - 1) / 2;
                }
            }
            int32_t room = 0;
            PCHK(masr_stream_room(p->e, sess[i]->sid, &room));
            if (emit > 0 && emit > room) {               // (masr_stream_room already holds back the Efficient-Conformer's group padding)
                refused.push_back(sess[i]->sid);
                continue;
            }
            keep_s.push_back(sess[i]);
            keep_e.push_back(end_flag[i]);
            keep_f.push_back(std::move(feeds_of[i]));
        }
        sess.swap(keep_s);
        end_flag.swap(keep_e);
        feeds_of.swap(keep_f);
    }
    if (sess.empty()) {
        for (int32_t h : refused) {
            p->handles.push_back(h);
            p->state.push_back(-1);
        }
        *n_sessions = (int32_t)refused.size();
        if (handles_out) *handles_out = p->handles.data();
        if (state_out) *state_out = p->state.data();
        return 0;
    }
    const int n = (int)sess.size();
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](int k) {
        const auto t = std::chrono::steady_clock::now();
        p->prof[k] += std::chrono::duration<double, std::milli>(t - t_last).count();
        t_last = t;
    };
    if (p->pending) {                                // the previous step's copies out of the pinned area (long done)
        PHIP(hipEventSynchronize(p->ev));
        p->pending = false;
    }
    if (!p->ev) PHIP(hipEventCreateWithFlags(&p->ev, hipEventDisableTiming));
    PCHK(p->pin.reset());
    int max_row = 0;
    for (Session* q : sess) max_row = std::max(max_row, q->row);
    PCHK(grow_state(p, max_row + 1, std::max(p->hist_cap, 256), s));

    // ---- [carried-over | fed] samples of every session, float32, in pinned memory (predict.py:260-281) -----------------------
    std::vector<long> L(n);
    long n_max = p->min_samples;
    for (int i = 0; i < n; ++i) {
        long tot = (long)sess[i]->remained.size();
        for (int k : feeds_of[i]) tot += feed_n[k];
        L[i] = tot;
        n_max = std::max(n_max, tot);
    }
    float* buf = nullptr;
    int32_t* lens_h = nullptr;
    PCHK(p->pin.take((size_t)n * n_max * 4, (void**)&buf));
    PCHK(p->pin.take((size_t)n * 4, (void**)&lens_h));
    for (int i = 0; i < n; ++i) {
        float* row = buf + (size_t)i * n_max;
        size_t at = sess[i]->remained.size();
        if (at) memcpy(row, sess[i]->remained.data(), at * 4);
        for (int k : feeds_of[i]) {
            const long m = feed_n[k];
            if (feed_format[k] == 1) {
                memcpy(row + at, feed_samples[k], (size_t)m * 4);
            } else {                                  // int16 PCM: x / 2^15 in float32 (buf_to_float, data_utils/utils.py:382-411)
                const int16_t* src = (const int16_t*)feed_samples[k];
                for (long j = 0; j < m; ++j) row[at + j] = (float)src[j] * (1.0f / 32768.0f);
            }
            at += (size_t)m;
        }
        if ((long)at < n_max) memset(row + at, 0, (size_t)(n_max - (long)at) * 4);
        lens_h[i] = (int32_t)L[i];
    }
    lap(0);
    PCHK(upload(p, p->samples, buf, (size_t)n * n_max * 4, s));
    PCHK(upload(p, p->lens, lens_h, (size_t)n * 4, s));

    // ---- gains: mean squares from the device, the scalar expressions of AudioSegment.normalize on the host ------------------
    float* gains_h = nullptr;
    if (p->use_db) {
        float* ms_h = nullptr;
        PCHK(p->pin.take((size_t)n * 4, (void**)&ms_h));
        PCHK(p->pin.take((size_t)n * 4, (void**)&gains_h));
        PCHK(p->ms.ensure((size_t)n * 4, 0, s));
        PCHK(masr_mean_square(p->e, p->samples.p, 1, p->lens.as<int32_t>(), n, (int32_t)n_max, p->ms.as<float>(), stream));
        PHIP(hipMemcpyAsync(ms_h, p->ms.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        PHIP(hipStreamSynchronize(s));
        lap(1);
        if (gain_fn) {
            if (gain_fn(ms_h, n, p->target_db, gains_h, gain_user)) PFAIL("masr_pool_step: the gain evaluator failed");
        } else {                                       // libm evaluation of audio.py:287-304,519-529 (float32 steps)
            for (int i = 0; i < n; ++i) {
                const float m = ms_h[i] == 0.f ? 1.f : ms_h[i];
                const float gain_db = p->target_db - 10.f * log10f(m);
                if (gain_db > 300.f) PFAIL("masr_pool_step: gain beyond max_gain_db (300 dB)");
                gains_h[i] = powf(10.f, gain_db / 20.f);
            }
        }
        PCHK(upload(p, p->gains, gains_h, (size_t)n * 4, s));
        lap(2);
    }
    // ---- ONE ragged feature launch over the pending samples ------------------------------------------------------------------
    const int tmax = p->method == 2 ? (int)((n_max - 320) / 160 + 1) : (int)(1 + (n_max - 400) / 160);
    PCHK(p->feats.ensure((size_t)n * tmax * p->F * 4, 0, s));
    const int mode = p->use_db ? 2 : 0;
    if (p->method == 0)
        PCHK(masr_fbank_batch(p->e, p->samples.p, 1, p->lens.as<int32_t>(), n, (int32_t)n_max, mode, p->target_db, p->feats.as<float>(),
                              nullptr, nullptr, p->gains.as<float>(), stream));
    else if (p->method == 1)
        PCHK(masr_mfcc_batch(p->e, p->samples.p, 1, p->lens.as<int32_t>(), n, (int32_t)n_max, mode, p->target_db, p->n_mfcc,
                             p->feats.as<float>(), nullptr, p->gains.as<float>(), stream));
    else
        PCHK(masr_linear_batch(p->e, p->samples.p, 1, p->lens.as<int32_t>(), n, (int32_t)n_max, mode, p->target_db, p->feats.as<float>(),
                               nullptr, p->gains.as<float>(), stream));

    // ---- bookkeeping: carried-over samples, new frames appended to the sessions' rows of the frame pool --------------------
    std::vector<int> fresh(n);
    int need = 0;
    for (int i = 0; i < n; ++i) {
        fresh[i] = feature_frames(p, L[i]);
        need = std::max(need, sess[i]->nf + fresh[i]);
    }
    if (need > p->feat_cap) PCHK(widen_feat(p, std::max(need, 2 * p->feat_cap), s));
    std::vector<PoolSeg> seg_h;
    int seg_rows = 0;
    for (int i = 0; i < n; ++i) {
        Session& q = *sess[i];
        const float* row = buf + (size_t)i * n_max;
        const long from = 160L * fresh[i];
        q.remained.assign(row + from, row + L[i]);                   // normalised in place, like AudioSegment.normalize
        if (p->use_db && L[i] > 0)
            for (float& v : q.remained) v *= gains_h[i];
        if (q.f0 + q.nf + fresh[i] > p->feat_cap) {                   // make room: the live frames move to the front of the row
            float* base = p->feat.as<float>() + (size_t)q.row * p->feat_cap * p->F;
            PCHK(p->win.ensure((size_t)q.nf * p->F * 4, 0, s));
            PHIP(hipMemcpyAsync(p->win.p, base + (size_t)q.f0 * p->F, (size_t)q.nf * p->F * 4, hipMemcpyDeviceToDevice, s));
            PHIP(hipMemcpyAsync(base, p->win.p, (size_t)q.nf * p->F * 4, hipMemcpyDeviceToDevice, s));
            q.f0 = 0;
        }
        if (fresh[i]) {
            seg_h.push_back(PoolSeg{(long)i * tmax, (long)q.row * p->feat_cap + q.f0 + q.nf, fresh[i], 0});
            seg_rows = std::max(seg_rows, fresh[i]);
        }
        q.nf += fresh[i];
    }
    // the step's segment tables live in ONE device array; every launch gets its own slice
    size_t seg_total = seg_h.size();
    // windows of every session (predict.py:283-306), advanced in lock-step
    std::vector<std::vector<std::pair<int, int>>> plans(n);
    size_t lockstep = 0;
    for (int i = 0; i < n; ++i) {
        const int nfr = sess[i]->nf;
        if ((nfr < kWindow && !end_flag[i]) || nfr < kContext) continue;
        const int left = end_flag[i] ? kContext : kWindow;
        for (int cur = 0; cur <= nfr - left; cur += kStride) plans[i].push_back({cur, std::min(cur + kWindow, nfr)});
        lockstep = std::max(lockstep, plans[i].size());
        seg_total += 2 * plans[i].size();
    }
    PCHK(p->segs.ensure((seg_total + 8) * sizeof(PoolSeg), 0, s));
    PoolSeg* seg_pin = nullptr;
    PCHK(p->pin.take((seg_total + 8) * sizeof(PoolSeg), (void**)&seg_pin));
    size_t seg_at = 0;
    auto push_segments = [&](const std::vector<PoolSeg>& v, const PoolSeg** dev) -> int {
        memcpy(seg_pin + seg_at, v.data(), v.size() * sizeof(PoolSeg));
        PHIP(hipMemcpyAsync(p->segs.as<PoolSeg>() + seg_at, seg_pin + seg_at, v.size() * sizeof(PoolSeg), hipMemcpyHostToDevice, s));
        *dev = p->segs.as<PoolSeg>() + seg_at;
        seg_at += v.size();
        return 0;
    };
    if (!seg_h.empty()) {
        const PoolSeg* d = nullptr;
        PCHK(push_segments(seg_h, &d));
        launch_copy_segments(p->feats.as<float>(), p->feat.as<float>(), nullptr, nullptr, d, (int)seg_h.size(), seg_rows, p->F, s);
    }
    std::vector<int32_t> ids;
    lap(3);
    for (size_t k = 0; k < lockstep; ++k) {
        // full windows together; a short last window on its own (lengths in ascending order of first appearance)
        std::vector<int> lengths;
        for (int i = 0; i < n; ++i)
            if (k < plans[i].size()) {
                const int len = plans[i][k].second - plans[i][k].first;
                if (std::find(lengths.begin(), lengths.end(), len) == lengths.end()) lengths.push_back(len);
            }
        for (int len : lengths) {
            std::vector<PoolSeg> gat, sca;
            ids.clear();
            int32_t tq = 0;
            PCHK(masr_encoder_frames(p->e, len, &tq));
            int hist_need = 0;
            for (int i = 0; i < n; ++i) {
                if (k >= plans[i].size() || plans[i][k].second - plans[i][k].first != len) continue;
                Session& q = *sess[i];
                const long item = (long)ids.size();
                gat.push_back(PoolSeg{(long)q.row * p->feat_cap + q.f0 + plans[i][k].first, item * len, len, 0});
                sca.push_back(PoolSeg{item * tq, 0, tq, i});            // dst filled in once the history width is final
                hist_need = std::max(hist_need, q.frames + tq);
                ids.push_back(q.sid);
            }
            const int m = (int)ids.size();
            PCHK(grow_state(p, 0, hist_need, s));
            for (PoolSeg& g : sca) {
                Session& q = *sess[g.pad_];
                g.dst = (long)q.row * p->hist_cap + q.frames;
                g.pad_ = 0;
            }
            PCHK(p->win.ensure((size_t)m * len * p->F * 4, 0, s));
            PCHK(p->idx.ensure((size_t)m * std::max(tq, 1) * 4, 0, s));
            PCHK(p->mp.ensure((size_t)m * std::max(tq, 1) * 4, 0, s));
            const PoolSeg *dg = nullptr, *ds = nullptr;
            PCHK(push_segments(gat, &dg));
            launch_copy_segments(p->feat.as<float>(), p->win.as<float>(), nullptr, nullptr, dg, m, len, p->F, s);
            PCHK(masr_encode_chunk(p->e, ids.data(), m, p->win.as<float>(), len, nullptr, p->idx.as<int32_t>(), p->mp.as<float>(), stream));
            if (tq > 0) {
                PCHK(push_segments(sca, &ds));
                launch_copy_segments(reinterpret_cast<const float*>(p->idx.p), p->hist_idx.as<float>(), p->mp.as<float>(),
                                     p->hist_mp.as<float>(), ds, m, tq, 1, s);
            }
            for (int i = 0; i < n; ++i)
                if (k < plans[i].size() && plans[i][k].second - plans[i][k].first == len) sess[i]->frames += tq;
        }
    }
    // ---- greedy: ONE collapse launch for every session that advanced (full-history best path + score, greedy_decoder_chunk
    // semantics, ctc_greedy_decoder.py:52-89); ONE copy back ------------------------------------------------------------------
    lap(4);
    std::vector<int> adv;
    int hmax = 0;
    for (int i = 0; i < n; ++i)
        if (!plans[i].empty()) {
            adv.push_back(i);
            hmax = std::max(hmax, sess[i]->frames);
        }
    const int na = (int)adv.size(), width = hmax + 2;
    int32_t* rows_h = nullptr;
    if (na) {
        int32_t* meta_h = nullptr;                 // [rows | frame counts]
        PCHK(p->pin.take((size_t)2 * na * 4, (void**)&meta_h));
        for (int j = 0; j < na; ++j) {
            meta_h[j] = sess[adv[j]]->row;
            meta_h[na + j] = sess[adv[j]]->frames;
        }
        PCHK(upload(p, p->meta, meta_h, (size_t)2 * na * 4, s));
        PCHK(p->rows_out.ensure((size_t)na * width * 4, 0, s));
        launch_ctc_collapse_hist(p->hist_idx.as<int>(), p->hist_mp.as<float>(), p->meta.as<int>() + na, p->meta.as<int>(), p->hist_cap,
                                 na, hmax, 0, p->rows_out.as<int>(), s);
        PCHK(p->pin.take((size_t)na * width * 4, (void**)&rows_h));
        PHIP(hipMemcpyAsync(rows_h, p->rows_out.p, (size_t)na * width * 4, hipMemcpyDeviceToHost, s));
    }
    PHIP(hipGetLastError());
    PHIP(hipEventRecord(p->ev, s));
    p->pending = true;
    if (na) PHIP(hipStreamSynchronize(s));
    lap(5);
    ++p->prof_steps;
    // ---- results + the windows' bookkeeping (predict.py:329: keep the overlap frames) -----------------------------------------
    for (int i = 0; i < n; ++i) {
        Session& q = *sess[i];
        if (!plans[i].empty()) {
            const int used = plans[i].back().second - kOverlap;
            q.f0 += used;
            q.nf -= used;
        }
        p->handles.push_back(q.sid);
        p->state.push_back(plans[i].empty() ? 0 : 1);
    }
    for (int32_t h : refused) {
        p->handles.push_back(h);
        p->state.push_back(-1);
    }
    *n_sessions = n + (int32_t)refused.size();
    if (handles_out) *handles_out = p->handles.data();
    if (state_out) *state_out = p->state.data();
    if (rows_host) *rows_host = rows_h;
    if (row_width) *row_width = na ? width : 0;
    if (rows_dev) *rows_dev = na ? p->rows_out.as<int32_t>() : nullptr;
    return 0;
}

int masr_pool_profile(masr_pool* p, double* phase_ms, int64_t* steps, int32_t reset) {
    if (!p || !phase_ms || !steps) PFAIL("null argument");
    for (int k = 0; k < 6; ++k) phase_ms[k] = p->prof[k];
    *steps = p->prof_steps;
    if (reset) {
        for (double& v : p->prof) v = 0;
        p->prof_steps = 0;
    }
    return 0;
}

}  // extern "C"
