// Host side of the boundary: the utterances of a device pass (separate host arrays of different lengths) into ONE padded, pinned
// staging buffer [B, n_max] that is then copied up in a single transfer -- the batch the reference forms with its collate_fn
// (masr/data_utils/collate_fn.py:6-34: zero-padded to the longest, lengths beside it), done on the audio instead of on the
// features.  Four worker threads that live with the library take rows in turn (20 MB per pass of 32 x 20 s: a copy bound by host
// memory bandwidth; python threads over numpy row assignments took 0.45 ms for it, one thread 0.66 ms).
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/masr_hip.h"

namespace {

struct StageJob {
    char* dst = nullptr;
    int64_t row_bytes = 0;
    const void* const* src = nullptr;
    const int32_t* n = nullptr;
    int32_t B = 0, sample_bytes = 0;
    std::atomic<int> next{0};
};

struct StagePool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> workers;
    StageJob* job = nullptr;
    uint64_t epoch = 0;
    int running = 0;
    bool stop = false;

    static void rows(StageJob* j) {
        for (;;) {
            const int i = j->next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j->B) return;
            const int64_t valid = (int64_t)j->n[i] * j->sample_bytes;
            char* d = j->dst + (int64_t)i * j->row_bytes;
            const int64_t m = valid < j->row_bytes ? valid : j->row_bytes;
            if (m > 0) memcpy(d, j->src[i], (size_t)m);
            if (m < j->row_bytes) memset(d + m, 0, (size_t)(j->row_bytes - m));
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            StageJob* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || epoch != seen; });
                if (stop) return;
                seen = epoch;
                j = job;
            }
            rows(j);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--running == 0) cv_done.notify_all();
            }
        }
    }
    void run(StageJob* j, int threads) {
        std::unique_lock<std::mutex> lk(mu);
        while ((int)workers.size() < threads - 1) workers.emplace_back([this] { loop(); });
        job = j;
        running = (int)workers.size();
        ++epoch;
        cv_work.notify_all();
        lk.unlock();
        rows(j);                                   // the calling thread works too
        lk.lock();
        cv_done.wait(lk, [&] { return running == 0; });
        job = nullptr;
    }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
};

StagePool& pool() {
    static StagePool p;
    return p;
}
std::mutex g_call_mu;       // one staging call at a time (a server's worker threads share the pool)

}  // namespace

extern "C" int masr_stage_rows(void* dst, int64_t row_bytes, const void* const* src, const int32_t* n_samples, int32_t B,
                               int32_t sample_bytes, int32_t threads) {
    if (!dst || !src || !n_samples || B < 0 || row_bytes < 0 || sample_bytes <= 0) return 1;
    if (B == 0 || row_bytes == 0) return 0;
    StageJob j;
    j.dst = (char*)dst;
    j.row_bytes = row_bytes;
    j.src = src;
    j.n = n_samples;
    j.B = B;
    j.sample_bytes = sample_bytes;
    std::lock_guard<std::mutex> lk(g_call_mu);
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if (threads == 1 || (int64_t)B * row_bytes < (1 << 20)) {
        StagePool::rows(&j);
        return 0;
    }
    pool().run(&j, threads);
    return 0;
}
