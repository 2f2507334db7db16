// External n-gram language model scorer of the CTC beam search (reference: the `Scorer` of the third-party
// paddlespeech_ctcdecoders that masr/decoders/beam_search_decoder.py:29-42 and swig_wrapper.py:4-18 construct; scorer.cpp of
// DeepSpeech2 / PaddleSpeech ctc_decoders on top of KenLM -- un-vendored, absent here: parity UNPINNED, the published algorithm
// is restated).  Character-based models only (every LM word is one vocabulary token, as the reference's Mandarin LMs are).
//
// Layout shared by the host search (beam_search.cpp) and the GPU kernel (beam_gpu.hip): ONE open-addressing hash table of all
// n-grams of all orders.  Key = 64-bit hash chain over the words from the last one backwards (+ the order) with w = vocabulary
// token id (+ <s>, </s> behind them); value = (ln P, ln backoff).  16-byte entries, linear probing, load <= 1/3, key 0 =
// empty slot.  A lookup is one 16-byte load per probe; the probes of one backoff recursion are issued together.  In HBM the table of a pruned 5-gram Mandarin LM (~10^8 n-grams) is ~4 GB -- resident next to the
// 138 MB of encoder weights; nothing is paged.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LM_HD __host__ __device__ __forceinline__
#else
#define LM_HD inline
#endif

namespace masr {

struct LmEntry {
    unsigned long long key;   // 0 = empty
    float prob;               // ln P(w_n | w_1 .. w_{n-1}) as stored in the ARPA file (log10 there)
    float backoff;            // ln backoff weight of this n-gram as a context (0 when the file has none)
};

struct LmView {
    const LmEntry* table;
    unsigned long long mask;          // slots - 1 (power of two)
    const unsigned char* known;       // [n_words]: 1 = the token is an LM word (unigram present and not <unk>)
    int max_order;                    // <= 5 (contexts are packed 4 x 16 bits)
    int n_words;                      // vocabulary size + 2
    int bos, eos;                     // ids of <s> and </s> (= vocabulary size, vocabulary size + 1)
};

static constexpr float LM_OOV_SCORE = -1000.0f;   // scorer.h OOV_SCORE: returned as it is (not converted from log10)

// one word into the 64-bit chain state, on 32-bit halves: two 32-bit multiplies (a 64-bit multiply is ~8 quarter-rate
// instructions on the vector ALU and this runs ~60 000 times per frame of a beam-300 search)
LM_HD unsigned long long lm_mix(unsigned long long h, unsigned long long w) {
    unsigned a = (unsigned)h, b = (unsigned)(h >> 32);
    a = (a ^ ((unsigned)w + 0x7F4A7C15u)) * 0x9E3779B1u;
    a ^= a >> 15;
    b = (b ^ a) * 0x85EBCA77u;
    b ^= b >> 13;
    a = (a + b) * 0xC2B2AE3Du;
    a ^= a >> 16;
    return ((unsigned long long)b << 32) | a;
}
// Keys are hash chains that start at the LAST word of the n-gram and walk backwards: h_1 = mix(seed, w_n), h_2 = mix(h_1, w_{n-1}),
// ... ; key of the n-gram = fin(h_n, n).  One pass over a prefix's packed words (most recent in the low bits) therefore yields the
// keys of ALL its suffix n-grams -- what the backoff recursion looks up.
static constexpr unsigned long long LM_SEED = 0x5851F42D4C957F2Dull;
LM_HD unsigned long long lm_fin(unsigned long long h, int n) {
    const unsigned long long k = h ^ ((unsigned long long)n * 0xA24BAED4963EE407ull);
    return k ? k : 1ull;
}
// key of the n-gram (the `len` most recent words of ctx, then w)
LM_HD unsigned long long lm_key(unsigned long long ctx, int len, int w) {
    unsigned long long h = lm_mix(LM_SEED, (unsigned long long)w);
    for (int j = 0; j < len; ++j) h = lm_mix(h, (ctx >> (16 * j)) & 0xFFFFull);
    return lm_fin(h, len + 1);
}
// key of the n-gram made of the `len` most recent words of ctx themselves (len >= 1)
LM_HD unsigned long long lm_key_ctx(unsigned long long ctx, int len) { return lm_key(ctx >> 16, len - 1, (int)(ctx & 0xFFFFull)); }

LM_HD void lm_load_entry(const LmView& lm, unsigned long long slot, unsigned long long* k, float* p, float* b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 raw = *reinterpret_cast<const uint4*>(lm.table + slot);          // one 16-byte load per probe
    *k = ((unsigned long long)raw.y << 32) | raw.x;
    *p = __uint_as_float(raw.z);
    *b = __uint_as_float(raw.w);
#else
    const LmEntry e = lm.table[slot];
    *k = e.key;
    *p = e.prob;
    *b = e.backoff;
#endif
}
LM_HD bool lm_find(const LmView& lm, unsigned long long key, float* prob, float* backoff) {
    unsigned long long h = key & lm.mask;
    while (true) {
        unsigned long long k;
        float p, b;
        lm_load_entry(lm, h, &k, &p, &b);
        if (k == key) {
            *prob = p;
            *backoff = b;
            return true;
        }
        if (k == 0ull) return false;
        h = (h + 1) & lm.mask;
    }
}

// What a live prefix carries for scoring its extensions: its last max_order - 1 words (<s>-padded, scorer.cpp make_ngram),
// the longest suffix of them that exists as an n-gram in the model (`m` words; ARPA models are prefix-closed, so no longer
// n-gram can match any extension), the backoff weights of those suffixes, and whether one of the words is unknown to the LM.
struct LmState {
    unsigned long long ctx;   // 4 x 16 bits, most recent word in the low bits
    float bo[4];              // bo[j] = ln backoff of the (j + 1)-word suffix, j < m
    int m;                    // 0 .. max_order - 1
    int oov;                  // 1: an unknown word among the last max_order - 1 words -> every extension scores OOV
};

LM_HD LmState lm_state_of(const LmView& lm, unsigned long long ctx) {
    LmState s;
    s.ctx = ctx;
    s.m = 0;
    s.oov = 0;
    const int k = lm.max_order - 1;
    for (int j = 0; j < 4; ++j) s.bo[j] = 0.f;
    for (int j = 0; j < k; ++j) {
        const int w = (int)((ctx >> (16 * j)) & 0xFFFFull);
        if (w >= lm.n_words || !lm.known[w]) s.oov = 1;
    }
    // keys of the 1 .. k word suffixes in one pass over the packed words, their first probes issued together (one memory round
    // trip instead of one per suffix length); linear probing continues only where a slot is taken by another n-gram
    unsigned long long key[4], k0[4];
    float p0[4], b0[4];
    unsigned long long h = LM_SEED;
#pragma unroll
    for (int len = 1; len <= 4; ++len) {
        if (len <= k) {
            h = lm_mix(h, (ctx >> (16 * (len - 1))) & 0xFFFFull);
            key[len - 1] = lm_fin(h, len);
        }
    }
#pragma unroll
    for (int len = 1; len <= 4; ++len)
        if (len <= k) lm_load_entry(lm, key[len - 1] & lm.mask, &k0[len - 1], &p0[len - 1], &b0[len - 1]);
    bool open = true;
#pragma unroll
    for (int len = 1; len <= 4; ++len) {
        if (len > k || !open) continue;
        bool hit = k0[len - 1] == key[len - 1];
        float p = p0[len - 1], b = b0[len - 1];
        if (!hit && k0[len - 1] != 0ull) hit = lm_find(lm, key[len - 1], &p, &b);
        if (!hit) {
            open = false;
            continue;
        }
        s.bo[len - 1] = b;
        s.m = len;
    }
    return s;
}
LM_HD unsigned long long lm_root_ctx(const LmView& lm) {      // the empty prefix: <s> <s> <s> <s>
    const unsigned long long b = (unsigned long long)lm.bos;
    return b | (b << 16) | (b << 32) | (b << 48);
}
LM_HD unsigned long long lm_push(unsigned long long ctx, int w) { return (ctx << 16) | (unsigned long long)(w & 0xFFFF); }

// ln P(w | the state's words): Scorer::get_log_cond_prob on make_ngram(prefix + w) -- KenLM BaseScore from the null context
// == the ARPA backoff recursion: the longest (suffix, w) n-gram in the model, plus the backoff weights of the longer suffixes
// (w_known: lm.known[w], looked up by the caller -- the GPU search does it once per frame and candidate, not once per pair)
LM_HD float lm_cond(const LmView& lm, const LmState& s, int w, bool w_known) {
    if (s.oov || !w_known) return LM_OOV_SCORE;
    // keys of (suffix of length len, w) for len = 0 .. m in one pass; their first probes are independent loads, all in flight
    // before the first is looked at (one memory round trip instead of m + 1); linear probing continues only on a collision
    unsigned long long key[5], k0[5];
    float p0[5], b0[5];
    unsigned long long h = lm_mix(LM_SEED, (unsigned long long)w);
#pragma unroll
    for (int len = 0; len < 5; ++len) {
        key[len] = lm_fin(h, len + 1);
        if (len < s.m) h = lm_mix(h, (s.ctx >> (16 * len)) & 0xFFFFull);
    }
#pragma unroll
    for (int len = 0; len < 5; ++len)
        if (len <= s.m) lm_load_entry(lm, key[len] & lm.mask, &k0[len], &p0[len], &b0[len]);
    float acc = 0.f;
#pragma unroll
    for (int len = 4; len >= 0; --len) {
        if (len > s.m) continue;
        bool hit = k0[len] == key[len];
        float p = p0[len], b;
        if (!hit && k0[len] != 0ull) hit = lm_find(lm, key[len], &p, &b);      // occupied by another n-gram: walk on
        if (hit) return acc + p;
        if (len > 0) acc += s.bo[len - 1];
    }
    return LM_OOV_SCORE;       // (a known word always has its unigram)
}
LM_HD float lm_cond(const LmView& lm, const LmState& s, int w) { return lm_cond(lm, s, w, w >= 0 && w < lm.n_words && lm.known[w] != 0); }

// The same score with as few table probes as possible, for the GPU search: that kernel is bound by the number of divergent
// 16-byte loads it issues (12 000 extensions per frame, measured: putting MORE probes in flight per thread makes it slower), not
// by their latency.  Longest n-gram first, stop at the first hit; the unigram never goes to memory -- `uni` is ln P(w) as the
// caller looked it up once per frame and candidate.  ORD >= the model's order bounds the keys held in registers.
template <int ORD>
LM_HD float lm_cond_desc(const LmView& lm, const LmState& s, int w, float uni) {
    unsigned long long key[ORD > 1 ? ORD - 1 : 1];
    unsigned long long h = lm_mix(LM_SEED, (unsigned long long)w);
#pragma unroll
    for (int len = 1; len < ORD; ++len) {
        if (len <= s.m) {
            h = lm_mix(h, (s.ctx >> (16 * (len - 1))) & 0xFFFFull);
            key[len - 1] = lm_fin(h, len + 1);
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int len = ORD - 1; len >= 1; --len) {
        if (len > s.m) continue;
        float p, b;
        if (lm_find(lm, key[len - 1], &p, &b)) return acc + p;
        acc += s.bo[len - 1];
    }
    return acc + uni;
}

// ln P(w | the state's words) exactly as lm_cond_desc returns it AND the scorer state of the prefix extended by w, i.e. m and bo[] of
// lm_state_of(lm, lm_push(s.ctx, w)), for a KNOWN word w behind a context without unknown words: the n-grams (j most recent words
// of the context, w), j = 1 .. s.m, are the ones the backoff recursion reads and -- for j <= max_order - 2 -- the (j + 1)-word
// suffixes of the new context (the one-word suffix is w's unigram: `uni` / `ubo` = its ln P and ln backoff, looked up by the caller
// once per frame and candidate; no longer suffix can exist because ARPA models are prefix-closed).  The first probes of all lengths
// are in flight together: one memory round trip for the score and the state, where lm_cond_desc + lm_state_of make 2 - 5.
// m_next: matched suffix length of the new context; bo_next[0 .. ORD-2]: its backoffs (bo_next[0] = ubo), 0 beyond m_next.
template <int ORD>
LM_HD float lm_cond_next(const LmView& lm, const LmState& s, int w, float uni, float ubo, int* m_next, float* bo_next) {
    constexpr int NK = ORD > 1 ? ORD - 1 : 1;
    unsigned long long key[NK], k0[NK];
    float p0[NK], b0[NK];
    bool hit[NK];
    unsigned long long h = lm_mix(LM_SEED, (unsigned long long)w);
#pragma unroll
    for (int len = 1; len < ORD; ++len) {
        key[len - 1] = 0ull; k0[len - 1] = 0ull; p0[len - 1] = 0.f; b0[len - 1] = 0.f;
        if (len <= s.m) {
            h = lm_mix(h, (s.ctx >> (16 * (len - 1))) & 0xFFFFull);
            key[len - 1] = lm_fin(h, len + 1);
        }
    }
#pragma unroll
    for (int len = 1; len < ORD; ++len)
        if (len <= s.m) lm_load_entry(lm, key[len - 1] & lm.mask, &k0[len - 1], &p0[len - 1], &b0[len - 1]);
#pragma unroll
    for (int len = 1; len < ORD; ++len) {
        hit[len - 1] = false;
        if (len <= s.m) {
            hit[len - 1] = k0[len - 1] == key[len - 1];
            if (!hit[len - 1] && k0[len - 1] != 0ull) hit[len - 1] = lm_find(lm, key[len - 1], &p0[len - 1], &b0[len - 1]);
        }
    }
    float acc = 0.f, val = 0.f;
    bool done = false;
#pragma unroll
    for (int len = ORD - 1; len >= 1; --len) {
        if (len > s.m || done) continue;
        if (hit[len - 1]) {
            val = acc + p0[len - 1];
            done = true;
        } else {
            acc += s.bo[len - 1];
        }
    }
    if (!done) val = acc + uni;
    const int kk = lm.max_order - 1;
    int m2 = 0;
#pragma unroll
    for (int j = 0; j < NK; ++j) bo_next[j] = 0.f;
    if (kk >= 1) {
        m2 = 1;
        bo_next[0] = ubo;
        bool open = true;
#pragma unroll
        for (int len = 1; len < ORD - 1; ++len) {
            if (open && len + 1 <= kk && len <= s.m && hit[len - 1]) {
                m2 = len + 1;
                bo_next[len] = b0[len - 1];
            } else {
                open = false;
            }
        }
    }
    *m_next = m2;
    return val;
}

}  // namespace masr
