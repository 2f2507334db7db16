// CTC prefix beam search (LM-free part) -- host side of the ctc_beam_search decoder.
//
// The reference delegates this to the third-party SWIG module `paddlespeech_ctcdecoders`
// (masr/decoders/swig_wrapper.py:35-121, beam_search_decoder.py:45-96; DeepSpeech2/PaddleSpeech
// ctc_beam_search_decoder.cpp + path_trie.cpp), which is NOT vendored, not version-pinned and needs a
// 2.8 GB KenLM model: parity for this row is UNPINNED.  This file restates the published algorithm
// without the external scorer (alpha = 0 path): per frame the vocabulary is pruned to the smallest
// prefix of the descending-probability order whose mass reaches cutoff_prob (at most cutoff_top_n
// entries) -- done on the GPU by topk_prune_kernel (elementwise.hip) -- and the candidates extend a
// prefix trie that tracks log P(prefix ending in blank) / log P(prefix ending in non-blank).
// Like the reference (a C++ thread pool of num_processes), the search itself runs on host threads.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/masr_hip.h"
#include "lm_scorer.h"

namespace masr {
LmView lm_host_view(const masr_lm* lm);       // lm_scorer.cpp
bool lm_word_based(const masr_lm* lm);
int lm_space_id(const masr_lm* lm);
int lm_dict_next(const masr_lm* lm, int state, int token);
bool lm_dict_final(const masr_lm* lm, int state);
int lm_dict_word(const masr_lm* lm, int state);
float lm_cond_words(const LmView& v, const int* ctx, int n_ctx, int w);
float lm_sentence_words(const LmView& v, const int* ids, int n);
}
using masr::LmState;
using masr::LmView;

namespace {

const float NEG_INF = -std::numeric_limits<float>::infinity();

inline float log_sum_exp(float x, float y) {
    if (x == NEG_INF) return y;
    if (y == NEG_INF) return x;
    const float m = x > y ? x : y;
    return m + logf(expf(x - m) + expf(y - m));
}

struct Node {
    float b_prev = NEG_INF, nb_prev = NEG_INF, b_cur = NEG_INF, nb_cur = NEG_INF, score = NEG_INF;
    int ch = -1;          // character of this node (-1 = root)
    int parent = -1;
    bool exists = true;
    LmState lm{};         // character-based scorer: state of this prefix (its last words; lm_scorer.h)
    // word-based scorer: state of the spelling dictionary after this prefix (PathTrie::dictionary_state_) and the ids of the
    // complete words before the one being spelled, most recent first (what Scorer::make_ngram walks the trie for)
    int dstate = 0;
    int wctx[4] = {0, 0, 0, 0};
    std::vector<std::pair<int, int>> kids;   // (character, node index)
};

struct Beam {
    std::vector<Node> pool;
    std::vector<int> prefixes;               // node indices, first min(size, beam) are live candidates
    int beam_size = 300, blank = 0;
    // external scorer (ctc_beam_search_decoder.cpp: a new prefix p + c adds alpha * ln P_LM(c | p) + beta); lm == nullptr: off
    const masr_lm* lm = nullptr;
    LmView view{};
    float alpha = 0.f, beta = 0.f;
    bool word_based = false;
    int space_id = -2;

    void set_lm(const masr_lm* m, float a, float b) {
        lm = m;
        alpha = a;
        beta = b;
        word_based = false;
        if (lm) {
            view = masr::lm_host_view(lm);
            word_based = masr::lm_word_based(lm);
            space_id = masr::lm_space_id(lm);
        }
    }

    void reset() {
        pool.clear();
        pool.emplace_back();
        pool[0].score = 0.f;
        pool[0].b_prev = 0.f;
        if (lm && !word_based) pool[0].lm = masr::lm_state_of(view, masr::lm_root_ctx(view));
        if (lm && word_based)
            for (int j = 0; j < 4; ++j) pool[0].wctx[j] = view.bos;
        prefixes.assign(1, 0);
    }

    // PathTrie::get_path_trie(new_char): find (reviving a dead node) or create; -1 when a word-based scorer's dictionary has no
    // such continuation -- and if the prefix just completed a word, THAT failed attempt moves it to the dictionary's start (the
    // published code's `if (is_final && reset) dictionary_state_ = dictionary_->Start()`): the next attempt starts a new word
    int child(int n, int c) {
        for (auto& kv : pool[n].kids)
            if (kv.first == c) {
                Node& k = pool[kv.second];
                if (!k.exists) {
                    k.exists = true;
                    k.b_prev = k.nb_prev = k.b_cur = k.nb_cur = k.score = NEG_INF;
                }
                return kv.second;
            }
        Node k;
        k.ch = c;
        k.parent = n;
        if (lm && !word_based) k.lm = masr::lm_state_of(view, masr::lm_push(pool[n].lm.ctx, c));
        if (lm && word_based) {
            const int nx = masr::lm_dict_next(lm, pool[n].dstate, c);
            if (nx < 0) {
                if (masr::lm_dict_final(lm, pool[n].dstate)) pool[n].dstate = 0;
                return -1;
            }
            k.dstate = nx;
            for (int j = 0; j < 4; ++j) k.wctx[j] = pool[n].wctx[j];
            if (c == space_id) {             // the word spelled up to n is complete: it joins the context of what follows
                for (int j = 3; j > 0; --j) k.wctx[j] = k.wctx[j - 1];
                k.wctx[0] = word_of(n);
            }
        }
        pool.push_back(k);
        const int id = (int)pool.size() - 1;
        pool[n].kids.emplace_back(c, id);
        return id;
    }

    // id of the LM word that the characters since the last space spell (-1: none, e.g. an unfinished word).  The dictionary state
    // cannot be used after the start-over quirk above, so the trie is walked like Scorer::make_ngram does
    int word_of(int n) const {
        std::vector<int> rev;
        while (n > 0 && pool[n].ch != space_id) {
            rev.push_back(pool[n].ch);
            n = pool[n].parent;
        }
        int st = 0;
        for (size_t i = rev.size(); i-- > 0 && st >= 0;) st = masr::lm_dict_next(lm, st, rev[i]);
        return st > 0 ? masr::lm_dict_word(lm, st) : -1;
    }

    // word-based: alpha * ln P_LM(word ending at n | the words before it) + beta  (make_ngram(prefix) + get_log_cond_prob)
    float word_score(int n) const {
        return alpha * masr::lm_cond_words(view, pool[n].wctx, view.max_order - 1, word_of(n)) + beta;
    }

    void collect(int n, std::vector<int>& out) {   // PathTrie::iterate_to_vec
        Node& nd = pool[n];
        if (nd.exists) {
            nd.b_prev = nd.b_cur;
            nd.nb_prev = nd.nb_cur;
            nd.b_cur = nd.nb_cur = NEG_INF;
            nd.score = log_sum_exp(nd.b_prev, nd.nb_prev);
            out.push_back(n);
        }
        for (size_t i = 0; i < pool[n].kids.size(); ++i) collect(pool[n].kids[i].second, out);
    }

    void remove(int n) {                     // PathTrie::remove: drop the node, and childless dead ancestors
        pool[n].exists = false;
        while (n > 0 && !pool[n].exists && pool[n].kids.empty()) {
            const int par = pool[n].parent;
            auto& pk = pool[par].kids;
            for (size_t i = 0; i < pk.size(); ++i)
                if (pk[i].second == n) {
                    pk.erase(pk.begin() + i);
                    break;
                }
            n = par;
        }
    }

    bool better(int a, int b) const {        // prefix_compare: score desc, then character asc
        const Node &x = pool[a], &y = pool[b];
        if (x.score == y.score) return x.ch != y.ch && x.ch < y.ch;
        return x.score > y.score;
    }

    // one time step: cand = (index, log prob) pairs in descending probability order.  `blank_lp` = ln p(blank) of the frame, the
    // input of the decoder's pruning rule (NaN: rule off): with a scorer bound and a full beam, (prefix, c) -- and every prefix
    // behind it in score order -- is skipped once log p(c) + score(prefix) < score(worst live prefix) + ln p(blank) - max(0, beta)
    void step(const int32_t* idx, const float* logp, int count, float blank_lp) {
        const size_t live = std::min(prefixes.size(), (size_t)beam_size);
        float min_cutoff = NEG_INF;
        bool full_beam = false;
        if (lm) {
            std::sort(prefixes.begin(), prefixes.begin() + live, [this](int a, int b) { return better(a, b); });
            if (blank_lp == blank_lp) {
                min_cutoff = (float)((double)pool[prefixes[live - 1]].score + (double)blank_lp - std::max(0.0, (double)beta));
                full_beam = (int)live == beam_size;
            }
        }
        for (int k = 0; k < count; ++k) {
            const int c = idx[k];
            const float lp = logp[k];
            for (size_t i = 0; i < live; ++i) {
                const int p = prefixes[i];
                if (full_beam && lp + pool[p].score < min_cutoff) break;
                if (c == blank) {
                    pool[p].b_cur = log_sum_exp(pool[p].b_cur, lp + pool[p].score);
                    continue;
                }
                if (c == pool[p].ch) pool[p].nb_cur = log_sum_exp(pool[p].nb_cur, lp + pool[p].nb_prev);
                const int pc = pool[p].ch;
                const float pscore = pool[p].score, pb = pool[p].b_prev;
                float add = NEG_INF;
                if (c == pc && pb > NEG_INF) add = lp + pb;
                else if (c != pc) add = lp + pscore;
                if (lm && !word_based && add > NEG_INF) add += alpha * masr::lm_cond(view, pool[p].lm, c) + beta;
                const int q = child(p, c);          // may reallocate the pool: no references held across it
                if (q < 0) continue;                // the dictionary of a word-based scorer has no such continuation
                if (lm && word_based && c == space_id && add > NEG_INF) add += word_score(p);
                pool[q].nb_cur = log_sum_exp(pool[q].nb_cur, add);
            }
        }
        prefixes.clear();
        collect(0, prefixes);
        if ((int)prefixes.size() >= beam_size) {
            std::nth_element(prefixes.begin(), prefixes.begin() + beam_size, prefixes.end(),
                             [this](int a, int b) { return better(a, b); });
            for (size_t i = beam_size; i < prefixes.size(); ++i) remove(prefixes[i]);
            prefixes.resize(beam_size);
        }
    }

    // best hypothesis so far: token ids (root -> leaf) and its log probability
    int best(int32_t* tokens, int max_len, float* score) {
        const size_t live = std::min(prefixes.size(), (size_t)beam_size);
        // a word-based scorer also scores the unfinished last word of every prefix (end of ctc_beam_search_decoder); done on a
        // copy of the scores so that a streaming search can be asked for its best prefix between chunks
        std::vector<float> final_score(live);
        for (size_t i = 0; i < live; ++i) {
            const int n = prefixes[i];
            final_score[i] = pool[n].score;
            if (lm && word_based && n > 0 && pool[n].ch != space_id) final_score[i] += word_score(n);
        }
        size_t bi = 0;
        for (size_t i = 1; i < live; ++i) {
            const Node &x = pool[prefixes[i]], &y = pool[prefixes[bi]];
            if (final_score[i] > final_score[bi] || (final_score[i] == final_score[bi] && x.ch < y.ch)) bi = i;
        }
        int n = prefixes[bi];
        *score = final_score[bi];
        std::vector<int> rev;
        while (n > 0) {
            rev.push_back(pool[n].ch);
            n = pool[n].parent;
        }
        const int len = std::min((int)rev.size(), max_len);
        for (int i = 0; i < len; ++i) tokens[i] = rev[rev.size() - 1 - i];
        if (lm) {
            // approx_ctc: the scorer's share is taken out of the reported score again -- |prefix| * beta and
            // alpha * ln P_LM(sentence), the sentence probability counting </s> too (Scorer::get_sent_log_prob)
            const int L = (int)rev.size();
            float sent = 0.f;
            if (!word_based) {
                unsigned long long ctx = masr::lm_root_ctx(view);
                sent = L == 0 ? masr::lm_cond(view, masr::lm_state_of(view, ctx), view.bos) : 0.f;
                for (int i = 0; i <= L; ++i) {
                    const int w = i < L ? rev[L - 1 - i] : view.eos;
                    sent += masr::lm_cond(view, masr::lm_state_of(view, ctx), w);
                    ctx = masr::lm_push(ctx, w);
                }
            } else {
                // Scorer::split_labels: the space-separated words of the transcript
                std::vector<int> words;
                if (L > 0) {
                    int st = 0;
                    bool bad = false;
                    for (int i = L - 1; i >= -1; --i) {
                        const int c = i >= 0 ? rev[i] : space_id;
                        if (c == space_id) {
                            words.push_back(!bad && st > 0 ? masr::lm_dict_word(lm, st) : -1);
                            st = 0;
                            bad = false;
                        } else if (!bad) {
                            st = masr::lm_dict_next(lm, st, c);
                            if (st < 0) bad = true;
                        }
                    }
                }
                sent = masr::lm_sentence_words(view, words.data(), (int)words.size());
            }
            *score = *score - (float)L * beta - alpha * sent;
        }
        return (int)rev.size();
    }
};

}  // namespace

struct masr_beam {
    Beam b;
};

extern "C" {

int masr_beam_create(int32_t beam_size, int32_t blank, masr_beam** out) {
    if (!out || beam_size <= 0) return 1;
    masr_beam* h = new masr_beam();
    h->b.beam_size = beam_size;
    h->b.blank = blank;
    h->b.reset();
    *out = h;
    return 0;
}

void masr_beam_destroy(masr_beam* h) { delete h; }

int masr_beam_set_lm(masr_beam* h, const masr_lm* lm, float alpha, float beta) {
    if (!h) return 1;
    h->b.set_lm(lm, alpha, beta);
    h->b.reset();
    return 0;
}

int masr_beam_reset(masr_beam* h) {
    if (!h) return 1;
    h->b.reset();
    return 0;
}

int masr_beam_advance(masr_beam* h, const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                      int32_t T, int32_t K) {
    return masr_beam_advance_lm(h, idx_host, logp_host, count_host, nullptr, T, K);
}

int masr_beam_advance_lm(masr_beam* h, const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                         const float* blank_logp_host, int32_t T, int32_t K) {
    if (!h || !idx_host || !logp_host || !count_host) return 1;
    for (int t = 0; t < T; ++t)
        h->b.step(idx_host + (size_t)t * K, logp_host + (size_t)t * K, std::min(count_host[t], K),
                  blank_logp_host ? blank_logp_host[t] : NAN);
    return 0;
}

int masr_beam_result(masr_beam* h, int32_t* tokens_host, int32_t max_len, int32_t* len, float* score) {
    if (!h || !tokens_host || !len || !score) return 1;
    *len = h->b.best(tokens_host, max_len, score);
    return 0;
}

int masr_beam_search_batch(const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                           const int32_t* frames_host, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size,
                           int32_t blank, int32_t num_threads, int32_t* tokens_host, int32_t max_len, int32_t* len_host,
                           float* score_host) {
    return masr_beam_search_batch_lm(idx_host, logp_host, count_host, frames_host, B, T_stride, K, beam_size, blank, num_threads,
                                     nullptr, 0.f, 0.f, nullptr, tokens_host, max_len, len_host, score_host);
}

int masr_beam_search_batch_lm(const int32_t* idx_host, const float* logp_host, const int32_t* count_host,
                              const int32_t* frames_host, int32_t B, int32_t T_stride, int32_t K, int32_t beam_size,
                              int32_t blank, int32_t num_threads, const masr_lm* lm, float alpha, float beta,
                              const float* blank_logp_host, int32_t* tokens_host, int32_t max_len, int32_t* len_host,
                              float* score_host) {
    if (B <= 0) return 0;
    if (num_threads <= 0) num_threads = 1;
    num_threads = std::min(num_threads, B);
    auto work = [&](int tid) {
        Beam bm;
        bm.beam_size = beam_size;
        bm.blank = blank;
        bm.set_lm(lm, alpha, beta);
        for (int b = tid; b < B; b += num_threads) {
            bm.reset();
            const size_t base = (size_t)b * T_stride;
            const int T = std::min(frames_host[b], T_stride);
            for (int t = 0; t < T; ++t)
                bm.step(idx_host + (base + t) * K, logp_host + (base + t) * K, std::min(count_host[base + t], K),
                        blank_logp_host ? blank_logp_host[base + t] : NAN);
            len_host[b] = bm.best(tokens_host + (size_t)b * max_len, max_len, score_host + b);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < num_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    return 0;
}

}  // extern "C"
