// ARPA n-gram language model -> the hash table of lm_scorer.h, on the host and (on first use per device) in HBM.
// Replaces the model side of paddlespeech_ctcdecoders' Scorer (scorer.cpp: setup / load_lm / fill_dictionary on KenLM), which the
// reference builds in masr/decoders/beam_search_decoder.py:29-35.  KenLM's binary formats (.klm / .trie.klm) are NOT parsed:
// `lmplz` / `build_binary` users keep the ARPA text they built the binary from (docs/beam_search.md trains LMs that way).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/masr_hip.h"
#include "lm_scorer.h"

using namespace masr;

struct masr_lm {
    std::vector<LmEntry> table;
    std::vector<unsigned char> known;
    int max_order = 0, V = 0;
    long long n_ngrams = 0, skipped = 0;
    bool char_based = true;
    // word-based models (scorer.cpp: is_character_based() false): LM words get their own ids (unigram order), and the scorer owns
    // the spelling dictionary of fill_dictionary -- every LM word whose characters are all vocabulary tokens, followed by the space
    // token.  The original builds an FST and determinises / minimises it; a plain trie accepts the same strings from every state.
    int n_words = 0, bos = 0, eos = 0, space_id = -1, dict_size = 0;
    std::unordered_map<unsigned long long, int> dict_arc;     // (state << 16 | token) -> next state; state 0 = start
    std::vector<unsigned char> dict_final;                    // [state]: a whole word + space was read
    std::vector<int> dict_word;                               // [state]: id of the LM word spelled by the path to the state, or -1
    std::unordered_map<std::string, int> word_ids;            // word-based models: LM word -> id
    struct Dev {
        LmEntry* table = nullptr;
        unsigned char* known = nullptr;
    };
    std::map<int, Dev> dev;
    std::mutex dev_lock;       // first use on a device from two threads (ctypes releases the GIL)

    LmView host_view() const {
        LmView v;
        v.table = table.data();
        v.mask = table.size() - 1;
        v.known = known.data();
        v.max_order = max_order;
        v.n_words = n_words;
        v.bos = bos;
        v.eos = eos;
        return v;
    }
};

static thread_local std::string g_lm_err;
const char* masr_lm_error_string() { return g_lm_err.c_str(); }
static int lm_fail(const std::string& m) {
    g_lm_err = m;
    return 1;
}

namespace masr {
// internal: the view of `lm` in the memory of the CURRENT device (uploaded once per device)
int lm_device_view(masr_lm* lm, LmView* out) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return lm_fail("hipGetDevice failed");
    std::lock_guard<std::mutex> guard(lm->dev_lock);
    auto it = lm->dev.find(d);
    if (it == lm->dev.end()) {
        masr_lm::Dev dv;
        bool ok = hipMalloc((void**)&dv.table, lm->table.size() * sizeof(LmEntry)) == hipSuccess;
        ok = ok && hipMalloc((void**)&dv.known, lm->known.size()) == hipSuccess;
        ok = ok && hipMemcpy(dv.table, lm->table.data(), lm->table.size() * sizeof(LmEntry), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(dv.known, lm->known.data(), lm->known.size(), hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) {
            if (dv.table) (void)hipFree(dv.table);
            if (dv.known) (void)hipFree(dv.known);
            return lm_fail("allocation / upload of the language model table on the device failed");
        }
        it = lm->dev.emplace(d, dv).first;
    }
    *out = lm->host_view();
    out->table = it->second.table;
    out->known = it->second.known;
    return 0;
}
LmView lm_host_view(const masr_lm* lm) { return lm->host_view(); }
// ---- word-based scorers: the spelling dictionary and n-gram scoring over word ids (host search only) ----
bool lm_word_based(const masr_lm* lm) { return !lm->char_based; }
int lm_space_id(const masr_lm* lm) { return lm->space_id; }
int lm_dict_next(const masr_lm* lm, int state, int token) {
    auto it = lm->dict_arc.find(((unsigned long long)state << 16) | (unsigned long long)(token & 0xFFFF));
    return it == lm->dict_arc.end() ? -1 : it->second;
}
bool lm_dict_final(const masr_lm* lm, int state) { return lm->dict_final[state] != 0; }
int lm_dict_word(const masr_lm* lm, int state) { return lm->dict_word[state]; }
// ln P(w | ctx[n_ctx-1] .. ctx[0]) (ctx most recent first, word ids, -1 = a string that is no LM word): Scorer::get_log_cond_prob ==
// the ARPA backoff recursion from the full context, every needed n-gram looked up (no cached per-prefix state: a word-based
// scorer is consulted once per word, not once per character)
float lm_cond_words(const LmView& v, const int* ctx, int n_ctx, int w) {
    if (w < 0 || w >= v.n_words || !v.known[w]) return LM_OOV_SCORE;
    for (int j = 0; j < n_ctx; ++j)
        if (ctx[j] < 0 || ctx[j] >= v.n_words || !v.known[ctx[j]]) return LM_OOV_SCORE;
    unsigned long long key[5], ckey[5];           // key[len]: (ctx[len-1..0], w);  ckey[len]: the context n-gram ctx[len-1..0]
    unsigned long long h = lm_mix(LM_SEED, (unsigned long long)w);
    key[0] = lm_fin(h, 1);
    for (int len = 1; len <= n_ctx; ++len) {
        h = lm_mix(h, (unsigned long long)ctx[len - 1]);
        key[len] = lm_fin(h, len + 1);
    }
    if (n_ctx > 0) {
        unsigned long long hc = lm_mix(LM_SEED, (unsigned long long)ctx[0]);
        ckey[1] = lm_fin(hc, 1);
        for (int len = 2; len <= n_ctx; ++len) {
            hc = lm_mix(hc, (unsigned long long)ctx[len - 1]);
            ckey[len] = lm_fin(hc, len);
        }
    }
    float acc = 0.f;
    for (int len = n_ctx; len >= 0; --len) {
        float p, b;
        if (lm_find(v, key[len], &p, &b)) return acc + p;
        if (len > 0 && lm_find(v, ckey[len], &p, &b)) acc += b;
    }
    return LM_OOV_SCORE;
}
// Scorer::get_sent_log_prob over word ids (-1 = no LM word): windows of max_order over <s> x (max_order - 1) + words + </s>
float lm_sentence_words(const LmView& v, const int* ids, int n) {
    const int k = v.max_order - 1;
    int ctx[4];
    for (int j = 0; j < 4; ++j) ctx[j] = v.bos;
    double tot = n == 0 ? lm_cond_words(v, ctx, k, v.bos) : 0.0;       // no words: max_order x <s>, so <s> itself is scored first
    for (int i = 0; i <= n; ++i) {
        const int w = i < n ? ids[i] : v.eos;
        tot += lm_cond_words(v, ctx, k, w);
        for (int j = 3; j > 0; --j) ctx[j] = ctx[j - 1];
        ctx[0] = w;
    }
    return (float)tot;
}
}  // namespace masr

static int utf8_chars(const std::string& s) {
    int n = 0;
    for (unsigned char c : s) n += (c & 0xC0) != 0x80;
    return n;
}

extern "C" {

const char* masr_lm_last_error(void) { return g_lm_err.c_str(); }

int masr_lm_load_arpa(const char* path, const char* const* vocab_utf8, int32_t V, masr_lm** out) {
    if (!path || !vocab_utf8 || !out || V <= 0) return lm_fail("null argument");
    if (V + 2 >= 0xFFFF) return lm_fail("vocabulary too large for the 16-bit packed LM context");
    FILE* f = fopen(path, "rb");
    if (!f) return lm_fail(std::string("cannot open language model ") + path);
    {   // KenLM binaries start with "mmap lm http://kheafield.com/code format version"
        char magic[16] = {0};
        const size_t got = fread(magic, 1, 7, f);
        if (got == 7 && memcmp(magic, "mmap lm", 7) == 0) {
            fclose(f);
            return lm_fail(std::string(path) + " is a KenLM binary; this scorer reads the ARPA text the binary was built from");
        }
        rewind(f);
    }
    std::unordered_map<std::string, int> token_id;           // vocabulary token -> index (set_char_map: later duplicates win)
    token_id.reserve((size_t)V * 2);
    int space_id = -1;
    for (int i = 0; i < V; ++i) {
        token_id[vocab_utf8[i]] = i;
        if (space_id < 0 && (!strcmp(vocab_utf8[i], "<space>") || !strcmp(vocab_utf8[i], " "))) space_id = i;
    }
    std::unordered_map<std::string, int> first_token;        // ... the n-gram words of a character-based model use the FIRST index
    first_token.reserve((size_t)V * 2);
    for (int i = 0; i < V; ++i) first_token.emplace(vocab_utf8[i], i);

    struct Raw {
        unsigned long long key;
        float prob, backoff;
    };
    struct Uni {
        std::string w;
        double lp, bo;
    };
    std::vector<Raw> rows;
    std::vector<Uni> unis;                       // the unigram section is held back until the model's kind is known
    std::vector<long long> declared, seen(6, 0);
    std::vector<unsigned char> known;
    std::unordered_map<std::string, int> word_id;            // LM word -> id used in the n-gram keys
    bool char_based = true, kind_known = false;
    long long skipped = 0;
    int order = 0, max_order = 0, n_words = 0, bos = -1, eos = -1;
    std::string line;
    std::vector<char> buf(1 << 16);
    const double LN10 = 2.302585092994046;
    std::vector<int> ids;

    auto is_space = [](char c) { return c == ' ' || c == '\t'; };
    // after the last unigram: character based (every word one UTF-8 character: scorer.cpp load_lm) or word based
    auto fix_kind = [&]() {
        kind_known = true;
        for (const Uni& u : unis)
            if (u.w != "<s>" && u.w != "</s>" && u.w != "<unk>" && utf8_chars(u.w) != 1) char_based = false;
        if (char_based) {
            // a word IS a vocabulary token: ids = token indices, <s> / </s> behind them; words the acoustic model cannot emit
            // are dropped (they can never be asked for)
            word_id = first_token;
            bos = V;
            eos = V + 1;
            word_id["<s>"] = bos;
            word_id["</s>"] = eos;
            n_words = V + 2;
        } else {
            for (const Uni& u : unis)
                if (u.w != "<unk>") word_id.emplace(u.w, (int)word_id.size());
            if (!word_id.count("<s>")) word_id.emplace("<s>", (int)word_id.size());
            if (!word_id.count("</s>")) word_id.emplace("</s>", (int)word_id.size());
            bos = word_id["<s>"];
            eos = word_id["</s>"];
            n_words = (int)word_id.size();
        }
        known.assign((size_t)n_words, 0);
        for (const Uni& u : unis) {
            auto it = word_id.find(u.w);
            if (it == word_id.end() || u.w == "<unk>") {      // <unk> is KenLM's index 0: a word mapped to it is an OOV
                ++skipped;
                continue;
            }
            Raw r;
            r.key = lm_key(0ull, 0, it->second);
            r.prob = (float)(u.lp * LN10);
            r.backoff = (float)(u.bo * LN10);
            rows.push_back(r);
            known[it->second] = 1;
        }
    };

    while (fgets(buf.data(), (int)buf.size(), f)) {
        line.assign(buf.data());
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '\\') {
            if (order == 1 && !kind_known) fix_kind();
            if (line == "\\data\\" || line == "\\end\\") {
                order = 0;
                continue;
            }
            int n = 0;
            if (sscanf(line.c_str(), "\\%d-grams:", &n) == 1) {
                order = n;
                max_order = std::max(max_order, n);
                if (n > 5) {
                    fclose(f);
                    return lm_fail("language models above order 5 are not supported (contexts are packed 4 x 16 bits)");
                }
                continue;
            }
            continue;
        }
        if (order == 0) {
            int n = 0;
            long long c = 0;
            if (sscanf(line.c_str(), "ngram %d=%lld", &n, &c) == 2 && n >= 1 && n <= 5) {
                if ((int)declared.size() < n) declared.resize(n, 0);
                declared[n - 1] = c;
            }
            continue;
        }
        // "<log10 p> <w1> ... <wn> [<log10 backoff>]", fields separated by tabs (KenLM, SRILM) or blanks
        std::vector<std::string> fld;
        for (size_t i = 0; i < line.size();) {
            while (i < line.size() && is_space(line[i])) ++i;
            size_t j = i;
            while (j < line.size() && !is_space(line[j])) ++j;
            if (j > i) fld.push_back(line.substr(i, j - i));
            i = j;
        }
        if ((int)fld.size() != order + 1 && (int)fld.size() != order + 2) {
            fclose(f);
            return lm_fail(std::string(path) + ": malformed " + std::to_string(order) + "-gram line: " + line);
        }
        ++seen[order];
        const double lp = atof(fld[0].c_str());
        const double bo = (int)fld.size() == order + 2 ? atof(fld[order + 1].c_str()) : 0.0;
        if (order == 1) {
            unis.push_back(Uni{fld[1], lp, bo});
            continue;
        }
        if (!kind_known) fix_kind();
        ids.clear();
        bool usable = true;
        for (int j = 1; j <= order; ++j) {
            auto it = word_id.find(fld[j]);
            if (it == word_id.end() || fld[j] == "<unk>") usable = false;
            else ids.push_back(it->second);
        }
        if (!usable) {
            ++skipped;
            continue;
        }
        unsigned long long h = lm_mix(LM_SEED, (unsigned long long)ids[order - 1]);
        for (int j = order - 2; j >= 0; --j) h = lm_mix(h, (unsigned long long)ids[j]);
        Raw r;
        r.key = lm_fin(h, order);
        r.prob = (float)(lp * LN10);
        r.backoff = (float)(bo * LN10);
        rows.push_back(r);
    }
    fclose(f);
    if (!kind_known && !unis.empty()) fix_kind();
    if (max_order == 0 || rows.empty()) return lm_fail(std::string(path) + ": no n-grams found (not an ARPA file?)");
    for (size_t n = 0; n < declared.size(); ++n)
        if (declared[n] != seen[n + 1])
            return lm_fail(std::string(path) + ": the header declares " + std::to_string(declared[n]) + " " + std::to_string(n + 1) +
                           "-grams, the file holds " + std::to_string(seen[n + 1]) + " (truncated?)");
    if (!char_based && space_id < 0)
        return lm_fail("a word-based language model needs a space token ('<space>') in the vocabulary");
    masr_lm* lm = new masr_lm();
    size_t slots = 16;
    while (slots < rows.size() * 3) slots <<= 1;          // load <= 1/3: ~1.2 probes per hit, ~1.4 per miss
    lm->table.assign(slots, LmEntry{0ull, 0.f, 0.f});
    const unsigned long long mask = slots - 1;
    for (const Raw& r : rows) {
        unsigned long long h = r.key & mask;
        while (lm->table[h].key != 0ull && lm->table[h].key != r.key) h = (h + 1) & mask;
        if (lm->table[h].key == r.key && (lm->table[h].prob != r.prob || lm->table[h].backoff != r.backoff)) {
            delete lm;                                    // the same n-gram twice with different values, or a 64-bit key collision
            return lm_fail(std::string(path) + ": two n-grams map to one table key with different values (duplicate n-gram in the file?)");
        }
        lm->table[h] = LmEntry{r.key, r.prob, r.backoff};
    }
    lm->known = known;
    lm->max_order = max_order;
    lm->V = V;
    lm->n_ngrams = (long long)rows.size();
    lm->skipped = skipped;
    lm->char_based = char_based;
    lm->n_words = n_words;
    lm->bos = bos;
    lm->eos = eos;
    lm->space_id = space_id;
    if (!char_based) {
        lm->word_ids = word_id;
        // fill_dictionary(add_space = true): the characters of every LM word as vocabulary tokens, then the space token
        lm->dict_final.assign(1, 0);
        lm->dict_word.assign(1, -1);
        std::vector<int> toks;
        for (const Uni& u : unis) {
            auto wi = word_id.find(u.w);
            if (wi == word_id.end()) continue;
            toks.clear();
            bool ok = !u.w.empty();
            for (size_t i = 0; i < u.w.size() && ok;) {
                size_t j = i + 1;
                while (j < u.w.size() && ((unsigned char)u.w[j] & 0xC0) == 0x80) ++j;
                auto it = token_id.find(u.w.substr(i, j - i));
                if (it == token_id.end()) ok = false;
                else toks.push_back(it->second);
                i = j;
            }
            if (!ok) continue;
            toks.push_back(space_id);
            int st = 0;
            for (size_t k = 0; k < toks.size(); ++k) {
                const unsigned long long key = ((unsigned long long)st << 16) | (unsigned long long)toks[k];
                auto it = lm->dict_arc.find(key);
                if (it == lm->dict_arc.end()) {
                    const int nx = (int)lm->dict_final.size();
                    lm->dict_final.push_back(0);
                    lm->dict_word.push_back(-1);
                    it = lm->dict_arc.emplace(key, nx).first;
                }
                st = it->second;
                if (k + 2 == toks.size()) lm->dict_word[st] = wi->second;        // the state the word's last character leads to
            }
            lm->dict_final[st] = 1;
            ++lm->dict_size;
        }
    }
    *out = lm;
    return 0;
}

void masr_lm_destroy(masr_lm* lm) {
    if (!lm) return;
    for (auto& kv : lm->dev) {
        (void)hipFree(kv.second.table);
        (void)hipFree(kv.second.known);
    }
    delete lm;
}

int masr_lm_info(const masr_lm* lm, int32_t* max_order, int64_t* n_ngrams, int32_t* char_based, int64_t* skipped) {
    if (!lm) return lm_fail("null language model");
    if (max_order) *max_order = lm->max_order;
    if (n_ngrams) *n_ngrams = lm->n_ngrams;
    if (char_based) *char_based = lm->char_based ? 1 : 0;
    if (skipped) *skipped = lm->skipped;
    return 0;
}

int masr_lm_word_id(const masr_lm* lm, const char* word_utf8, int32_t* id) {
    if (!lm || !word_utf8 || !id) return lm_fail("null argument");
    if (lm->char_based) return lm_fail("masr_lm_word_id: a character-based model scores vocabulary token ids");
    auto it = lm->word_ids.find(word_utf8);
    *id = it == lm->word_ids.end() ? -1 : it->second;
    return 0;
}

int masr_lm_dict_size(const masr_lm* lm, int32_t* dict_size) {
    if (!lm || !dict_size) return lm_fail("null argument");
    *dict_size = lm->dict_size;
    return 0;
}

int masr_lm_cond_log_prob(const masr_lm* lm, const int32_t* ids, int32_t n, float* out) {
    if (!lm || !ids || !out || n <= 0) return lm_fail("null argument");
    const LmView v = lm->host_view();
    if (!lm->char_based) {                   // ids are LM word ids (masr_lm_word_id), oldest first; <s>-padded like make_ngram
        int ctx[4];
        const int k = v.max_order - 1;
        for (int j = 0; j < k; ++j) ctx[j] = n - 2 - j >= 0 ? ids[n - 2 - j] : v.bos;
        *out = lm_cond_words(v, ctx, k, ids[n - 1]);
        return 0;
    }
    unsigned long long ctx = lm_root_ctx(v);
    for (int i = std::max(0, n - v.max_order); i + 1 < n; ++i) ctx = lm_push(ctx, ids[i]);
    *out = lm_cond(v, lm_state_of(v, ctx), ids[n - 1]);
    return 0;
}

int masr_lm_sentence_log_prob(const masr_lm* lm, const int32_t* ids, int32_t n, float* out) {
    if (!lm || !out || n < 0) return lm_fail("null argument");
    const LmView v = lm->host_view();
    if (!lm->char_based) {
        *out = lm_sentence_words(v, ids, n);
        return 0;
    }
    unsigned long long ctx = lm_root_ctx(v);
    // no words: the reference pads with max_order x <s> (not max_order - 1), so a window that scores <s> itself comes first
    double tot = n == 0 ? lm_cond(v, lm_state_of(v, ctx), v.bos) : 0.0;
    for (int i = 0; i <= n; ++i) {
        const int w = i < n ? ids[i] : v.eos;
        tot += lm_cond(v, lm_state_of(v, ctx), w);
        ctx = lm_push(ctx, w);
    }
    *out = (float)tot;
    return 0;
}

}  // extern "C"
