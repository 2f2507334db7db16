// ARPA n-gram language model -> the hash table of lm_scorer.h, on the host and (on first use per device) in HBM.
// Replaces the model side of paddlespeech_ctcdecoders' Scorer (scorer.cpp: setup / load_lm / fill_dictionary on KenLM), which the
// reference builds in masr/decoders/beam_search_decoder.py:29-35.  KenLM's binary formats (.klm / .trie.klm) are NOT parsed:
// `lmplz` / `build_binary` users keep the ARPA text they built the binary from (docs/beam_search.md trains LMs that way).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
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
    struct Dev {
        LmEntry* table = nullptr;
        unsigned char* known = nullptr;
    };
    std::map<int, Dev> dev;

    LmView host_view() const {
        LmView v;
        v.table = table.data();
        v.mask = table.size() - 1;
        v.known = known.data();
        v.max_order = max_order;
        v.n_words = V + 2;
        v.bos = V;
        v.eos = V + 1;
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
    auto it = lm->dev.find(d);
    if (it == lm->dev.end()) {
        masr_lm::Dev dv;
        if (hipMalloc((void**)&dv.table, lm->table.size() * sizeof(LmEntry)) != hipSuccess ||
            hipMalloc((void**)&dv.known, lm->known.size()) != hipSuccess)
            return lm_fail("hipMalloc of the language model table failed");
        if (hipMemcpy(dv.table, lm->table.data(), lm->table.size() * sizeof(LmEntry), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dv.known, lm->known.data(), lm->known.size(), hipMemcpyHostToDevice) != hipSuccess)
            return lm_fail("upload of the language model table failed");
        it = lm->dev.emplace(d, dv).first;
    }
    *out = lm->host_view();
    out->table = it->second.table;
    out->known = it->second.known;
    return 0;
}
LmView lm_host_view(const masr_lm* lm) { return lm->host_view(); }
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
    std::unordered_map<std::string, int> word_id;
    word_id.reserve((size_t)V * 2);
    for (int i = 0; i < V; ++i) word_id.emplace(vocab_utf8[i], i);     // first occurrence wins, like vocabulary.index()
    word_id["<s>"] = V;
    word_id["</s>"] = V + 1;

    struct Raw {
        unsigned long long key;
        float prob, backoff;
    };
    std::vector<Raw> rows;
    std::vector<long long> declared;
    std::vector<unsigned char> known((size_t)V + 2, 0);
    bool char_based = true;
    long long skipped = 0;
    int order = 0, max_order = 0;
    std::string line;
    std::vector<char> buf(1 << 16);
    const double LN10 = 2.302585092994046;
    std::vector<int> ids;
    while (fgets(buf.data(), (int)buf.size(), f)) {
        line.assign(buf.data());
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '\\') {
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
            if (sscanf(line.c_str(), "ngram %d=%lld", &n, &c) == 2) {
                if ((int)declared.size() < n) declared.resize(n, 0);
                declared[n - 1] = c;
            }
            continue;
        }
        // "<log10 p>\t<w1> ... <wn>[\t<log10 backoff>]"
        const size_t t1 = line.find('\t');
        if (t1 == std::string::npos) continue;
        const size_t t2 = line.find('\t', t1 + 1);
        const double lp = atof(line.substr(0, t1).c_str());
        const std::string words = line.substr(t1 + 1, t2 == std::string::npos ? std::string::npos : t2 - t1 - 1);
        const double bo = t2 == std::string::npos ? 0.0 : atof(line.substr(t2 + 1).c_str());
        ids.clear();
        bool usable = true;
        size_t pos = 0;
        while (pos <= words.size()) {
            size_t sp = words.find(' ', pos);
            if (sp == std::string::npos) sp = words.size();
            const std::string w = words.substr(pos, sp - pos);
            pos = sp + 1;
            if (w.empty()) continue;
            if (order == 1 && w != "<s>" && w != "</s>" && w != "<unk>" && utf8_chars(w) != 1) char_based = false;
            auto it = word_id.find(w);
            if (it == word_id.end() || w == "<unk>") usable = false;      // a word the acoustic model cannot emit (or <unk>: KenLM index 0)
            else ids.push_back(it->second);
        }
        if (!usable || (int)ids.size() != order) {
            ++skipped;
            continue;
        }
        unsigned long long ctx = 0;
        for (int j = 0; j + 1 < order; ++j) ctx = lm_push(ctx, ids[j]);
        Raw r;
        r.key = lm_key(ctx, order - 1, ids[order - 1]);
        r.prob = (float)(lp * LN10);
        r.backoff = (float)(bo * LN10);
        rows.push_back(r);
        if (order == 1) known[ids[0]] = 1;
    }
    fclose(f);
    if (max_order == 0 || rows.empty()) return lm_fail(std::string(path) + ": no n-grams found (not an ARPA file?)");
    if (!char_based)
        return lm_fail("word-based language models (space-delimited vocabularies) are not implemented: the scorer is "
                       "character-based like the reference's Mandarin models");
    masr_lm* lm = new masr_lm();
    size_t slots = 16;
    while (slots < rows.size() * 3) slots <<= 1;          // load <= 1/3: ~1.2 probes per hit, ~1.4 per miss
    lm->table.assign(slots, LmEntry{0ull, 0.f, 0.f});
    const unsigned long long mask = slots - 1;
    for (const Raw& r : rows) {
        unsigned long long h = r.key & mask;
        while (lm->table[h].key != 0ull && lm->table[h].key != r.key) h = (h + 1) & mask;
        lm->table[h] = LmEntry{r.key, r.prob, r.backoff};
    }
    lm->known = known;
    lm->max_order = max_order;
    lm->V = V;
    lm->n_ngrams = (long long)rows.size();
    lm->skipped = skipped;
    lm->char_based = char_based;
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

int masr_lm_cond_log_prob(const masr_lm* lm, const int32_t* ids, int32_t n, float* out) {
    if (!lm || !ids || !out || n <= 0) return lm_fail("null argument");
    const LmView v = lm->host_view();
    unsigned long long ctx = lm_root_ctx(v);
    for (int i = std::max(0, n - v.max_order); i + 1 < n; ++i) ctx = lm_push(ctx, ids[i]);
    *out = lm_cond(v, lm_state_of(v, ctx), ids[n - 1]);
    return 0;
}

int masr_lm_sentence_log_prob(const masr_lm* lm, const int32_t* ids, int32_t n, float* out) {
    if (!lm || !out || n < 0) return lm_fail("null argument");
    const LmView v = lm->host_view();
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
