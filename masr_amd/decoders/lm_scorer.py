"""External n-gram language model of the CTC beam search: the Python handle on ``masr_lm_*`` (libmasr_hip.so, csrc/lm_scorer.*).

Stands where the reference constructs ``Scorer(alpha, beta, language_model_path, vocab_list)`` (masr/decoders/
beam_search_decoder.py:29-35 -> paddlespeech_ctcdecoders on KenLM).  Character-based ARPA text models up to order 5; KenLM's
binary ``.klm`` files are not parsed (keep / regenerate the ARPA the binary was built from, docs/beam_search.md).  The n-gram
table lives in host memory for the host-thread search and is uploaded to HBM the first time a GPU search uses it.
"""
import ctypes as C

import numpy as np

from masr_amd import _lib


class LanguageModel:
    def __init__(self, path, vocab_list):
        self._lib = _lib.lib()
        toks = [t.encode('utf-8') for t in vocab_list]
        arr = (C.c_char_p * len(toks))(*toks)
        h = C.c_void_p()
        if self._lib.masr_lm_load_arpa(str(path).encode('utf-8'), arr, len(toks), C.byref(h)) != 0:
            raise _lib.MasrError(self._lib.masr_lm_last_error().decode('utf-8', 'replace'))
        self.h = h
        self.path = path
        self.vocab_size = len(toks)
        mo, n, cb, sk = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64()
        self._lib.masr_lm_info(self.h, C.byref(mo), C.byref(n), C.byref(cb), C.byref(sk))
        self.max_order, self.n_ngrams, self.is_character_based, self.skipped = mo.value, n.value, bool(cb.value), sk.value
        self.bos, self.eos = self.vocab_size, self.vocab_size + 1
        ds = C.c_int32()
        self._lib.masr_lm_dict_size(self.h, C.byref(ds))
        self.dict_size = ds.value         # word-based models: words in the spelling dictionary (Scorer::get_dict_size)

    def close(self):
        if getattr(self, 'h', None):
            self._lib.masr_lm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # the three accessors of the reference Scorer that beam_search_decoder.py:36-38 prints
    def get_max_order(self):
        return self.max_order

    def get_dict_size(self):
        return self.dict_size

    def word_id(self, word):
        """word-based models: the id ``cond_log_prob`` / ``sentence_log_prob`` take for an LM word (-1: not an LM word)"""
        out = C.c_int32()
        if self._lib.masr_lm_word_id(self.h, word.encode('utf-8'), C.byref(out)) != 0:
            raise _lib.MasrError(self._lib.masr_lm_last_error().decode('utf-8', 'replace'))
        return out.value

    def cond_log_prob(self, ids):
        """ln P(ids[-1] | ids[:-1]) with the <s>-padded window of Scorer::make_ngram / get_log_cond_prob"""
        a = (C.c_int32 * len(ids))(*[int(i) for i in ids])
        out = C.c_float()
        if self._lib.masr_lm_cond_log_prob(self.h, a, len(ids), C.byref(out)) != 0:
            raise _lib.MasrError(self._lib.masr_lm_last_error().decode('utf-8', 'replace'))
        return out.value

    def sentence_log_prob(self, ids):
        a = (C.c_int32 * max(len(ids), 1))(*[int(i) for i in ids])
        out = C.c_float()
        if self._lib.masr_lm_sentence_log_prob(self.h, a, len(ids), C.byref(out)) != 0:
            raise _lib.MasrError(self._lib.masr_lm_last_error().decode('utf-8', 'replace'))
        return out.value


def write_synthetic_arpa(path, vocab_list, order=3, seed=0, n_higher=20000, skip=('<blank>', '<unk>', '<space>', '<eos>')):
    """A random, prefix- and suffix-closed character ``order``-gram model in ARPA text over the single-character tokens of
    ``vocab_list`` (benchmarks and tests only: there is no text corpus here to estimate a real one from).  Returns ``path``."""
    rng = np.random.default_rng(seed)
    chars = [t for t in vocab_list if t not in skip and len(t) == 1]
    grams = [dict() for _ in range(order)]
    for w in ['<unk>', '<s>', '</s>'] + chars:
        grams[0][(w,)] = (-99.0 if w == '<s>' else float(rng.uniform(-5.0, -1.5)), float(rng.uniform(-1.0, -0.05)))
    ctxs = chars + ['<s>']
    for n in range(1, order):
        for _ in range(n_higher):
            g = tuple(ctxs[int(i)] for i in rng.integers(0, len(ctxs), n)) + ((chars + ['</s>'])[int(rng.integers(0, len(chars) + 1))],)
            if any(w == '<s>' for w in g[1:]):                       # <s> only ever starts an n-gram
                continue
            grams[n][g] = (float(rng.uniform(-3.0, -0.1)), float(rng.uniform(-0.8, -0.02)))
        for g in list(grams[n]):                                      # closure: every prefix and suffix n-gram exists
            for sub_ in (g[:-1], g[1:]):
                if sub_ not in grams[n - 1] and sub_[-1] != '<s>' or (len(sub_) == 1 and sub_ not in grams[0]):
                    if sub_[0] == '</s>' or '<s>' in sub_[1:]:
                        continue
                    grams[n - 1].setdefault(sub_, (float(rng.uniform(-3.5, -0.5)), float(rng.uniform(-0.8, -0.02))))
        for k in range(n - 1, 0, -1):                                 # ... recursively down the orders
            for g in list(grams[k]):
                for sub_ in (g[:-1], g[1:]):
                    if '<s>' in sub_[1:] or sub_[0] == '</s>':
                        continue
                    grams[k - 1].setdefault(sub_, (float(rng.uniform(-3.5, -0.5)), float(rng.uniform(-0.8, -0.02))))
    with open(path, 'w', encoding='utf-8') as f:
        f.write('\\data\\\n')
        for n in range(order):
            f.write(f'ngram {n + 1}={len(grams[n])}\n')
        for n in range(order):
            f.write(f'\n\\{n + 1}-grams:\n')
            for g, (p, b) in grams[n].items():
                f.write(f'{p:.6f}\t{" ".join(g)}' + (f'\t{b:.6f}\n' if n + 1 < order else '\n'))
        f.write('\n\\end\\\n')
    return path


def write_synthetic_word_arpa(path, words, order=3, seed=0, n_higher=400):
    """A random, prefix- and suffix-closed WORD ``order``-gram model in ARPA text over ``words`` (tests / benchmarks of the
    word-based scorer: spelling dictionary, scoring at the space token).  Returns ``path``."""
    rng = np.random.default_rng(seed)
    words = list(words)
    grams = [dict() for _ in range(order)]
    for w in ['<unk>', '<s>', '</s>'] + words:
        grams[0][(w,)] = (-99.0 if w == '<s>' else float(rng.uniform(-4.0, -1.0)), float(rng.uniform(-1.0, -0.05)))
    ctxs = words + ['<s>']
    for n in range(1, order):
        for _ in range(n_higher):
            g = tuple(ctxs[int(i)] for i in rng.integers(0, len(ctxs), n)) + ((words + ['</s>'])[int(rng.integers(0, len(words) + 1))],)
            if any(w == '<s>' for w in g[1:]):
                continue
            grams[n][g] = (float(rng.uniform(-2.5, -0.1)), float(rng.uniform(-0.8, -0.02)))
    for n in range(order - 1, 0, -1):                                 # closure: every prefix and suffix n-gram exists
        for g in list(grams[n]):
            for sub_ in (g[:-1], g[1:]):
                if '<s>' in sub_[1:] or sub_[0] == '</s>':
                    continue
                grams[n - 1].setdefault(sub_, (float(rng.uniform(-3.0, -0.5)), float(rng.uniform(-0.8, -0.02))))
    with open(path, 'w', encoding='utf-8') as f:
        f.write('\\data\\\n')
        for n in range(order):
            f.write(f'ngram {n + 1}={len(grams[n])}\n')
        for n in range(order):
            f.write(f'\n\\{n + 1}-grams:\n')
            for g, (p, b) in grams[n].items():
                f.write(f'{p:.6f}\t{" ".join(g)}' + (f'\t{b:.6f}\n' if n + 1 < order else '\n'))
        f.write('\n\\end\\\n')
    return path
