"""TEST INFRASTRUCTURE ONLY -- pure-Python restatement of the LM-free CTC prefix beam search of the
third-party decoder the reference wraps (``paddlespeech_ctcdecoders``: ctc_beam_search_decoder.cpp /
path_trie.cpp of DeepSpeech2 / PaddleSpeech; call sites ``masr/decoders/swig_wrapper.py:35-64``,
``beam_search_decoder.py:45-56``).  That module is not vendored, not version-pinned and absent here:
**parity unpinned**; this file states the published algorithm (no external scorer, i.e. alpha = 0) and
is what the HIP/host implementation is checked against.  Small inputs only (pure Python)."""
import math

import numpy as np

NEG_INF = -float('inf')
FLT_MIN = 1.17549435e-38


def _lse(x, y):
    if x == NEG_INF:
        return y
    if y == NEG_INF:
        return x
    m = max(x, y)
    return np.float32(m + np.float32(math.log(math.exp(x - m) + math.exp(y - m))))


def pruned_log_probs(prob_step, cutoff_prob=1.0, cutoff_top_n=40):
    """get_pruned_log_probs: descending sort, cumulative cutoff, top-n cap; returns [(index, log p)]."""
    prob_step = np.asarray(prob_step)
    order = sorted(range(len(prob_step)), key=lambda i: (-float(prob_step[i]), i))
    cutoff_len = len(order)
    if cutoff_prob < 1.0 or cutoff_top_n < cutoff_len:
        if cutoff_prob < 1.0:
            cum, cutoff_len = 0.0, 0
            for i in order:
                cum += float(prob_step[i])
                cutoff_len += 1
                if cum >= cutoff_prob or cutoff_len >= cutoff_top_n:
                    break
        else:
            cutoff_len = min(cutoff_len, cutoff_top_n)
        order = order[:cutoff_len]
    return [(i, np.float32(math.log(float(prob_step[i]) + FLT_MIN))) for i in order]


class _Node:
    __slots__ = ('b_prev', 'nb_prev', 'b_cur', 'nb_cur', 'score', 'ch', 'parent', 'exists', 'kids')

    def __init__(self, ch=-1, parent=None):
        self.b_prev = self.nb_prev = self.b_cur = self.nb_cur = self.score = NEG_INF
        self.ch, self.parent, self.exists, self.kids = ch, parent, True, []


def prefix_beam_search(cands_per_frame, beam_size=300, blank=0):
    """cands_per_frame: list over frames of [(index, log prob)] in descending probability.
    Returns (log prob of the best prefix, token id list)."""
    root = _Node()
    root.score = root.b_prev = np.float32(0.0)
    prefixes = [root]

    def child(n, c):
        for k in n.kids:
            if k.ch == c:
                if not k.exists:
                    k.exists = True
                    k.b_prev = k.nb_prev = k.b_cur = k.nb_cur = k.score = NEG_INF
                return k
        k = _Node(c, n)
        n.kids.append(k)
        return k

    def collect(n, out):
        if n.exists:
            n.b_prev, n.nb_prev = n.b_cur, n.nb_cur
            n.b_cur = n.nb_cur = NEG_INF
            n.score = _lse(n.b_prev, n.nb_prev)
            out.append(n)
        for k in list(n.kids):
            collect(k, out)

    def remove(n):
        n.exists = False
        while n.parent is not None and not n.exists and not n.kids:
            n.parent.kids.remove(n)
            n = n.parent

    def key(n):
        return (-n.score, n.ch)

    for cands in cands_per_frame:
        live = prefixes[:beam_size]
        for c, lp in cands:
            for p in live:
                if c == blank:
                    p.b_cur = _lse(p.b_cur, np.float32(lp + p.score))
                    continue
                if c == p.ch:
                    p.nb_cur = _lse(p.nb_cur, np.float32(lp + p.nb_prev))
                q = child(p, c)
                add = NEG_INF
                if c == p.ch and p.b_prev > NEG_INF:
                    add = np.float32(lp + p.b_prev)
                elif c != p.ch:
                    add = np.float32(lp + p.score)
                q.nb_cur = _lse(q.nb_cur, add)
        prefixes = []
        collect(root, prefixes)
        if len(prefixes) >= beam_size:
            prefixes.sort(key=key)
            for n in prefixes[beam_size:]:
                remove(n)
            prefixes = prefixes[:beam_size]
    prefixes.sort(key=key)
    n = prefixes[0]
    score = float(n.score)
    toks = []
    while n.parent is not None:
        toks.append(n.ch)
        n = n.parent
    return score, toks[::-1]


def decode(probs, vocabulary, beam_size=300, cutoff_prob=1.0, cutoff_top_n=40, blank=0):
    cands = [pruned_log_probs(p, cutoff_prob, cutoff_top_n) for p in np.asarray(probs)]
    score, toks = prefix_beam_search(cands, beam_size, blank)
    return score, ''.join(vocabulary[t] for t in toks).replace('<space>', ' ')


# ---- external scorer (alpha / beta) ------------------------------------------------------------------------------------------
# Restatement of paddlespeech_ctcdecoders' Scorer (scorer.cpp: make_ngram / get_log_cond_prob / get_sent_log_prob on KenLM) for
# CHARACTER-based models, and of where ctc_beam_search_decoder.cpp applies it.  Un-vendored third-party code, absent here:
# **parity unpinned** (module docstring).  KenLM's BaseScore is stated as what it computes -- the ARPA backoff recursion.
LN10 = math.log(10.0)
OOV_SCORE = -1000.0
START_TOKEN, END_TOKEN, UNK_TOKEN = '<s>', '</s>', '<unk>'


class ArpaLM:
    """ARPA text -> {n-gram tuple: (log10 prob, log10 backoff)}; plain dictionaries, small models only"""

    def __init__(self, path):
        self.grams, self.max_order = {}, 0
        order = 0
        with open(path, encoding='utf-8') as f:
            for line in f:
                line = line.rstrip('\n')
                if not line:
                    continue
                if line.startswith('\\'):
                    order = int(line[1:line.index('-')]) if line.endswith('-grams:') else 0
                    self.max_order = max(self.max_order, order)
                    continue
                if order == 0:
                    continue
                parts = line.split('\t')
                words = tuple(parts[1].split(' '))
                self.grams[words] = (float(parts[0]), float(parts[2]) if len(parts) > 2 else 0.0)

    def known(self, w):
        return (w,) in self.grams and w != UNK_TOKEN

    def cond_log_prob(self, ngram):
        """Scorer::get_log_cond_prob: ln P(last word | the others); the first unknown word anywhere returns OOV_SCORE"""
        for w in ngram:
            if not self.known(w):
                return OOV_SCORE
        ctx, w, acc = tuple(ngram[:-1]), ngram[-1], 0.0
        while True:
            hit = self.grams.get(ctx + (w,))
            if hit is not None:
                return (acc + hit[0]) * LN10
            if not ctx:
                return OOV_SCORE
            acc += self.grams.get(ctx, (0.0, 0.0))[1]
            ctx = ctx[1:]

    def make_ngram(self, words):
        """Scorer::make_ngram (character based): the last max_order characters, <s>-padded in front"""
        tail = list(words[-self.max_order:])
        return [START_TOKEN] * (self.max_order - len(tail)) + tail

    def sent_log_prob(self, words):
        """Scorer::get_sent_log_prob: sliding windows over <s> x (max_order - 1) + words + </s> (max_order x <s> + </s> when
        there are no words)"""
        sent = [START_TOKEN] * (self.max_order if not words else self.max_order - 1) + list(words) + [END_TOKEN]
        return sum(self.cond_log_prob(sent[i:i + self.max_order]) for i in range(len(sent) - self.max_order + 1))


def prefix_beam_search_lm(cands_per_frame, vocabulary, lm, alpha, beta, beam_size=300, blank=0):
    """``prefix_beam_search`` with the external scorer: when candidate c extends prefix p to a NEW prefix p + c its
    probability term gets alpha * ln P_LM(c | last characters of p) + beta (ctc_beam_search_decoder.cpp: ``log_p += score;
    log_p += ext_scorer->beta`` for character-based scorers).  The reported score is the decoder's approx_ctc: the best
    prefix's score minus |prefix| * beta minus alpha * ln P_LM(sentence).  Returns (approx_ctc, token ids)."""
    root = _Node()
    root.score = root.b_prev = np.float32(0.0)
    prefixes = [root]

    def words_of(n):
        out = []
        while n.parent is not None:
            out.append(vocabulary[n.ch])
            n = n.parent
        return out[::-1]

    def child(n, c):
        for k in n.kids:
            if k.ch == c:
                if not k.exists:
                    k.exists = True
                    k.b_prev = k.nb_prev = k.b_cur = k.nb_cur = k.score = NEG_INF
                return k
        k = _Node(c, n)
        n.kids.append(k)
        return k

    def collect(n, out):
        if n.exists:
            n.b_prev, n.nb_prev = n.b_cur, n.nb_cur
            n.b_cur = n.nb_cur = NEG_INF
            n.score = _lse(n.b_prev, n.nb_prev)
            out.append(n)
        for k in list(n.kids):
            collect(k, out)

    def remove(n):
        n.exists = False
        while n.parent is not None and not n.exists and not n.kids:
            n.parent.kids.remove(n)
            n = n.parent

    key = lambda n: (-n.score, n.ch)
    for cands in cands_per_frame:
        live = prefixes[:beam_size]
        for c, lp in cands:
            for p in live:
                if c == blank:
                    p.b_cur = _lse(p.b_cur, np.float32(lp + p.score))
                    continue
                if c == p.ch:
                    p.nb_cur = _lse(p.nb_cur, np.float32(lp + p.nb_prev))
                add = NEG_INF
                if c == p.ch and p.b_prev > NEG_INF:
                    add = np.float32(lp + p.b_prev)
                elif c != p.ch:
                    add = np.float32(lp + p.score)
                q = child(p, c)
                if add > NEG_INF:
                    add = np.float32(add + alpha * lm.cond_log_prob(lm.make_ngram(words_of(q))) + beta)
                q.nb_cur = _lse(q.nb_cur, add)
        prefixes = []
        collect(root, prefixes)
        if len(prefixes) >= beam_size:
            prefixes.sort(key=key)
            for n in prefixes[beam_size:]:
                remove(n)
            prefixes = prefixes[:beam_size]
    prefixes.sort(key=key)
    best = prefixes[0]
    words = words_of(best)
    approx_ctc = float(best.score) - len(words) * beta - alpha * lm.sent_log_prob(words)
    toks = []
    n = best
    while n.parent is not None:
        toks.append(n.ch)
        n = n.parent
    return approx_ctc, toks[::-1]


def decode_lm(probs, vocabulary, lm, alpha, beta, beam_size=300, cutoff_prob=1.0, cutoff_top_n=40, blank=0):
    cands = [pruned_log_probs(p, cutoff_prob, cutoff_top_n) for p in np.asarray(probs)]
    score, toks = prefix_beam_search_lm(cands, vocabulary, lm, alpha, beta, beam_size, blank)
    return score, ''.join(vocabulary[t] for t in toks).replace('<space>', ' ')
