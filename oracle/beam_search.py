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


# ---- the whole decoder as published: min_cutoff / full_beam pruning, word-based scorers, dictionary ----------------------------
# ctc_beam_search_decoder.cpp (ctc_beam_search_decoder), path_trie.cpp (get_path_trie / get_path_vec / iterate_to_vec / remove),
# scorer.cpp (setup / set_char_map / fill_dictionary / make_ngram / split_labels) of paddlespeech_ctcdecoders -- the module
# masr/decoders/swig_wrapper.py:35-64 calls.  Absent third-party code, restated from the published sources: **parity unpinned**.
# Vocabulary convention of the reference: the space token is the string '<space>' (text_featurizer.py:24; a literal ' ' is
# accepted too).
SPACE_TOKENS = ('<space>', ' ')


class Scorer:
    """Scorer(alpha, beta, lm_path, vocabulary) of swig_wrapper.py:4-18 around an ``ArpaLM``.

    ``is_character_based``: every LM word other than <unk>/<s>/</s> is one UTF-8 character (scorer.cpp load_lm).  A word-based
    scorer owns the spelling dictionary (fill_dictionary: every LM word whose characters are all vocabulary tokens, followed by
    the space token; the FST of the original is determinised + minimised, which accepts the same strings from every state as
    this plain trie)."""

    def __init__(self, lm, vocabulary, alpha, beta):
        self.lm, self.alpha, self.beta = lm, float(alpha), float(beta)
        self.vocab = list(vocabulary)
        self.max_order = lm.max_order
        self.space_id = next((i for i, t in enumerate(self.vocab) if t in SPACE_TOKENS), -1)
        words = [g[0] for g in lm.grams if len(g) == 1]
        self.is_character_based = all(len(w) == 1 for w in words if w not in (UNK_TOKEN, START_TOKEN, END_TOKEN))
        self.dict_next, self.dict_final, self.dict_size = None, None, 0
        if not self.is_character_based:
            char_map = {}
            for i, t in enumerate(self.vocab):
                char_map[t] = i                      # later duplicates overwrite, as std::unordered_map::operator[] does
            self.dict_next, self.dict_final = [{}], [False]
            for w in words:
                ids = [char_map.get(ch) for ch in w]
                if not ids or any(i is None for i in ids) or self.space_id < 0:
                    continue
                s = 0
                for i in ids + [self.space_id]:
                    nxt = self.dict_next[s].get(i)
                    if nxt is None:
                        nxt = len(self.dict_next)
                        self.dict_next.append({})
                        self.dict_final.append(False)
                        self.dict_next[s][i] = nxt
                    s = nxt
                self.dict_final[s] = True
                self.dict_size += 1

    def words_of(self, labels):
        """Scorer::split_labels: characters (character based) or the space-separated words"""
        toks = [self.vocab[i] for i in labels]
        if self.is_character_based:
            return toks
        if not labels:
            return []
        out, cur = [], ''
        for i, t in zip(labels, toks):
            if i == self.space_id:
                out.append(cur)
                cur = ''
            else:
                cur += t
        out.append(cur)
        return out

    def make_ngram(self, node):
        """Scorer::make_ngram on a trie node: the last max_order words of its prefix (the unfinished one included), <s> in front"""
        ngram, cur = [], node
        for order in range(self.max_order):
            chars = []
            if self.is_character_based:
                # get_path_vec(prefix_vec, SPACE_ID_, 1): one character, nothing when the node is a space or the root
                if cur.ch != self.space_id and cur.ch != -1:
                    chars.append(cur.ch)
                    new = cur.parent
                else:
                    new = cur
                cur = new
            else:
                new = cur
                while new.ch != self.space_id and new.ch != -1:
                    chars.append(new.ch)
                    new = new.parent
                cur = new.parent                     # skipping the space
            ngram.append(''.join(self.vocab[c] for c in reversed(chars)))
            if new.ch == -1:
                ngram.extend([START_TOKEN] * (self.max_order - order - 1))
                break
        return ngram[::-1]


class _DNode(_Node):
    __slots__ = ('dstate',)

    def __init__(self, ch=-1, parent=None, dstate=0):
        super().__init__(ch, parent)
        self.dstate = dstate


def ctc_beam_search_decoder(probs_seq, vocabulary, beam_size, cutoff_prob=1.0, cutoff_top_n=40, scorer=None, blank_id=0,
                            prune=True, cands=None, blank_logp=None):
    """The published decoder end to end.  Returns (approx_ctc score of the best prefix, token ids, its search score).

    ``scorer is None``: the LM-free search (== ``prefix_beam_search``).  With a scorer: the candidates of a frame are visited
    for the live prefixes in descending score order and, when the beam is full, the pair (prefix, c) and every lower prefix is
    skipped once ``log p(c) + score(prefix) < min_cutoff = score(worst live prefix) + log p(blank) - max(0, beta)``
    (``prune=False`` switches that off).  ``cands`` / ``blank_logp`` replace probs_seq by already pruned candidate lists and the
    per-frame ln p(blank) (fixtures)."""
    if cands is None:
        probs_seq = np.asarray(probs_seq)
        cands = [pruned_log_probs(p, cutoff_prob, cutoff_top_n) for p in probs_seq]
        blank_logp = [math.log(float(p[blank_id])) if p[blank_id] > 0 else NEG_INF for p in probs_seq]
    space_id = next((i for i, t in enumerate(vocabulary) if t in SPACE_TOKENS), -2)
    root = _DNode()
    root.score = root.b_prev = np.float32(0.0)
    prefixes = [root]
    use_dict = scorer is not None and not scorer.is_character_based
    key = lambda n: (-n.score, n.ch)

    def get_path_trie(n, c):
        for k in n.kids:
            if k.ch == c:
                if not k.exists:
                    k.exists = True
                    k.b_prev = k.nb_prev = k.b_cur = k.nb_cur = k.score = NEG_INF
                return k
        dstate = 0
        if use_dict:
            nxt = scorer.dict_next[n.dstate].get(c)
            if nxt is None:                       # the character leaves the dictionary
                if scorer.dict_final[n.dstate]:   # ... after a complete word: the NEXT attempt starts a new word
                    n.dstate = 0
                return None
            dstate = nxt
        k = _DNode(c, n, dstate)
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

    for t, frame in enumerate(cands):
        min_cutoff, full_beam = NEG_INF, False
        live = prefixes[:beam_size]
        if scorer is not None:
            live.sort(key=key)
            prefixes[:len(live)] = live
            if prune:
                min_cutoff = np.float32(float(live[-1].score) + blank_logp[t] - max(0.0, scorer.beta))
                full_beam = len(live) == beam_size
        for c, lp in frame:
            for p in live:
                if full_beam and np.float32(lp + p.score) < min_cutoff:
                    break
                if c == blank_id:
                    p.b_cur = _lse(p.b_cur, np.float32(lp + p.score))
                    continue
                if c == p.ch:
                    p.nb_cur = _lse(p.nb_cur, np.float32(lp + p.nb_prev))
                q = get_path_trie(p, c)
                if q is None:
                    continue
                add = NEG_INF
                if c == p.ch and p.b_prev > NEG_INF:
                    add = np.float32(lp + p.b_prev)
                elif c != p.ch:
                    add = np.float32(lp + p.score)
                if scorer is not None and (c == space_id or scorer.is_character_based):
                    to_score = q if scorer.is_character_based else p
                    if add > NEG_INF:
                        add = np.float32(add + scorer.alpha * scorer.lm.cond_log_prob(scorer.make_ngram(to_score)) + scorer.beta)
                q.nb_cur = _lse(q.nb_cur, add)
        prefixes = []
        collect(root, prefixes)
        if len(prefixes) >= beam_size:
            prefixes.sort(key=key)
            for n in prefixes[beam_size:]:
                remove(n)
            prefixes = prefixes[:beam_size]
    # a word-based scorer scores the unfinished last word of every prefix
    if scorer is not None and not scorer.is_character_based:
        for p in prefixes[:beam_size]:
            if p.parent is not None and p.ch != space_id:
                p.score = np.float32(p.score + scorer.alpha * scorer.lm.cond_log_prob(scorer.make_ngram(p)) + scorer.beta)
    prefixes.sort(key=key)
    best = prefixes[0]
    toks, n = [], best
    while n.parent is not None:
        toks.append(n.ch)
        n = n.parent
    toks = toks[::-1]
    approx = float(best.score)
    if scorer is not None:
        approx = approx - len(toks) * scorer.beta - scorer.alpha * scorer.lm.sent_log_prob(scorer.words_of(toks))
    return approx, toks, float(best.score)
