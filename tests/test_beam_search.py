"""CTC prefix beam search: host search (CPU, no GPU needed) against the pure-Python oracle restatement;
GPU candidate pruning + the drop-in BeamSearchDecoder (gpu-marked)."""
import ctypes as C

import numpy as np
import pytest

from oracle import beam_search as obs
from oracle import decoders as od


def _host_search(cands, beam, blank=0):
    from masr_amd import _lib
    lib = _lib.lib()
    T = len(cands)
    K = max(len(c) for c in cands)
    idx = np.zeros((T, K), np.int32)
    logp = np.zeros((T, K), np.float32)
    cnt = np.zeros(T, np.int32)
    for t, c in enumerate(cands):
        cnt[t] = len(c)
        for k, (i, lp) in enumerate(c):
            idx[t, k], logp[t, k] = i, lp
    frames = np.array([T], np.int32)
    toks = np.zeros((1, T + 1), np.int32)
    lens = np.zeros(1, np.int32)
    score = np.zeros(1, np.float32)
    rc = lib.masr_beam_search_batch(idx.ctypes.data_as(C.c_void_p), logp.ctypes.data_as(C.c_void_p),
                                    cnt.ctypes.data_as(C.c_void_p), frames.ctypes.data_as(C.c_void_p), 1, T, K, beam, blank,
                                    2, toks.ctypes.data_as(C.c_void_p), T + 1, lens.ctypes.data_as(C.c_void_p),
                                    score.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return float(score[0]), list(toks[0, :lens[0]])


@pytest.mark.parametrize('seed,V,T,beam,cut,topn', [(0, 6, 12, 4, 1.0, 40), (1, 12, 30, 8, 0.99, 5), (2, 30, 25, 20, 0.95, 10),
                                                    (3, 5, 40, 300, 1.0, 40), (4, 50, 18, 3, 0.9, 40)])
def test_host_prefix_search_matches_oracle(built_lib, seed, V, T, beam, cut, topn):
    rng = np.random.default_rng(seed)
    probs = rng.dirichlet(np.ones(V) * 0.4, size=T).astype(np.float32)
    cands = [obs.pruned_log_probs(p, cut, topn) for p in probs]
    s_ref, t_ref = obs.prefix_beam_search(cands, beam, 0)
    s, t = _host_search(cands, beam)
    assert t == t_ref
    assert abs(s - s_ref) < 1e-4


def test_streaming_equals_offline(built_lib):
    from masr_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(9)
    probs = rng.dirichlet(np.ones(8) * 0.3, size=40).astype(np.float32)
    cands = [obs.pruned_log_probs(p, 1.0, 40) for p in probs]
    s_ref, t_ref = _host_search(cands, 16)
    h = C.c_void_p()
    assert lib.masr_beam_create(16, 0, C.byref(h)) == 0
    for lo in range(0, 40, 7):
        part = cands[lo:lo + 7]
        idx = np.array([[i for i, _ in c] for c in part], np.int32)
        lp = np.array([[l for _, l in c] for c in part], np.float32)
        cnt = np.full(len(part), 8, np.int32)
        assert lib.masr_beam_advance(h, idx.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p),
                                     cnt.ctypes.data_as(C.c_void_p), len(part), 8) == 0
    toks = np.zeros(64, np.int32)
    n, sc = C.c_int32(), C.c_float()
    lib.masr_beam_result(h, toks.ctypes.data_as(C.c_void_p), 64, C.byref(n), C.byref(sc))
    assert list(toks[:n.value]) == t_ref and abs(sc.value - s_ref) < 1e-5
    lib.masr_beam_reset(h)
    lib.masr_beam_result(h, toks.ctypes.data_as(C.c_void_p), 64, C.byref(n), C.byref(sc))
    assert n.value == 0
    lib.masr_beam_destroy(h)


def test_peaked_distribution_equals_greedy(built_lib):
    """with (almost) one-hot frames the best prefix is the greedy best path"""
    rng = np.random.default_rng(3)
    V, T = 10, 50
    ids = rng.integers(0, V, T)
    probs = np.full((T, V), 1e-4, np.float32)
    probs[np.arange(T), ids] = 1.0 - 1e-4 * (V - 1)
    vocab = ['<blank>'] + [chr(97 + i) for i in range(V - 1)]
    _, t_greedy = od.greedy_decoder(probs, vocab)
    s, toks = _host_search([obs.pruned_log_probs(p, 1.0, 40) for p in probs], 10)
    assert ''.join(vocab[t] for t in toks) == t_greedy


@pytest.mark.gpu
def test_gpu_pruning_and_decoder_api():
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    rng = np.random.default_rng(11)
    V = 300
    vocab = ['<blank>', '<unk>', '<space>'] + [chr(0x4e00 + i) for i in range(V - 4)] + ['<eos>']
    dec = BeamSearchDecoder(alpha=2.2, beta=4.3, beam_size=20, cutoff_prob=0.99, cutoff_top_n=40, vocab_list=vocab,
                            num_processes=4, language_model_path=None)
    probs = [rng.dirichlet(np.ones(V) * 0.02, size=T).astype(np.float32) for T in (37, 12, 25)]
    # pruning kernel == oracle pruning
    idx, logp, cnt, _, K = dec._candidates(probs[0])
    for t in range(37):
        ref = obs.pruned_log_probs(probs[0][t], 0.99, 40)
        assert cnt[t] == len(ref)
        assert [int(i) for i in idx[t, :cnt[t]]] == [i for i, _ in ref]
        assert np.allclose(logp[t, :cnt[t]], [l for _, l in ref], atol=1e-6)
    texts = dec.decode_batch_beam_search_offline(probs)
    for p, text in zip(probs, texts):
        s_ref, t_ref = obs.decode(p, vocab, 20, 0.99, 40)
        assert text == t_ref
    s, t = dec.decode_beam_search_offline(probs[1])
    assert t == texts[1]
    # streaming chunks give the offline result
    dec.reset_decoder()
    out = None
    for lo in range(0, 37, 16):
        out = dec.decode_chunk(probs[0][None, lo:lo + 16], np.array([min(16, 37 - lo)]))
    assert out[1] == texts[0]
    dec.reset_decoder()


def _gpu_vs_host(probs_list, beam, cut, topn):
    """GPU search (masr_beam_search_gpu) and host search (masr_beam_search_batch) on the same pruned candidates"""
    import torch
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    V = probs_list[0].shape[1]
    vocab = ['<blank>'] + [chr(0x4e00 + i) for i in range(V - 1)]
    dec = BeamSearchDecoder(alpha=0, beta=0, beam_size=beam, cutoff_prob=cut, cutoff_top_n=topn, vocab_list=vocab,
                            num_processes=4, language_model_path=None)
    assert dec.gpu_search_supported(max(p.shape[0] for p in probs_list), V)
    gpu = dec._batch(probs_list)
    dec.use_gpu_search = False
    host = dec._batch(probs_list)
    return gpu, host


@pytest.mark.gpu
@pytest.mark.parametrize('seed,V,Ts,beam,cut,topn,conc', [
    (0, 6, (12, 5, 9), 4, 1.0, 40, 0.4), (1, 50, (30, 30), 8, 0.99, 5, 0.4), (2, 300, (60, 41, 1), 20, 0.99, 40, 0.05),
    (3, 5, (40,), 300, 1.0, 40, 0.4), (4, 4233, (120, 77, 33, 120), 300, 0.99, 40, 0.002), (5, 600, (90, 64), 64, 0.95, 64, 0.01)])
def test_gpu_prefix_search_matches_host_search(seed, V, Ts, beam, cut, topn, conc):
    rng = np.random.default_rng(seed)
    probs = [rng.dirichlet(np.ones(V) * conc, size=T).astype(np.float32) for T in Ts]
    gpu, host = _gpu_vs_host(probs, beam, cut, topn)
    for (sg, tg), (sh, th) in zip(gpu, host):
        assert tg == th
        assert abs(sg - sh) < 1e-3 * max(1.0, abs(sh))


@pytest.mark.gpu
def test_gpu_prefix_search_peaked_equals_greedy_and_oracle():
    rng = np.random.default_rng(3)
    V, T = 40, 200
    ids = rng.integers(0, V, T)
    ids[rng.random(T) < 0.5] = 0                       # plenty of blanks and repeats
    probs = np.full((T, V), 2e-4, np.float32)
    probs[np.arange(T), ids] = 1.0 - 2e-4 * (V - 1)
    vocab = ['<blank>'] + [chr(0x4e00 + i) for i in range(V - 1)]
    _, t_greedy = od.greedy_decoder(probs, vocab)
    gpu, host = _gpu_vs_host([probs], 300, 0.99, 40)
    assert gpu[0][1] == t_greedy == host[0][1]
    s_ref, t_ref = obs.decode(probs[:60], vocab, 16, 1.0, 40)
    g2, _ = _gpu_vs_host([probs[:60]], 16, 1.0, 40)
    assert g2[0][1] == t_ref and abs(g2[0][0] - s_ref) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('seed,beam', [(0, 2), (1, 3), (2, 5), (3, 10), (4, 10), (5, 16)])
def test_gpu_prefix_search_small_beam_revived_prefixes(seed, beam):
    """few dominant symbols + a small beam: prefixes are dropped and re-created all the time while their extensions stay
    live -- a re-created prefix must be the SAME prefix (the reference keeps dead trie nodes that have live descendants)"""
    rng = np.random.default_rng(100 + seed)
    V, T = 24, 150
    alpha = np.full(V, 0.03)
    alpha[[0, 3, 7]] = 1.5
    probs = [rng.dirichlet(alpha, size=T).astype(np.float32), rng.dirichlet(alpha, size=T // 2).astype(np.float32)]
    gpu, host = _gpu_vs_host(probs, beam, 0.99, 40)
    for (sg, tg), (sh, th) in zip(gpu, host):
        assert tg == th
        assert abs(sg - sh) < 1e-3 * max(1.0, abs(sh))


@pytest.mark.gpu
@pytest.mark.parametrize('beam', [5, 300])
def test_gpu_streaming_search_equals_offline_and_host(beam):
    """decode_chunk with the device-resident search state (masr_gbeam_*) == whole-utterance search == host stream"""
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    rng = np.random.default_rng(77)
    V, T = 60, 131
    alpha = np.full(V, 0.02)
    alpha[[0, 5, 9, 11]] = 1.0
    probs = rng.dirichlet(alpha, size=T).astype(np.float32)
    vocab = ['<blank>'] + [chr(0x4e00 + i) for i in range(V - 1)]
    dec = BeamSearchDecoder(0, 0, beam, 0.99, 40, vocab, language_model_path=None)
    off = dec.decode_beam_search_offline(probs)
    for use_gpu in (True, False, True):                 # the second GPU pass also checks reset_decoder on a used stream
        dec.use_gpu_search = use_gpu
        dec.reset_decoder()
        out = None
        for lo in range(0, T, 16):
            n = min(16, T - lo)
            out = dec.decode_chunk(probs[None, lo:lo + n], np.array([n]))
        assert out[1] == off[1]
        assert abs(out[0] - off[0]) < 1e-3 * max(1.0, abs(off[0]))
    dec.reset_decoder()


# ---- external scorer (alpha / beta): ARPA loader + scoring + host search against the oracle restatement (CPU) ----------------
@pytest.fixture(scope='module')
def lm_pair(built_lib, tmp_path_factory):
    from masr_amd.decoders.lm_scorer import LanguageModel, write_synthetic_arpa
    from masr_amd.utils import synthetic
    vocab = synthetic.synthetic_vocab(60)
    path = write_synthetic_arpa(str(tmp_path_factory.mktemp('lm') / 'lm3.arpa'), vocab, order=3, seed=4, n_higher=1500)
    lm = LanguageModel(path, vocab)
    return vocab, lm, obs.ArpaLM(path)


def test_arpa_loader_and_scoring_match_oracle(lm_pair):
    vocab, lm, olm = lm_pair
    assert (lm.max_order, lm.is_character_based) == (3, True) and olm.max_order == 3
    assert lm.n_ngrams + lm.skipped == len(olm.grams) and lm.skipped == 1          # <unk> is KenLM's index 0: never a hit
    rng = np.random.default_rng(0)
    known = [i for i, t in enumerate(vocab) if olm.known(t)]
    seen_backoff = seen_hit = 0
    for k in range(3000):
        n = int(rng.integers(1, 6))
        ids = [int(rng.choice(known)) for _ in range(n)]
        if k % 7 == 0:
            ids[int(rng.integers(0, n))] = 1                                        # '<unk>' of the acoustic vocabulary: OOV
        words = [vocab[i] for i in ids]
        want = olm.cond_log_prob(olm.make_ngram(words))
        got = lm.cond_log_prob(ids)
        assert abs(got - want) < 2e-4 * max(1.0, abs(want)), (ids, got, want)
        seen_hit += tuple(olm.make_ngram(words)) in olm.grams
        seen_backoff += tuple(olm.make_ngram(words)) not in olm.grams and want > -999
        sw, sg = olm.sent_log_prob(words), lm.sentence_log_prob(ids)
        assert abs(sg - sw) < 2e-4 * max(1.0, abs(sw)), (ids, sg, sw)
    assert seen_hit > 20 and seen_backoff > 500                                     # both branches of the recursion were exercised
    assert abs(lm.sentence_log_prob([]) - olm.sent_log_prob([])) < 1e-4             # the empty-sentence quirk
    assert lm.cond_log_prob([1]) == -1000.0 and olm.cond_log_prob(olm.make_ngram([vocab[1]])) == -1000.0


def test_arpa_loader_rejects_what_it_cannot_score(built_lib, tmp_path):
    from masr_amd import _lib
    from masr_amd.decoders.lm_scorer import LanguageModel
    vocab = ['<blank>', '<unk>', 'a', 'b']
    words = tmp_path / 'word.arpa'
    words.write_text('\\data\\\nngram 1=4\n\n\\1-grams:\n-1.0\t<s>\t-0.5\n-1.0\t</s>\n-1.0\tab\t-0.3\n-1.2\ta\t-0.2\n\n\\end\\\n', encoding='utf-8')
    with pytest.raises(_lib.MasrError, match='word-based'):
        LanguageModel(str(words), vocab)
    klm = tmp_path / 'lm.klm'
    klm.write_bytes(b'mmap lm http://kheafield.com/code format version 5\n\0' + bytes(64))
    with pytest.raises(_lib.MasrError, match='KenLM binary'):
        LanguageModel(str(klm), vocab)
    with pytest.raises(_lib.MasrError, match='cannot open'):
        LanguageModel(str(tmp_path / 'missing.arpa'), vocab)


def _host_search_lm(cands, beam, lm, alpha, beta, blank=0):
    from masr_amd import _lib
    lib = _lib.lib()
    T, K = len(cands), max(len(c) for c in cands)
    idx, logp, cnt = np.zeros((T, K), np.int32), np.zeros((T, K), np.float32), np.zeros(T, np.int32)
    for t, c in enumerate(cands):
        cnt[t] = len(c)
        for k, (i, lp) in enumerate(c):
            idx[t, k], logp[t, k] = i, lp
    frames = np.array([T], np.int32)
    toks, lens, score = np.zeros((1, T + 1), np.int32), np.zeros(1, np.int32), np.zeros(1, np.float32)
    rc = lib.masr_beam_search_batch_lm(idx.ctypes.data_as(C.c_void_p), logp.ctypes.data_as(C.c_void_p),
                                       cnt.ctypes.data_as(C.c_void_p), frames.ctypes.data_as(C.c_void_p), 1, T, K, beam, blank, 2,
                                       lm.h, C.c_float(alpha), C.c_float(beta), None, toks.ctypes.data_as(C.c_void_p), T + 1,
                                       lens.ctypes.data_as(C.c_void_p), score.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return float(score[0]), list(toks[0, :lens[0]])


@pytest.mark.parametrize('seed,T,beam,cut,topn,alpha,beta', [(0, 14, 6, 1.0, 40, 2.2, 4.3), (1, 30, 12, 0.99, 8, 2.2, 4.3),
                                                             (2, 25, 40, 0.95, 10, 0.7, 1.5), (3, 20, 300, 1.0, 40, 2.2, 4.3),
                                                             (4, 18, 5, 0.9, 40, 0.0, 3.0), (5, 22, 9, 1.0, 12, 1.3, 0.0)])
def test_host_prefix_search_with_language_model_matches_oracle(lm_pair, seed, T, beam, cut, topn, alpha, beta):
    vocab, lm, olm = lm_pair
    rng = np.random.default_rng(seed)
    probs = rng.dirichlet(np.ones(len(vocab)) * 0.25, size=T).astype(np.float32)
    cands = [obs.pruned_log_probs(p, cut, topn) for p in probs]
    s_ref, t_ref = obs.prefix_beam_search_lm(cands, vocab, olm, alpha, beta, beam, 0)
    s, t = _host_search_lm(cands, beam, lm, alpha, beta)
    assert t == t_ref, (t, t_ref)
    assert abs(s - s_ref) < 2e-3 * max(1.0, abs(s_ref)), (s, s_ref)
    s0, t0 = obs.prefix_beam_search(cands, beam, 0)                  # and the scorer does change the search on these inputs
    if alpha > 1:
        assert (t0 != t_ref) or abs(s0 - s_ref) > 1e-3


def test_streaming_host_search_with_language_model(lm_pair):
    from masr_amd import _lib
    lib = _lib.lib()
    vocab, lm, olm = lm_pair
    rng = np.random.default_rng(11)
    probs = rng.dirichlet(np.ones(len(vocab)) * 0.25, size=35).astype(np.float32)
    cands = [obs.pruned_log_probs(p, 1.0, 10) for p in probs]
    s_ref, t_ref = _host_search_lm(cands, 16, lm, 2.2, 4.3)
    h = C.c_void_p()
    assert lib.masr_beam_create(16, 0, C.byref(h)) == 0
    assert lib.masr_beam_set_lm(h, lm.h, C.c_float(2.2), C.c_float(4.3)) == 0
    for lo in range(0, 35, 6):
        part = cands[lo:lo + 6]
        idx = np.array([[i for i, _ in c] for c in part], np.int32)
        lp = np.array([[l for _, l in c] for c in part], np.float32)
        cnt = np.full(len(part), 10, np.int32)
        assert lib.masr_beam_advance(h, idx.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p),
                                     cnt.ctypes.data_as(C.c_void_p), len(part), 10) == 0
    toks = np.zeros(64, np.int32)
    n, sc = C.c_int32(), C.c_float()
    lib.masr_beam_result(h, toks.ctypes.data_as(C.c_void_p), 64, C.byref(n), C.byref(sc))
    assert list(toks[:n.value]) == t_ref and abs(sc.value - s_ref) < 1e-4
    lib.masr_beam_destroy(h)


# ---- external scorer on the GPU: kernel == host search == oracle restatement at the shipped alpha = 2.2, beta = 4.3 ----------
def _lm_decoder(tmp, V, beam, cut, topn, alpha=2.2, beta=4.3, order=3, seed=4):
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    from masr_amd.decoders.lm_scorer import write_synthetic_arpa
    vocab = ['<blank>', '<unk>', '<space>'] + [chr(0x4e00 + i) for i in range(V - 3)]
    path = write_synthetic_arpa(str(tmp / f'lm{order}_{V}.arpa'), vocab, order=order, seed=seed, n_higher=40 * V)
    dec = BeamSearchDecoder(alpha=alpha, beta=beta, beam_size=beam, cutoff_prob=cut, cutoff_top_n=topn, vocab_list=vocab,
                            num_processes=4, language_model_path=path)
    assert dec._ext_scorer is not None and dec._ext_scorer.max_order == order
    return dec, vocab, obs.ArpaLM(path)


@pytest.mark.gpu
@pytest.mark.parametrize('prune', [True, False])
@pytest.mark.parametrize('seed,V,Ts,beam,cut,topn,conc,order', [
    (0, 30, (14, 9), 6, 1.0, 40, 0.3, 3), (1, 60, (30, 30, 7), 12, 0.99, 8, 0.2, 3), (2, 300, (60, 41, 1), 20, 0.99, 40, 0.05, 4),
    (3, 12, (40,), 300, 1.0, 40, 0.4, 5), (4, 4233, (100, 77, 33), 300, 0.99, 40, 0.002, 3)])
def test_gpu_search_with_language_model_matches_host_and_oracle(tmp_path, seed, V, Ts, beam, cut, topn, conc, order, prune):
    """prune=True: the decoder as the reference runs it (min_cutoff / full_beam rule on, ln p(blank) from the pruning kernel);
    prune=False: every candidate of a frame scored"""
    rng = np.random.default_rng(seed)
    probs = [rng.dirichlet(np.ones(V) * conc, size=T).astype(np.float32) for T in Ts]
    dec, vocab, olm = _lm_decoder(tmp_path, V, beam, cut, topn, order=order)
    dec.prune_min_cutoff = prune
    assert dec.gpu_search_supported(max(Ts), V)
    gpu = dec._batch(probs)
    dec.use_gpu_search = False
    host = dec._batch(probs)
    for (sg, tg), (sh, th) in zip(gpu, host):
        assert tg == th, (tg, th)
        assert abs(sg - sh) < 2e-3 * max(1.0, abs(sh)), (sg, sh)
    if V <= 300:                                           # the pure-Python oracle on the smaller cases
        sc = obs.Scorer(olm, vocab, 2.2, 4.3)
        for p, (sg, tg) in zip(probs, gpu):
            s_ref, toks_ref, _ = obs.ctc_beam_search_decoder(p, vocab, beam, cut, topn, sc, 0, prune=prune)
            t_ref = ''.join(vocab[t] for t in toks_ref).replace('<space>', ' ')
            assert tg == t_ref and abs(sg - s_ref) < 2e-3 * max(1.0, abs(s_ref)), (sg, s_ref)
    # the scorer is live: the LM-free search of the same candidates differs somewhere
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    plain = BeamSearchDecoder(0, 0, beam, cut, topn, vocab, language_model_path=None)._batch(probs)
    assert any(tp != tg or abs(sp - sg) > 1e-2 for (sp, tp), (sg, tg) in zip(plain, gpu))


def _mixed_posteriors(rng, V, T, sharp_every=3):
    """frames of a trained model's kind (1-3 candidates pass cutoff_prob) with flat frames (cutoff_top_n candidates) in between:
    the GPU search switches between its narrow and its wide step inside one utterance"""
    logits = rng.normal(0, 1.0, (T, V))
    for t in range(T):
        if t % sharp_every:
            logits[t] *= 14.0
    e = np.exp(logits - logits.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('lm_order,beam,V,prune', [(0, 300, 4233, True), (3, 300, 4233, True), (3, 300, 4233, False), (5, 120, 900, True),
                                                   (0, 20, 300, True), (3, 500, 600, True), (3, 16, 50, True)])
def test_gpu_narrow_step_equals_wide_step_on_the_same_frames(tmp_path, lm_order, beam, V, prune):
    """masr_debug_set key 37: 1 (default) runs frames of <= 1024 extension entries on the narrow step of beam_gpu.hip, 0 runs every
    frame on the wide step.  Same candidates, same frames: transcripts AND scores must be identical (not close), offline and
    through the streaming state; both equal the host search."""
    from masr_amd import runtime
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    rng = np.random.default_rng(77 + beam)
    probs = [_mixed_posteriors(rng, V, T, sharp_every=se) for T, se in ((90, 3), (61, 1000), (40, 2), (17, 1))]
    if lm_order:
        dec, vocab, _ = _lm_decoder(tmp_path, V, beam, 0.99, 40, order=lm_order)
        dec.prune_min_cutoff = prune
    else:
        vocab = ['<blank>'] + [chr(0x4e00 + i) for i in range(V - 1)]
        dec = BeamSearchDecoder(0, 0, beam, 0.99, 40, vocab, num_processes=4, language_model_path=None)
    eng = runtime.aux_engine()
    out = {}
    try:
        for mode in (1, 0):
            assert eng.lib.masr_debug_set(eng.h, 37, mode) == 0
            dec.use_gpu_search = True
            off = dec._batch(probs)
            dec.reset_decoder()
            chunks = []
            for lo in range(0, 90, 16):
                chunks.append(dec.decode_chunk(probs[0][None, lo:lo + 16], [min(16, 90 - lo)]))
            dec.reset_decoder()
            out[mode] = (off, chunks)
    finally:
        eng.lib.masr_debug_set(eng.h, 37, 1)
    assert out[1] == out[0]
    assert out[1][1][-1][1] == out[1][0][0][1] and abs(out[1][1][-1][0] - out[1][0][0][0]) < 1e-3 * max(1.0, abs(out[1][0][0][0]))
    dec.use_gpu_search = False
    host = dec._batch(probs)
    for (sg, tg), (sh, th) in zip(out[1][0], host):
        assert tg == th and abs(sg - sh) < 2e-3 * max(1.0, abs(sh))


@pytest.mark.gpu
def test_two_searches_on_two_streams_do_not_share_workspaces(tmp_path):
    """predict_batch launches the prefix searches of consecutive passes on two side streams so that they run next to each other
    (one workgroup per utterance each): searches launched on different streams must work in workspaces of their own -- same
    transcripts and scores as the searches run one after the other."""
    import torch
    V, beam = 600, 300
    rng = np.random.default_rng(11)
    sets = [[rng.dirichlet(np.ones(V) * 0.02, size=T).astype(np.float32) for T in Ts] for Ts in ((180, 150, 120, 90), (170, 60, 140))]
    dec, vocab, _ = _lm_decoder(tmp_path, V, beam, 0.99, 40, order=3)
    dev_sets = [[torch.from_numpy(p).cuda() for p in ps] for ps in sets]
    sequential = [dec._batch(ps) for ps in dev_sets]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(3):
        pending = []
        for st, ps in zip(streams, dev_sets):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                pending.append(dec._batch(ps, defer=True))
        both = [dec._batch_collect(h) for h in pending]
        for seq, con in zip(sequential, both):
            for (ss, ts), (sc, tc) in zip(seq, con):
                assert ts == tc and abs(ss - sc) < 1e-4 * max(1.0, abs(ss)), (ts, tc, ss, sc)


@pytest.mark.gpu
@pytest.mark.parametrize('seed,alpha,beta,beam', [(0, 1.0, 0.0, 8), (1, 0.5, 0.3, 300), (2, 1.0, -0.5, 40)])
def test_gpu_pruning_rule_at_settings_where_it_cuts(tmp_path, seed, alpha, beta, beam):
    """low beta: every candidate less likely than blank is cut for the prefixes at the bottom of a full beam -- GPU == host ==
    oracle with the rule on, and the rule-off search differs"""
    rng = np.random.default_rng(300 + seed)
    V = 40
    conc = np.full(V, 0.05)
    conc[0] = 2.0
    probs = [rng.dirichlet(conc, size=T).astype(np.float32) for T in (80, 55)]
    dec, vocab, olm = _lm_decoder(tmp_path, V, beam, 1.0, 40, alpha=alpha, beta=beta)
    gpu = dec._batch(probs)
    dec.use_gpu_search = False
    host = dec._batch(probs)
    sc = obs.Scorer(olm, vocab, alpha, beta)
    differs = False
    for p, (sg, tg), (sh, th) in zip(probs, gpu, host):
        assert tg == th and abs(sg - sh) < 2e-3 * max(1.0, abs(sh))
        s_ref, toks_ref, _ = obs.ctc_beam_search_decoder(p, vocab, beam, 1.0, 40, sc, 0, prune=True)
        assert tg == ''.join(vocab[t] for t in toks_ref).replace('<space>', ' ')
        assert abs(sg - s_ref) < 2e-3 * max(1.0, abs(s_ref))
        s_off, toks_off, _ = obs.ctc_beam_search_decoder(p, vocab, beam, 1.0, 40, sc, 0, prune=False)
        differs |= toks_off != toks_ref or abs(s_off - s_ref) > 1e-3
    assert differs


@pytest.mark.gpu
def test_word_based_scorer_through_the_decoder_api(word_lm):
    """a space-delimited LM through BeamSearchDecoder: the GPU prunes the vocabulary, the prefix search (spelling dictionary)
    runs on host threads; offline == streaming == oracle"""
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    vocab, words, lm, olm, path = word_lm
    rng = np.random.default_rng(5)
    probs = [_word_probs(rng, vocab, words, T, 0.3) for T in (60, 37)]
    dec = BeamSearchDecoder(2.2, 4.3, 20, 0.99, 40, vocab, num_processes=2, language_model_path=path)
    assert not dec.gpu_search_supported(60, len(vocab)) and dec._ext_scorer.get_dict_size() == len(words) - 1
    texts = dec.decode_batch_beam_search_offline(probs)
    sc = obs.Scorer(olm, vocab, 2.2, 4.3)
    for p, text in zip(probs, texts):
        s_ref, toks_ref, _ = obs.ctc_beam_search_decoder(p, vocab, 20, 0.99, 40, sc, 0)
        assert text == ''.join(vocab[t] for t in toks_ref).replace('<space>', ' ')
    out = None
    for lo in range(0, 60, 16):
        out = dec.decode_chunk(probs[0][None, lo:lo + 16], np.array([min(16, 60 - lo)]))
    assert out[1] == texts[0]
    dec.reset_decoder()


@pytest.mark.gpu
@pytest.mark.parametrize('beam', [5, 300])
def test_gpu_streaming_search_with_language_model(tmp_path, beam):
    """decode_chunk on the device-resident search state with the scorer bound == whole-utterance search == host stream;
    a forked decoder (one per serving session) has its own search state over the same LM table"""
    rng = np.random.default_rng(78)
    V, T = 60, 131
    conc = np.full(V, 0.02)
    conc[[0, 5, 9, 11]] = 1.0
    probs = rng.dirichlet(conc, size=T).astype(np.float32)
    dec, vocab, olm = _lm_decoder(tmp_path, V, beam, 0.99, 40)
    off = dec.decode_beam_search_offline(probs)
    twin = dec.fork()
    for d, use_gpu in ((dec, True), (dec, False), (twin, True), (dec, True)):
        d.use_gpu_search = use_gpu
        d.reset_decoder()
        out = None
        for lo in range(0, T, 16):
            n = min(16, T - lo)
            out = d.decode_chunk(probs[None, lo:lo + n], np.array([n]))
        assert out[1] == off[1]
        assert abs(out[0] - off[0]) < 2e-3 * max(1.0, abs(off[0]))
    twin.close()
    dec.reset_decoder()


# ---- the decoder's min_cutoff / full_beam pruning (with a scorer bound) and word-based scorers: host search == oracle (CPU) ------
def _arrays(cands):
    T, K = len(cands), max(len(c) for c in cands)
    idx, logp, cnt = np.zeros((T, K), np.int32), np.zeros((T, K), np.float32), np.zeros(T, np.int32)
    for t, c in enumerate(cands):
        cnt[t] = len(c)
        for k, (i, lp) in enumerate(c):
            idx[t, k], logp[t, k] = i, lp
    return idx, logp, cnt, T, K


def _host_search_pruned(cands, blank_lp, beam, lm, alpha, beta, blank=0):
    """masr_beam_search_batch_lm with ln p(blank) per frame: the pruning rule is on"""
    from masr_amd import _lib
    lib = _lib.lib()
    idx, logp, cnt, T, K = _arrays(cands)
    blp = np.asarray(blank_lp, np.float32)
    frames = np.array([T], np.int32)
    toks, lens, score = np.zeros((1, T + 1), np.int32), np.zeros(1, np.int32), np.zeros(1, np.float32)
    rc = lib.masr_beam_search_batch_lm(idx.ctypes.data_as(C.c_void_p), logp.ctypes.data_as(C.c_void_p),
                                       cnt.ctypes.data_as(C.c_void_p), frames.ctypes.data_as(C.c_void_p), 1, T, K, beam, blank, 2,
                                       lm.h, C.c_float(alpha), C.c_float(beta), blp.ctypes.data_as(C.c_void_p),
                                       toks.ctypes.data_as(C.c_void_p), T + 1, lens.ctypes.data_as(C.c_void_p),
                                       score.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return float(score[0]), list(toks[0, :lens[0]])


def _blank_lp(probs, blank=0):
    return [float(np.log(np.float32(p[blank]))) for p in probs]


@pytest.mark.parametrize('seed,T,beam,cut,topn,alpha,beta', [(0, 30, 6, 1.0, 40, 2.2, 4.3), (1, 40, 12, 0.99, 8, 2.2, 4.3),
                                                             (2, 35, 40, 0.95, 10, 0.7, 1.5), (3, 25, 300, 1.0, 40, 2.2, 4.3),
                                                             (4, 30, 5, 0.9, 40, 0.0, 3.0), (5, 30, 9, 1.0, 12, 1.3, -0.5)])
def test_host_search_with_min_cutoff_pruning_matches_oracle(lm_pair, seed, T, beam, cut, topn, alpha, beta):
    vocab, lm, olm = lm_pair
    rng = np.random.default_rng(50 + seed)
    probs = rng.dirichlet(np.ones(len(vocab)) * 0.25, size=T).astype(np.float32)
    cands = [obs.pruned_log_probs(p, cut, topn) for p in probs]
    blp = _blank_lp(probs)
    sc = obs.Scorer(olm, vocab, alpha, beta)
    a_ref, t_ref, _ = obs.ctc_beam_search_decoder(None, vocab, beam, scorer=sc, cands=cands, blank_logp=blp)
    s, t = _host_search_pruned(cands, blp, beam, lm, alpha, beta)
    assert t == t_ref, (t, t_ref)
    assert abs(s - a_ref) < 2e-3 * max(1.0, abs(a_ref)), (s, a_ref)


def test_min_cutoff_pruning_changes_the_search(built_lib, tmp_path):
    """five frames, beam 2, on which the decoder's rule changes the transcript: without it the search returns 'ba', with it
    (what the published decoder does whenever a scorer is bound) the extension that leads there is cut in a full beam and the
    result is 'b'.  Oracle and host search agree in both modes."""
    from masr_amd.decoders.lm_scorer import LanguageModel
    vocab = ['<blank>', 'a', 'b', 'c']
    path = str(tmp_path / 'uni.arpa')
    with open(path, 'w') as f:
        f.write('\\data\\\nngram 1=6\n\n\\1-grams:\n-99\t<s>\n-1\t</s>\n-1\t<unk>\n-0.5\ta\n-0.7\tb\n-0.9\tc\n\n\\end\\\n')
    olm = obs.ArpaLM(path)
    sc = obs.Scorer(olm, vocab, 0.5, 0.3)
    frames = np.array([[0.23, 0.08, 0.64, 0.05], [0.39, 0.37, 0.05, 0.19], [0.91, 0.01, 0.03, 0.05], [0.61, 0.18, 0.11, 0.10],
                       [0.59, 0.21, 0.15, 0.05]], np.float32)
    full, tf, _ = obs.ctc_beam_search_decoder(frames, vocab, 2, 1.0, 40, sc, 0, prune=False)
    cut_, tc, _ = obs.ctc_beam_search_decoder(frames, vocab, 2, 1.0, 40, sc, 0, prune=True)
    assert tf == [2, 1] and tc == [2]
    lm = LanguageModel(path, vocab)
    cands = [obs.pruned_log_probs(p, 1.0, 40) for p in frames]
    s_cut, t_cut = _host_search_pruned(cands, _blank_lp(frames), 2, lm, 0.5, 0.3)
    s_full, t_full = _host_search_lm(cands, 2, lm, 0.5, 0.3)
    assert t_cut == tc and t_full == tf
    assert abs(s_cut - cut_) < 1e-4 and abs(s_full - full) < 1e-4


@pytest.fixture(scope='module')
def word_lm(built_lib, tmp_path_factory):
    from masr_amd.decoders.lm_scorer import LanguageModel, write_synthetic_word_arpa
    vocab = ['<blank>', '<unk>', '<space>', "'"] + [chr(97 + i) for i in range(12)]
    rng = np.random.default_rng(8)
    letters = vocab[3:]
    words = sorted({''.join(letters[int(i)] for i in rng.integers(0, len(letters), int(rng.integers(1, 5)))) for _ in range(60)})
    words += ['z' + words[0]]                                          # a word the acoustic model cannot spell: not in the dictionary
    path = write_synthetic_word_arpa(str(tmp_path_factory.mktemp('wlm') / 'words.arpa'), words, order=3, seed=2)
    return vocab, words, LanguageModel(path, vocab), obs.ArpaLM(path), path


def test_word_based_scorer_loads_and_scores_like_the_oracle(word_lm):
    vocab, words, lm, olm, _ = word_lm
    sc = obs.Scorer(olm, vocab, 1.0, 1.0)
    assert not lm.is_character_based and not sc.is_character_based and lm.max_order == 3
    assert lm.get_dict_size() == sc.dict_size == len(words) - 1 and lm.word_id('nope') == -1
    rng = np.random.default_rng(1)
    hit = back = 0
    for k in range(2000):
        n = int(rng.integers(1, 5))
        ws = [words[int(rng.integers(0, len(words)))] for _ in range(n)]
        if k % 9 == 0:
            ws[int(rng.integers(0, n))] = 'qqq'                        # not an LM word: OOV
        ids = [lm.word_id(w) for w in ws]
        want = olm.cond_log_prob(olm.make_ngram(ws))
        got = lm.cond_log_prob(ids)
        assert abs(got - want) < 2e-4 * max(1.0, abs(want)), (ws, got, want)
        hit += tuple(olm.make_ngram(ws)) in olm.grams
        back += tuple(olm.make_ngram(ws)) not in olm.grams and want > -999
        sw, sg = olm.sent_log_prob(ws), lm.sentence_log_prob(ids)
        assert abs(sg - sw) < 2e-4 * max(1.0, abs(sw)), (ws, sg, sw)
    assert hit > 5 and back > 300
    assert abs(lm.sentence_log_prob([]) - olm.sent_log_prob([])) < 1e-4


def _word_probs(rng, vocab, words, T, noise):
    """frames that spell a few dictionary words (with blanks, repeats and noise) so that spaces and word scores matter"""
    ids = []
    while len(ids) < T:
        w = words[int(rng.integers(0, len(words) - 1))]
        for ch in w:
            ids += [vocab.index(ch)] * int(rng.integers(1, 3)) + [0] * int(rng.integers(0, 2))
        ids += [2] * int(rng.integers(1, 3))
    ids = ids[:T]
    probs = rng.dirichlet(np.ones(len(vocab)) * noise, size=T)
    for t, i in enumerate(ids):
        probs[t] = 0.55 * probs[t]
        probs[t, i] += 0.45
    return probs.astype(np.float32)


@pytest.mark.parametrize('seed,T,beam,cut,topn,alpha,beta,prune', [(0, 40, 8, 1.0, 40, 2.2, 4.3, True), (1, 60, 25, 0.99, 10, 1.5, 0.8, True),
                                                                   (2, 50, 300, 1.0, 40, 2.2, 4.3, True), (3, 45, 6, 0.95, 40, 0.6, 2.0, False),
                                                                   (4, 70, 16, 1.0, 40, 2.2, 4.3, False), (5, 30, 3, 1.0, 40, 3.0, 1.0, True)])
def test_host_search_with_word_based_scorer_matches_oracle(word_lm, seed, T, beam, cut, topn, alpha, beta, prune):
    """space-delimited LM: extensions are confined to the spelling dictionary (with the published code's start-over quirk after
    a complete word), a word is scored when its space arrives, the unfinished last word at the end, approx_ctc over the words"""
    vocab, words, lm, olm, _ = word_lm
    rng = np.random.default_rng(70 + seed)
    probs = _word_probs(rng, vocab, words, T, 0.3)
    cands = [obs.pruned_log_probs(p, cut, topn) for p in probs]
    blp = _blank_lp(probs)
    sc = obs.Scorer(olm, vocab, alpha, beta)
    a_ref, t_ref, _ = obs.ctc_beam_search_decoder(None, vocab, beam, scorer=sc, cands=cands, blank_logp=blp, prune=prune)
    if prune:
        s, t = _host_search_pruned(cands, blp, beam, lm, alpha, beta)
    else:
        s, t = _host_search_lm(cands, beam, lm, alpha, beta)
    assert t == t_ref, (t, t_ref)
    assert abs(s - a_ref) < 2e-3 * max(1.0, abs(a_ref)), (s, a_ref)
    text = ''.join(vocab[i] for i in t).replace('<space>', ' ')
    assert all(w in words for w in text.split(' ')[:-1])             # every finished word is a dictionary word
    # the LM-free search of the same frames leaves the dictionary
    _, t0 = obs.prefix_beam_search(cands, beam, 0)
    assert t0 != t_ref


def test_streaming_host_search_with_word_based_scorer(word_lm):
    from masr_amd import _lib
    lib = _lib.lib()
    vocab, words, lm, olm, _ = word_lm
    rng = np.random.default_rng(123)
    probs = _word_probs(rng, vocab, words, 48, 0.3)
    cands = [obs.pruned_log_probs(p, 1.0, 16) for p in probs]
    blp = np.asarray(_blank_lp(probs), np.float32)
    s_ref, t_ref = _host_search_pruned(cands, blp, 12, lm, 2.2, 4.3)
    h = C.c_void_p()
    assert lib.masr_beam_create(12, 0, C.byref(h)) == 0
    assert lib.masr_beam_set_lm(h, lm.h, C.c_float(2.2), C.c_float(4.3)) == 0
    toks = np.zeros(64, np.int32)
    n, sc = C.c_int32(), C.c_float()
    for lo in range(0, 48, 7):
        idx, lp, cnt, Tn, K = _arrays(cands[lo:lo + 7])
        b = blp[lo:lo + 7].copy()
        assert lib.masr_beam_advance_lm(h, idx.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p),
                                        cnt.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), Tn, K) == 0
        lib.masr_beam_result(h, toks.ctypes.data_as(C.c_void_p), 64, C.byref(n), C.byref(sc))   # asking in between changes nothing
    assert list(toks[:n.value]) == t_ref and abs(sc.value - s_ref) < 1e-4
    lib.masr_beam_destroy(h)


def test_missing_language_model_is_the_reference_assertion(built_lib, tmp_path):
    """beam_search_decoder.py:28 asserts on a missing model file; the scorer-free search needs language_model_path=None"""
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    vocab = ['<blank>', 'a', 'b']
    with pytest.raises(AssertionError, match='语言模型不存在'):
        BeamSearchDecoder(2.2, 4.3, 10, 0.99, 40, vocab, language_model_path=str(tmp_path / 'absent.klm'))
    with pytest.raises(AssertionError, match='语言模型不存在'):
        BeamSearchDecoder(2.2, 4.3, 10, 0.99, 40, vocab)              # the reference's default path: nothing to download here
    dec = BeamSearchDecoder(2.2, 4.3, 10, 0.99, 40, vocab, language_model_path=None)
    assert dec._ext_scorer is None and dec.alpha == 0 and dec.beta == 0
    dec.close()


# ---- BASELINE configs[2]'s size regime: T' = 498, V = 4233, beam 300, top-n 40, alpha 2.2 / beta 4.3, 3- and 5-gram LM ---------
# The pure-Python oracle needs minutes here: its answers are the committed fixture tests/golden/beam_regime.npz
# (oracle/make_beam_golden.py); the language models are re-created from the same seeds.
import os

REGIME = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'beam_regime.npz')
REGIME_CASES = [('lm3', 2.2, 4.3, True), ('lm3', 2.2, 4.3, False), ('lm5', 2.2, 4.3, True), ('lm5', 2.2, 4.3, False),
                ('lm3_a1b0', 1.0, 0.0, True), ('lm3_a1b0', 1.0, 0.0, False)]


@pytest.fixture(scope='module')
def regime(built_lib, tmp_path_factory):
    from masr_amd.decoders.lm_scorer import LanguageModel, write_synthetic_arpa
    from masr_amd.utils import synthetic
    from oracle.make_beam_golden import LMS
    z = np.load(REGIME)
    T, V, beam, K = (int(x) for x in z['meta'])
    vocab = synthetic.synthetic_vocab(V)
    d = tmp_path_factory.mktemp('regime_lm')
    lms = {name: LanguageModel(write_synthetic_arpa(str(d / (name + '.arpa')), vocab, **kw), vocab) for name, kw in LMS.items()}
    lms['lm3_a1b0'] = lms['lm3']
    return z, T, V, beam, K, vocab, lms


@pytest.mark.parametrize('name,alpha,beta,prune', REGIME_CASES)
def test_host_search_at_config2_size_matches_the_oracle_fixture(regime, name, alpha, beta, prune):
    from masr_amd import _lib
    lib = _lib.lib()
    z, T, V, beam, K, vocab, lms = regime
    idx, logp, cnt = z['idx'].astype(np.int32), z['logp'], z['cnt']
    blp = z['blank_lp'].astype(np.float32)
    frames = np.array([T], np.int32)
    toks, lens, score = np.zeros((1, T + 1), np.int32), np.zeros(1, np.int32), np.zeros(1, np.float32)
    rc = lib.masr_beam_search_batch_lm(idx.ctypes.data_as(C.c_void_p), logp.ctypes.data_as(C.c_void_p),
                                       cnt.ctypes.data_as(C.c_void_p), frames.ctypes.data_as(C.c_void_p), 1, T, K, beam, 0, 1,
                                       lms[name].h, C.c_float(alpha), C.c_float(beta),
                                       blp.ctypes.data_as(C.c_void_p) if prune else None, toks.ctypes.data_as(C.c_void_p),
                                       T + 1, lens.ctypes.data_as(C.c_void_p), score.ctypes.data_as(C.c_void_p))
    assert rc == 0
    tag = f'{name}_{"pruned" if prune else "full"}'
    assert list(toks[0, :lens[0]]) == list(z[tag + '_tokens'])
    want = float(z[tag + '_score'][0])
    assert abs(float(score[0]) - want) < 1e-3 * abs(want), (float(score[0]), want)


def test_the_regime_fixture_is_the_size_it_claims(regime):
    z, T, V, beam, K = regime[:5]
    assert (T, V, beam, K) == (498, 4233, 300, 40)
    assert (z['cnt'] == 40).sum() > 100 and len(z['lm3_pruned_tokens']) > 200 and len(z['lm5_pruned_tokens']) > 200
    assert list(z['lm3_pruned_tokens']) != list(z['lm5_pruned_tokens']) != list(z['nolm_tokens'])


@pytest.mark.gpu
@pytest.mark.parametrize('name,alpha,beta,prune', REGIME_CASES)
def test_gpu_search_at_config2_size_matches_host_and_the_oracle_fixture(regime, name, alpha, beta, prune):
    """the GPU kernel on the stored candidates of one 498-frame utterance (x 3 copies in one launch, one of them cut short):
    150 k-entry LM table in HBM, 64-bit prefix identities over ~500 extensions, fp32 log-sum-exp over 498 frames"""
    import torch
    from masr_amd import _lib, runtime
    from masr_amd._lib import check
    lib = _lib.lib()
    z, T, V, beam, K, vocab, lms = regime
    eng = runtime.aux_engine()
    dev = eng.device
    B = 3
    fr_h = np.array([T, T, 301], np.int32)
    idx = torch.from_numpy(np.tile(z['idx'].astype(np.int32)[None], (B, 1, 1))).to(dev).contiguous()
    logp = torch.from_numpy(np.tile(z['logp'][None], (B, 1, 1))).to(dev).contiguous()
    cnt = torch.from_numpy(np.tile(z['cnt'][None], (B, 1))).to(dev).contiguous()
    blp = torch.from_numpy(np.tile(z['blank_lp'].astype(np.float32)[None], (B, 1))).to(dev).contiguous()
    fr = torch.from_numpy(fr_h).to(dev)
    toks = torch.zeros(B, T + 1, dtype=torch.int32, device=dev)
    lens = torch.zeros(B, dtype=torch.int32, device=dev)
    score = torch.zeros(B, dtype=torch.float32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    check(lib.masr_beam_search_gpu_lm(eng.h, P(idx), P(logp), P(cnt), P(fr), B, T, K, beam, 0, lms[name].h, C.c_float(alpha),
                                      C.c_float(beta), P(blp) if prune else None, P(toks), T + 1, P(lens), P(score),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    toks, lens, score = toks.cpu().numpy(), lens.cpu().numpy(), score.cpu().numpy()
    tag = f'{name}_{"pruned" if prune else "full"}'
    want_t, want_s = list(z[tag + '_tokens']), float(z[tag + '_score'][0])
    for b in (0, 1):
        assert list(toks[b, :lens[b]]) == want_t
        assert abs(float(score[b]) - want_s) < 1e-3 * abs(want_s), (float(score[b]), want_s)
    # the shortened copy against the host search of the same 301 frames
    h_idx, h_logp, h_cnt = z['idx'].astype(np.int32)[:301].copy(), z['logp'][:301].copy(), z['cnt'][:301].copy()
    h_blp = z['blank_lp'].astype(np.float32)[:301].copy()
    ht, hl, hs = np.zeros((1, 302), np.int32), np.zeros(1, np.int32), np.zeros(1, np.float32)
    f1 = np.array([301], np.int32)
    assert lib.masr_beam_search_batch_lm(h_idx.ctypes.data_as(C.c_void_p), h_logp.ctypes.data_as(C.c_void_p),
                                         h_cnt.ctypes.data_as(C.c_void_p), f1.ctypes.data_as(C.c_void_p), 1, 301, K, beam, 0, 1,
                                         lms[name].h, C.c_float(alpha), C.c_float(beta),
                                         h_blp.ctypes.data_as(C.c_void_p) if prune else None, ht.ctypes.data_as(C.c_void_p), 302,
                                         hl.ctypes.data_as(C.c_void_p), hs.ctypes.data_as(C.c_void_p)) == 0
    assert list(toks[2, :lens[2]]) == list(ht[0, :hl[0]])
    assert abs(float(score[2]) - float(hs[0])) < 1e-3 * abs(float(hs[0]))


@pytest.mark.gpu
def test_gpu_search_without_scorer_at_config2_size_matches_the_oracle_fixture(regime):
    import torch
    from masr_amd import _lib, runtime
    from masr_amd._lib import check
    lib = _lib.lib()
    z, T, V, beam, K, vocab, lms = regime
    eng = runtime.aux_engine()
    dev = eng.device
    idx = torch.from_numpy(z['idx'].astype(np.int32)).to(dev).contiguous()
    logp = torch.from_numpy(z['logp']).to(dev).contiguous()
    cnt = torch.from_numpy(z['cnt']).to(dev).contiguous()
    fr = torch.tensor([T], dtype=torch.int32, device=dev)
    toks = torch.zeros(1, T + 1, dtype=torch.int32, device=dev)
    lens = torch.zeros(1, dtype=torch.int32, device=dev)
    score = torch.zeros(1, dtype=torch.float32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    check(lib.masr_beam_search_gpu(eng.h, P(idx), P(logp), P(cnt), P(fr), 1, T, K, beam, 0, P(toks), T + 1, P(lens), P(score),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    n = int(lens.item())
    assert toks[0, :n].cpu().tolist() == list(z['nolm_tokens'])
    assert abs(float(score.item()) - float(z['nolm_score'][0])) < 1e-3 * abs(float(z['nolm_score'][0]))
