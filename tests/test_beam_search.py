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
                            num_processes=4)
    probs = [rng.dirichlet(np.ones(V) * 0.02, size=T).astype(np.float32) for T in (37, 12, 25)]
    # pruning kernel == oracle pruning
    idx, logp, cnt, K = dec._candidates(probs[0])
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
                            num_processes=4)
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
    dec = BeamSearchDecoder(0, 0, beam, 0.99, 40, vocab)
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
                                       lm.h, C.c_float(alpha), C.c_float(beta), toks.ctypes.data_as(C.c_void_p), T + 1,
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
@pytest.mark.parametrize('seed,V,Ts,beam,cut,topn,conc,order', [
    (0, 30, (14, 9), 6, 1.0, 40, 0.3, 3), (1, 60, (30, 30, 7), 12, 0.99, 8, 0.2, 3), (2, 300, (60, 41, 1), 20, 0.99, 40, 0.05, 4),
    (3, 12, (40,), 300, 1.0, 40, 0.4, 5), (4, 4233, (100, 77, 33), 300, 0.99, 40, 0.002, 3)])
def test_gpu_search_with_language_model_matches_host_and_oracle(tmp_path, seed, V, Ts, beam, cut, topn, conc, order):
    rng = np.random.default_rng(seed)
    probs = [rng.dirichlet(np.ones(V) * conc, size=T).astype(np.float32) for T in Ts]
    dec, vocab, olm = _lm_decoder(tmp_path, V, beam, cut, topn, order=order)
    assert dec.gpu_search_supported(max(Ts), V)
    gpu = dec._batch(probs)
    dec.use_gpu_search = False
    host = dec._batch(probs)
    for (sg, tg), (sh, th) in zip(gpu, host):
        assert tg == th, (tg, th)
        assert abs(sg - sh) < 2e-3 * max(1.0, abs(sh)), (sg, sh)
    if V <= 300:                                           # the pure-Python oracle on the smaller cases
        for p, (sg, tg) in zip(probs, gpu):
            s_ref, t_ref = obs.decode_lm(p, vocab, olm, 2.2, 4.3, beam, cut, topn)
            assert tg == t_ref and abs(sg - s_ref) < 2e-3 * max(1.0, abs(s_ref)), (sg, s_ref)
    # the scorer is live: the LM-free search of the same candidates differs somewhere
    from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    plain = BeamSearchDecoder(0, 0, beam, cut, topn, vocab)._batch(probs)
    assert any(tp != tg or abs(sp - sg) > 1e-2 for (sp, tp), (sg, tg) in zip(plain, gpu))


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
