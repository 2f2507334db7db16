"""GPU: size-regime parity.  The fixtures of test_gpu_parity.py are T = 331 frames (T' <= 82), B = 3; the kernels change
regime with size -- attention key-tile loops, positional-table offsets, the FFN / GEMM split heuristics, >= 64 row blocks,
more workgroups than CUs.  Here the HIP path is compared with the CPU oracle (pinned bit-identical to the reference modules,
tests/test_oracle_golden.py) at BASELINE.json's own sizes:

  * one 20 s utterance (T = 1998 -> T' = 498 keys; Squeezeformer 249 after its time reduction) -- configs[2]'s longest;
  * a ragged batch of 32 utterances up to 10 s (T = 998 -> T' = 248, M = 7936 rows) -- configs[1] / [3];
  * chunk steps with a long history (offset ~ 1000 encoder frames) for one stream and for >= 128 lock-step streams -- configs[4].

Bar: encoder output / probabilities <= 1e-3 (fp32), as north_star states.  The achieved error is printed (-s) and asserted.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def _inputs(B, T, seed, ragged):
    gen = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, 80, generator=gen) * 3 + 13
    lens = torch.full((B,), T, dtype=torch.int64)
    if ragged and B > 1:
        lens[1:] = torch.randint(T // 3, T + 1, (B - 1,), generator=gen)
        lens[B // 2] = T - 1
    feats = feats * (torch.arange(T)[None, :, None] < lens[:, None, None])      # collate_fn zero padding
    return feats, lens


def _families():
    """(name, engine factory, oracle encoder) for the four model families as their YAMLs ship them"""
    from masr_amd.engine import HipEngine
    from oracle import conformer as oc, efficient_conformer as oe, squeezeformer as osq, weights
    V = 512
    return {
        'conformer': (lambda: (HipEngine(sd := weights.conformer_state_dict(0, V), vocab_size=V), sd),
                      lambda sd, f, l: oc.encoder_full(sd, f, l, -1)),
        'squeezeformer': (lambda: (HipEngine(sd := weights.squeezeformer_state_dict(0, V), vocab_size=V, streaming=False,
                                             use_model='squeezeformer'), sd),
                          lambda sd, f, l: osq.encoder_full(sd, f, l)),
        'squeezeformer_streaming': (lambda: (HipEngine(sd := weights.squeezeformer_state_dict(0, V, streaming=True), vocab_size=V,
                                                       streaming=True, use_model='squeezeformer'), sd),
                                    lambda sd, f, l: osq.encoder_full(sd, f, l, causal=True)),
        'efficient_conformer': (lambda: (HipEngine(sd := weights.efficient_conformer_state_dict(0, V), vocab_size=V, streaming=True,
                                                   use_model='efficient_conformer'), sd),
                                lambda sd, f, l: oe.encoder_full(sd, f, l)),
    }


@pytest.fixture(scope='module', params=['conformer', 'squeezeformer', 'squeezeformer_streaming', 'efficient_conformer'])
def family(request):
    make, oracle = _families()[request.param]
    eng, sd = make()
    yield request.param, eng, sd, oracle
    eng.close()


def _compare(name, what, enc, ref, lens_enc=None):
    assert tuple(enc.shape) == tuple(ref.shape), (enc.shape, ref.shape)
    d = (enc - ref).abs()
    if lens_enc is not None:                       # rows behind an utterance's own frames are padding-defined garbage in both
        keep = torch.arange(enc.shape[1])[None, :] < lens_enc[:, None]
        d = d * keep[:, :, None]
    err = d.max().item()
    print(f'{name} {what}: max |enc - oracle| = {err:.3e} over {tuple(enc.shape)}')
    assert err < 1e-3, f'{name} {what}: {err}'


def test_long_utterance_20s_against_oracle(family):
    name, eng, sd, oracle = family
    feats, lens = _inputs(1, 1998, seed=20, ragged=False)
    with torch.no_grad():
        ref = oracle(sd, feats, lens)
    enc = eng.encode_full(dev(feats), dev(lens, torch.int32), -1).cpu()
    _compare(name, '1 x 20 s', enc, ref)


def test_batch32_10s_against_oracle(family):
    name, eng, sd, oracle = family
    feats, lens = _inputs(32, 998, seed=32, ragged=True)
    with torch.no_grad():
        ref = oracle(sd, feats, lens)
    enc = eng.encode_full(dev(feats), dev(lens, torch.int32), -1).cpu()
    _compare(name, '32 x <= 10 s (ragged)', enc, ref, eng.enc_frames(lens))
    # and the whole device path at that size: CTC head + collapse agree with the oracle's argmax path where the oracle's
    # top-2 margin is above the numerical noise
    probs = torch.softmax(torch.nn.functional.linear(ref, sd['ctc.ctc_lo.weight'], sd['ctc.ctc_lo.bias']), dim=-1)
    idx, mp = eng.ctc_greedy_frames(dev(enc))
    top2 = probs.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2e-3
    keep = torch.arange(enc.shape[1])[None, :] < eng.enc_frames(lens)[:, None]
    agree = (idx.cpu() == probs.argmax(-1))[safe & keep]
    print(f'{name}: {int((safe & keep).sum())} decided frames of {int(keep.sum())}, argmax agreement {agree.float().mean():.6f}')
    assert agree.all()


def test_efficient_conformer_pass_of_64_equals_its_two_halves():
    """BASELINE configs[3] at N <= 4: device passes of 64 x 10 s (behind the stride layer 64 utterances are the 248 row blocks
    that fill the chip; 32 are 124).  Size-independent property: an utterance's encoder output does not depend on its batch
    mates -- the pass of 64 equals the two passes of 32 (which test_batch32_10s_against_oracle holds against the oracle); the
    half-rate layers of the two run on different kernels (full row-block launches vs the d_ff-split ones), so the bar is the
    path's 1e-3, measured ~1e-5."""
    make, _ = _families()['efficient_conformer']
    eng, sd = make()
    try:
        feats, lens = _inputs(64, 998, seed=64, ragged=True)
        # equal padded length per half (the pad mask keeps one key past an utterance's frames: only the SAME padded length
        # gives the same function), so the halves see the same [*, 998, 80] inputs
        enc64 = eng.encode_full(dev(feats), dev(lens, torch.int32), -1).cpu()
        enc32 = torch.cat([eng.encode_full(dev(feats[h:h + 32]), dev(lens[h:h + 32], torch.int32), -1).cpu() for h in (0, 32)])
        n = eng.enc_frames(lens)
        keep = (torch.arange(enc64.shape[1])[None, :] < n[:, None])[:, :, None]
        err = ((enc64 - enc32).abs() * keep).max().item()
        print(f'efficient_conformer 64 x <= 10 s vs 2 x 32: max diff {err:.3e}')
        assert err < 1e-3
        tok64 = eng.ctc_greedy_frames(dev(enc64))[0].cpu()
        tok32 = eng.ctc_greedy_frames(dev(enc32))[0].cpu()
        same = ((tok64 == tok32) | ~keep[:, :, 0]).float().mean().item()
        assert same > 0.999, same
    finally:
        eng.close()


def test_deepspeech2_sizes_against_oracle():
    """bi-directional DeepSpeech2: one 20 s utterance (498 LSTM steps on the per-unit kernel) and 32 ragged utterances of up
    to 3 s (the MFMA recurrence for 5..32 sequences)"""
    from masr_amd.engine import HipEngine
    from oracle import deepspeech2 as ods, weights
    sd = weights.deepspeech2_state_dict(0, 300, bidirectional=True)
    eng = HipEngine(sd, encoder_conf={'num_rnn_layers': 5, 'rnn_size': 1024}, streaming=False, use_model='deepspeech2')
    try:
        for B, T, what in ((1, 1998, '1 x 20 s'), (32, 298, '32 x <= 3 s (ragged)')):
            feats, lens = _inputs(B, T, seed=B, ragged=True)
            with torch.no_grad():
                ref = ods.get_encoder_out(sd, feats, lens)
            probs = eng.ctc_probs(eng.encode_full(dev(feats), dev(lens, torch.int32))).cpu()[:, :ref.shape[1]]
            n_enc = ((lens - 1) // 2 - 1) // 2
            keep = torch.arange(ref.shape[1])[None, :] < n_enc[:, None]
            err = ((probs - ref).abs() * keep[:, :, None]).max().item()
            print(f'deepspeech2 {what}: max |probs - oracle| = {err:.3e}')
            assert err < 1e-3
    finally:
        eng.close()


_LONG = {}


def _long_history_reference():
    """the oracle's 62-step chunk run of the two inputs (CPU, the slow part): computed once for the three stream counts"""
    if not _LONG:
        from oracle import conformer as oc, weights
        sd = weights.conformer_state_dict(0, 512)
        steps, win, stride = 62, 67, 64
        gen = torch.Generator().manual_seed(5)
        feats = torch.randn(2, stride * (steps - 1) + win, 80, generator=gen) * 3 + 13
        ref = []
        with torch.no_grad():
            for u in range(2):
                att, cnn, off, tail = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0), 0, []
                for k in range(steps):
                    p, att, cnn = oc.get_encoder_out_chunk(sd, feats[u:u + 1, k * stride:k * stride + win], off, -1, att, cnn)
                    off += p.shape[1]
                    if k >= steps - 3:
                        tail.append(p[0])
                ref.append(torch.stack(tail))
        assert off == 16 * steps
        _LONG.update(sd=sd, feats=feats, ref=ref, steps=steps, win=win, stride=stride)
    return _LONG


@pytest.mark.parametrize('n_streams', [1, 130, 260])
def test_chunk_steps_with_long_history_against_oracle(n_streams):
    """62 chunk steps of 16 encoder frames: the last ones attend over ~1000 cached keys (positional table offsets ~ 1000,
    many key tiles, cache appends far into the stream's buffer).  260 lock-step streams take the throughput kernels, 130 the split ones, a single
    stream the latency-cut ones; streams alternate between two inputs and every checked stream must match ITS oracle run."""
    from masr_amd.engine import HipEngine
    L = _long_history_reference()
    sd, feats, ref, steps, win, stride = L['sd'], L['feats'], L['ref'], L['steps'], L['win'], L['stride']
    eng = HipEngine(sd, vocab_size=512)
    check_from = steps - 3
    try:
        sids = [eng.stream_open(16 * steps + 16) for _ in range(n_streams)]
        x = dev(feats[torch.arange(n_streams) % 2])
        worst = 0.0
        for k in range(steps):
            probs, _, _ = eng.encode_chunk(sids, x[:, k * stride:k * stride + win].contiguous())
            if k >= check_from:
                for s in sorted({0, min(1, n_streams - 1), n_streams - 1}):
                    worst = max(worst, (probs[s].cpu() - ref[s % 2][k - check_from]).abs().max().item())
        print(f'{n_streams} stream(s), offset {eng.stream_offset(sids[0])}: max |probs - oracle| over the last 3 chunk steps = {worst:.3e}')
        assert eng.stream_offset(sids[0]) == 16 * steps and worst < 1e-3
    finally:
        eng.close()


@pytest.mark.parametrize('required', [0, 16, 40])
def test_bounded_attention_history_against_oracle(required):
    """forward_chunk with required_cache_size >= 0 (conformer/encoder.py:397-410; the oracle's handling of it is pinned against
    the live reference in tests/test_oracle_golden.py): probabilities of every chunk step and the exported cache"""
    from masr_amd.engine import HipEngine
    from oracle import conformer as oc, weights
    V = 512
    sd = weights.conformer_state_dict(0, V)
    eng = HipEngine(sd, vocab_size=V)
    gen = torch.Generator().manual_seed(6)
    feats = torch.randn(2, 64 * 5 + 67, 80, generator=gen) * 3 + 13
    try:
        sids = [eng.stream_open(200), eng.stream_open(200)]
        eng.stream_set_history(sids[0], required)               # stream 1 keeps all history: lock-step streams may differ
        att0, cnn0 = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
        att1, cnn1 = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
        off = 0
        for cur in range(0, feats.shape[1] - 67 + 1, 64):
            with torch.no_grad():
                p0, att0, cnn0 = oc.get_encoder_out_chunk(sd, feats[:1, cur:cur + 67], off, required, att0, cnn0)
                p1, att1, cnn1 = oc.get_encoder_out_chunk(sd, feats[1:, cur:cur + 67], off, -1, att1, cnn1)
            off += p0.shape[1]
            probs, _, _ = eng.encode_chunk(sids, dev(feats[:, cur:cur + 67]))
            assert (probs[0].cpu() - p0[0]).abs().max() < 1e-3 and (probs[1].cpu() - p1[0]).abs().max() < 1e-3
        att, cnn = eng.stream_export_cache(sids[0])
        assert tuple(att.shape) == tuple(att0.shape) and att0.shape[2] == min(required, off)
        if att0.numel():
            assert (att.cpu() - att0).abs().max() < 1e-3
        assert (cnn.cpu() - cnn0).abs().max() < 1e-3
        # the facade-level runner takes the reference's argument (inference_predictor.py:80-94)
    finally:
        eng.close()


@pytest.mark.parametrize('which', ['squeezeformer', 'efficient_conformer'])
@pytest.mark.parametrize('required', [0, 16, 24, 41])
def test_bounded_attention_history_of_the_sibling_encoders_against_oracle(which, required):
    """forward_chunk with required_cache_size >= 0 for the Squeezeformer and the Efficient-Conformer (squeezeformer/encoder.py:
    292-297,338-347; efficient_conformer/encoder.py:323-336,365-381; the oracle's handling is pinned against the live reference
    in tests/test_oracle_golden.py): probabilities of every chunk step and the exported caches -- the half-rate layers' windows
    follow the reference's next_cache_start // 2 trimming of the repeat-interleaved cache (even, odd and zero sizes); a second
    stream of the same lock-step call keeps all history."""
    from masr_amd.engine import HipEngine
    from oracle import efficient_conformer as oe, squeezeformer as osq, weights
    V = 64
    if which == 'squeezeformer':
        sd, orc = weights.squeezeformer_state_dict(0, V, streaming=True), osq
    else:
        sd, orc = weights.efficient_conformer_state_dict(0, V), oe
    eng = HipEngine(sd, vocab_size=V, use_model=which, streaming=True)
    gen = torch.Generator().manual_seed(8)
    feats = torch.randn(2, 64 * 5 + 67, 80, generator=gen) * 3 + 13
    try:
        sids = [eng.stream_open(200), eng.stream_open(200)]
        eng.stream_set_history(sids[0], required)
        att0, cnn0 = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
        att1, cnn1 = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
        off, worst = 0, 0.0
        for cur in range(0, feats.shape[1] - 67 + 1, 64):
            with torch.no_grad():
                p0, att0, cnn0 = orc.get_encoder_out_chunk(sd, feats[:1, cur:cur + 67], off, required, att0, cnn0)
                p1, att1, cnn1 = orc.get_encoder_out_chunk(sd, feats[1:, cur:cur + 67], off, -1, att1, cnn1)
            off += p0.shape[1]
            probs, _, _ = eng.encode_chunk(sids, dev(feats[:, cur:cur + 67]))
            worst = max(worst, (probs[0].cpu() - p0[0]).abs().max().item(), (probs[1].cpu() - p1[0]).abs().max().item())
            assert worst < 1e-3, (cur, worst)
            att, cnn = eng.stream_export_cache(sids[0])
            assert tuple(att.shape) == tuple(att0.shape), (cur, att.shape, att0.shape)
            if att0.numel():
                assert (att.cpu() - att0).abs().max() < 1e-3, cur
            assert (cnn.cpu() - cnn0).abs().max() < 1e-3
        att, cnn = eng.stream_export_cache(sids[1])                     # the keep-all stream of the same calls
        assert tuple(att.shape) == tuple(att1.shape) and (att.cpu() - att1).abs().max() < 1e-3
        print(f'{which}, required_cache_size {required}: max |probs - oracle| = {worst:.2e}, kept cache {tuple(att0.shape)}')
    finally:
        eng.close()


def test_bounded_history_is_rejected_where_there_is_no_attention_cache():
    from masr_amd._lib import MasrError
    from masr_amd.engine import HipEngine
    from oracle import weights
    eng = HipEngine(weights.deepspeech2_state_dict(0, 64, bidirectional=False), vocab_size=64, streaming=True, use_model='deepspeech2')
    try:
        sid = eng.stream_open(100)
        with pytest.raises(MasrError, match='required_cache_size'):
            eng.stream_set_history(sid, 16)
        eng.stream_set_history(sid, -1)
    finally:
        eng.close()


@pytest.mark.parametrize('which,chunk', [('squeezeformer_streaming', 16), ('squeezeformer_streaming', 4), ('efficient_conformer', 16),
                                         ('efficient_conformer', 6), ('efficient_nonstreaming', -1), ('squeezeformer', 16)])
def test_chunk_masked_full_forward_and_nonstreaming_efficient_against_oracle(which, chunk):
    """decoding_chunk_size > 0 in ``masr_encode_full`` for the Squeezeformer / Efficient-Conformer (the chunk mask is thinned
    by the time reduction, the stride layer and the grouped attention) and the Efficient-Conformer ``streaming: False`` build;
    the oracle branches used here are pinned against the live reference (tests/test_oracle_golden.py).  A non-streaming build
    ignores the argument like the reference (use_dynamic_chunk off)."""
    from masr_amd.engine import HipEngine
    from oracle import efficient_conformer as oe, squeezeformer as osq, weights
    V = 64
    if which.startswith('squeezeformer'):
        streaming = which.endswith('streaming')
        sd = weights.squeezeformer_state_dict(0, V, streaming=streaming)
        eng = HipEngine(sd, vocab_size=V, streaming=streaming, use_model='squeezeformer')
        oracle = lambda f, l: osq.encoder_full(sd, f, l, causal=streaming, decoding_chunk_size=chunk)
    else:
        streaming = which == 'efficient_conformer'
        sd = weights.efficient_conformer_state_dict(0, V)
        eng = HipEngine(sd, vocab_size=V, streaming=streaming, use_model='efficient_conformer')
        oracle = lambda f, l: oe.encoder_full(sd, f, l, streaming=streaming, decoding_chunk_size=chunk)
    try:
        for T in (331, 203):
            feats, lens = _inputs(4, T, seed=T + chunk, ragged=True)
            with torch.no_grad():
                ref = oracle(feats, lens)
            enc = eng.encode_full(dev(feats), dev(lens, torch.int32), chunk).cpu()
            _compare(which, f'chunk {chunk}, T = {T}', enc, ref, eng.enc_frames(lens))
        if chunk > 0 and streaming:                  # the mask is live: full-context output differs
            full = eng.encode_full(dev(feats), dev(lens, torch.int32), -1).cpu()
            assert (full - enc).abs().max() > 1e-3
    finally:
        eng.close()
