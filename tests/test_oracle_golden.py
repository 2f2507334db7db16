"""CPU: the oracle restatement against the committed fixtures produced by the REAL reference
(oracle/make_golden.py) and, when /root/reference is present, against the reference live."""
import os

import numpy as np
import pytest
import torch

from oracle import conformer as oc
from oracle import decoders as od
from oracle import fbank as ofb
from oracle import shims, weights
from oracle.make_golden import golden_inputs

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def g(name):
    return np.load(os.path.join(GOLDEN, name))


def test_greedy_toy_known_answer():
    # derived by running the reference greedy_decoder
    vocab = ['<blank>', '<unk>', 'a', 'b', '<space>', '<eos>']
    p = np.array([[.1, 0, .8, .1, 0, 0], [.1, 0, .7, .2, 0, 0], [.9, 0, .05, .05, 0, 0], [.1, 0, .6, .3, 0, 0],
                  [0, 0, .1, .2, .7, 0], [.2, 0, .1, .7, 0, 0]], np.float32)
    assert od.greedy_decoder(p, vocab) == (69.9999988079071, 'aa b')
    s, t, a, b = od.greedy_decoder_chunk(p[:3], vocab)
    assert (s, t) == (75.0, 'a')
    s, t, a, b = od.greedy_decoder_chunk(p[3:], vocab, a, b)
    assert (s, t) == (69.9999988079071, 'aa b')


def test_greedy_against_reference_fixture():
    z = g('greedy.npz')
    v6 = ['<blank>', '<unk>', 'a', 'b', '<space>', '<eos>']
    s, t = od.greedy_decoder(z['probs'], v6)
    assert s == float(z['score']) and t == str(z['text'])
    a = b = None
    for k, (rs, rt) in enumerate(zip(z['chunk_scores'], z['chunk_texts'])):
        sc, tx, a, b = od.greedy_decoder_chunk(z['probs'][8 * k:8 * k + 8], v6, a, b)
        assert sc == float(rs) and tx == str(rt)


def test_audio_segment_and_fbank_fixture():
    z = g('testwav.npz')
    feat, i16 = ofb.featurize_pcm16(z['pcm'])
    assert np.array_equal(i16, z['norm_i16'])          # reference AudioSegment.normalize + to('int16')
    assert feat.shape == (837, 80)
    np.testing.assert_allclose(feat, z['fbank'], atol=1e-5)
    f64, _ = ofb.featurize_pcm16(z['pcm'], dtype=np.float64)
    assert np.abs(f64 - feat).max() < 1e-3


def test_fbank_cross_check_transformers():
    """torchaudio is absent (parity unpinned at the reference level): cross-check the restatement
    against the independent numpy Kaldi mimic shipped with `transformers`."""
    au = pytest.importorskip('transformers.audio_utils')
    z = g('testwav.npz')
    mf = au.mel_filter_bank(257, 80, 20, 8000, 16000, norm=None, mel_scale='kaldi', triangularize_in_mel_space=True)
    ref = au.spectrogram(z['norm_i16'].astype(np.float64), au.window_function(400, 'povey', periodic=False),
                         frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False, preemphasis=0.97,
                         mel_filters=mf, log_mel='log', mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    mine = ofb.kaldi_fbank(z['norm_i16'], 80, np.float64)
    assert ref.shape == mine.shape
    assert np.abs(ref - mine).max() < 1e-3


def test_conformer_full_and_chunk_fixture():
    z = g('conformer_v512.npz')
    feats, lens = golden_inputs()
    sd = weights.conformer_state_dict(0, 512)
    with torch.no_grad():
        enc = oc.encoder_full(sd, feats, lens, -1)
        np.testing.assert_allclose(enc.numpy(), z['enc'], atol=2e-5)
        np.testing.assert_allclose(oc.ctc_probs(sd, enc).numpy(), z['probs'], atol=2e-6)
        np.testing.assert_allclose(oc.encoder_full(sd, feats, lens, 16).numpy(), z['enc16'], atol=2e-5)
        att = torch.zeros(0, 0, 0, 0)
        cnn = torch.zeros(0, 0, 0, 0)
        off = 0
        for k, cur in enumerate(range(0, 331 - 67 + 1, 64)):
            p, att, cnn = oc.get_encoder_out_chunk(sd, feats[:1, cur:cur + 67], off, -16, att, cnn)
            off += p.shape[1]
            np.testing.assert_allclose(p[0].numpy(), z['chunk_probs'][k], atol=2e-6)
        assert list(att.shape) == list(z['att_shape'])
        np.testing.assert_allclose(att[:, :, -16:].numpy(), z['att_tail'], atol=2e-5)
        np.testing.assert_allclose(cnn.numpy(), z['cnn'], atol=2e-5)


def test_conformer_v4233_top4_fixture():
    z = g('conformer_v4233.npz')
    feats, lens = golden_inputs()
    sd = weights.conformer_state_dict(0, 4233)
    with torch.no_grad():
        probs = oc.get_encoder_out(sd, feats, lens)
    top = torch.topk(probs, 4, dim=-1)
    np.testing.assert_allclose(top.values.numpy(), z['top_p'], atol=2e-6)
    assert (top.indices.numpy()[..., 0] == z['top_i'][..., 0]).mean() > 0.999


def test_squeezeformer_fixture():
    from oracle import squeezeformer as osq
    z = g('squeezeformer_v512.npz')
    feats, lens = golden_inputs()
    sd = weights.squeezeformer_state_dict(0, 512)
    with torch.no_grad():
        enc = osq.encoder_full(sd, feats, lens)
        np.testing.assert_allclose(enc.numpy(), z['enc'], atol=2e-5)
        np.testing.assert_allclose(osq.get_encoder_out(sd, feats, lens).numpy(), z['probs'], atol=2e-6)


def test_efficient_conformer_fixture():
    from oracle import efficient_conformer as oe
    z = g('efficient_conformer_v512.npz')
    feats, lens = golden_inputs()
    sd = weights.efficient_conformer_state_dict(0, 512)
    with torch.no_grad():
        enc = oe.encoder_full(sd, feats, lens)
        assert enc.shape[1] == 41                      # T' = 82 -> 41 after the stride layer
        np.testing.assert_allclose(enc.numpy(), z['enc'], atol=2e-5)
        np.testing.assert_allclose(oe.get_encoder_out(sd, feats, lens).numpy(), z['probs'], atol=2e-6)


def test_flop_model_matches_survey():
    # 23.18 GFLOP per 10 s utterance (+ the once-per-batch pos projection)
    per_utt = oc.conformer_flops(998) - 12 * 2 * 256 * 256 * 248
    assert abs(per_utt / 1e9 - 23.18) < 0.02


@pytest.mark.skipif(not shims.reference_available(), reason='/root/reference not present')
@pytest.mark.parametrize('streaming', [False, True])
def test_oracle_batch_norm_conv_module_against_live_reference(streaming):
    """encoder_conf.cnn_module_norm: batch_norm (conformer/convolution.py:60-67,122-125) -- the oracle's conv module against the
    reference ConformerModel built with that option, full-context forward, symmetric and causal conv"""
    import json
    import tempfile
    import yaml
    shims.install()
    from masr.model_utils.conformer.model import ConformerModel
    cfg = yaml.safe_load(open(os.path.join(shims.REFERENCE_ROOT, 'configs', 'conformer.yml'), encoding='utf-8'))
    enc_conf = dict(cfg['encoder_conf'], cnn_module_norm='batch_norm')
    sd = weights.conformer_state_dict(4, 300, cnn_module_norm='batch_norm')
    p = os.path.join(tempfile.mkdtemp(), 'm.json')
    json.dump({'mean': [0.0] * 80, 'istd': [1.0] * 80}, open(p, 'w'))
    m = ConformerModel(input_dim=80, vocab_size=300, mean_istd_path=p, streaming=streaming,
                       encoder_conf=enc_conf, decoder_conf=cfg['decoder_conf'], **cfg['model_conf']).eval()
    missing = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if 'conv_module.norm' in k]
    torch.manual_seed(6)
    x = torch.randn(2, 150, 80) * 3 + 13
    lens = torch.tensor([150, 99])
    with torch.no_grad():
        ref = m.get_encoder_out(x, lens)
        got = oc.ctc_probs(sd, oc.encoder_full(sd, x, lens, streaming=streaming))
        assert (ref - got).abs().max() < 1e-6


@pytest.mark.skipif(not shims.reference_available(), reason='/root/reference not present')
def test_oracle_against_live_reference():
    import json
    import tempfile
    import yaml
    shims.install()
    from masr.model_utils.conformer.model import ConformerModel
    cfg = yaml.safe_load(open(os.path.join(shims.REFERENCE_ROOT, 'configs', 'conformer.yml'), encoding='utf-8'))
    sd = weights.conformer_state_dict(3, 300)
    p = os.path.join(tempfile.mkdtemp(), 'm.json')
    json.dump({'mean': [0.0] * 80, 'istd': [1.0] * 80}, open(p, 'w'))
    m = ConformerModel(input_dim=80, vocab_size=300, mean_istd_path=p, streaming=True,
                       encoder_conf=cfg['encoder_conf'], decoder_conf=cfg['decoder_conf'], **cfg['model_conf']).eval()
    m.load_state_dict(sd, strict=False)
    torch.manual_seed(5)
    x = torch.randn(2, 150, 80) * 3 + 13
    lens = torch.tensor([150, 99])
    with torch.no_grad():
        assert (m.get_encoder_out(x, lens) - oc.get_encoder_out(sd, x, lens)).abs().max() < 1e-6
        # bounded attention history (required_cache_size >= 0, conformer/encoder.py:397-410): the kept cache and the positional
        # offset of its first key must follow the reference step by step
        for req in (0, 16, 24, 40):
            ra, rc = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
            oa, oc_ = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
            off = 0
            for cur in range(0, 150 - 67 + 1 + 64, 64):
                ch = x[:1, cur:cur + 67]
                if ch.shape[1] < 7:
                    break
                pr, ra, rc = m.get_encoder_out_chunk(ch, torch.tensor([off]), torch.tensor([req]), ra, rc)
                po, oa, oc_ = oc.get_encoder_out_chunk(sd, ch, off, req, oa, oc_)
                off += pr.shape[1]
                assert (pr - po).abs().max() < 1e-6 and ra.shape == oa.shape
                assert ra.numel() == 0 or (ra - oa).abs().max() < 1e-6
            assert ra.shape[2] == min(req, off)


def test_deepspeech2_fixture():
    # oracle LSTM stack (cell written out, pack/pad semantics) vs the reference nn.LSTM model, bi- and uni-directional
    from oracle import deepspeech2 as ods
    z = g('deepspeech2_v300.npz')
    feats, lens = golden_inputs()
    with torch.no_grad():
        sd = weights.deepspeech2_state_dict(0, 300, bidirectional=True)
        np.testing.assert_allclose(ods.get_encoder_out(sd, feats, lens).numpy(), z['bi_probs'], atol=5e-6)
        sd = weights.deepspeech2_state_dict(0, 300, bidirectional=False)
        np.testing.assert_allclose(ods.get_encoder_out(sd, feats, lens).numpy(), z['uni_probs'], atol=5e-6)
        h = c = None
        for i, cur in enumerate(range(0, 331 - 67 + 1, 64)):
            x = feats[:1, cur:cur + 67]
            p, xl, h, c = ods.get_encoder_out_chunk(sd, x, torch.tensor([67]), h, c)
            np.testing.assert_allclose(p[0].numpy(), z['chunk_probs'][i], atol=5e-6)
        np.testing.assert_allclose(h.numpy(), z['h'], atol=5e-6)
        np.testing.assert_allclose(c.numpy(), z['c'], atol=5e-6)


def test_metrics_match_reference_definitions():
    # masr/utils/metrics.py:4-29 (Levenshtein package) restated with a numpy DP; KATs by hand + the oracle's plain DP
    from masr_amd.utils.metrics import cer, wer
    assert cer('a b c', 'abd') == pytest.approx(1 / 3)
    assert cer('今天天气', '今天天气') == 0.0
    assert cer('', 'abc') == 1.0
    assert wer('the cat sat', 'the bat sat down') == 0.5
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = ''.join(rng.choice(list('abcd'), rng.integers(0, 12)))
        b = ''.join(rng.choice(list('abcd'), rng.integers(1, 12)))
        assert cer(a, b) == pytest.approx(od.cer(b, a))          # oracle signature is (ref, hyp)


def test_squeezeformer_streaming_build_fixture():
    # squeezeformer.yml as shipped (streaming: True): causal conv module + TimeReductionLayerStream, full-context decode
    from oracle import squeezeformer as osq
    z = g('squeezeformer_streaming_v512.npz')
    feats, lens = golden_inputs()
    sd = weights.squeezeformer_state_dict(0, 512, streaming=True)
    with torch.no_grad():
        enc = osq.encoder_full(sd, feats, lens, causal=True)
        np.testing.assert_allclose(enc.numpy(), z['enc'], atol=2e-5)
        np.testing.assert_allclose(osq.get_encoder_out(sd, feats, lens, causal=True).numpy(), z['probs'], atol=2e-6)


def test_conformer_nonstreaming_build_fixture():
    # conformer.yml with streaming: False -> non-causal conv module (symmetric padding), no chunk masks
    z = g('conformer_nonstreaming_v512.npz')
    feats, lens = golden_inputs()
    sd = weights.conformer_state_dict(0, 512)
    with torch.no_grad():
        enc = oc.encoder_full(sd, feats, lens, streaming=False)
        np.testing.assert_allclose(enc.numpy(), z['enc'], atol=2e-5)
        np.testing.assert_allclose(oc.get_encoder_out(sd, feats, lens, streaming=False).numpy(), z['probs'], atol=2e-6)
        # decoding_chunk_size is ignored by the non-streaming build
        assert torch.equal(oc.encoder_full(sd, feats, lens, decoding_chunk_size=16, streaming=False), enc)


def test_linear_and_mfcc_front_end_fixture():
    """feature_method linear / mfcc (audio_featurizer.py:73-117): the linear restatement against the reference's own function
    (fixture from oracle/make_golden.py --only-features), the MFCC restatement against an independent DCT (scipy), and the
    oracle Conformer with input_dim 161 / 40 against the reference ConformerModel on those features."""
    import scipy.fft
    from oracle import fbank as ofb
    z = g('features.npz')
    s = ofb.pcm16_to_float32(z['pcm'])
    lin = ofb.featurize_samples(s, 'linear')
    assert lin.shape == (199, 161)
    np.testing.assert_allclose(lin, z['linear'], atol=1e-6)
    mf = ofb.featurize_samples(s, 'mfcc')
    assert mf.shape == (198, 40)
    fb64 = ofb.featurize_samples(s, 'fbank').astype(np.float64)
    ref = scipy.fft.dct(fb64, type=2, norm='ortho', axis=1)[:, :40] * ofb.lifter_coeffs(40).astype(np.float64)
    assert np.abs(mf - ref).max() / np.abs(ref).max() < 2e-5
    np.testing.assert_allclose(mf, z['mfcc'], atol=1e-5)
    for method, dim in (('linear', 161), ('mfcc', 40)):
        sd = weights.conformer_state_dict(0, 512, n_mels=dim)
        sd['encoder.global_cmvn.mean'] = torch.from_numpy(z[method + '_cmvn'][0])
        sd['encoder.global_cmvn.istd'] = torch.from_numpy(z[method + '_cmvn'][1])
        x = torch.from_numpy(z[method])[None]
        with torch.no_grad():
            probs = oc.get_encoder_out(sd, x, torch.tensor([x.shape[1]]))[0].numpy()
        np.testing.assert_allclose(probs, z[method + '_probs'], atol=5e-6)


@pytest.mark.skipif(not shims.reference_available(), reason='/root/reference not present')
def test_chunk_masks_and_nonstreaming_efficient_against_live_reference():
    """decoding_chunk_size > 0 in the full forward of the streaming-trained Squeezeformer / Efficient-Conformer (the time
    reduction / stride layer / grouped attention thin the chunk mask out) and the Efficient-Conformer ``streaming: False``
    build (symmetric conv padding, stride layer included): oracle == the live reference encoders"""
    import json
    import tempfile
    import yaml
    from oracle import efficient_conformer as oe, squeezeformer as osq
    shims.install()
    from masr.model_utils.efficient_conformer.model import EfficientConformerModel
    from masr.model_utils.squeezeformer.model import SqueezeformerModel
    tmp = tempfile.mkdtemp()
    torch.manual_seed(7)
    x = torch.randn(2, 203, 80) * 3 + 13
    lens = torch.tensor([203, 150])
    x = x * (torch.arange(203)[None, :, None] < lens[:, None, None])

    def build(cls, name, sd, streaming):
        cfg = yaml.safe_load(open(os.path.join(shims.REFERENCE_ROOT, 'configs', name), encoding='utf-8'))
        p = os.path.join(tmp, f'{name}_{streaming}.json')
        json.dump({'mean': sd['encoder.global_cmvn.mean'].tolist(), 'istd': sd['encoder.global_cmvn.istd'].tolist()}, open(p, 'w'))
        m = cls(input_dim=80, vocab_size=64, mean_istd_path=p, streaming=streaming, encoder_conf=cfg['encoder_conf'],
                decoder_conf=cfg['decoder_conf'], **cfg['model_conf']).eval()
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected
        return m

    with torch.no_grad():
        sd = weights.squeezeformer_state_dict(0, 64, streaming=True)
        m = build(SqueezeformerModel, 'squeezeformer.yml', sd, True)
        for chunk in (16, 8, 4):
            ref, _ = m.encoder(x, lens, chunk, -1)
            assert (ref - osq.encoder_full(sd, x, lens, causal=True, decoding_chunk_size=chunk)).abs().max() < 2e-5, chunk
        sd = weights.efficient_conformer_state_dict(0, 64)
        m = build(EfficientConformerModel, 'efficient_conformer.yml', sd, True)
        for chunk in (16, 6):
            ref, _ = m.encoder(x, lens, chunk, -1)
            assert (ref - oe.encoder_full(sd, x, lens, decoding_chunk_size=chunk)).abs().max() < 2e-5, chunk
        m = build(EfficientConformerModel, 'efficient_conformer.yml', sd, False)
        ref, _ = m.encoder(x, lens, -1, -1)
        assert (ref - oe.encoder_full(sd, x, lens, streaming=False)).abs().max() < 2e-5
        ref16, _ = m.encoder(x, lens, 16, -1)                   # use_dynamic_chunk is off in this build: the argument is ignored
        assert torch.equal(ref, ref16)


@pytest.mark.skipif(not shims.reference_available(), reason='/root/reference not present')
@pytest.mark.parametrize('which', ['squeezeformer', 'efficient_conformer'])
def test_bounded_history_of_the_sibling_encoders_against_live_reference(which):
    """forward_chunk with required_cache_size >= 0 for the Squeezeformer (squeezeformer/encoder.py:292-297,338-347: the half-rate
    layers read every second entry of a cache that is trimmed at next_cache_start // 2 and stored repeat-interleaved) and the
    Efficient-Conformer (efficient_conformer/encoder.py:323-336,365-372): the oracle's probabilities and carried caches follow
    the live reference chunk by chunk, for even, odd and zero cache sizes, the last chunk short."""
    import json
    import tempfile
    import yaml
    from oracle import efficient_conformer as oe, squeezeformer as osq
    shims.install()
    tmp = tempfile.mkdtemp()
    if which == 'squeezeformer':
        from masr.model_utils.squeezeformer.model import SqueezeformerModel as cls
        sd = weights.squeezeformer_state_dict(0, 64, streaming=True)
        name, orc = 'squeezeformer.yml', osq
    else:
        from masr.model_utils.efficient_conformer.model import EfficientConformerModel as cls
        sd = weights.efficient_conformer_state_dict(0, 64)
        name, orc = 'efficient_conformer.yml', oe
    cfg = yaml.safe_load(open(os.path.join(shims.REFERENCE_ROOT, 'configs', name), encoding='utf-8'))
    p = os.path.join(tmp, 'cmvn.json')
    json.dump({'mean': sd['encoder.global_cmvn.mean'].tolist(), 'istd': sd['encoder.global_cmvn.istd'].tolist()}, open(p, 'w'))
    m = cls(input_dim=80, vocab_size=64, mean_istd_path=p, streaming=True, encoder_conf=cfg['encoder_conf'],
            decoder_conf=cfg['decoder_conf'], **cfg['model_conf']).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    torch.manual_seed(9)
    x = torch.randn(1, 64 * 5 + 41, 80) * 3 + 13          # five full chunks and a short one
    with torch.no_grad():
        for req in (0, 16, 24, 41, 40):
            ra, rc = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
            oa, oc_ = torch.zeros(0, 0, 0, 0), torch.zeros(0, 0, 0, 0)
            off = 0
            for cur in range(0, x.shape[1], 64):
                ch = x[:, cur:cur + 67]
                if ch.shape[1] < 7:
                    break
                try:
                    pr, ra, rc = m.get_encoder_out_chunk(ch, torch.tensor([off]), torch.tensor([req]), ra, rc)
                except RuntimeError:
                    # the reference Efficient-Conformer itself fails on a SHORT chunk behind a zero or odd cache size (its stride
                    # layer / torch.cat see mismatched lengths): nothing to follow there
                    assert which == 'efficient_conformer' and ch.shape[1] < 67 and (req == 0 or req % 2 == 1), (which, req, cur)
                    break
                po, oa, oc_ = orc.get_encoder_out_chunk(sd, ch, off, req, oa, oc_)
                off += pr.shape[1]
                assert (pr - po).abs().max() < 2e-5, (which, req, cur)
                assert ra.shape == oa.shape, (which, req, cur, ra.shape, oa.shape)
                assert ra.numel() == 0 or (ra - oa).abs().max() < 2e-5
                assert (rc - oc_).abs().max() < 2e-5
            assert off >= 5 * (16 if which == 'squeezeformer' else 8)
