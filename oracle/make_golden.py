"""TEST INFRASTRUCTURE ONLY -- generate ``tests/golden/*.npz`` by running the REAL
reference (``/root/reference/masr``, imported unmodified through ``oracle/shims``)
on CPU.  Run in the build container:  ``python -m oracle.make_golden``.

Fixtures (all inputs are either stored or re-derivable from seeds):
* ``testwav.npz``      int16 PCM of the reference's only fixture ``dataset/test.wav``
                       + reference ``AudioFeaturizer.featurize`` output (fbank itself comes
                       from the oracle restatement: torchaudio is absent -> unpinned).
* ``conformer_v512.npz``  reference ConformerModel (configs/conformer.yml, streaming=True,
                       synthetic weights seed 0, V=512): get_encoder_out probs, encoder output,
                       chunk-16 masked encoder output, and a 5-step get_encoder_out_chunk run.
* ``conformer_v4233.npz`` same model family at V=4233: per-frame top-4 (index, prob).
* ``deepspeech2_v300.npz`` reference DeepSpeech2Model (configs/deepspeech2.yml: 5 x LSTM-1024), V=300, synthetic weights:
                       bi-directional (streaming=False) get_encoder_out on the ragged batch, and the uni-directional
                       (streaming=True) model: get_encoder_out + a 5-chunk get_encoder_out_chunk run with carried (h, c).
* ``predictor_deepspeech2.npz`` BASELINE config 1: reference MASRPredictor(use_gpu=False) on the TorchScript export of the
                       synthetic DeepSpeech2 models (V=4233): predict(test.wav) for bi/uni + every predict_stream partial.
* ``predictor.npz``    reference MASRPredictor(use_gpu=False) on TorchScript export of the
                       synthetic model: predict(test.wav) and every predict_stream partial.
* ``greedy.npz``       reference greedy_decoder / greedy_decoder_chunk outputs on seeded probs.
"""
import json
import os
import tempfile
import wave

import numpy as np
import torch
import yaml

from oracle import shims, weights

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
REF = shims.REFERENCE_ROOT


def build_reference_conformer(sd, vocab_size, tmp):
    from masr.model_utils.conformer.model import ConformerModel
    cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'conformer.yml'), encoding='utf-8'))
    p = os.path.join(tmp, 'mean_istd.json')
    json.dump({'mean': sd['encoder.global_cmvn.mean'].tolist(), 'istd': sd['encoder.global_cmvn.istd'].tolist(),
               'feature_method': 'fbank'}, open(p, 'w'))
    torch.manual_seed(0)
    m = ConformerModel(input_dim=80, vocab_size=vocab_size, mean_istd_path=p, streaming=True,
                       encoder_conf=cfg['encoder_conf'], decoder_conf=cfg['decoder_conf'], **cfg['model_conf'])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith('decoder.') for k in missing)
    return m.eval(), cfg, p


def reference_gain(pcm):
    """what AudioSegment.normalize(-20) multiplied the samples of ``pcm`` by ON THIS HOST (audio.py:287-304: float32 log10 /
    power of this numpy build) and the mean square it started from"""
    from masr.data_utils.audio import AudioSegment
    seg = AudioSegment.from_ndarray(pcm, 16000)
    before = seg.samples.copy()
    ms = np.mean(before ** 2)
    gain = np.float32(10.) ** (np.float32(-20 - 10 * np.log10(ms)) / np.float32(20.))
    seg.normalize(target_db=-20)
    assert np.array_equal(before * gain, seg.samples)
    return {'gain': np.float32(gain), 'mean_square': np.float32(ms)}


def record_facade(pred, pcm, step=8000):
    """predict(pcm) and every predict_stream partial of a reference MASRPredictor, fed ``step``-sample chunks"""
    off = pred.predict(audio_data=pcm.copy())
    texts, scores, valid = [], [], []
    for s in range(0, len(pcm), step):
        r = pred.predict_stream(audio_data=pcm[s:s + step].tobytes(), is_end=(s + step >= len(pcm)))
        valid.append(r is not None and r['text'] is not None)
        texts.append('' if not valid[-1] else r['text'])
        scores.append(0.0 if not valid[-1] else float(r['score']))
    pred.reset_stream()
    return {'offline_text': np.array(off['text']), 'offline_score': np.array(off['score'], np.float64),
            'stream_text': np.array(texts), 'stream_score': np.array(scores, np.float64), 'stream_valid': np.array(valid)}


def nonorm_fixture(tmp):
    """predictor_nonorm.npz: the reference facade with ``use_dB_normalization: False`` (conformer.yml otherwise) on
    dataset/test.wav -- no machine-dependent float32 log10 / power anywhere on the path, so every partial transcript of this
    fixture must be reproduced EXACTLY on any host."""
    from masr.predict import MASRPredictor
    w = wave.open(os.path.join(REF, 'dataset', 'test.wav'))
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16).copy()
    sd = weights.conformer_state_dict(0, 4233)
    m, cfg, mean_istd = build_reference_conformer(sd, 4233, tmp)
    vpath = os.path.join(tmp, 'vocabulary_nn.txt')
    with open(vpath, 'w', encoding='utf-8') as f:
        for t in weights.synthetic_vocab(4233):
            f.write(f'{t}\t1\n')
    mdir = os.path.join(tmp, 'models', 'conformer_nonorm')
    os.makedirs(mdir, exist_ok=True)
    torch.jit.save(m.export(), os.path.join(mdir, 'inference.pt'))
    cfg['dataset_conf']['dataset_vocab'] = vpath
    cfg['dataset_conf']['mean_istd_path'] = mean_istd
    cfg['decoder'] = 'ctc_greedy'
    cfg['preprocess_conf']['use_dB_normalization'] = False
    pred = MASRPredictor(configs=cfg, model_path=os.path.join(mdir, 'inference.pt'), use_gpu=False)
    rec = record_facade(pred, pcm)
    np.savez_compressed(os.path.join(OUT, 'predictor_nonorm.npz'), **rec)
    print('nonorm facade:', rec['offline_text'], rec['offline_score'], rec['stream_text'][-1])


def golden_inputs():
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(3, 331, 80, generator=g) * 3 + 13
    lens = torch.tensor([331, 200, 97])
    feats = feats * (torch.arange(331)[None, :, None] < lens[:, None, None])   # collate_fn zero padding
    return feats, lens


def build_reference_deepspeech2(sd, vocab_size, streaming, tmp):
    from masr.model_utils.deepspeech2.model import DeepSpeech2Model
    cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'deepspeech2.yml'), encoding='utf-8'))
    p = os.path.join(tmp, 'mean_istd_ds2.json')
    json.dump({'mean': sd['encoder.global_cmvn.mean'].tolist(), 'istd': sd['encoder.global_cmvn.istd'].tolist(),
               'feature_method': 'fbank'}, open(p, 'w'))
    m = DeepSpeech2Model(input_dim=80, vocab_size=vocab_size, mean_istd_path=p, streaming=streaming,
                         encoder_conf=cfg['encoder_conf'], decoder_conf=cfg['decoder_conf'])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m.eval()


def efficient_chunk_run(ef, feats):
    """chunked streaming of the Efficient Conformer: five 67-frame windows (stride 64) + a short last one (11 frames)"""
    att = torch.zeros(0, 0, 0, 0)
    cnn = torch.zeros(0, 0, 0, 0)
    off = 0
    chunks = []
    for cur, n in [(c, 67) for c in range(0, 331 - 67 + 1, 64)] + [(320, 11)]:
        r, att, cnn = ef.get_encoder_out_chunk(feats[:1, cur:cur + n], off, -16, att, cnn)
        off += r.shape[1]
        chunks.append(r[0].numpy())
    lay = np.array([0, 3, 4, 11])           # a grouped layer, the stride layer, two half-rate layers (keeps the file small)
    return np.concatenate(chunks), (lay, att.numpy()[lay]), cnn.numpy()[lay]


def conformer_nonstreaming_fixture(tmp):
    """conformer.yml with streaming: False (non-causal conv module, no dynamic chunk masks), ragged batch"""
    from masr.model_utils.conformer.model import ConformerModel
    feats, lens = golden_inputs()
    sd = weights.conformer_state_dict(0, 512)
    cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'conformer.yml'), encoding='utf-8'))
    p = os.path.join(tmp, 'mean_istd_ns.json')
    json.dump({'mean': sd['encoder.global_cmvn.mean'].tolist(), 'istd': sd['encoder.global_cmvn.istd'].tolist(),
               'feature_method': 'fbank'}, open(p, 'w'))
    m = ConformerModel(input_dim=80, vocab_size=512, mean_istd_path=p, streaming=False,
                       encoder_conf=cfg['encoder_conf'], decoder_conf=cfg['decoder_conf'], **cfg['model_conf'])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith('decoder.') for k in missing)
    m.eval()
    enc, _ = m.encoder(feats, lens, -1, -1)
    np.savez_compressed(os.path.join(OUT, 'conformer_nonstreaming_v512.npz'), enc=enc.numpy(),
                        probs=m.get_encoder_out(feats, lens).numpy())


def squeezeformer_streaming_fixture(mean_istd):
    """squeezeformer.yml as shipped (streaming: True -> causal conv module + TimeReductionLayerStream), full-context
    get_encoder_out on the ragged batch"""
    from masr.model_utils.squeezeformer.model import SqueezeformerModel
    feats, lens = golden_inputs()
    cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'squeezeformer.yml'), encoding='utf-8'))
    sd = weights.squeezeformer_state_dict(0, 512, streaming=True)
    m = SqueezeformerModel(input_dim=80, vocab_size=512, mean_istd_path=mean_istd, streaming=True,
                           encoder_conf=cfg['encoder_conf'], decoder_conf=cfg['decoder_conf'], **cfg['model_conf'])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith('decoder.') for k in missing), (missing[:4], unexpected[:4])
    m.eval()
    enc, _ = m.encoder(feats, lens, -1, -1)
    probs = m.get_encoder_out(feats, lens)
    # chunked streaming: five 67-frame windows (stride 64) and a short last one (11 frames -> 2 output frames)
    att = torch.zeros(0, 0, 0, 0)
    cnn = torch.zeros(0, 0, 0, 0)
    off = 0
    chunks = []
    for cur, n in [(c, 67) for c in range(0, 331 - 67 + 1, 64)] + [(320, 11)]:
        r, att, cnn = m.get_encoder_out_chunk(feats[:1, cur:cur + n], off, -16, att, cnn)
        off += r.shape[1]
        chunks.append(r[0].numpy())
    lay = np.array([0, 5, 10, 11])          # caches of a full-rate, two half-rate and the recovery layer (keeps the file small)
    np.savez_compressed(os.path.join(OUT, 'squeezeformer_streaming_v512.npz'), enc=enc.numpy(), probs=probs.numpy(),
                        chunk_probs=np.concatenate(chunks), att_layers=lay, att=att.numpy()[lay], cnn=cnn.numpy()[lay])


def deepspeech2_fixture(tmp):
    feats, lens = golden_inputs()
    out = {}
    sd = weights.deepspeech2_state_dict(0, 300, bidirectional=True)
    m = build_reference_deepspeech2(sd, 300, False, tmp)
    out['bi_probs'] = m.get_encoder_out(feats, lens).numpy()
    sd = weights.deepspeech2_state_dict(0, 300, bidirectional=False)
    m = build_reference_deepspeech2(sd, 300, True, tmp)
    out['uni_probs'] = m.get_encoder_out(feats, lens).numpy()
    h = torch.zeros(0, 0, 0, 0)
    c = torch.zeros(0, 0, 0, 0)
    chunks = []
    for cur in range(0, 331 - 67 + 1, 64):
        x = feats[:1, cur:cur + 67]
        r, _, h, c = m.get_encoder_out_chunk(x, torch.tensor([x.shape[1]]), h, c)
        chunks.append(r[0].numpy())
    out['chunk_probs'] = np.stack(chunks)
    out['h'] = h.numpy()
    out['c'] = c.numpy()
    np.savez_compressed(os.path.join(OUT, 'deepspeech2_v300.npz'), **out)

    # BASELINE config 1: deepspeech2.yml, streaming False, ctc_greedy, B=1 on dataset/test.wav through the reference
    # MASRPredictor(use_gpu=False); plus the streaming (uni-directional) model through predict_stream
    from masr.predict import MASRPredictor
    w = wave.open(os.path.join(REF, 'dataset', 'test.wav'))
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16).copy()
    vocab = weights.synthetic_vocab(4233)
    vpath = os.path.join(tmp, 'vocabulary_ds2.txt')
    with open(vpath, 'w', encoding='utf-8') as f:
        for t in vocab:
            f.write(f'{t}\t1\n')
    res = {}
    for streaming in (False, True):
        sd = weights.deepspeech2_state_dict(0, 4233, bidirectional=not streaming)
        m = build_reference_deepspeech2(sd, 4233, streaming, tmp)
        mdir = os.path.join(tmp, 'models', f'deepspeech2_{streaming}')
        os.makedirs(mdir)
        torch.jit.save(m.export(), os.path.join(mdir, 'inference.pt'))
        cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'deepspeech2.yml'), encoding='utf-8'))
        cfg['dataset_conf']['dataset_vocab'] = vpath
        cfg['dataset_conf']['mean_istd_path'] = os.path.join(tmp, 'mean_istd_ds2.json')
        cfg['decoder'] = 'ctc_greedy'
        cfg['streaming'] = streaming
        pred = MASRPredictor(configs=cfg, model_path=os.path.join(mdir, 'inference.pt'), use_gpu=False)
        r = pred.predict(audio_data=pcm.copy())
        key = 'uni' if streaming else 'bi'
        res[key + '_text'] = np.array(r['text'])
        res[key + '_score'] = np.array(r['score'], np.float64)
        if streaming:
            texts, scores, valid = [], [], []
            for s0 in range(0, len(pcm), 8000):
                q = pred.predict_stream(audio_data=pcm[s0:s0 + 8000].tobytes(), is_end=(s0 + 8000 >= len(pcm)))
                valid.append(q is not None and q['text'] is not None)
                texts.append('' if not valid[-1] else q['text'])
                scores.append(0.0 if not valid[-1] else float(q['score']))
            pred.reset_stream()
            res['stream_text'], res['stream_score'], res['stream_valid'] = np.array(texts), np.array(scores), np.array(valid)
    np.savez_compressed(os.path.join(OUT, 'predictor_deepspeech2.npz'), **res)
    print('deepspeech2 facade:', res['bi_text'], res['bi_score'], '| stream:', res['stream_text'][-1], res['stream_score'][-1])


def features_fixture(tmp):
    """linear / mfcc front-ends: the reference AudioFeaturizer itself on a 2 s clip of dataset/test.wav (linear is the
    reference's own numpy code; mfcc goes through the kaldi shim = oracle restatement), plus the reference ConformerModel with
    input_dim 161 / 40 on those features (get_encoder_out)."""
    from masr.data_utils.audio import AudioSegment
    from masr.data_utils.featurizer.audio_featurizer import AudioFeaturizer
    from masr.model_utils.conformer.model import ConformerModel
    w = wave.open(os.path.join(REF, 'dataset', 'test.wav'))
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16).copy()[24000:56000]
    out = {'pcm': pcm}
    cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'conformer.yml'), encoding='utf-8'))
    for method, dim in (('linear', 161), ('mfcc', 40)):
        seg = AudioSegment(pcm.copy(), 16000)
        fz = AudioFeaturizer(feature_method=method, n_mels=80, n_mfcc=40, use_dB_normalization=True, target_dB=-20)
        feat = np.asarray(fz.featurize(seg))
        assert fz.feature_dim == dim and feat.shape[1] == dim
        out[method] = feat.astype(np.float32)
        sd = weights.conformer_state_dict(0, 512, n_mels=dim)
        # CMVN statistics of the right order of magnitude for this feature type
        sd['encoder.global_cmvn.mean'] = torch.from_numpy(feat.mean(0).astype(np.float32))
        sd['encoder.global_cmvn.istd'] = torch.from_numpy((1.0 / (feat.std(0) + 1e-3)).astype(np.float32))
        p = os.path.join(tmp, f'mean_istd_{method}.json')
        json.dump({'mean': sd['encoder.global_cmvn.mean'].tolist(), 'istd': sd['encoder.global_cmvn.istd'].tolist(),
                   'feature_method': method}, open(p, 'w'))
        m = ConformerModel(input_dim=dim, vocab_size=512, mean_istd_path=p, streaming=True, encoder_conf=cfg['encoder_conf'],
                           decoder_conf=cfg['decoder_conf'], **cfg['model_conf']).eval()
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith('decoder.') for k in missing)
        x = torch.from_numpy(out[method])[None]
        out[method + '_probs'] = m.get_encoder_out(x, torch.tensor([x.shape[1]])).numpy()[0]
        out[method + '_cmvn'] = np.stack([sd['encoder.global_cmvn.mean'].numpy(), sd['encoder.global_cmvn.istd'].numpy()])
    np.savez_compressed(os.path.join(OUT, 'features.npz'), **out)


def main():
    import sys
    shims.install()
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    torch.set_grad_enabled(False)
    if '--only-efficient' in sys.argv:
        from masr.model_utils.efficient_conformer.model import EfficientConformerModel
        feats, lens = golden_inputs()
        ef_cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'efficient_conformer.yml'), encoding='utf-8'))
        ef_sd = weights.efficient_conformer_state_dict(0, 512)
        p = os.path.join(tmp, 'mean_istd_ef.json')
        json.dump({'mean': ef_sd['encoder.global_cmvn.mean'].tolist(), 'istd': ef_sd['encoder.global_cmvn.istd'].tolist(),
                   'feature_method': 'fbank'}, open(p, 'w'))
        ef = EfficientConformerModel(input_dim=80, vocab_size=512, mean_istd_path=p, streaming=True,
                                     encoder_conf=ef_cfg['encoder_conf'], decoder_conf=ef_cfg['decoder_conf'],
                                     **ef_cfg['model_conf']).eval()
        ef.load_state_dict(ef_sd, strict=False)
        ef_enc, _ = ef.encoder(feats, lens, -1, -1)
        ef_chunks, ef_att, ef_cnn = efficient_chunk_run(ef, feats)
        np.savez_compressed(os.path.join(OUT, 'efficient_conformer_v512.npz'), enc=ef_enc.numpy(),
                            probs=ef.get_encoder_out(feats, lens).numpy(), chunk_probs=ef_chunks, att_layers=ef_att[0],
                            att=ef_att[1], cnn=ef_cnn)
        print('efficient conformer fixture written')
        return
    if '--only-conformer-nonstreaming' in sys.argv:
        conformer_nonstreaming_fixture(tmp)
        print('conformer non-streaming fixture written')
        return
    if '--only-squeezeformer-streaming' in sys.argv:
        p = os.path.join(tmp, 'mean_istd_sq.json')
        sd0 = weights.squeezeformer_state_dict(0, 512, streaming=True)
        json.dump({'mean': sd0['encoder.global_cmvn.mean'].tolist(), 'istd': sd0['encoder.global_cmvn.istd'].tolist(),
                   'feature_method': 'fbank'}, open(p, 'w'))
        squeezeformer_streaming_fixture(p)
        print('squeezeformer streaming fixture written')
        return
    if '--only-nonorm' in sys.argv:
        nonorm_fixture(tmp)
        z = dict(np.load(os.path.join(OUT, 'testwav.npz')))
        z.update(reference_gain(z['pcm']))
        np.savez_compressed(os.path.join(OUT, 'testwav.npz'), **z)
        print('nonorm + testwav gain fixtures written', z['gain'], z['mean_square'])
        return
    if '--only-features' in sys.argv:
        features_fixture(tmp)
        print('features fixture written')
        return
    if '--only-deepspeech2' in sys.argv:
        deepspeech2_fixture(tmp)
        print('deepspeech2 fixture written')
        return
    deepspeech2_fixture(tmp)

    # ---- test.wav ---------------------------------------------------------
    w = wave.open(os.path.join(REF, 'dataset', 'test.wav'))
    pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16).copy()
    from masr.data_utils.audio import AudioSegment
    from masr.data_utils.featurizer.audio_featurizer import AudioFeaturizer
    seg = AudioSegment.from_ndarray(pcm, 16000)
    feat = AudioFeaturizer(feature_method='fbank', n_mels=80, sample_rate=16000, use_dB_normalization=True,
                           target_dB=-20).featurize(seg)
    i16 = seg.to('int16')
    np.savez_compressed(os.path.join(OUT, 'testwav.npz'), pcm=pcm, norm_i16=i16, fbank=feat.astype(np.float32),
                        **reference_gain(pcm))

    # ---- conformer V=512 ----------------------------------------------------
    feats, lens = golden_inputs()
    sd = weights.conformer_state_dict(0, 512)
    m, cfg, _ = build_reference_conformer(sd, 512, tmp)
    probs = m.get_encoder_out(feats, lens)
    enc, _ = m.encoder(feats, lens, -1, -1)
    enc16, _ = m.encoder(feats, lens, 16, -1)
    att = torch.zeros(0, 0, 0, 0)
    cnn = torch.zeros(0, 0, 0, 0)
    off = 0
    chunk_probs = []
    for cur in range(0, 331 - 67 + 1, 64):
        r, att, cnn = m.get_encoder_out_chunk(feats[:1, cur:cur + 67], off, -16, att, cnn)
        off += r.shape[1]
        chunk_probs.append(r[0].numpy())
    np.savez_compressed(os.path.join(OUT, 'conformer_v512.npz'), probs=probs.numpy(), enc=enc.numpy(),
                        enc16=enc16.numpy(), chunk_probs=np.stack(chunk_probs), att_tail=att[:, :, -16:].numpy(),
                        att_shape=np.array(att.shape), cnn=cnn.numpy())

    # ---- conformer V=4233: top-4 ---------------------------------------------
    sd = weights.conformer_state_dict(0, 4233)
    m, cfg, mean_istd = build_reference_conformer(sd, 4233, tmp)
    probs = m.get_encoder_out(feats, lens)
    top = torch.topk(probs, 4, dim=-1)
    np.savez_compressed(os.path.join(OUT, 'conformer_v4233.npz'), top_p=top.values.numpy(),
                        top_i=top.indices.numpy().astype(np.int32))

    # ---- squeezeformer (configs/squeezeformer.yml, streaming: False) V=512 -------------------------
    from masr.model_utils.squeezeformer.model import SqueezeformerModel
    sq_cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'squeezeformer.yml'), encoding='utf-8'))
    sq_sd = weights.squeezeformer_state_dict(0, 512)
    sq = SqueezeformerModel(input_dim=80, vocab_size=512, mean_istd_path=mean_istd, streaming=False,
                            encoder_conf=sq_cfg['encoder_conf'], decoder_conf=sq_cfg['decoder_conf'],
                            **sq_cfg['model_conf'])
    missing, unexpected = sq.load_state_dict(sq_sd, strict=False)
    assert not unexpected and all(k.startswith('decoder.') for k in missing)
    sq.eval()
    sq_enc, _ = sq.encoder(feats, lens, -1, -1)
    sq_probs = sq.get_encoder_out(feats, lens)
    np.savez_compressed(os.path.join(OUT, 'squeezeformer_v512.npz'), enc=sq_enc.numpy(), probs=sq_probs.numpy())
    squeezeformer_streaming_fixture(mean_istd)
    conformer_nonstreaming_fixture(tmp)

    # ---- efficient conformer (configs/efficient_conformer.yml, streaming: True) V=512 ------------------
    from masr.model_utils.efficient_conformer.model import EfficientConformerModel
    ef_cfg = yaml.safe_load(open(os.path.join(REF, 'configs', 'efficient_conformer.yml'), encoding='utf-8'))
    ef_sd = weights.efficient_conformer_state_dict(0, 512)
    ef = EfficientConformerModel(input_dim=80, vocab_size=512, mean_istd_path=mean_istd, streaming=True,
                                 encoder_conf=ef_cfg['encoder_conf'], decoder_conf=ef_cfg['decoder_conf'],
                                 **ef_cfg['model_conf'])
    missing, unexpected = ef.load_state_dict(ef_sd, strict=False)
    assert not unexpected and all(k.startswith('decoder.') or 'concat_linear' in k for k in missing)
    ef.eval()
    ef_enc, _ = ef.encoder(feats, lens, -1, -1)
    ef_probs = ef.get_encoder_out(feats, lens)
    ef_chunks, ef_att, ef_cnn = efficient_chunk_run(ef, feats)
    np.savez_compressed(os.path.join(OUT, 'efficient_conformer_v512.npz'), enc=ef_enc.numpy(), probs=ef_probs.numpy(),
                        chunk_probs=ef_chunks, att_layers=ef_att[0], att=ef_att[1], cnn=ef_cnn)

    # ---- MASRPredictor facade on the TorchScript export ---------------------------
    from masr.predict import MASRPredictor
    vocab = weights.synthetic_vocab(4233)
    vpath = os.path.join(tmp, 'vocabulary.txt')
    with open(vpath, 'w', encoding='utf-8') as f:
        for t in vocab:
            f.write(f'{t}\t1\n')
    mdir = os.path.join(tmp, 'models', 'conformer_streaming_fbank')
    os.makedirs(mdir)
    torch.jit.save(m.export(), os.path.join(mdir, 'inference.pt'))
    cfg['dataset_conf']['dataset_vocab'] = vpath
    cfg['dataset_conf']['mean_istd_path'] = mean_istd
    cfg['decoder'] = 'ctc_greedy'
    pred = MASRPredictor(configs=cfg, model_path=os.path.join(mdir, 'inference.pt'), use_gpu=False)
    off_res = pred.predict(audio_data=pcm.copy())
    texts, scores, valid = [], [], []
    n = len(pcm)
    step = 8000
    for s in range(0, n, step):
        chunk = pcm[s:s + step].tobytes()
        r = pred.predict_stream(audio_data=chunk, is_end=(s + step >= n))
        valid.append(r is not None and r['text'] is not None)
        texts.append('' if not valid[-1] else r['text'])
        scores.append(0.0 if not valid[-1] else float(r['score']))
    pred.reset_stream()
    np.savez_compressed(os.path.join(OUT, 'predictor.npz'), offline_text=np.array(off_res['text']),
                        offline_score=np.array(off_res['score'], np.float64), stream_text=np.array(texts),
                        stream_score=np.array(scores, np.float64), stream_valid=np.array(valid))

    # ---- greedy decoders --------------------------------------------------------
    from masr.decoders.ctc_greedy_decoder import greedy_decoder, greedy_decoder_chunk
    rng = np.random.default_rng(7)
    v6 = ['<blank>', '<unk>', 'a', 'b', '<space>', '<eos>']
    p = rng.dirichlet(np.ones(6) * 0.3, size=40).astype(np.float32)
    s, t = greedy_decoder(p, v6)
    a = b = None
    cs, ct = [], []
    for k in range(0, 40, 8):
        sc, tx, a, b = greedy_decoder_chunk(p[k:k + 8], v6, a, b)
        cs.append(sc)
        ct.append(tx)
    np.savez_compressed(os.path.join(OUT, 'greedy.npz'), probs=p, score=np.array(s, np.float64), text=np.array(t),
                        chunk_scores=np.array(cs, np.float64), chunk_texts=np.array(ct))
    print('golden fixtures written to', OUT)
    for f in sorted(os.listdir(OUT)):
        print(' ', f, os.path.getsize(os.path.join(OUT, f)))
    print('offline:', off_res, 'stream:', texts[-1], scores[-1])


if __name__ == '__main__':
    main()
