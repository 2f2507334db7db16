"""EXPLORATORY split-bf16 precision mode (masr_debug_set key 20; csrc/gemm_bf16x3.hip): conv2, the embed projection and the FFN
GEMMs as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the bf16 matrix pipe with fp32 accumulation.  Not the reference's arithmetic and
never the contract path; these tests hold it to the SAME bars as the fp32 path (logits / probabilities <= 1e-3 against the fp32
oracle, identical greedy transcripts) and measure how far it sits from the exact-fp32 kernels."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import needs_experiments

pytestmark = [pytest.mark.gpu, needs_experiments()]


@pytest.fixture()
def x3():
    """switches the mode on for one test and off again whatever happens (the switch is process-wide)"""
    from masr_amd import _lib
    lib = _lib.lib()
    engines = []

    def on(eng, value=3):                 # 3 = every kernel of the mode, the unfused split-bf16 FFN included
        engines.append(eng)
        lib.masr_debug_set(eng.h, 20, value)

    def off(eng):
        lib.masr_debug_set(eng.h, 20, 0)
    yield on, off
    for e in engines:
        lib.masr_debug_set(e.h, 20, 0)


def _op_gemm(eng, a, w, bias, res, act, alpha):
    from masr_amd._lib import check
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty(M, N, device=a.device)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(eng.lib.masr_op_gemm(eng.h, P(a), P(w), P(bias), P(res), P(c), M, N, K, act, C.c_float(alpha),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return c


@pytest.mark.parametrize('M,N,K,act,alpha,with_res', [(7936, 256, 4864, 0, 16.0, False), (7936, 2048, 256, 2, 1.0, False),
                                                      (7936, 256, 2048, 0, 0.5, True), (333, 100, 96, 1, 1.0, True),
                                                      (65, 4233, 256, 0, 1.0, False)])
def test_split_bf16_gemm_against_float64(x3, M, N, K, act, alpha, with_res):
    from masr_amd import runtime
    on, off = x3
    eng = runtime.aux_engine()
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 2.0 + 0.5).cuda()
    w = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda() if with_res else None
    ref = a.double() @ w.double().T + bias.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = ref * torch.sigmoid(ref)
    ref = ref * alpha + (res.double() if with_res else 0.0)
    f32 = _op_gemm(eng, a, w, bias, res, act, alpha)
    on(eng)
    b3 = _op_gemm(eng, a, w, bias, res, act, alpha)
    off(eng)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e32 = float((f32.double() - ref).abs().max()) / scale
    e3 = float((b3.double() - ref).abs().max()) / scale
    assert not torch.equal(f32, b3)                       # the mode really took the launch
    assert e3 < 2e-5, (e3, e32)                           # ~16 mantissa bits per product, fp32 accumulation
    assert e3 < 30 * max(e32, 1e-7), (e3, e32)


def test_conformer_forward_in_split_bf16_mode_keeps_the_parity_bars(x3):
    """32 ragged utterances of <= 10 s: encoder output and probabilities against the fp32 ORACLE within the contract's 1e-3,
    greedy token ids identical to the exact-fp32 kernels' on every utterance"""
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    from oracle import conformer as oc
    on, off = x3
    V = 512
    sd = synthetic.conformer_state_dict(0, V)
    eng = HipEngine(sd, vocab_size=V)
    rng = np.random.default_rng(5)
    lens = rng.integers(40000, 160001, 32).astype(np.int32)
    lens[0] = 160000
    pcm = synthetic.synthetic_pcm(32, 160000, seed=21)
    for i, l in enumerate(lens):
        pcm[i, l:] = 0
    pcm_d, n_d = torch.from_numpy(pcm).cuda(), torch.from_numpy(lens).cuda()
    feats, frames = eng.fbank_batch(pcm_d, n_d)
    enc32 = eng.encode_full(feats, frames, -1).clone()
    tok32, nt32, sc32 = [t.clone() for t in eng.transcribe_batch(pcm_d, n_d)]
    on(eng)
    enc3 = eng.encode_full(feats, frames, -1).clone()
    probs3 = eng.ctc_probs(enc3).clone()
    tok3, nt3, sc3 = [t.clone() for t in eng.transcribe_batch(pcm_d, n_d)]
    off(eng)
    torch.cuda.synchronize()
    assert not torch.equal(enc32, enc3)
    assert float((enc32 - enc3).abs().max()) < 2e-4
    assert torch.equal(nt32, nt3) and torch.equal(tok32, tok3)
    assert float((sc32 - sc3).abs().max()) < 1e-4
    with torch.no_grad():                                   # the fp32 oracle on four of the utterances
        for i in (0, 3, 17, 31):
            f = feats[i:i + 1, :int(frames[i])].cpu()
            ref = oc.encoder_full(sd, f, frames[i:i + 1].cpu().long(), -1)
            t = ref.shape[1]
            single = eng_forward_single(eng, on, off, f)
            assert float((single[0, :t] - ref[0]).abs().max()) < 1e-3
    eng.close()


def eng_forward_single(eng, on, off, feats_cpu):
    on(eng)
    out = eng.encode_full(feats_cpu.cuda(), torch.tensor([feats_cpu.shape[1]], dtype=torch.int32).cuda(), -1).cpu()
    off(eng)
    return out


@pytest.mark.parametrize('kind', ['squeezeformer', 'efficient_conformer'])
def test_sibling_encoders_in_split_bf16_mode(x3, kind):
    from masr_amd.engine import HipEngine
    from masr_amd.utils import synthetic
    on, off = x3
    V = 512
    sd = getattr(synthetic, kind + '_state_dict')(0, V)
    eng = HipEngine(sd, vocab_size=V, use_model=kind, streaming=(kind != 'squeezeformer'))
    pcm = torch.from_numpy(synthetic.synthetic_pcm(8, 120000, seed=4)).cuda()
    n = torch.full((8,), 120000, dtype=torch.int32).cuda()
    feats, frames = eng.fbank_batch(pcm, n)
    a = eng.encode_full(feats, frames, -1).clone()
    on(eng)
    b = eng.encode_full(feats, frames, -1).clone()
    off(eng)
    torch.cuda.synchronize()
    assert not torch.equal(a, b) and float((a - b).abs().max()) < 3e-4
    eng.close()


def test_reference_facade_transcript_identical_in_split_bf16_mode(x3, tmp_path):
    """the facade fixture of tests/test_gpu_identity.py (a): the REFERENCE MASRPredictor's offline transcript of test.wav with
    normalisation off -- identical with the mode on (text ==, score to 1e-3); the streaming path keeps the fp32 kernels"""
    import os
    from tests.test_gpu_identity import GOLDEN, _predictor
    on, off = x3
    pred = _predictor(str(tmp_path), False)
    z = np.load(os.path.join(GOLDEN, 'predictor_nonorm.npz'), allow_pickle=True)
    pcm = np.load(os.path.join(GOLDEN, 'testwav.npz'))['pcm']
    base = pred.predict(audio_data=pcm.copy())
    on(pred.predictor.engine)
    got = pred.predict(audio_data=pcm.copy())
    batch = pred.predict_batch([pcm.copy(), pcm[:70000].copy(), pcm.copy()])
    off(pred.predictor.engine)
    assert got['text'] == base['text'] == str(z['offline_text'])
    assert abs(got['score'] - float(z['offline_score'])) < 1e-3 * max(1.0, abs(float(z['offline_score'])))
    assert batch[0]['text'] == batch[2]['text'] == got['text']
    pred.predictor.engine.close()
