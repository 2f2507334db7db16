"""CPU: the on-disk formats either side of the path and the host-side gain arithmetic.

* ``inference.pt``: the REAL reference models (via oracle/shims) exported with ``model.export()`` + ``torch.jit.save`` exactly
  like ``MASRTrainer.export`` (trainer.py:653-697) must come back from ``load_state_dict`` key for key, value for value;
* packed artefact: DeepSpeech2 keeps its ``decoder.ctc_lo.*`` head; the bf16 blob widens back to the rounded weights;
* ``engine.reference_gains`` == what ``AudioSegment.normalize`` multiplies by on this host (bit-exact, random utterances).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import shims, weights

needs_ref = pytest.mark.skipif(not shims.reference_available(), reason='reference checkout not present')


@needs_ref
@pytest.mark.parametrize('family', ['conformer', 'deepspeech2'])
def test_torchscript_export_round_trips_key_for_key(tmp_path, family):
    from masr_amd.infer_utils.inference_predictor import load_state_dict
    from oracle import make_golden
    shims.install()
    if family == 'conformer':
        sd = weights.conformer_state_dict(0, 64)
        model = make_golden.build_reference_conformer(sd, 64, str(tmp_path))[0]
        head = 'ctc.ctc_lo.weight'
    else:
        sd = weights.deepspeech2_state_dict(0, 64, bidirectional=False)
        model = make_golden.build_reference_deepspeech2(sd, 64, True, str(tmp_path))
        model = model[0] if isinstance(model, tuple) else model
        head = 'decoder.ctc_lo.weight'
    path = os.path.join(tmp_path, 'inference.pt')
    torch.jit.save(model.export(), path)                       # trainer.py:684-689
    back = load_state_dict(path)
    want = model.state_dict()
    assert set(back) == set(want) and head in back            # TorchScript keeps every parameter and buffer under its name
    for k, v in want.items():
        assert tuple(back[k].shape) == tuple(v.shape) and torch.equal(back[k].to(v.dtype), v), k
    for k, v in sd.items():                                    # and those are the tensors the engine is keyed by
        if k in back:
            assert torch.equal(back[k].float(), torch.as_tensor(v).float()), k
    # model.pt (plain state_dict, trainer.py:308) takes the other branch of the loader
    mpath = os.path.join(tmp_path, 'model.pt')
    torch.save(model.state_dict(), mpath)
    assert set(load_state_dict(mpath)) == set(want)


def test_packed_deepspeech2_keeps_its_head_and_bf16_blob(tmp_path):
    from masr_amd.infer_utils.inference_predictor import load_state_dict
    from masr_amd.utils import packed, synthetic
    sd = synthetic.deepspeech2_state_dict(0, 50, rnn_size=64, num_rnn_layers=2, bidirectional=True)
    p32, p16 = os.path.join(tmp_path, 'ds2.masr'), os.path.join(tmp_path, 'ds2_bf16.masr')
    n = packed.export_packed(sd, p32)
    assert n == len(sd) and {'decoder.ctc_lo.weight', 'decoder.ctc_lo.bias'} <= set(load_state_dict(p32))
    for k, v in load_state_dict(p32).items():
        assert torch.equal(v, torch.as_tensor(sd[k]).float())
    assert packed.export_packed(sd, p16, dtype='bf16') == n
    assert os.path.getsize(p16) < 0.55 * os.path.getsize(p32) + 4096
    back, _ = packed.load_packed(p16)
    for k, v in back.items():
        want = torch.as_tensor(sd[k]).float().to(torch.bfloat16).float()
        assert v.dtype == torch.float32 and torch.equal(v, want), k
    with pytest.raises(ValueError):
        packed.export_packed(sd, p16, dtype='fp8')


@needs_ref
def test_reference_gains_equal_audio_segment_normalize():
    from masr_amd.engine import reference_gains
    shims.install()
    from masr.data_utils.audio import AudioSegment
    rng = np.random.default_rng(11)
    for k in range(200):
        n = int(rng.integers(400, 40000))
        x = (rng.normal(0, 10 ** rng.uniform(-4, -0.3), n)).astype(np.float32)
        seg = AudioSegment.from_ndarray(x.copy(), 16000)
        ms = np.mean(seg.samples ** 2)
        target = int(rng.integers(-30, -9)) if k % 2 else float(rng.uniform(-30, -10))
        seg.normalize(target_db=target)
        g = reference_gains(np.array([ms], np.float32), target)
        assert g.dtype == np.float32 and np.array_equal(x * g[0], seg.samples), (k, g)
    with pytest.raises(ValueError):                              # silence: gain above max_gain_db (audio.py:300-303)
        reference_gains(np.array([1e-38], np.float32), -20, max_gain_db=300.0)
    assert reference_gains(np.array([0.0], np.float32), -20)[0] == np.float32(10.) ** (np.float32(-20) / np.float32(20.))


def test_committed_bench_line_keeps_the_contract():
    """profiles/r04_bench_line.json -- the line `python bench.py` printed on the GPU box for the final build of the round --
    carries every field of the driver's contract, and its numbers are consistent with each other"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = open(os.path.join(root, 'profiles', 'r04_bench_line.json')).read().strip().splitlines()[-1]
    r = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in r, key
    assert r['n_gpus'] == 1 and r['higher_is_better'] is True and r['scaling'] == 'weak' and r['vs_baseline'] is None
    assert r['dtype'] == 'f32' and 'synthetic' in r['data'] and 'workload' in r['config'] and 'model' not in r['config']
    assert abs(r['value'] - 320.0 / (r['ms_per_step'] * 1e-3)) / r['value'] < 1e-3          # 32 x 10 s per step
    rf = r['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in rf, key
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    assert abs(rf['achieved'] - rf['flops_per_launch'] / (rf['avg_us'] * 1e-6) / 1e12) / rf['achieved'] < 1e-2
    cb = r['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in cb, key
    assert cb['kind'] in ('port', 'reference') and cb['cores'] >= 1
    assert set(r['extra']) >= {'efficient_b256', 'stream128', 'stream16', 'squeezeformer_b64_beam', 'conformer_b32_bf16x3_exploratory'}
    assert r['extra']['conformer_b32_bf16x3_exploratory']['dtype'] == 'bf16x3' and 'value_host_to_host' in r
    assert all(r['extra'][k]['steps'] >= 10 for k in ('efficient_b256', 'stream128', 'stream16', 'squeezeformer_b64_beam'))
    # round 4: the drop-in surface itself and a roofline block per BASELINE config
    assert set(r['extra']) >= {'facade_b32', 'predict_b1', 'squeezeformer_b64_beam_sharp', 'squeezeformer_b64_beam_wordlm_host'}
    for k in ('efficient_b256', 'stream128', 'stream16', 'squeezeformer_b64_beam', 'facade_b32', 'predict_b1'):
        rk = r['extra'][k]['roofline']
        assert rk['bound'] == 'mfma' and abs(rk['frac'] - rk['achieved'] / rk['peak']) < 1e-3 and 0 < rk['frac'] < 1
    assert r['extra']['facade_b32']['transcripts'] == 32 and r['extra']['predict_b1']['latency_ms']['p50'] > 0


def test_batched_reference_gains_equal_the_scalar_expressions():
    """engine.reference_gains evaluates a batch with float32 array operations (+ scalar powers): bit-identical to the
    reference's scalar expressions element by element, digital silence (float64 branch) and several targets included"""
    from masr_amd.engine import reference_gains, reference_gains_scalar
    rng = np.random.default_rng(3)
    ms = (10 ** rng.uniform(-9, 0.5, 100000)).astype(np.float32)
    ms[::997] = 0
    for target in (-20, -20.0, -23.5, -17, -3, -20.1):
        assert np.array_equal(reference_gains(ms, target), reference_gains_scalar(ms, target)), target
    with pytest.raises(ValueError):
        reference_gains(np.array([0.5, 1e-38], np.float32), -20)
