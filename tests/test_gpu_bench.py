"""bench.py's timed step on the GPU: the pipelined exchange (side stream, double-buffered outputs and host buffers, packed
hypothesis rows) must hand every step's own transcripts to the host -- compared with the engine called directly."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize('force_dist', [False, True])
def test_contract_step_texts_are_the_steps_own(force_dist, monkeypatch):
    import bench
    from masr_amd import parallel
    from masr_amd.utils import synthetic
    made_group = False
    if force_dist:
        monkeypatch.setenv('MASR_FORCE_DIST', '1')
        if not torch.distributed.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29613')
            torch.distributed.init_process_group('nccl', rank=0, world_size=1)
            made_group = True
    torch.cuda.set_device(0)
    eng = bench.make_engine('conformer', 0)
    try:
        vocab = synthetic.synthetic_vocab(bench.VOCAB)
        cs = bench.ContractStep(eng, 0, 1, vocab)
        # a different batch per step: the texts of step k must come from batch k, whatever is in flight
        batches = [torch.from_numpy(synthetic.synthetic_pcm(bench.BATCH, bench.N_SAMPLES, seed=900 + k)).to(eng.device)
                   for k in range(5)]
        def direct(b):
            """the bit-exact route called directly: mean squares -> this host's numpy -> supplied gains -> packed rows"""
            gains = eng.host_gains(b, cs.n, cs.TARGET_DB)
            rows = eng.transcribe_rows(b, cs.n, True, cs.TARGET_DB, gain_in=gains)
            tok, nt, _ = parallel.unpack_hypothesis_rows(rows.cpu().numpy())
            return parallel.tokens_to_text(tok, nt, np.array(vocab, dtype=object)), gains.cpu().numpy()

        want, want_gains = zip(*[direct(b) for b in batches])
        want = list(want)
        torch.cuda.synchronize()
        assert len({tuple(w) for w in want}) == 5
        for announce in (True, False):                # the next batch announced (its gains prepared under this step) or not
            got, gains = [], []
            cs.n_texts = 0
            for k, b in enumerate(batches):
                cs.pcm = b
                cs.step(k, 'full', next_pcm=batches[k + 1] if announce and k + 1 < len(batches) else None)
                gains.append(cs.last_gains.copy())
                if k:
                    got.append(list(cs.texts))        # step k delivers the text of step k - 1
            cs.flush()
            got.append(list(cs.texts))
            assert got == want
            assert all(np.array_equal(g, w) for g, w in zip(gains, want_gains))     # every step ran on ITS batch's gains
            assert cs.n_texts == 5 * bench.BATCH
        # consecutive steps on the engine's two lanes (two workspace sets, two streams): the same texts from the same batches
        got = []
        for k, b in enumerate(batches):
            cs.pcm = b
            cs.step(k, 'lanes', next_pcm=batches[k + 1] if k + 1 < len(batches) else None)
            if k:
                got.append(list(cs.texts))
        cs.flush()
        got.append(list(cs.texts))
        assert got == want
        torch.cuda.synchronize()
        # 'host' mode: the PCM comes over PCIe on a copy stream, one step ahead; every step must still transcribe that batch
        host_want, _ = direct(cs.pcm_host.to(eng.device))
        for k in range(4):
            cs.step(k, 'host')
            if k:
                assert list(cs.texts) == host_want
        cs.flush()
        assert list(cs.texts) == host_want
        for k in range(3):                            # the unsynchronised mode leaves nothing pending
            cs.step(k, 'device')
        torch.cuda.synchronize()
        assert cs.pending is None
    finally:
        eng.close()
        if made_group:
            torch.cuda.synchronize()
            torch.distributed.destroy_process_group()


def test_beam_call_time_does_not_depend_on_what_the_process_created_before():
    """BASELINE configs[2] (sharpened head) through predict_batch with the GPU prefix search: in a process that starts with it, and
    in a process that first created twelve busy torch streams, two engines, a predictor and a stream pool (a server's history).
    The library's side streams (masr_side_stream) are what the searches run on: the call must take the same time (rounds 3-5:
    63 vs 46 ms, then repaired with GPU_MAX_HW_QUEUES=16 from the package's __init__ -- gone now).  GPU_MAX_HW_QUEUES unset."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop('GPU_MAX_HW_QUEUES', None)
    ms = {}
    # the runtime deals hardware queues round-robin in creation order: 12, 13 and 14 earlier streams and an RCCL communicator
    # shift the deal by 0, 1, 2 and 1 of 4 -- the first half of round 6 passed with 12 and lost 30 % behind RCCL (search stream 0
    # on the NULL stream's queue); the side streams are now chosen by probing which queue a candidate landed on (engine.hip)
    def probe(pre):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'beam_history_probe.py')] + [str(v) for v in pre], env=env,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('RESULT ')]
        return json.loads(line[-1][7:])['ms']
    for pre in ((0,), (12,), (13,), (14,), (0, 'rccl')):
        ms[pre] = probe(pre)
        if pre != (0,) and ms[pre] > 1.10 * ms[(0,)]:
            ms[pre] = min(ms[pre], probe(pre))        # (one run in twenty reads 5 - 8 % high on a shared box: a dependence reproduces)
    for pre, v in ms.items():
        assert v <= 1.10 * ms[(0,)], f'cold process {ms[(0,)]} ms per call, after the history {pre}: {v} ms ({ms})'
