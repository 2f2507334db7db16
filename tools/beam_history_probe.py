"""BASELINE configs[2] (sharpened head, GPU prefix search) in a process with a SERVER-LIKE HISTORY: argv[1] torch streams created and
used first, then two engines, a predictor and a stream pool, and only then the beam predictor whose predict_batch call is timed
(argv[1] = 0: a process that starts with the workload).  Prints ``RESULT {"ms": ...}``; tests/test_gpu_bench.py compares the two.
usage: python tools/beam_history_probe.py [n_streams]"""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
keep = []
if pre:                                 # a server that touched the GPU first: torch streams with work on them ...
    keep = [torch.cuda.Stream() for _ in range(pre)]
    x = torch.ones(1 << 20, device='cuda')
    for s in keep:
        with torch.cuda.stream(s):
            x = x * 1.0
    torch.cuda.synchronize()
import bench  # noqa: E402

if pre:                                 # ... then two engines, a predictor and a pool
    from masr_amd.serving import StreamPool
    e1 = bench.make_engine('conformer', 0)
    e2 = bench.make_engine('efficient_conformer', 0)
    p1 = bench.facade('conformer', 'ctc_greedy', 0)
    keep += [e1, e2, p1, StreamPool(p1)]
args = types.SimpleNamespace(steps=10, warmup=3)
r = bench.extra_squeezeformer_beam(args, 0, 1, 0, sharp=True)
print('RESULT ' + json.dumps({'ms': r['ms_per_step'], 'history_streams': pre}))
