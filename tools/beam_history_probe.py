"""BASELINE configs[2] (sharpened head, GPU prefix search) in a process with a SERVER-LIKE HISTORY: argv[1] torch streams created and
used first, then two engines, a predictor and a stream pool, and only then the beam predictor whose predict_batch call is timed
(argv[1] = 0: a process that starts with the workload); argv[2] = rccl: torch.distributed is initialised (one rank, RCCL) before
anything else, as on every rank of a multi-GPU job.  The HIP runtime deals hardware queues to streams round-robin in creation
order, so each history shifts the deal differently; the library chooses its side streams by probing (engine.hip).  Prints
``RESULT {"ms": ...}``; tests/test_gpu_bench.py compares the histories.
usage: python tools/beam_history_probe.py [n_streams] [rccl]"""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
keep = []
if len(sys.argv) > 2 and sys.argv[2] == 'rccl':          # a rank of a multi-GPU job: torch.distributed (RCCL) comes up before any engine
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29641')
    torch.cuda.set_device(0)
    torch.distributed.init_process_group('nccl', rank=0, world_size=1)
    t = torch.ones(8, device='cuda')
    torch.distributed.all_reduce(t)
    torch.cuda.synchronize()
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
print('RESULT ' + json.dumps({'ms': r['ms_per_step'], 'history_streams': pre, 'rccl_first': len(sys.argv) > 2 and sys.argv[2] == 'rccl'}))
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()
