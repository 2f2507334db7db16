"""Phase cycle counters of the GPU prefix search INSIDE the configs[2] bench workload (masr_debug_set key 2 on the device's auxiliary
engine: workgroup 0 of the last search launch -- the second pass's 32 utterances): what tools/beam_profile.py measures on synthetic
posteriors, on the posteriors the random-init Squeezeformer really emits.  Round 6 found the flat line's search there: 85 000 of
130 000 cycles per frame in the extension phase, walking the children lists of a few parents with hundreds of live children.
usage: python tools/beam_bench_phases.py [sharp (0 | 1)]"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from masr_amd import runtime  # noqa: E402

sharp = len(sys.argv) > 1 and sys.argv[1] == '1'
eng = runtime.aux_engine()
eng.lib.masr_debug_set(eng.h, 2, 1)
r = bench.extra_squeezeformer_beam(types.SimpleNamespace(steps=2, warmup=1), 0, 1, 0, sharp=sharp)
torch.cuda.synchronize()
eng.lib.masr_debug_set(eng.h, 2, 0)          # prints the counters
print(f"{'sharpened head' if sharp else 'flat posteriors'}: {r['ms_per_step']} ms per call")
