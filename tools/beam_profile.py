"""Phase breakdown (cycles of workgroup 0) of the GPU CTC prefix beam search on flat posteriors (worst case: every
frame keeps cutoff_top_n candidates).  usage: python tools/beam_profile.py [T] [V] [beam]"""
import sys
import time

import numpy as np
import torch

from masr_amd import runtime
from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
V = int(sys.argv[2]) if len(sys.argv) > 2 else 4233
beam = int(sys.argv[3]) if len(sys.argv) > 3 else 300
rng = np.random.default_rng(0)
logits = rng.normal(0, 1.0, (8, T, V)).astype(np.float32)
probs = torch.softmax(torch.from_numpy(logits), -1).cuda()
dec = BeamSearchDecoder(0, 0, beam, 0.99, 40, ['<blank>'] + [chr(0x4e00 + i) for i in range(V - 1)])
eng = runtime.aux_engine()
dec._batch([probs[i] for i in range(8)])
torch.cuda.synchronize()
eng.lib.masr_debug_set(eng.h, 2, 1)
t0 = time.perf_counter()
dec._batch([probs[i] for i in range(8)])
torch.cuda.synchronize()
print(f'batch of 8 x {T} frames: {1e3 * (time.perf_counter() - t0):.2f} ms  ({1e6 * (time.perf_counter() - t0) / T:.1f} us per frame step incl. pruning)')
eng.lib.masr_debug_set(eng.h, 2, 0)
