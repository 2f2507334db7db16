"""Phase breakdown (cycles of workgroup 0) of the GPU CTC prefix beam search on flat posteriors (worst case: every
frame keeps cutoff_top_n candidates), without and with the external n-gram scorer.
usage: python tools/beam_profile.py [T] [V] [beam] [lm_order (0 = no LM)] [prune (1)] [lm_cache (1: one scorer probe per distinct context and candidate; 0: per pair)]
       [logit scale (1: flat posteriors, 40 candidates in every frame; 8: ~12; 14: ~3 -- what a trained model gives the search)]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from masr_amd import runtime
from masr_amd.decoders.beam_search_decoder import BeamSearchDecoder
from masr_amd.decoders.lm_scorer import write_synthetic_arpa

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
V = int(sys.argv[2]) if len(sys.argv) > 2 else 4233
beam = int(sys.argv[3]) if len(sys.argv) > 3 else 300
order = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(0)
scale = float(sys.argv[7]) if len(sys.argv) > 7 else 1.0
NB = int(os.environ.get("BEAM_PROFILE_B", "8"))
logits = (rng.normal(0, 1.0, (NB, T, V)) * scale).astype(np.float32)
probs = torch.softmax(torch.from_numpy(logits), -1).cuda()
vocab = ['<blank>', '<unk>', '<space>'] + [chr(0x4e00 + i) for i in range(V - 3)]
kw = {'language_model_path': None}
prune = int(sys.argv[5]) if len(sys.argv) > 5 else 1      # 0: the decoder's min_cutoff rule off (every candidate scored)
if order:
    kw['language_model_path'] = write_synthetic_arpa(os.path.join(tempfile.mkdtemp(), 'lm.arpa'), vocab, order=order, seed=5)
dec = BeamSearchDecoder(2.2 if order else 0, 4.3 if order else 0, beam, 0.99, 40, vocab, **kw)
dec.prune_min_cutoff = bool(prune)
eng = runtime.aux_engine()
cache = int(sys.argv[6]) if len(sys.argv) > 6 else 1
eng.lib.masr_debug_set(eng.h, 32, cache)
eng.lib.masr_debug_set(eng.h, 37, int(os.environ.get('BEAM_PROFILE_NARROW', '1')))      # 0: every frame on the wide step; 3: narrow step on, the wide step's pass skipping off
cand = dec._candidates(probs[0], to_host=False)[2].float().mean().item()
print(f'logit scale {scale}: {cand:.1f} candidates per frame')
ref = dec._batch([probs[i] for i in range(NB)])
if cache and order:                    # the table must not change a score: same transcripts and scores as per-pair probing
    eng.lib.masr_debug_set(eng.h, 32, 0)
    base = dec._batch([probs[i] for i in range(NB)])
    eng.lib.masr_debug_set(eng.h, 32, 1)
    assert base == ref, 'scorer table changed the search result'
    print('identical transcripts and scores with and without the per-frame scorer table')
torch.cuda.synchronize()
eng.lib.masr_debug_set(eng.h, 2, 1)
t0 = time.perf_counter()
dec._batch([probs[i] for i in range(NB)])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'batch of {NB} x {T} frames, LM order {order}, scorer table {cache}: {1e3 * dt:.2f} ms  ({1e6 * dt / T:.1f} us per frame step incl. pruning)')
eng.lib.masr_debug_set(eng.h, 2, 0)
