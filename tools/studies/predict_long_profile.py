"""predict_long on the 296 s test recording of tools/serve_bench.py, three times in a row: total, VAD and recognition parts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from masr_amd.predict import MASRPredictor  # noqa: E402
from masr_amd.utils import synthetic  # noqa: E402

CONFIG = """
encoder_conf: {output_size: 256, attention_heads: 4, linear_units: 2048, num_blocks: 12, input_layer: conv2d,
  normalize_before: True, cnn_module_kernel: 15, use_cnn_module: True, activation_type: swish, pos_enc_layer_type: rel_pos}
preprocess_conf: {feature_method: fbank, n_mels: 80, n_mfcc: 40, sample_rate: 16000, use_dB_normalization: True, target_dB: -20}
dataset_conf: {dataset_vocab: VOCAB}
use_model: conformer
streaming: True
decoder: ctc_greedy
metrics_type: cer
"""
V = 4233
vpath = '/tmp/serve_bench_vocab.txt'
with open(vpath, 'w', encoding='utf-8') as f:
    for t in synthetic.synthetic_vocab(V):
        f.write(f'{t}\t1\n')
p = MASRPredictor(configs=yaml.safe_load(CONFIG.replace('VOCAB', vpath)), use_gpu=True, state_dict=synthetic.conformer_state_dict(0, V))
here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(0)
speech = np.load(os.path.join(here, '..', 'tests', 'golden', 'testwav.npz'))['pcm']
long_pcm = np.concatenate([np.concatenate([speech, rng.normal(0, 3, 24000).astype(np.int16)]) for i in range(30)])
from masr_amd.infer_utils.silero_vad import SileroVAD  # noqa: E402
from masr_amd.infer_utils.vad_predictor import VADPredictor  # noqa: E402
zz = np.load(os.path.join(here, '..', 'tests', 'golden', 'silero_testwav.npz'))
vad = VADPredictor(session=SileroVAD(weights={16000: {k[4:]: np.asarray(zz[k], np.float32).reshape(zz[k].shape or (1,))
                                                       for k in zz.files if k.startswith('w16.')}}))
p.predict_long(long_pcm[:480000], vad_predictor=vad)
f32 = long_pcm.astype(np.float32) / 32768
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = p.predict_long(long_pcm, batch_size=32, vad_predictor=vad)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    segs = vad.get_speech_timestamps(f32, 16000)
    t2 = time.perf_counter()
    clips = [long_pcm[s['start']:s['end']] for s in segs]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    p.predict_batch(clips, batch_size=32)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f'rep {rep}: predict_long {1e3 * (t1 - t0):.1f} ms | VAD alone {1e3 * (t2 - t1):.1f} ms | predict_batch of the {len(segs)} segments alone {1e3 * (t4 - t3):.1f} ms')
