"""Measurement for the SURVEY 8(f) rows (not part of the library): (1) what dynamic batching buys -- 64 ten-second requests
recognised one `predict` at a time (the reference server's behaviour, infer_server.py:63) vs through the EngineWorker;
(2) 16 websocket-style sessions stepped by the worker vs one `predict_stream` after the other; (3) the feature front-ends
(fbank / mfcc / linear) on 32 x 10 s; (4) predict_long on a 5-minute recording."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import yaml
from masr_amd.predict import MASRPredictor
from masr_amd.server import EngineWorker
from masr_amd.serving import StreamPool
from masr_amd.utils import synthetic

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
out = {}
clips = [c for c in synthetic.synthetic_pcm(64, 160000, seed=3)]
audio_s = 64 * 10.0

for c in clips[:2]:
    p.predict(audio_data=c)
t0 = time.perf_counter()
for c in clips:
    p.predict(audio_data=c)
torch.cuda.synchronize()
out['offline_one_by_one_audio_s_per_s'] = round(audio_s / (time.perf_counter() - t0), 1)

w = EngineWorker(p, StreamPool(p, max_frames_out=300), max_batch=32, max_wait_ms=5.0)
for _ in range(3):                                             # warm-up: the staging buffers and the pinned result / mean-square rings (four slots) exist afterwards
    [f.result() for f in [w.recognize(c) for c in clips]]
bursts = []
for _ in range(7):                                              # a burst of 64 requests is ~15 ms: seven of them, the median reported
    t0 = time.perf_counter()
    futs = [w.recognize(c) for c in clips]
    [f.result() for f in futs]
    bursts.append(round(audio_s / (time.perf_counter() - t0), 1))
out['offline_engine_worker_audio_s_per_s'] = float(np.median(bursts))
out['offline_engine_worker_bursts'] = bursts
out['worker_batches'] = dict(w.stats)

# streaming: 16 sessions, 0.5 s chunks (8000 samples), 10 s each
chunks = [[c[i:i + 8000].astype('<i2').tobytes() for i in range(0, 160000, 8000)] for c in clips[:16]]
t0 = time.perf_counter()
for ch in chunks:
    p.reset_stream()
    for k, b in enumerate(ch):
        p.predict_stream(audio_data=b, is_end=k == len(ch) - 1)
out['stream_one_session_at_a_time_audio_s_per_s'] = round(160.0 / (time.perf_counter() - t0), 1)
p.reset_stream()
hs = [w.stream_open().result() for _ in range(16)]
lat = []
t0 = time.perf_counter()
for k in range(20):
    t1 = time.perf_counter()
    fs = [w.stream_feed(h, chunks[i][k], k == 19) for i, h in enumerate(hs)]
    [f.result() for f in fs]
    lat.append(time.perf_counter() - t1)
out['stream_16_sessions_worker_audio_s_per_s'] = round(160.0 / (time.perf_counter() - t0), 1)
out['stream_16_sessions_tick_ms_p50'] = round(float(np.percentile(lat, 50)) * 1e3, 2)
[w.stream_close(h).result() for h in hs]

# feature front-ends, 32 x 10 s resident in HBM
eng = p.predictor.engine
pcm = torch.from_numpy(synthetic.synthetic_pcm(32, 160000, seed=4)).cuda()
n = torch.full((32,), 160000, dtype=torch.int32, device='cuda')
for m in ('fbank', 'mfcc', 'linear'):
    for _ in range(3):
        eng.features_batch(m, pcm, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.features_batch(m, pcm, n)
    torch.cuda.synchronize()
    out[f'features_{m}_ms_per_32x10s'] = round((time.perf_counter() - t0) / 20 * 1e3, 3)

# predict_long: ~5 minutes of speech = the reference's test recording (tests/golden/testwav.npz) 30 times, 1.5 s of faint noise between
rng = np.random.default_rng(0)
speech = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'testwav.npz'))['pcm']
long_pcm = np.concatenate([np.concatenate([speech, rng.normal(0, 3, 24000).astype(np.int16)]) for i in range(30)])
# the Silero network on the GPU with the reference's 16 kHz weights (the test fixture's copy: the ONNX file is not on this box)
from masr_amd.infer_utils.silero_vad import SileroVAD
from masr_amd.infer_utils.vad_predictor import VADPredictor
zz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'silero_testwav.npz'))
vad = VADPredictor(session=SileroVAD(weights={16000: {k[4:]: np.asarray(zz[k], np.float32).reshape(zz[k].shape or (1,))
                                                       for k in zz.files if k.startswith('w16.')}}))
p.predict_long(long_pcm[:480000], vad_predictor=vad)
t0 = time.perf_counter()
res = p.predict_long(long_pcm, batch_size=32, vad_predictor=vad)
out['predict_long_296s_recording_first_call_ms'] = round((time.perf_counter() - t0) * 1e3, 1)     # incl. the workspaces growing to this size
t0 = time.perf_counter()
res = p.predict_long(long_pcm, batch_size=32, vad_predictor=vad)
dt = time.perf_counter() - t0
out['predict_long_296s_recording_ms'] = round(dt * 1e3, 1)
f32 = long_pcm.astype(np.float32) / 32768
t0 = time.perf_counter()
segs = vad.get_speech_timestamps(f32, 16000)
out['silero_vad_296s_recording_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
out['predict_long_segments'] = len(segs)
w.shutdown()
print(json.dumps(out))
