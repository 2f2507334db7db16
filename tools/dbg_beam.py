import os, numpy as np, torch
from masr_amd.predict import MASRPredictor
from masr_amd.utils import synthetic
from oracle import beam_search as obs, fbank as ofb, squeezeformer as osq
V = 300
vocab = synthetic.synthetic_vocab(V)
vpath = '/tmp/vocabulary.txt'
with open(vpath, 'w', encoding='utf-8') as f:
    for t in vocab:
        f.write(f'{t}\t1\n')
cfg = {'encoder_conf': {'encoder_dim': 256, 'output_size': 256, 'attention_heads': 4, 'num_blocks': 12, 'reduce_idx': 5,
                        'recover_idx': 11, 'feed_forward_expansion_factor': 8, 'cnn_module_kernel': 31},
       'preprocess_conf': {'feature_method': 'fbank', 'n_mels': 80, 'n_mfcc': 40, 'sample_rate': 16000,
                           'use_dB_normalization': True, 'target_dB': -20},
       'ctc_beam_search_decoder_conf': {'alpha': 2.2, 'beta': 4.3, 'beam_size': 10, 'num_processes': 4,
                                        'cutoff_prob': 0.99, 'cutoff_top_n': 40, 'language_model_path': 'lm/absent.klm'},
       'dataset_conf': {'dataset_vocab': vpath}, 'use_model': 'squeezeformer', 'streaming': False,
       'decoder': 'ctc_beam_search', 'metrics_type': 'cer'}
sd = synthetic.squeezeformer_state_dict(0, V)
pred = MASRPredictor(configs=cfg, use_gpu=True, state_dict=sd)
pcm = np.load('tests/golden/testwav.npz')['pcm'][:48000]
r1 = pred.predict(audio_data=pcm.copy())
pred.beam_search_decoder.use_gpu_search = False
r2 = pred.predict(audio_data=pcm.copy())
print('gpu ', r1)
print('host', r2)
feat, _ = ofb.featurize_pcm16(pcm)
with torch.no_grad():
    probs = osq.get_encoder_out(sd, torch.from_numpy(feat)[None], torch.tensor([feat.shape[0]]))[0].numpy()
print('orac', obs.decode(probs, vocab, 10, 0.99, 40))
# same oracle probabilities through both searches
dec = pred.beam_search_decoder
dec.use_gpu_search = True
print('gpu  on oracle probs', dec._batch([probs]))
dec.use_gpu_search = False
print('host on oracle probs', dec._batch([probs]))
