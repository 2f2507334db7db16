"""The Silero VAD network on the GPU, behind onnxruntime's session interface.

The reference's VADPredictor constructs ``onnxruntime.InferenceSession(silero_vad.onnx)`` and calls
``session.run(None, {'input', 'h', 'c', 'sr'}) -> (prob, h, c)`` once per 512-sample window
(masr/infer_utils/vad_predictor.py:36, :83-104).  ``SileroVAD`` stands there: it reads the weights out of that same ONNX file
(the reference ships it next to vad_predictor.py; masr_amd/utils/onnx_lite.py parses it, no onnx / onnxruntime package needed),
uploads them once and runs the network with the HIP kernels of csrc/silero.hip (``masr_vad_*``).  Besides ``run`` it offers
``speech_probs``: all windows of a recording in two launches -- what ``get_speech_timestamps`` (:122-129) loops over.

The file is located like the reference does (next to this module), or through ``MASR_SILERO_VAD`` / an installed ``masr``
package; it is third-party weights and is not redistributed with this repository.  There is no CPU path."""
import ctypes as C
import importlib.util
import os

import numpy as np
import torch

from masr_amd import _lib
from masr_amd.utils import onnx_lite

_BLOCKS = (('first_layer.0', True), ('encoder.3.0', True), ('encoder.7.0', False), ('encoder.11.0', True))


def find_model(path=None):
    """the ONNX file: explicit path | $MASR_SILERO_VAD | next to this module (the reference's own default, vad_predictor.py:33-35)
    | next to an installed reference package's vad_predictor.py"""
    cands = [path, os.environ.get('MASR_SILERO_VAD'), os.path.join(os.path.dirname(os.path.realpath(__file__)), 'silero_vad.onnx')]
    try:
        spec = importlib.util.find_spec('masr')
        if spec is not None and spec.submodule_search_locations:
            cands.append(os.path.join(list(spec.submodule_search_locations)[0], 'infer_utils', 'silero_vad.onnx'))
    except Exception:
        pass
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError('Silero VAD model not found: pass path=<silero_vad.onnx> (the file the reference ships in masr/infer_utils/), '
                    'set MASR_SILERO_VAD, or copy it next to masr_amd/infer_utils/silero_vad.py')


def weights_from_onnx(path):
    """{16000: {name: float32 array}, 8000: {...}} under the names csrc/silero.hip expects"""
    g = onnx_lite.load(path)
    top = [n for n in g['nodes'] if n['op'] == 'If']
    if len(top) != 1 or 'onnx::Equal_124' not in g['init'] and not any(n['op'] == 'Equal' for n in g['nodes']):
        raise Exception(f'{path}: not the Silero VAD graph this loader knows (one top-level If on the sample rate)')
    out = {}
    for sr, branch, pre in ((16000, 'then_branch', 'model.'), (8000, 'else_branch', 'model_8k.')):
        br, init = top[0]['attr'][branch], g['init']
        w = {'basis': init[pre + 'feature_extractor.forward_basis_buffer'][:, 0, :],
             'norm_filter': init[pre + 'adaptive_normalization.filter_'].reshape(-1)}
        # the 1 x 1 convolutions between the blocks carry anonymous initializers: they are the Convs of the branch whose weight
        # is not one of the named model tensors, in graph order
        between = [n for n in br['nodes'] if n['op'] == 'Conv' and not n['input'][1].startswith(pre)]
        if len(between) != 4:
            raise Exception(f'{path}: unexpected Silero graph ({len(between)} inter-block convolutions)')
        for k, (name, has_proj) in enumerate(_BLOCKS):
            w[f'b{k}.dw.w'] = init[pre + name + '.dw_conv.0.weight'][:, 0, :]
            w[f'b{k}.dw.b'] = init[pre + name + '.dw_conv.0.bias']
            w[f'b{k}.pw.w'] = init[pre + name + '.pw_conv.0.weight'][:, :, 0]
            w[f'b{k}.pw.b'] = init[pre + name + '.pw_conv.0.bias']
            if has_proj:
                w[f'b{k}.proj.w'] = init[pre + name + '.proj.weight'][:, :, 0]
                w[f'b{k}.proj.b'] = init[pre + name + '.proj.bias']
            w[f'b{k}.out.w'] = init[between[k]['input'][1]][:, :, 0]
            w[f'b{k}.out.b'] = init[between[k]['input'][2]]
            w[f'b{k}.out.stride'] = np.array([between[k]['attr']['strides'][0]], np.float32)
        lstm_if = [n for n in br['nodes'] if n['op'] == 'If' and len(n['output']) == 3]
        if len(lstm_if) != 1:
            raise Exception(f'{path}: unexpected Silero graph (LSTM branch)')
        sub = lstm_if[0]['attr']['then_branch']           # the branch taken when a state is passed in (the reference always does)
        lstms = [n for n in sub['nodes'] if n['op'] == 'LSTM']
        if len(lstms) != 2:
            raise Exception(f'{path}: unexpected Silero graph ({len(lstms)} LSTM layers)')
        for k, n in enumerate(lstms):
            if n['attr'].get('hidden_size') != 64 or n['attr'].get('direction', b'forward') != b'forward':
                raise Exception(f'{path}: unexpected LSTM configuration')
            w[f'lstm{k}.W'] = sub['init'][n['input'][1]][0]
            w[f'lstm{k}.R'] = sub['init'][n['input'][2]][0]
            bias = sub['init'][n['input'][3]][0]
            w[f'lstm{k}.b'] = bias[:256] + bias[256:]
        w['dec.w'] = init[pre + 'decoder.decoder.1.weight'].reshape(-1)
        w['dec.b'] = init[pre + 'decoder.decoder.1.bias'].reshape(-1)
        out[sr] = {k: np.ascontiguousarray(v, np.float32) for k, v in w.items()}
    return out


class SileroVAD:
    """``SileroVAD(path)`` or ``SileroVAD(weights={16000: {...}, 8000: {...}})``; ``run`` has onnxruntime's signature"""

    def __init__(self, path=None, weights=None, device=None):
        self._lib = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.MasrError('SileroVAD needs a GPU (there is no CPU path)')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if weights is None:
            self.path = find_model(path)
            weights = weights_from_onnx(self.path)
        h = C.c_void_p()
        self._check(self._lib.masr_vad_create(self.device.index or 0, C.byref(h)))
        self.h = h
        for sr, tensors in weights.items():
            for name, a in tensors.items():
                a = np.ascontiguousarray(a, np.float32)
                self._check(self._lib.masr_vad_load_tensor(self.h, int(sr), name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
            self._check(self._lib.masr_vad_finalize(self.h, int(sr)))
        self.sample_rates = sorted(int(s) for s in weights)
        self.intra_op_num_threads = self.inter_op_num_threads = 1        # attributes the reference sets on its session (:37-38)

    def _check(self, rc):
        if rc != 0:
            raise _lib.MasrError(self._lib.masr_vad_last_error().decode('utf-8', 'replace'))

    def close(self):
        if getattr(self, 'h', None):
            self._lib.masr_vad_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _forward(self, x, sr, h, c, n_win, window):
        """x device [B, n_win * window]; h, c device [2, B, 64] (updated in place) -> probs device [B, n_win]"""
        B = x.shape[0]
        probs = torch.empty(B, n_win, dtype=torch.float32, device=self.device)
        self._check(self._lib.masr_vad_forward(self.h, int(sr), C.c_void_p(x.data_ptr()), B, n_win, window,
                                               C.c_void_p(h.data_ptr()), C.c_void_p(c.data_ptr()), C.c_void_p(probs.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return probs

    def run(self, _names, feeds):
        """onnxruntime's ``session.run(None, {'input': [B, N], 'h', 'c': [2, B, 64], 'sr'})`` -> [prob [B, 1], h, c].
        N (the window): any length from sr / 31.25 samples (512 at 16 kHz, 256 at 8 kHz -- the ONNX graph's own lower bound) up to
        1536, which covers the sizes the reference class asks for (512 / 1024 / 1536 at 16 kHz, 256 / 512 / 768 at 8 kHz,
        vad_predictor.py:23,31) and every odd length between; longer chunks are refused with that message (masr_vad_forward)."""
        x = torch.as_tensor(np.ascontiguousarray(feeds['input'], np.float32)).to(self.device)
        h = torch.as_tensor(np.ascontiguousarray(feeds['h'], np.float32)).to(self.device)
        c = torch.as_tensor(np.ascontiguousarray(feeds['c'], np.float32)).to(self.device)
        p = self._forward(x, int(feeds['sr']), h, c, 1, x.shape[1])
        return [p.cpu().numpy(), h.cpu().numpy(), c.cpu().numpy()]

    def speech_probs(self, audio, sr, window, h=None, c=None):
        """every window of a recording (the last one zero-padded, vad_predictor.py:126-127) from the given state (default: zeros)
        -> (probs float32 [n_win], h, c) as numpy; one upload, two launches, one download"""
        audio = np.ascontiguousarray(audio, np.float32).reshape(-1)
        n_win = (len(audio) + window - 1) // window
        if n_win == 0:
            return np.zeros(0, np.float32), h, c
        x = torch.zeros(1, n_win * window, dtype=torch.float32, device=self.device)
        x[0, :len(audio)] = torch.from_numpy(audio).to(self.device)
        hd = torch.zeros(2, 1, 64, device=self.device) if h is None else torch.as_tensor(np.asarray(h, np.float32)).to(self.device)
        cd = torch.zeros(2, 1, 64, device=self.device) if c is None else torch.as_tensor(np.asarray(c, np.float32)).to(self.device)
        p = self._forward(x, sr, hd, cd, n_win, window)
        return p[0].cpu().numpy(), hd.cpu().numpy(), cd.cpu().numpy()
