"""TEST INFRASTRUCTURE ONLY -- the Silero VAD network of the reference (masr/infer_utils/silero_vad.onnx, evaluated by onnxruntime
in vad_predictor.py:83-104) restated as plain numpy: what the graph computes for one window, written out layer by layer.
Pinned against the operator-by-operator evaluation of the reference's own file (oracle/onnx_run.py; tests/test_vad_cpu.py) --
onnxruntime itself is absent: **parity unpinned vs onnxruntime**.

  x [B, N] (N = 512 | 1024 | 1536 at 16 kHz; 256 | 512 | 768 at 8 kHz), state h, c [2, B, 64]
  reflect-pad 96 | STFT as a strided convolution (258 = 129 real + 129 imaginary basis rows of 256 taps, hop 64)
  magnitude [B, 129, F]; spect = ln(1 + 2^20 * magnitude); its per-frame mean over the bins, reflect-padded by 3, smoothed with
  a 7-tap filter and averaged over the window, is subtracted (adaptive normalisation); features = [magnitude | normalised spect]
  four blocks of depthwise (k = 5) -> ReLU -> pointwise (+ projection or identity shortcut) -> ReLU, each followed by a
  1 x 1 convolution (stride 2, 2, 2 | 1 at 16 | 8 kHz, 1) + ReLU: 258 -> 16 -> 32 -> 32 -> 64 channels
  two LSTM layers (hidden 64, ONNX gate order i, o, f, c), ReLU, 1 x 1 convolution to one logit, sigmoid, mean over the steps."""
import numpy as np


def weights_from_graph(graph, sample_rate=16000):
    """the tensors of one of the file's two models (top-level ``If sr == 16000``) under fixed names"""
    br = graph['nodes'][1]['attr']['then_branch' if sample_rate == 16000 else 'else_branch']
    pre = 'model.' if sample_rate == 16000 else 'model_8k.'
    init = graph['init']
    w = {'basis': init[pre + 'feature_extractor.forward_basis_buffer'][:, 0, :], 'norm_filter': init[pre + 'adaptive_normalization.filter_'][0, 0]}
    blocks = [('first_layer.0', True), ('encoder.3.0', True), ('encoder.7.0', False), ('encoder.11.0', True)]
    convs = [n for n in br['nodes'] if n['op'] == 'Conv']
    # the four 1 x 1 convolutions between the blocks carry anonymous initializers: take them from the graph in order
    between = [n for n in convs if n['attr']['kernel_shape'] == [1] and not n['input'][1].startswith(pre)]
    for k, (name, has_proj) in enumerate(blocks):
        w[f'b{k}.dw.w'] = init[pre + name + '.dw_conv.0.weight'][:, 0, :]
        w[f'b{k}.dw.b'] = init[pre + name + '.dw_conv.0.bias']
        w[f'b{k}.pw.w'] = init[pre + name + '.pw_conv.0.weight'][:, :, 0]
        w[f'b{k}.pw.b'] = init[pre + name + '.pw_conv.0.bias']
        if has_proj:
            w[f'b{k}.proj.w'] = init[pre + name + '.proj.weight'][:, :, 0]
            w[f'b{k}.proj.b'] = init[pre + name + '.proj.bias']
        w[f'b{k}.out.w'] = init[between[k]['input'][1]][:, :, 0]
        w[f'b{k}.out.b'] = init[between[k]['input'][2]]
        w[f'b{k}.out.stride'] = between[k]['attr']['strides'][0]
    # LSTM weights live in the branch that runs when a state is passed (the reference always passes one)
    lstm_if = [n for n in br['nodes'] if n['op'] == 'If' and len(n['output']) == 3][0]['attr']['then_branch']
    for k, n in enumerate([n for n in lstm_if['nodes'] if n['op'] == 'LSTM']):
        w[f'lstm{k}.W'] = lstm_if['init'][n['input'][1]][0]
        w[f'lstm{k}.R'] = lstm_if['init'][n['input'][2]][0]
        B = lstm_if['init'][n['input'][3]][0]
        w[f'lstm{k}.b'] = B[:256] + B[256:]
    w['dec.w'] = init[pre + 'decoder.decoder.1.weight'][0, :, 0]
    w['dec.b'] = init[pre + 'decoder.decoder.1.bias']
    return w


def _conv1d(x, w, b, stride=1):
    """x [B, C, T], pointwise w [O, C] -> [B, O, ceil(T / stride)]"""
    return np.einsum('oc,bct->bot', w, x[:, :, ::stride]) + b[None, :, None]


def _dwconv5(x, w, b):
    xp = np.pad(x, ((0, 0), (0, 0), (2, 2)))
    T = x.shape[2]
    return sum(w[None, :, k, None] * xp[:, :, k:k + T] for k in range(5)) + b[None, :, None]


def features(w, x):
    """x [B, N] float32 -> encoder output [B, 64, T]"""
    x = np.asarray(x, np.float32)
    xp = np.pad(x, ((0, 0), (96, 96)), mode='reflect')
    F = (xp.shape[1] - 256) // 64 + 1
    frames = np.stack([xp[:, 64 * f:64 * f + 256] for f in range(F)], axis=2)          # [B, 256, F]
    st = np.einsum('rk,bkf->brf', w['basis'], frames)
    mag = np.sqrt(st[:, :129] ** 2 + st[:, 129:] ** 2)
    spect = np.log(1.0 + 1048576.0 * mag)
    mean = spect.mean(axis=1, keepdims=True)                                          # [B, 1, F]
    mp = np.concatenate([mean[:, :, 1:4][:, :, ::-1], mean, mean[:, :, -4:-1][:, :, ::-1]], axis=2)
    sm = sum(w['norm_filter'][k] * mp[:, :, k:k + F] for k in range(7))
    h = np.concatenate([mag, spect - sm.mean(axis=-1, keepdims=True)], axis=1).astype(np.float32)
    for k in range(4):
        y = np.maximum(_dwconv5(h, w[f'b{k}.dw.w'], w[f'b{k}.dw.b']), 0)
        y = _conv1d(y, w[f'b{k}.pw.w'], w[f'b{k}.pw.b'])
        sc = _conv1d(h, w[f'b{k}.proj.w'], w[f'b{k}.proj.b']) if f'b{k}.proj.w' in w else h
        h = np.maximum(y + sc, 0)
        h = np.maximum(_conv1d(h, w[f'b{k}.out.w'], w[f'b{k}.out.b'], w[f'b{k}.out.stride']), 0).astype(np.float32)
    return h


def forward(w, x, h, c):
    """one call of the network: x [B, N], h, c [2, B, 64] -> (prob [B, 1], h', c')"""
    feat = features(w, x)                                                             # [B, 64, T]
    h, c = np.array(h, np.float32), np.array(c, np.float32)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    acc = np.zeros(feat.shape[0], np.float32)
    for t in range(feat.shape[2]):
        inp = feat[:, :, t]
        for k in range(2):
            g = inp @ w[f'lstm{k}.W'].T + h[k] @ w[f'lstm{k}.R'].T + w[f'lstm{k}.b']
            i, o, f, cc = (g[:, j * 64:(j + 1) * 64] for j in range(4))
            c[k] = sig(f) * c[k] + sig(i) * np.tanh(cc)
            h[k] = sig(o) * np.tanh(c[k])
            inp = h[k]
        acc += sig(np.maximum(inp, 0) @ w['dec.w'] + w['dec.b'][0])
    return (acc / feat.shape[2])[:, None].astype(np.float32), h, c


class OracleSession:
    """stands where the reference constructs ``onnxruntime.InferenceSession(path)`` (vad_predictor.py:36): ``run(None, feeds)``
    with the restated network (``graph=None``) or the operator-by-operator evaluation of the file (``graph=`` the parsed model)"""

    def __init__(self, weights_by_rate, graph=None):
        self.w, self.graph = weights_by_rate, graph
        self.intra_op_num_threads = self.inter_op_num_threads = 1

    def run(self, _, feeds):
        if self.graph is not None:
            from oracle import onnx_run
            return onnx_run.run(self.graph, feeds)
        p, h, c = forward(self.w[int(feeds['sr'])], feeds['input'], feeds['h'], feeds['c'])
        return [p, h, c]
