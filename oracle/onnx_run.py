"""TEST INFRASTRUCTURE ONLY -- evaluates an ONNX graph (parsed by masr_amd.utils.onnx_lite) operator by operator with numpy /
torch-CPU, following the ONNX operator specifications, for the operators the reference's Silero VAD file uses
(masr/infer_utils/silero_vad.onnx, run by onnxruntime in vad_predictor.py:83-104).  onnxruntime is absent from this image:
**parity unpinned vs onnxruntime** -- the file itself is the reference's, the operator semantics are the published ONNX ones."""
import numpy as np
import torch
import torch.nn.functional as F


def _lstm(X, W, R, B, h0, c0, hidden):
    """ONNX LSTM, forward direction, layout 0: X [T, B, I]; W [1, 4H, I], R [1, 4H, H] with gate order i, o, f, c;
    B [1, 8H] = Wb | Rb; f = sigmoid, g = h = tanh; returns Y [T, 1, B, H], Y_h [1, B, H], Y_c [1, B, H]"""
    W, R = W[0], R[0]
    b = B[0][:4 * hidden] + B[0][4 * hidden:] if B is not None else np.zeros(4 * hidden, np.float32)
    h, c = h0[0].astype(np.float32), c0[0].astype(np.float32)
    ys = []
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    for t in range(X.shape[0]):
        g = X[t] @ W.T + h @ R.T + b
        i, o, f, cc = (g[:, k * hidden:(k + 1) * hidden] for k in range(4))
        c = sig(f) * c + sig(i) * np.tanh(cc)
        h = sig(o) * np.tanh(c)
        ys.append(h.astype(np.float32))
    Y = np.stack(ys)[:, None]
    return Y.astype(np.float32), h[None].astype(np.float32), c[None].astype(np.float32)


def run(graph, feeds, outer=None):
    """returns the list of the graph's outputs"""
    env = dict(outer or {})
    env.update(graph['init'])
    env.update(feeds)
    for n in graph['nodes']:
        op, a = n['op'], n['attr']
        x = [env[i] if i else None for i in n['input']]
        if op == 'If':
            br = a['then_branch'] if bool(np.asarray(x[0]).reshape(-1)[0]) else a['else_branch']
            out = run(br, {}, env)
        elif op == 'Equal':
            out = [np.equal(x[0], x[1])]
        elif op == 'Identity':
            out = [x[0]]
        elif op == 'Shape':
            out = [np.array(np.asarray(x[0]).shape[a.get('start', 0):], np.int64)]
        elif op == 'Gather':
            out = [np.take(x[0], x[1], axis=a.get('axis', 0))]
        elif op == 'Unsqueeze':
            y = np.asarray(x[0])
            for ax in sorted(int(v) for v in np.asarray(x[1]).reshape(-1)):
                y = np.expand_dims(y, ax)
            out = [y]
        elif op == 'Squeeze':
            out = [np.squeeze(x[0], axis=tuple(int(v) for v in np.asarray(x[1]).reshape(-1)))]
        elif op == 'Concat':
            out = [np.concatenate([np.asarray(v) for v in x], axis=a['axis'])]
        elif op == 'Reshape':
            out = [np.reshape(x[0], [int(v) for v in x[1]])]
        elif op == 'Cast':
            out = [np.asarray(x[0]).astype({9: np.bool_, 1: np.float32, 7: np.int64}[a['to']])]
        elif op == 'Pad':
            p = [int(v) for v in x[1]]
            r = len(p) // 2
            out = [np.pad(x[0], [(p[i], p[i + r]) for i in range(r)], mode=a.get('mode', b'constant').decode())]
        elif op == 'Conv':
            pads = a.get('pads', [0, 0])
            assert pads[0] == pads[1]
            y = F.conv1d(torch.from_numpy(np.ascontiguousarray(x[0])), torch.from_numpy(x[1]),
                         None if len(x) < 3 or x[2] is None else torch.from_numpy(x[2]), stride=a['strides'][0],
                         padding=pads[0], dilation=a['dilations'][0], groups=a['group'])
            out = [y.numpy()]
        elif op == 'Slice':
            data, starts, ends = x[0], x[1], x[2]
            axes = x[3] if len(x) > 3 and x[3] is not None else np.arange(len(starts))
            steps = x[4] if len(x) > 4 and x[4] is not None else np.ones(len(starts), np.int64)
            sl = [slice(None)] * np.asarray(data).ndim
            for s, e, ax, st in zip(starts, ends, axes, steps):
                s, e, st = int(s), int(e), int(st)
                e = None if (st < 0 and e < -(1 << 62)) else (None if e > (1 << 62) else e)
                sl[int(ax)] = slice(s, e, st)
            out = [np.asarray(data)[tuple(sl)]]
        elif op == 'Pow':
            out = [np.power(x[0], x[1]).astype(np.float32)]
        elif op == 'Sqrt':
            out = [np.sqrt(x[0])]
        elif op == 'Log':
            out = [np.log(x[0])]
        elif op == 'Neg':
            out = [-x[0]]
        elif op == 'Add':
            out = [x[0] + x[1]]
        elif op == 'Mul':
            out = [x[0] * x[1]]
        elif op == 'Relu':
            out = [np.maximum(x[0], 0)]
        elif op == 'Sigmoid':
            out = [(1.0 / (1.0 + np.exp(-x[0].astype(np.float32)))).astype(np.float32)]
        elif op == 'ReduceMean':
            out = [np.mean(x[0], axis=tuple(a['axes']), keepdims=bool(a.get('keepdims', 1))).astype(np.float32)]
        elif op == 'Transpose':
            out = [np.transpose(x[0], a['perm'])]
        elif op == 'ConstantOfShape':
            out = [np.full([int(v) for v in x[0]], a['value'].reshape(-1)[0], a['value'].dtype)]
        elif op == 'LSTM':
            assert a.get('direction', b'forward') == b'forward' and a.get('layout', 0) == 0
            out = list(_lstm(x[0], x[1], x[2], x[3], x[5], x[6], a['hidden_size']))
        else:
            raise NotImplementedError(f'ONNX operator {op}')
        for name, v in zip(n['output'], out):
            if name:
                env[name] = v
    return [env[o] for o in graph['outputs']]
