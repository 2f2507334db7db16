"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the reference DeepSpeech2 inference forward
(``masr/model_utils/deepspeech2/``): GlobalCMVN -> Conv2dSubsampling4Pure (conv.py:5-22) ->
``num_rnn_layers`` x [LSTM (uni-directional when ``streaming`` else bi-directional) over the packed sequence ->
LayerNorm] (encoder.py:36-45,96-129) -> CTC softmax (model.py:64-77).  The LSTM cell is written out
(PyTorch gate order i, f, g, o); pack_padded / pad_packed semantics: a sequence only advances its state while
t < len, padded outputs are zero, the reverse direction starts at each sequence's own last frame.
Pinned against the real reference module (nn.LSTM) by tests/test_oracle_golden.py."""
import torch
import torch.nn.functional as F


def conv_frontend(sd, feats, lens):
    x = (feats - sd['encoder.global_cmvn.mean']) * sd['encoder.global_cmvn.istd']
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd['encoder.conv.conv.0.weight'], sd['encoder.conv.conv.0.bias'], stride=2))
    x = F.relu(F.conv2d(x, sd['encoder.conv.conv.2.weight'], sd['encoder.conv.conv.2.bias'], stride=2))
    x = x.permute(0, 2, 1, 3)
    x = x.reshape(x.shape[0], x.shape[1], -1)
    xl = torch.div(torch.div(lens - 1, 2, rounding_mode='trunc') - 1, 2, rounding_mode='trunc')
    return x, xl


def _lstm_dir(x, xl, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse):
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = F.linear(x, w_ih, b_ih + b_hh)
    h, c = h0.clone(), c0.clone()
    out = torch.zeros(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gx[:, t] + F.linear(h, w_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        act = (t < xl).unsqueeze(1)
        c = torch.where(act, c_new, c)
        h = torch.where(act, h_new, h)
        out[:, t] = torch.where(act, h_new, torch.zeros_like(h_new))
    return out, h, c


def encoder(sd, feats, lens, h0=None, c0=None):
    """returns (encoder_out [B,T',D], lens', h [L,ndir... as the reference: [L, ndir, B, H]], c)"""
    x, xl = conv_frontend(sd, feats, lens)
    L = 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('encoder.rnns.'))
    bi = 'encoder.rnns.0.rnn.weight_ih_l0_reverse' in sd
    B = x.shape[0]
    H = sd['encoder.rnns.0.rnn.weight_hh_l0'].shape[1]
    hs, cs = [], []
    for i in range(L):
        p = f'encoder.rnns.{i}.rnn.'
        outs, hh, cc = [], [], []
        for d, suf in enumerate([''] + (['_reverse'] if bi else [])):
            h_init = torch.zeros(B, H) if h0 is None or h0.numel() == 0 else h0[i, d]
            c_init = torch.zeros(B, H) if c0 is None or c0.numel() == 0 else c0[i, d]
            o, h, c = _lstm_dir(x, xl, sd[p + 'weight_ih_l0' + suf], sd[p + 'weight_hh_l0' + suf],
                                sd[p + 'bias_ih_l0' + suf], sd[p + 'bias_hh_l0' + suf], h_init, c_init, d == 1)
            outs.append(o)
            hh.append(h)
            cc.append(c)
        x = torch.cat(outs, dim=-1)
        x = x[:, :int(xl.max())]                  # pad_packed_sequence trims to the longest sequence
        x = F.layer_norm(x, (x.shape[-1],), sd[f'encoder.rnns.{i}.layer_norm.weight'],
                         sd[f'encoder.rnns.{i}.layer_norm.bias'], 1e-5)
        hs.append(torch.stack(hh))
        cs.append(torch.stack(cc))
    return x, xl, torch.stack(hs), torch.stack(cs)


def get_encoder_out(sd, feats, lens):
    x, _, _, _ = encoder(sd, feats, lens)
    return torch.softmax(F.linear(x, sd['decoder.ctc_lo.weight'], sd['decoder.ctc_lo.bias']), dim=2)


def get_encoder_out_chunk(sd, feats, lens, h0, c0):
    x, xl, h, c = encoder(sd, feats, lens, h0, c0)
    return torch.softmax(F.linear(x, sd['decoder.ctc_lo.weight'], sd['decoder.ctc_lo.bias']), dim=2), xl, h, c
