"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the reference Squeezeformer inference
forward, non-streaming build (``streaming: False`` -> symmetric depthwise conv + BatchNorm,
TimeReductionLayer1D, pad masks only; ``configs/squeezeformer.yml``).  Driven by a ``state_dict``
with the reference key names; paths below are relative to ``masr/model_utils``.  Pinned against the
real reference modules by tests/test_oracle_golden.py (live + committed fixture)."""
import math

import torch
import torch.nn.functional as F

from oracle.conformer import positional_table


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def _ada(sd, p, x):
    """adaptive scale / bias (squeezeformer/attention.py:112-115, positionwise.py:57-58, convolution.py:109-110)."""
    return sd[p + '.ada_scale'] * x + sd[p + '.ada_bias']


def embed(sd, feats):
    """GlobalCMVN + DepthwiseConv2DSubsampling4.forward (squeezeformer/subsampling.py:60-76; dw_stride=False
    => the second conv is a full 256->256 conv).  NB the sqrt(d) scaling is applied BEFORE input_proj."""
    x = (feats - sd['encoder.global_cmvn.mean']) * sd['encoder.global_cmvn.istd']
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd['encoder.embed.pw_conv.weight'], sd['encoder.embed.pw_conv.bias'], stride=2))
    x = F.relu(F.conv2d(x, sd['encoder.embed.dw_conv.weight'], sd['encoder.embed.dw_conv.bias'], stride=2))
    b, c, t, f = x.shape
    x = x.permute(0, 2, 1, 3).contiguous().view(b, t, c * f)
    x = x * math.sqrt(c)
    return F.linear(x, sd['encoder.embed.input_proj.0.weight'], sd['encoder.embed.input_proj.0.bias'])


def _attention(sd, p, x, pos_emb, key_mask, heads):
    """squeezeformer/attention.py:88-167 (rel_shift removed, ada scale on q/k/v inputs)."""
    B, T, d = x.shape
    dk = d // heads
    xs = _ada(sd, p, x)
    q = F.linear(xs, sd[p + '.linear_q.weight'], sd[p + '.linear_q.bias']).view(B, T, heads, dk)
    k = F.linear(xs, sd[p + '.linear_k.weight'], sd[p + '.linear_k.bias']).view(B, T, heads, dk).transpose(1, 2)
    v = F.linear(xs, sd[p + '.linear_v.weight'], sd[p + '.linear_v.bias']).view(B, T, heads, dk).transpose(1, 2)
    pp = F.linear(pos_emb, sd[p + '.linear_pos.weight']).view(1, -1, heads, dk).transpose(1, 2)
    qu = (q + sd[p + '.pos_bias_u']).transpose(1, 2)
    qv = (q + sd[p + '.pos_bias_v']).transpose(1, 2)
    scores = (qu @ k.transpose(-2, -1) + qv @ pp.transpose(-2, -1)) / math.sqrt(dk)
    m = ~key_mask.unsqueeze(1)                         # (B,1,1,T)
    scores = scores.masked_fill(m, -float('inf'))
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    o = (attn @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, sd[p + '.linear_out.weight'], sd[p + '.linear_out.bias'])


def _ffn(sd, p, x):
    """squeezeformer/positionwise.py:49-59."""
    h = F.silu(F.linear(_ada(sd, p, x), sd[p + '.w_1.weight'], sd[p + '.w_1.bias']))
    return F.linear(h, sd[p + '.w_2.weight'], sd[p + '.w_2.bias'])


def _conv_module(sd, p, x, pad_mask, kernel, causal=False):
    """squeezeformer/convolution.py:92-148, BatchNorm1d in eval mode; causal (streaming-trained model): the input is
    left-padded with kernel-1 zero frames BEFORE pointwise_conv1 (:117-120), symmetric otherwise (Conv1d padding)."""
    x = _ada(sd, p, x).transpose(1, 2)
    x = x.masked_fill(~pad_mask.unsqueeze(1), 0.0)
    if causal:
        x = F.pad(x, (kernel - 1, 0), 'constant', 0.0)
    x = F.conv1d(x, sd[p + '.pointwise_conv1.weight'], sd[p + '.pointwise_conv1.bias'])
    x = F.glu(x, dim=1)
    x = F.conv1d(x, sd[p + '.depthwise_conv.weight'], sd[p + '.depthwise_conv.bias'],
                 padding=0 if causal else (kernel - 1) // 2, groups=x.shape[1])
    x = F.batch_norm(x, sd[p + '.norm.running_mean'], sd[p + '.norm.running_var'], sd[p + '.norm.weight'],
                     sd[p + '.norm.bias'], False, 0.1, 1e-5)
    x = F.silu(x)
    x = F.conv1d(x, sd[p + '.pointwise_conv2.weight'], sd[p + '.pointwise_conv2.bias'])
    x = x.masked_fill(~pad_mask.unsqueeze(1), 0.0)
    return x.transpose(1, 2)


def _layer(sd, i, x, pos_emb, pad_mask, heads, kernel, causal=False, att_mask=None):
    """SqueezeformerEncoderLayer.forward, normalize_before=False (squeezeformer/encoder.py:412-463).  att_mask: the
    (B, T, T) chunk mask of the attention, or None = the pad mask (B, 1, T)."""
    p = f'encoder.encoders.{i}'
    am = pad_mask.unsqueeze(1) if att_mask is None else att_mask
    x = _ln(sd, p + '.layer_norm1', x + _attention(sd, p + '.self_attn', x, pos_emb, am, heads))
    x = _ln(sd, p + '.layer_norm2', x + _ffn(sd, p + '.ffn1', x))
    x = _ln(sd, p + '.layer_norm3', x + _conv_module(sd, p + '.conv_module', x, pad_mask, kernel, causal))
    x = _ln(sd, p + '.layer_norm4', x + _ffn(sd, p + '.ffn2', x))
    return x


def _time_reduce(sd, x, pad_mask):
    """TimeReductionLayer1D.forward (squeezeformer/time_reduction.py:53-76): dw k=5 s=2 pad=3 + pw; the streaming build
    uses TimeReductionLayerStream (:131-200): dw k=1 s=2 pad=0 -- told apart by the depthwise kernel size."""
    p = 'encoder.time_reduction_layer'
    y = x.transpose(1, 2).masked_fill(~pad_mask.unsqueeze(1), 0.0)
    k = sd[p + '.dw_conv.weight'].shape[-1]
    y = F.conv1d(y, sd[p + '.dw_conv.weight'], sd[p + '.dw_conv.bias'], stride=2, padding=max(0, k - 2), groups=y.shape[1])
    y = F.conv1d(y, sd[p + '.pw_conv.weight'], sd[p + '.pw_conv.bias']).transpose(1, 2)
    pm = pad_mask[:, ::2]
    L, T = pm.shape[1], y.shape[1]
    if L - T < 0:
        y = y[:, :L - T, :].contiguous()
    else:
        y = torch.cat([y, torch.zeros(y.shape[0], L - T, y.shape[2])], dim=1)
    return y, pm


def num_blocks_of(sd):
    return 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('encoder.encoders.'))


def encoder_full(sd, feats, lens, heads=4, kernel=31, reduce_idx=5, recover_idx=11, causal=False, decoding_chunk_size=-1):
    """SqueezeformerEncoder.forward (squeezeformer/encoder.py:168-216); ``causal=True`` for the streaming-trained build
    (model.py:37-41: causal convolution + stream time reduction + use_dynamic_chunk, so that a positive
    ``decoding_chunk_size`` limits the attention to chunks: add_optional_chunk_mask, utils/mask.py:78-143; the time reduction
    keeps every second row and column of that mask, time_reduction.py:62)."""
    B, T, _ = feats.shape
    pad = torch.arange(T)[None, :] < lens[:, None]
    x = embed(sd, feats)
    Tp = x.shape[1]
    pad_s = pad[:, :-2:2][:, :-2:2]
    pos_emb = positional_table(5000, x.shape[-1])[:Tp].unsqueeze(0)
    x = _ln(sd, 'encoder.preln', x)
    att = None
    if causal and decoding_chunk_size > 0:
        idx = torch.arange(Tp)
        chunk = idx[None, :] < ((idx[:, None] // decoding_chunk_size + 1) * decoding_chunk_size)
        att = pad_s[:, None, :] & chunk[None]                          # (B, T', T')
    saved = None
    for i in range(num_blocks_of(sd)):
        if i == reduce_idx:
            saved = (x, pad_s, pos_emb, att)
            x, pad_s = _time_reduce(sd, x, pad_s)
            pos_emb = pos_emb[:, ::2, :]
            att = None if att is None else att[:, ::2, ::2]
        if i == recover_idx:
            rx, rpad, rpos, ratt = saved
            y = torch.repeat_interleave(x, repeats=2, dim=1)
            y = F.linear(y, sd['encoder.time_recover_layer.weight'], sd['encoder.time_recover_layer.bias'])
            x = rx + y[:, :rx.shape[1], :]
            pad_s, pos_emb, att = rpad, rpos, ratt
        x = _layer(sd, i, x, pos_emb, pad_s, heads, kernel, causal, att)
    return x


def get_encoder_out(sd, feats, lens, **kw):
    """SqueezeformerModel.get_encoder_out (squeezeformer/model.py) -> CTC softmax probabilities."""
    enc = encoder_full(sd, feats, lens, **kw)
    return torch.softmax(F.linear(enc, sd['ctc.ctc_lo.weight'], sd['ctc.ctc_lo.bias']), dim=2)


# ---- chunked streaming (streaming-trained build) ------------------------------------------------------------------------
def _attention_chunk(sd, p, x, pos_emb, heads, cache):
    """attention.py:88-167 with the key/value cache of forward_chunk: cache [1,H,t,2dk] or None; no masks (fake mask)."""
    B, T, d = x.shape
    dk = d // heads
    xs = _ada(sd, p, x)
    q = F.linear(xs, sd[p + '.linear_q.weight'], sd[p + '.linear_q.bias']).view(B, T, heads, dk)
    k = F.linear(xs, sd[p + '.linear_k.weight'], sd[p + '.linear_k.bias']).view(B, T, heads, dk).transpose(1, 2)
    v = F.linear(xs, sd[p + '.linear_v.weight'], sd[p + '.linear_v.bias']).view(B, T, heads, dk).transpose(1, 2)
    if cache is not None and cache.shape[2] > 0:
        k = torch.cat([cache[..., :dk], k], dim=2)
        v = torch.cat([cache[..., dk:], v], dim=2)
    new_cache = torch.cat([k, v], dim=-1)
    pp = F.linear(pos_emb, sd[p + '.linear_pos.weight']).view(1, -1, heads, dk).transpose(1, 2)
    qu = (q + sd[p + '.pos_bias_u']).transpose(1, 2)
    qv = (q + sd[p + '.pos_bias_v']).transpose(1, 2)
    scores = (qu @ k.transpose(-2, -1) + qv @ pp.transpose(-2, -1)) / math.sqrt(dk)
    o = (torch.softmax(scores, dim=-1) @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, sd[p + '.linear_out.weight'], sd[p + '.linear_out.bias']), new_cache


def _conv_module_chunk(sd, p, x, kernel, cache):
    """convolution.py:92-148, causal, with the left-context cache [1,d,kernel-1] (or None -> zero padding)."""
    x = _ada(sd, p, x).transpose(1, 2)
    if cache is None or cache.shape[2] == 0:
        x = F.pad(x, (kernel - 1, 0), 'constant', 0.0)
    else:
        x = torch.cat([cache, x], dim=2)
    new_cache = x[:, :, -(kernel - 1):]
    x = F.conv1d(x, sd[p + '.pointwise_conv1.weight'], sd[p + '.pointwise_conv1.bias'])
    x = F.glu(x, dim=1)
    x = F.conv1d(x, sd[p + '.depthwise_conv.weight'], sd[p + '.depthwise_conv.bias'], groups=x.shape[1])
    x = F.batch_norm(x, sd[p + '.norm.running_mean'], sd[p + '.norm.running_var'], sd[p + '.norm.weight'],
                     sd[p + '.norm.bias'], False, 0.1, 1e-5)
    x = F.conv1d(F.silu(x), sd[p + '.pointwise_conv2.weight'], sd[p + '.pointwise_conv2.bias'])
    return x.transpose(1, 2), new_cache


def get_encoder_out_chunk(sd, feats, offset, required_cache_size, att_cache, cnn_cache, heads=4, kernel=31, reduce_idx=5,
                          recover_idx=11):
    """SqueezeformerModel.get_encoder_out_chunk -> SqueezeformerEncoder.forward_chunk (squeezeformer/encoder.py:240-362):
    att_cache [L,H,t,2dk] at the FULL frame rate (layers reduce_idx..recover_idx-1 read every second entry and write their
    cache back repeat-interleaved), cnn_cache [L,1,d,kernel-1]; returns (probs, att_cache, cnn_cache)."""
    assert feats.shape[0] == 1
    x = embed(sd, feats)
    L = num_blocks_of(sd)
    have = att_cache.numel() > 0
    cache_t1 = att_cache.shape[2] if have else 0
    chunk = x.shape[1]
    key_size = cache_t1 + chunk
    pos_emb = positional_table(5000, x.shape[-1])[offset - cache_t1: offset - cache_t1 + key_size].unsqueeze(0)
    if required_cache_size < 0:
        start = 0
    elif required_cache_size == 0:
        start = key_size
    else:
        start = max(key_size - required_cache_size, 0)
    x = _ln(sd, 'encoder.preln', x)
    r_att, r_cnn = [], []
    saved = None
    max_att_len = 0
    for i in range(L):
        if i == reduce_idx:
            saved = (x, pos_emb)
            x, _ = _time_reduce(sd, x, torch.ones(1, x.shape[1], dtype=torch.bool))
            pos_emb = pos_emb[:, ::2, :]
        if i == recover_idx:
            rx, rpos = saved
            y = torch.repeat_interleave(x, repeats=2, dim=1)
            y = F.linear(y, sd['encoder.time_recover_layer.weight'], sd['encoder.time_recover_layer.bias'])
            x = rx + y[:, :rx.shape[1], :]
            pos_emb = rpos
        factor = 2 if reduce_idx <= i < recover_idx else 1
        ac = att_cache[i:i + 1][:, :, ::factor, :][:, :, :pos_emb.shape[1] - x.shape[1], :] if have else None
        p = f'encoder.encoders.{i}'
        a, new_att = _attention_chunk(sd, p + '.self_attn', x, pos_emb, heads, ac)
        x = _ln(sd, p + '.layer_norm1', x + a)
        x = _ln(sd, p + '.layer_norm2', x + _ffn(sd, p + '.ffn1', x))
        c, new_cnn = _conv_module_chunk(sd, p + '.conv_module', x, kernel, cnn_cache[i] if cnn_cache.numel() > 0 else None)
        x = _ln(sd, p + '.layer_norm3', x + c)
        x = _ln(sd, p + '.layer_norm4', x + _ffn(sd, p + '.ffn2', x))
        cached = new_att[:, :, start // factor:, :].repeat_interleave(repeats=factor, dim=2)
        if i == 0:
            max_att_len = cached.shape[2]
        r_att.append(cached[:, :, :max_att_len, :])
        r_cnn.append(new_cnn.unsqueeze(0))
    probs = torch.softmax(F.linear(x, sd['ctc.ctc_lo.weight'], sd['ctc.ctc_lo.bias']), dim=2)
    return probs, torch.cat(r_att, dim=0), torch.cat(r_cnn, dim=0)
