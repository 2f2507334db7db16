"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the reference
Conformer inference forward (full-context and chunked streaming) + CTC head.

Written from the reference's algorithm, driven directly by a ``state_dict`` with
the reference key names; every function cites the reference lines it follows
(paths relative to ``masr/model_utils``).  Pinned against the real reference
modules by ``tests/test_oracle_conformer.py`` (live, when /root/reference is
present) and by the fixtures of ``oracle/make_golden.py`` (committed).
"""
import math

import torch
import torch.nn.functional as F


def positional_table(max_len, d):
    """conformer/embedding.py:31-37 -- sin on even, cos on odd feature indices."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def subsampled_len(t):
    """Conv2d 3x3 stride 2, twice (conformer/subsampling.py:86-89)."""
    return ((t - 1) // 2 - 1) // 2


def embed(sd, feats):
    """GlobalCMVN (utils/cmvn.py:21-32) + Conv2dSubsampling4.forward
    (conformer/subsampling.py:95-112) incl. x*sqrt(d) (embedding.py:97)."""
    x = (feats - sd['encoder.global_cmvn.mean']) * sd['encoder.global_cmvn.istd']
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd['encoder.embed.conv.0.weight'], sd['encoder.embed.conv.0.bias'], stride=2))
    x = F.relu(F.conv2d(x, sd['encoder.embed.conv.2.weight'], sd['encoder.embed.conv.2.bias'], stride=2))
    b, c, t, f = x.shape
    x = x.transpose(1, 2).reshape(b, t, c * f)          # feature index = c*f2 + f
    x = F.linear(x, sd['encoder.embed.out.0.weight'], sd['encoder.embed.out.0.bias'])
    d = x.shape[-1]
    return x * math.sqrt(d)


def _ffn(sd, p, x):
    """conformer/positionwise.py:30-37 with Swish (utils/common.py:143-155)."""
    h = F.silu(F.linear(x, sd[p + '.w_1.weight'], sd[p + '.w_1.bias']))
    return F.linear(h, sd[p + '.w_2.weight'], sd[p + '.w_2.bias'])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def _attention(sd, p, x, pos_emb, key_mask, heads, cache=None):
    """RelPositionMultiHeadedAttention.forward (conformer/attention.py:190-251)
    + forward_attention (:81-119).  key_mask: bool [B, Tq or 1, Tk] (True = keep)
    or None.  cache: [B, H, t, 2*dk] or None.  Returns (out, new_cache)."""
    B, T, d = x.shape
    dk = d // heads
    q = F.linear(x, sd[p + '.linear_q.weight'], sd[p + '.linear_q.bias']).view(B, T, heads, dk)
    k = F.linear(x, sd[p + '.linear_k.weight'], sd[p + '.linear_k.bias']).view(B, T, heads, dk).transpose(1, 2)
    v = F.linear(x, sd[p + '.linear_v.weight'], sd[p + '.linear_v.bias']).view(B, T, heads, dk).transpose(1, 2)
    if cache is not None and cache.shape[2] > 0:
        k = torch.cat([cache[..., :dk], k], dim=2)
        v = torch.cat([cache[..., dk:], v], dim=2)
    new_cache = torch.cat([k, v], dim=-1)
    pp = F.linear(pos_emb, sd[p + '.linear_pos.weight']).view(1, -1, heads, dk).transpose(1, 2)
    qu = (q + sd[p + '.pos_bias_u']).transpose(1, 2)
    qv = (q + sd[p + '.pos_bias_v']).transpose(1, 2)
    scores = (qu @ k.transpose(-2, -1) + qv @ pp.transpose(-2, -1)) / math.sqrt(dk)
    if key_mask is not None:
        m = ~key_mask.unsqueeze(1)
        scores = scores.masked_fill(m, -float('inf'))
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(scores, dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, sd[p + '.linear_out.weight'], sd[p + '.linear_out.bias']), new_cache


def _conv_module(sd, p, x, pad_mask, kernel, cache=None, causal=True):
    """ConvolutionModule.forward, layer_norm and batch_norm variants (conformer/convolution.py:76-132).  x [B,T,d]; pad_mask bool [B,T]
    (True=valid) or None; cache [B,d,kernel-1] or None.  causal (streaming-trained model): NB the left zero padding (or
    cache) is concatenated BEFORE pointwise_conv1, so padded frames carry glu(bias).  Non-causal (streaming: False):
    symmetric zero padding inside the depthwise Conv1d (:53-61)."""
    x = x.transpose(1, 2)
    if pad_mask is not None:
        x = x.masked_fill(~pad_mask.unsqueeze(1), 0.0)
    lorder = kernel - 1
    new_cache = None
    if causal:
        if cache is None or cache.shape[2] == 0:
            x = F.pad(x, (lorder, 0))
        else:
            x = torch.cat([cache, x], dim=2)
        new_cache = x[:, :, -lorder:]
    x = F.conv1d(x, sd[p + '.pointwise_conv1.weight'], sd[p + '.pointwise_conv1.bias'])
    x = F.glu(x, dim=1)
    x = F.conv1d(x, sd[p + '.depthwise_conv.weight'], sd[p + '.depthwise_conv.bias'], groups=x.shape[1],
                 padding=0 if causal else lorder // 2)
    if p + '.norm.running_mean' in sd:    # cnn_module_norm: batch_norm (convolution.py:60-63,122-125): eval-mode BatchNorm1d on [B, C, T]
        x = F.silu(F.batch_norm(x, sd[p + '.norm.running_mean'], sd[p + '.norm.running_var'], sd[p + '.norm.weight'],
                                sd[p + '.norm.bias'], False, 0.0, 1e-5))
    else:
        x = F.silu(_ln(sd, p + '.norm', x.transpose(1, 2))).transpose(1, 2)
    x = F.conv1d(x, sd[p + '.pointwise_conv2.weight'], sd[p + '.pointwise_conv2.bias'])
    if pad_mask is not None:
        x = x.masked_fill(~pad_mask.unsqueeze(1), 0.0)
    return x.transpose(1, 2), new_cache


def _layer(sd, i, x, pos_emb, att_mask, pad_mask, heads, kernel, att_cache=None, cnn_cache=None, causal=True):
    """ConformerEncoderLayer.forward, normalize_before=True, macaron
    (conformer/encoder.py:82-163)."""
    p = f'encoder.encoders.{i}'
    x = x + 0.5 * _ffn(sd, p + '.feed_forward_macaron', _ln(sd, p + '.norm_ff_macaron', x))
    a, new_att = _attention(sd, p + '.self_attn', _ln(sd, p + '.norm_mha', x), pos_emb, att_mask, heads, att_cache)
    x = x + a
    c, new_cnn = _conv_module(sd, p + '.conv_module', _ln(sd, p + '.norm_conv', x), pad_mask, kernel, cnn_cache, causal)
    x = x + c
    x = x + 0.5 * _ffn(sd, p + '.feed_forward', _ln(sd, p + '.norm_ff', x))
    return _ln(sd, p + '.norm_final', x), new_att, new_cnn


def num_blocks_of(sd):
    return 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('encoder.encoders.'))


def encoder_full(sd, feats, lens, decoding_chunk_size=-1, heads=4, kernel=15, streaming=True,
                 return_layers=False):
    """ConformerEncoder.forward (conformer/encoder.py:305-346) with
    num_decoding_left_chunks=-1.  ``streaming`` (model.py:37-42) selects
    use_dynamic_chunk (=> (B,T',T') chunk mask, utils/mask.py:78-143).
    Returns encoder_out [B,T',d] (after_norm applied) [, per-layer outputs]."""
    if not streaming:                 # use_dynamic_chunk = False: decoding_chunk_size is ignored (mask.py:117-143)
        decoding_chunk_size = -1
    B, T, _ = feats.shape
    pad = torch.arange(T)[None, :] < lens[:, None]                      # ~make_pad_mask (mask.py:146-172)
    x = embed(sd, feats)
    Tp = x.shape[1]
    pad_s = pad[:, :-2:2][:, :-2:2]                                     # subsampling.py:112
    pos_emb = positional_table(5000, x.shape[-1])[:Tp].unsqueeze(0)
    idx = torch.arange(Tp)
    if decoding_chunk_size < 0:
        chunk = torch.ones(Tp, Tp, dtype=torch.bool)
    else:                                                               # subsequent_chunk_mask (mask.py:40-75)
        chunk = idx[None, :] < ((idx[:, None] // decoding_chunk_size + 1) * decoding_chunk_size)
    att_mask = pad_s[:, None, :] & chunk[None]                          # (B,T',T')
    outs = []
    for i in range(num_blocks_of(sd)):
        x, _, _ = _layer(sd, i, x, pos_emb, att_mask, pad_s, heads, kernel, causal=streaming)
        outs.append(x)
    x = _ln(sd, 'encoder.after_norm', x)
    return (x, outs) if return_layers else x


def ctc_probs(sd, enc):
    """CTCLoss.softmax (loss/ctc.py:62-70)."""
    return torch.softmax(F.linear(enc, sd['ctc.ctc_lo.weight'], sd['ctc.ctc_lo.bias']), dim=2)


def ctc_logits(sd, enc):
    return F.linear(enc, sd['ctc.ctc_lo.weight'], sd['ctc.ctc_lo.bias'])


def get_encoder_out(sd, feats, lens, **kw):
    """ConformerModel.get_encoder_out (conformer/model.py:152-167): full-context attention."""
    return ctc_probs(sd, encoder_full(sd, feats, lens, decoding_chunk_size=-1, **kw))


def get_encoder_out_chunk(sd, feats, offset, required_cache_size, att_cache, cnn_cache, heads=4, kernel=15):
    """ConformerModel.get_encoder_out_chunk (conformer/model.py:169-190) ->
    ConformerEncoder.forward_chunk (conformer/encoder.py:348-420).
    att_cache [L,H,t,2dk] (or numel 0), cnn_cache [L,1,d,kernel-1] (or numel 0)."""
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
    r_att, r_cnn = [], []
    for i in range(L):
        x, na, nc = _layer(sd, i, x, pos_emb, None, None, heads, kernel,
                           att_cache[i:i + 1] if have else None,
                           cnn_cache[i] if cnn_cache.numel() > 0 else None)
        r_att.append(na[:, :, start:, :])
        r_cnn.append(nc)
    x = _ln(sd, 'encoder.after_norm', x)
    return ctc_probs(sd, x), torch.cat(r_att, dim=0), torch.stack(r_cnn, dim=0)


# ALGORITHMIC FLOPs (2*MAC, GEMM/conv terms only) -- SURVEY.md section 8(d)
def conformer_flops(T, V=4233, d=256, d_ff=2048, L=12, K=15, n_mels=80, batch=1):
    T1 = (T - 1) // 2
    Tp = (T1 - 1) // 2
    F1 = (n_mels - 1) // 2
    F2 = (F1 - 1) // 2
    T2 = Tp
    per_utt = 2 * 9 * d * T1 * F1 + 2 * 9 * d * d * Tp * F2 + 2 * (F2 * d) * d * Tp
    per_layer = 8 * d * d_ff * Tp + 8 * d * d * Tp + 4 * d * Tp * T2 + 2 * d * Tp * T2 \
        + 4 * d * d * Tp + 2 * K * d * Tp + 2 * d * d * Tp
    per_utt += L * per_layer + 2 * d * V * Tp
    pos = L * 2 * d * d * T2                      # once per layer per batch
    return batch * per_utt + pos
