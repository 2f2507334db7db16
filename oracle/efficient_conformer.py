"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the reference Efficient-Conformer full-context
forward (``configs/efficient_conformer.yml``: streaming-trained => causal convs; conv2d (x4) front-end;
GroupedRelPositionMultiHeadedAttention (group 3) in blocks 0-3; block 3 = StrideConformerEncoderLayer with a
stride-2 depthwise conv and an AvgPool1d residual; blocks >= 4 run at half the frame rate with kernel 7).
Paths are relative to ``masr/model_utils``.  Pinned against the real reference modules by
tests/test_oracle_golden.py (live + committed fixture)."""
import math

import torch
import torch.nn.functional as F

from oracle import conformer as oc


def _grouped_attention(sd, p, x, pos_emb, key_mask, heads, g=3):
    """GroupedRelPositionMultiHeadedAttention.forward / pad4group (efficient_conformer/attention.py:35-69,120-182).
    key_mask bool [B, T] (True = keep); mask[:, ::g, ::g] -> grouped key j keeps key_mask[:, g*j]."""
    B, T, d = x.shape
    dk = d // heads
    q = F.linear(x, sd[p + '.linear_q.weight'], sd[p + '.linear_q.bias'])       # [B,T,d] (head-major features)
    k = F.linear(x, sd[p + '.linear_k.weight'], sd[p + '.linear_k.bias'])
    v = F.linear(x, sd[p + '.linear_v.weight'], sd[p + '.linear_v.bias'])
    pp = F.linear(pos_emb, sd[p + '.linear_pos.weight'])                        # [1,T,d]
    pad = (g - T % g) % g

    def grp(t):   # zero-pad time to a multiple of g, then a flat reshape [T, H*dk] -> [T/g, H, g*dk]
        t = F.pad(t, (0, 0, 0, pad))
        return t.reshape(t.shape[0], -1, heads, dk * g).transpose(1, 2)

    Q, K, V, P = grp(q), grp(k), grp(v), grp(pp)
    qu = Q + sd[p + '.pos_bias_u'][None, :, None, :]
    qv = Q + sd[p + '.pos_bias_v'][None, :, None, :]
    scores = (qu @ K.transpose(-2, -1) + qv @ P.transpose(-2, -1)) / math.sqrt(dk * g)
    if key_mask.dim() == 2:
        m = ~key_mask[:, ::g][:, None, None, :]
    else:                                              # (B, T, T) chunk mask: pad4group keeps rows and columns 0, g, 2g, ...
        m = ~key_mask[:, ::g, ::g][:, None]
    scores = scores.masked_fill(m, -float('inf'))
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    o = (attn @ V).transpose(1, 2).reshape(B, -1, d)[:, :T]
    return F.linear(o, sd[p + '.linear_out.weight'], sd[p + '.linear_out.bias'])


def _conv_module(sd, p, x, pad_mask, stride=1, causal=True):
    """efficient_conformer/convolution.py:71-134, layer_norm, optional stride in the depthwise conv; causal (streaming-trained
    build: kernel - 1 zero frames in front of pointwise_conv1) or symmetric (Conv1d padding (kernel - 1) // 2)."""
    kernel = sd[p + '.depthwise_conv.weight'].shape[-1]
    x = x.transpose(1, 2).masked_fill(~pad_mask.unsqueeze(1), 0.0)
    if causal:
        x = F.pad(x, (kernel - 1, 0))
    x = F.conv1d(x, sd[p + '.pointwise_conv1.weight'], sd[p + '.pointwise_conv1.bias'])
    x = F.glu(x, dim=1)
    x = F.conv1d(x, sd[p + '.depthwise_conv.weight'], sd[p + '.depthwise_conv.bias'], stride=stride, groups=x.shape[1],
                 padding=0 if causal else (kernel - 1) // 2)
    x = F.silu(oc._ln(sd, p + '.norm', x.transpose(1, 2))).transpose(1, 2)
    x = F.conv1d(x, sd[p + '.pointwise_conv2.weight'], sd[p + '.pointwise_conv2.bias'])
    pm = pad_mask if pad_mask.shape[1] == x.shape[2] else pad_mask[:, ::stride]
    return x.masked_fill(~pm.unsqueeze(1), 0.0).transpose(1, 2)


def encoder_full(sd, feats, lens, heads=4, stride_layer_idx=(3,), stride=(2,), group_layer_idx=(0, 1, 2, 3), group_size=3,
                 streaming=True, decoding_chunk_size=-1):
    """EfficientConformerEncoder.forward (efficient_conformer/encoder.py:213-265); layer bodies: ConformerEncoderLayer
    (conformer/encoder.py:82-163) / StrideConformerEncoderLayer (:454-545).  ``streaming`` (model.py: causal convs +
    use_dynamic_chunk): a positive decoding_chunk_size then masks the attention by chunks (utils/mask.py:78-143); the stride
    layer keeps every second row and column of the mask (:254-256)."""
    B, T, _ = feats.shape
    pad = torch.arange(T)[None, :] < lens[:, None]
    x = oc.embed(sd, feats)
    Tp = x.shape[1]
    pad_s = pad[:, :-2:2][:, :-2:2]
    pos_emb = oc.positional_table(5000, x.shape[-1])[:Tp].unsqueeze(0)
    att = None
    if streaming and decoding_chunk_size > 0:
        idx = torch.arange(Tp)
        chunk = idx[None, :] < ((idx[:, None] // decoding_chunk_size + 1) * decoding_chunk_size)
        att = pad_s[:, None, :] & chunk[None]                          # (B, T', T')
    for i in range(oc.num_blocks_of(sd)):
        p = f'encoder.encoders.{i}'
        x = x + 0.5 * oc._ffn(sd, p + '.feed_forward_macaron', oc._ln(sd, p + '.norm_ff_macaron', x))
        xn = oc._ln(sd, p + '.norm_mha', x)
        if i in group_layer_idx:
            a = _grouped_attention(sd, p + '.self_attn', xn, pos_emb, pad_s if att is None else att, heads, group_size)
        else:
            a, _ = oc._attention(sd, p + '.self_attn', xn, pos_emb, pad_s[:, None, :] if att is None else att, heads)
        x = x + a
        if i in stride_layer_idx:
            st = stride[list(stride_layer_idx).index(i)]
            c = _conv_module(sd, p + '.conv_module', oc._ln(sd, p + '.norm_conv', x), pad_s, st, streaming)
            res = F.avg_pool1d(x.transpose(1, 2), st, st, 0, True, False).transpose(1, 2)
            x = res + c
            pad_s = pad_s[:, ::st]
            pos_emb = pos_emb[:, ::st, :]
            att = None if att is None else att[:, ::st, ::st]
        else:
            x = x + _conv_module(sd, p + '.conv_module', oc._ln(sd, p + '.norm_conv', x), pad_s, causal=streaming)
        x = x + 0.5 * oc._ffn(sd, p + '.feed_forward', oc._ln(sd, p + '.norm_ff', x))
        x = oc._ln(sd, p + '.norm_final', x)
    return oc._ln(sd, 'encoder.after_norm', x)


def get_encoder_out(sd, feats, lens, **kw):
    return oc.ctc_probs(sd, encoder_full(sd, feats, lens, **kw))


# ---- chunked streaming ----------------------------------------------------------------------------------------------------
def _grouped_attention_chunk(sd, p, x, pos_emb, heads, cache, g=3):
    """GroupedRelPositionMultiHeadedAttention.forward with the key/value cache of forward_chunk (attention.py:120-182): the
    cache is concatenated in front of the new keys BEFORE pad4group, i.e. the flat [T, H*dk] -> [T/g, H, g*dk] regrouping runs
    over cache + chunk together; no masks."""
    B, T, d = x.shape
    dk = d // heads
    q = F.linear(x, sd[p + '.linear_q.weight'], sd[p + '.linear_q.bias']).view(B, T, heads, dk).transpose(1, 2)
    k = F.linear(x, sd[p + '.linear_k.weight'], sd[p + '.linear_k.bias']).view(B, T, heads, dk).transpose(1, 2)
    v = F.linear(x, sd[p + '.linear_v.weight'], sd[p + '.linear_v.bias']).view(B, T, heads, dk).transpose(1, 2)
    pp = F.linear(pos_emb, sd[p + '.linear_pos.weight'])                        # [1, key_size, d]
    if cache is not None and cache.shape[2] > 0:
        k = torch.cat([cache[..., :dk], k], dim=2)
        v = torch.cat([cache[..., dk:], v], dim=2)
    new_cache = torch.cat([k, v], dim=-1)

    def grp(t):   # [B,H,T,dk] -> zero-pad time to a multiple of g -> [B,T,H,dk] flat -> [B, T/g, H, g*dk] -> heads first
        t = F.pad(t, (0, 0, 0, (g - t.shape[2] % g) % g))
        return t.transpose(1, 2).contiguous().view(B, -1, heads, dk * g).transpose(1, 2)

    Q, K, V = grp(q), grp(k), grp(v)
    P = F.pad(pp, (0, 0, 0, (g - pp.shape[1] % g) % g)).view(1, -1, heads, dk * g).transpose(1, 2)
    qu = Q + sd[p + '.pos_bias_u'][None, :, None, :]
    qv = Q + sd[p + '.pos_bias_v'][None, :, None, :]
    scores = (qu @ K.transpose(-2, -1) + qv @ P.transpose(-2, -1)) / math.sqrt(dk * g)
    o = (torch.softmax(scores, dim=-1) @ V).transpose(1, 2).reshape(B, -1, d)[:, :T]
    return F.linear(o, sd[p + '.linear_out.weight'], sd[p + '.linear_out.bias']), new_cache


def _conv_module_chunk(sd, p, x, cache, stride=1):
    """efficient_conformer/convolution.py:71-134 with the left-context cache ([1,d,>=lorder]; the last lorder frames are
    used, :104) or zero padding."""
    kernel = sd[p + '.depthwise_conv.weight'].shape[-1]
    lorder = kernel - 1
    x = x.transpose(1, 2)
    if cache is None or cache.shape[2] == 0:
        x = F.pad(x, (lorder, 0))
    else:
        x = torch.cat([cache[:, :, -lorder:], x], dim=2)
    new_cache = x[:, :, -lorder:]
    x = F.conv1d(x, sd[p + '.pointwise_conv1.weight'], sd[p + '.pointwise_conv1.bias'])
    x = F.glu(x, dim=1)
    x = F.conv1d(x, sd[p + '.depthwise_conv.weight'], sd[p + '.depthwise_conv.bias'], stride=stride, groups=x.shape[1])
    x = F.silu(oc._ln(sd, p + '.norm', x.transpose(1, 2))).transpose(1, 2)
    x = F.conv1d(x, sd[p + '.pointwise_conv2.weight'], sd[p + '.pointwise_conv2.bias'])
    return x.transpose(1, 2), new_cache


def get_encoder_out_chunk(sd, feats, offset, required_cache_size, att_cache, cnn_cache, heads=4, stride_layer_idx=(3,),
                          stride=(2,), group_layer_idx=(0, 1, 2, 3), group_size=3, cnn_module_kernel=15):
    """EfficientConformerModel.get_encoder_out_chunk -> EfficientConformerEncoder.forward_chunk
    (efficient_conformer/encoder.py:267-392).  ``offset`` counts OUTPUT frames and is multiplied by the total
    down-sampling factor (:306); att_cache [L,H,t,2dk] at the input frame rate (layers behind the stride layer read every
    second entry and write back repeat-interleaved, :352,370), cnn_cache [L,1,d,14] left-padded with zeros (:372)."""
    assert feats.shape[0] == 1
    total = 1
    for s_ in stride:
        total *= s_
    offset = offset * total
    x = oc.embed(sd, feats)
    L = oc.num_blocks_of(sd)
    have = att_cache.numel() > 0
    cache_t1 = att_cache.shape[2] if have else 0
    chunk = x.shape[1]
    key_size = cache_t1 + chunk
    pos_emb = oc.positional_table(5000, x.shape[-1])[offset - cache_t1: offset - cache_t1 + key_size].unsqueeze(0)
    if required_cache_size < 0:
        start = 0
    elif required_cache_size == 0:
        start = key_size
    else:
        start = max(key_size - required_cache_size, 0)
    r_att, r_cnn = [], []
    max_att_len = max_cnn_len = 0
    for i in range(L):
        factor = 1
        for idx, si in enumerate(stride_layer_idx):
            if i > si:
                factor *= stride[idx]
        p = f'encoder.encoders.{i}'
        ac = att_cache[i:i + 1, :, ::factor, :] if have else None
        cc = cnn_cache[i] if cnn_cache.numel() > 0 else None
        x = x + 0.5 * oc._ffn(sd, p + '.feed_forward_macaron', oc._ln(sd, p + '.norm_ff_macaron', x))
        xn = oc._ln(sd, p + '.norm_mha', x)
        if i in group_layer_idx:
            a, new_att = _grouped_attention_chunk(sd, p + '.self_attn', xn, pos_emb, heads, ac, group_size)
        else:
            a, new_att = oc._attention(sd, p + '.self_attn', xn, pos_emb, None, heads, ac)
        x = x + a
        if i in stride_layer_idx:
            st = stride[list(stride_layer_idx).index(i)]
            c, new_cnn = _conv_module_chunk(sd, p + '.conv_module', oc._ln(sd, p + '.norm_conv', x), cc, st)
            x = F.avg_pool1d(x.transpose(1, 2), st, st, 0, True, False).transpose(1, 2) + c
            pos_emb = pos_emb[:, ::st, :]
        else:
            c, new_cnn = _conv_module_chunk(sd, p + '.conv_module', oc._ln(sd, p + '.norm_conv', x), cc)
            x = x + c
        x = x + 0.5 * oc._ffn(sd, p + '.feed_forward', oc._ln(sd, p + '.norm_ff', x))
        x = oc._ln(sd, p + '.norm_final', x)
        new_att = new_att[:, :, start // factor:, :].repeat_interleave(repeats=factor, dim=2)
        new_cnn = F.pad(new_cnn.unsqueeze(0), (cnn_module_kernel - 1 - new_cnn.shape[2], 0))
        if i == 0:
            max_att_len, max_cnn_len = new_att.shape[2], new_cnn.shape[3]
        r_att.append(new_att[:, :, -max_att_len:, :])
        r_cnn.append(new_cnn[:, :, :, -max_cnn_len:])
    x = oc._ln(sd, 'encoder.after_norm', x)
    return oc.ctc_probs(sd, x), torch.cat(r_att, dim=0), torch.cat(r_cnn, dim=0)
