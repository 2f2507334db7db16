"""Deterministic synthetic checkpoints / inputs (random-init weights of the reference architecture).

The reference ships no weights (checkpoints are a separate download), so benches and parity
tests run on synthetic ``state_dict``s whose key names / shapes are exactly the
reference's (``ConformerModel.state_dict()``, layout listed in SURVEY.md 3.5;
builder ``masr/trainer.py:167-203``).  Generation depends only on numpy so the
same tensors can be rebuilt on the GPU box without ``/root/reference``.
"""
import math
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _uniform(seed, name, shape, bound):
    return torch.from_numpy(_rng(seed, name).uniform(-bound, bound, shape).astype(np.float32))


def conformer_state_dict(seed=0, vocab_size=4233, d=256, heads=4, d_ff=2048, num_blocks=12,
                         kernel=15, n_mels=80, ctc_gain=6.0, cnn_module_norm='layer_norm'):
    """Keys/shapes == reference ``encoder.*`` + ``ctc.*`` entries (attention-decoder
    ``decoder.*`` entries are never used by get_encoder_out*, model.py:152-190).  ``cnn_module_norm='batch_norm'``
    (conformer/convolution.py:60-67): the conv module's norm carries running statistics (same weight / bias shapes)."""
    sd = {}
    f2 = ((n_mels - 1) // 2 - 1) // 2

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        b = gain / math.sqrt(in_f)
        sd[name + '.weight'] = _uniform(seed, name + '.weight', (out_f, in_f), b * math.sqrt(3.0))
        if bias:
            sd[name + '.bias'] = _uniform(seed, name + '.bias', (out_f,), 0.1)

    def ln(name, n=d):
        sd[name + '.weight'] = 1.0 + _uniform(seed, name + '.weight', (n,), 0.2)
        sd[name + '.bias'] = _uniform(seed, name + '.bias', (n,), 0.1)

    sd['encoder.global_cmvn.mean'] = 13.5 + _uniform(seed, 'cmvn.mean', (n_mels,), 1.0)
    sd['encoder.global_cmvn.istd'] = 0.3 + _uniform(seed, 'cmvn.istd', (n_mels,), 0.05)
    sd['encoder.embed.conv.0.weight'] = _uniform(seed, 'conv0.w', (d, 1, 3, 3), math.sqrt(3.0 / 9))
    sd['encoder.embed.conv.0.bias'] = _uniform(seed, 'conv0.b', (d,), 0.1)
    sd['encoder.embed.conv.2.weight'] = _uniform(seed, 'conv2.w', (d, d, 3, 3), math.sqrt(3.0 / (9 * d)))
    sd['encoder.embed.conv.2.bias'] = _uniform(seed, 'conv2.b', (d,), 0.1)
    lin('encoder.embed.out.0', d, d * f2)
    for i in range(num_blocks):
        p = f'encoder.encoders.{i}.'
        for q in ('linear_q', 'linear_k', 'linear_v', 'linear_out'):
            lin(p + 'self_attn.' + q, d, d)
        lin(p + 'self_attn.linear_pos', d, d, bias=False)
        sd[p + 'self_attn.pos_bias_u'] = _uniform(seed, p + 'u', (heads, d // heads), 0.3)
        sd[p + 'self_attn.pos_bias_v'] = _uniform(seed, p + 'v', (heads, d // heads), 0.3)
        for ff in ('feed_forward', 'feed_forward_macaron'):
            lin(p + ff + '.w_1', d_ff, d)
            lin(p + ff + '.w_2', d, d_ff)
        sd[p + 'conv_module.pointwise_conv1.weight'] = _uniform(seed, p + 'pw1.w', (2 * d, d, 1), math.sqrt(3.0 / d))
        sd[p + 'conv_module.pointwise_conv1.bias'] = _uniform(seed, p + 'pw1.b', (2 * d,), 0.1)
        sd[p + 'conv_module.depthwise_conv.weight'] = _uniform(seed, p + 'dw.w', (d, 1, kernel), math.sqrt(3.0 / kernel))
        sd[p + 'conv_module.depthwise_conv.bias'] = _uniform(seed, p + 'dw.b', (d,), 0.1)
        ln(p + 'conv_module.norm')
        if cnn_module_norm == 'batch_norm':
            sd[p + 'conv_module.norm.running_mean'] = _uniform(seed, p + 'cnorm.mean', (d,), 0.3)
            sd[p + 'conv_module.norm.running_var'] = 1.0 + _uniform(seed, p + 'cnorm.var', (d,), 0.5)
        sd[p + 'conv_module.pointwise_conv2.weight'] = _uniform(seed, p + 'pw2.w', (d, d, 1), math.sqrt(3.0 / d))
        sd[p + 'conv_module.pointwise_conv2.bias'] = _uniform(seed, p + 'pw2.b', (d,), 0.1)
        for n in ('norm_ff', 'norm_mha', 'norm_ff_macaron', 'norm_conv', 'norm_final'):
            ln(p + n)
    ln('encoder.after_norm')
    # sharpened CTC head: with gain 1 the softmax over V is nearly flat and the
    # greedy argmax is decided by 1e-6 gaps
    lin('ctc.ctc_lo', vocab_size, d, gain=ctc_gain)
    return sd


def synthetic_vocab(vocab_size=4233):
    """vocabulary.txt convention of trainer.py:480-488: <blank>, <unk>, tokens..., <eos>."""
    toks = ['<blank>', '<unk>', '<space>']
    cp = 0x4E00
    while len(toks) < vocab_size - 1:
        toks.append(chr(cp))
        cp += 1
    toks.append('<eos>')
    return toks


def synthetic_pcm(batch, n_samples, seed=1234):
    """BASELINE.md config-2 generator: N(0, 3000) -> rint -> clip -> int16."""
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 3000, (batch, n_samples))
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def squeezeformer_state_dict(seed=0, vocab_size=4233, d=256, heads=4, ff_factor=8, num_blocks=12, kernel=31,
                             n_mels=80, ctc_gain=6.0, streaming=False):
    """Keys/shapes == reference SqueezeformerModel ``encoder.*`` + ``ctc.*`` (masr/model_utils/squeezeformer/):
    BatchNorm conv module; ``streaming=False`` -> TimeReductionLayer1D (kernel 5), ``streaming=True`` (the shipped YAML
    default) -> TimeReductionLayerStream (kernel 1) -- the causal convolution changes no shapes."""
    sd = {}
    f2 = ((n_mels - 1) // 2 - 1) // 2
    d_ff = d * ff_factor

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        b = gain / math.sqrt(in_f)
        sd[name + '.weight'] = _uniform(seed, name + '.weight', (out_f, in_f), b * math.sqrt(3.0))
        if bias:
            sd[name + '.bias'] = _uniform(seed, name + '.bias', (out_f,), 0.1)

    def ln(name, n=d):
        sd[name + '.weight'] = 1.0 + _uniform(seed, name + '.weight', (n,), 0.2)
        sd[name + '.bias'] = _uniform(seed, name + '.bias', (n,), 0.1)

    def ada(p):
        sd[p + '.ada_scale'] = (1.0 + _uniform(seed, p + '.ada_scale', (1, 1, d), 0.2))
        sd[p + '.ada_bias'] = _uniform(seed, p + '.ada_bias', (1, 1, d), 0.1)

    sd['encoder.global_cmvn.mean'] = 13.5 + _uniform(seed, 'cmvn.mean', (n_mels,), 1.0)
    sd['encoder.global_cmvn.istd'] = 0.3 + _uniform(seed, 'cmvn.istd', (n_mels,), 0.05)
    sd['encoder.embed.pw_conv.weight'] = _uniform(seed, 'sq.conv0.w', (d, 1, 3, 3), math.sqrt(3.0 / 9))
    sd['encoder.embed.pw_conv.bias'] = _uniform(seed, 'sq.conv0.b', (d,), 0.1)
    sd['encoder.embed.dw_conv.weight'] = _uniform(seed, 'sq.conv2.w', (d, d, 3, 3), math.sqrt(3.0 / (9 * d)))
    sd['encoder.embed.dw_conv.bias'] = _uniform(seed, 'sq.conv2.b', (d,), 0.1)
    # input_proj sees x * sqrt(d) (pos_enc scaling happens BEFORE the projection, subsampling.py:72-75)
    lin('encoder.embed.input_proj.0', d, d * f2, gain=1.0 / math.sqrt(d))
    ln('encoder.preln')
    for i in range(num_blocks):
        p = f'encoder.encoders.{i}.'
        for q in ('linear_q', 'linear_k', 'linear_v', 'linear_out'):
            lin(p + 'self_attn.' + q, d, d)
        lin(p + 'self_attn.linear_pos', d, d, bias=False)
        sd[p + 'self_attn.pos_bias_u'] = _uniform(seed, p + 'u', (heads, d // heads), 0.3)
        sd[p + 'self_attn.pos_bias_v'] = _uniform(seed, p + 'v', (heads, d // heads), 0.3)
        ada(p + 'self_attn')
        for ff in ('ffn1', 'ffn2'):
            lin(p + ff + '.w_1', d_ff, d)
            lin(p + ff + '.w_2', d, d_ff)
            ada(p + ff)
        ada(p + 'conv_module')
        sd[p + 'conv_module.pointwise_conv1.weight'] = _uniform(seed, p + 'pw1.w', (2 * d, d, 1), math.sqrt(3.0 / d))
        sd[p + 'conv_module.pointwise_conv1.bias'] = _uniform(seed, p + 'pw1.b', (2 * d,), 0.1)
        sd[p + 'conv_module.depthwise_conv.weight'] = _uniform(seed, p + 'dw.w', (d, 1, kernel), math.sqrt(3.0 / kernel))
        sd[p + 'conv_module.depthwise_conv.bias'] = _uniform(seed, p + 'dw.b', (d,), 0.1)
        ln(p + 'conv_module.norm')
        sd[p + 'conv_module.norm.running_mean'] = _uniform(seed, p + 'bn.mean', (d,), 0.2)
        sd[p + 'conv_module.norm.running_var'] = 0.25 + _uniform(seed, p + 'bn.var', (d,), 0.1)
        sd[p + 'conv_module.norm.num_batches_tracked'] = torch.tensor(100, dtype=torch.long)
        sd[p + 'conv_module.pointwise_conv2.weight'] = _uniform(seed, p + 'pw2.w', (d, d, 1), math.sqrt(3.0 / d))
        sd[p + 'conv_module.pointwise_conv2.bias'] = _uniform(seed, p + 'pw2.b', (d,), 0.1)
        for n in ('layer_norm1', 'layer_norm2', 'layer_norm3', 'layer_norm4'):
            ln(p + n)
    tk = 1 if streaming else 5
    sd['encoder.time_reduction_layer.dw_conv.weight'] = _uniform(seed, 'tr.dw.w', (d, 1, tk), math.sqrt(3.0 / tk))
    sd['encoder.time_reduction_layer.dw_conv.bias'] = _uniform(seed, 'tr.dw.b', (d,), 0.1)
    sd['encoder.time_reduction_layer.pw_conv.weight'] = _uniform(seed, 'tr.pw.w', (d, d, 1), math.sqrt(3.0 / d))
    sd['encoder.time_reduction_layer.pw_conv.bias'] = _uniform(seed, 'tr.pw.b', (d,), 0.1)
    lin('encoder.time_recover_layer', d, d)
    lin('ctc.ctc_lo', vocab_size, d, gain=ctc_gain)
    return sd


def efficient_conformer_state_dict(seed=0, vocab_size=4233, d=256, heads=4, d_ff=2048, num_blocks=12, kernel=15,
                                   n_mels=80, ctc_gain=6.0, stride_layer_idx=(3,), stride=(2,),
                                   group_layer_idx=(0, 1, 2, 3), group_size=3):
    """Keys/shapes == reference EfficientConformerModel ``encoder.*`` + ``ctc.*``
    (configs/efficient_conformer.yml: streaming, layer_norm conv module, grouped attention in blocks 0-3,
    strided depthwise conv in block 3, kernel 15 -> 7 afterwards; masr/model_utils/efficient_conformer/)."""
    sd = conformer_state_dict(seed, vocab_size, d, heads, d_ff, num_blocks, kernel, n_mels, ctc_gain)
    k = kernel
    for i in range(num_blocks):
        p = f'encoder.encoders.{i}.'
        if i in group_layer_idx:
            sd[p + 'self_attn.pos_bias_u'] = _uniform(seed, p + 'gu', (heads, (d // heads) * group_size), 0.3)
            sd[p + 'self_attn.pos_bias_v'] = _uniform(seed, p + 'gv', (heads, (d // heads) * group_size), 0.3)
        if k != kernel:
            sd[p + 'conv_module.depthwise_conv.weight'] = _uniform(seed, p + 'dw.w', (d, 1, k), math.sqrt(3.0 / k))
        if i in stride_layer_idx:
            k = k // stride[list(stride_layer_idx).index(i)]
    return sd


def deepspeech2_state_dict(seed=0, vocab_size=4233, rnn_size=1024, num_rnn_layers=5, bidirectional=True, n_mels=80,
                           ctc_gain=2.0):
    """Keys/shapes == reference DeepSpeech2Model state_dict (masr/model_utils/deepspeech2/): conv front-end
    1->32->32, ``num_rnn_layers`` x (LSTM(rnn_size) [+ reverse] + LayerNorm), CTC head under ``decoder.ctc_lo``."""
    sd = {}
    f2 = ((n_mels - 1) // 2 - 1) // 2
    sd['encoder.global_cmvn.mean'] = 13.5 + _uniform(seed, 'cmvn.mean', (n_mels,), 1.0)
    sd['encoder.global_cmvn.istd'] = 0.3 + _uniform(seed, 'cmvn.istd', (n_mels,), 0.05)
    sd['encoder.conv.conv.0.weight'] = _uniform(seed, 'ds2.c0.w', (32, 1, 3, 3), math.sqrt(3.0 / 9))
    sd['encoder.conv.conv.0.bias'] = _uniform(seed, 'ds2.c0.b', (32,), 0.1)
    sd['encoder.conv.conv.2.weight'] = _uniform(seed, 'ds2.c2.w', (32, 32, 3, 3), math.sqrt(3.0 / (9 * 32)))
    sd['encoder.conv.conv.2.bias'] = _uniform(seed, 'ds2.c2.b', (32,), 0.1)
    ndir = 2 if bidirectional else 1
    out = rnn_size * ndir
    for i in range(num_rnn_layers):
        isz = 32 * f2 if i == 0 else out
        for suf in ([''] + (['_reverse'] if bidirectional else [])):
            p = f'encoder.rnns.{i}.rnn.'
            sd[p + 'weight_ih_l0' + suf] = _uniform(seed, p + 'wih' + suf, (4 * rnn_size, isz), math.sqrt(3.0 / isz))
            sd[p + 'weight_hh_l0' + suf] = _uniform(seed, p + 'whh' + suf, (4 * rnn_size, rnn_size), math.sqrt(3.0 / rnn_size))
            sd[p + 'bias_ih_l0' + suf] = _uniform(seed, p + 'bih' + suf, (4 * rnn_size,), 0.1)
            sd[p + 'bias_hh_l0' + suf] = _uniform(seed, p + 'bhh' + suf, (4 * rnn_size,), 0.1)
        sd[f'encoder.rnns.{i}.layer_norm.weight'] = 1.0 + _uniform(seed, f'ds2.ln{i}.w', (out,), 0.2)
        sd[f'encoder.rnns.{i}.layer_norm.bias'] = _uniform(seed, f'ds2.ln{i}.b', (out,), 0.1)
    b = ctc_gain / math.sqrt(out)
    sd['decoder.ctc_lo.weight'] = _uniform(seed, 'ds2.ctc.w', (vocab_size, out), b * math.sqrt(3.0))
    sd['decoder.ctc_lo.bias'] = _uniform(seed, 'ds2.ctc.b', (vocab_size,), 0.1)
    return sd
