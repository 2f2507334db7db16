"""Packed weight artefact: the tensors the inference path reads from the reference's exports --
``inference.pt`` (TorchScript of the whole model, trainer.py:684-689) or ``model.pt`` (state_dict, trainer.py:308) -- written
once as ONE flat little-endian float32 file with a JSON header.  Loading it needs neither ``torch.jit`` nor the model code and
is a single ``np.memmap`` (the engine copies every tensor to the device as it is named, ``masr_load_tensor``).

    file = b'MASRPACK' | u32 version | u32 header bytes | header JSON (utf-8) | padding to 64 B | float32 data
    header = {"tensors": {name: {"shape": [...], "offset": first float index}}, "meta": {...}}

Only what ``get_encoder_out`` / ``get_encoder_out_chunk`` touch is kept (``encoder.*``, ``ctc.*`` and DeepSpeech2's
``decoder.ctc_lo.*``; the attention
decoder of the reference model is never run on this path, conformer/model.py:152-190); BatchNorm / adaptive-scale folding
stays where it is done for every source format, at engine load.
"""
import json
import struct

import numpy as np
import torch

MAGIC = b'MASRPACK'
VERSION = 1


def _keep(name):
    # encoder.* and the CTC head: ctc.ctc_lo.* (Conformer family) / decoder.ctc_lo.* (DeepSpeech2, deepspeech2/model.py:36)
    return (name.startswith('encoder.') or name.startswith('ctc.') or name.startswith('decoder.ctc_lo.')) \
        and not name.endswith('num_batches_tracked')


def export_packed(state_dict, path, meta=None):
    """state_dict ({name: tensor}, e.g. from ``load_state_dict(inference.pt)``) -> ``path``; returns the number of tensors"""
    tensors, offset = {}, 0
    arrays = []
    for name, t in state_dict.items():
        if not _keep(name):
            continue
        a = np.ascontiguousarray(torch.as_tensor(t).detach().cpu().to(torch.float32).numpy())
        tensors[name] = {'shape': list(a.shape), 'offset': offset}
        offset += a.size
        arrays.append(a.reshape(-1))
    header = json.dumps({'tensors': tensors, 'meta': meta or {}}, ensure_ascii=False).encode('utf-8')
    with open(path, 'wb') as f:
        f.write(MAGIC + struct.pack('<II', VERSION, len(header)) + header)
        f.write(b'\0' * (-f.tell() % 64))
        for a in arrays:
            f.write(a.astype('<f4', copy=False).tobytes())
    return len(tensors)


def is_packed(path):
    with open(path, 'rb') as f:
        return f.read(len(MAGIC)) == MAGIC


def load_packed(path):
    """``path`` -> ({name: float32 tensor (views of one memory map)}, meta)"""
    with open(path, 'rb') as f:
        head = f.read(len(MAGIC) + 8)
        if head[:len(MAGIC)] != MAGIC:
            raise ValueError(f'{path} is not a packed MASR weight file')
        version, hlen = struct.unpack('<II', head[len(MAGIC):])
        if version != VERSION:
            raise ValueError(f'unsupported packed weight version {version}')
        header = json.loads(f.read(hlen).decode('utf-8'))
        data_at = f.tell() + (-f.tell() % 64)
    flat = np.memmap(path, dtype='<f4', mode='r', offset=data_at)
    out = {}
    for name, info in header['tensors'].items():
        n = int(np.prod(info['shape'])) if info['shape'] else 1
        out[name] = torch.from_numpy(np.array(flat[info['offset']:info['offset'] + n]).reshape(info['shape']))
    return out, header.get('meta', {})
