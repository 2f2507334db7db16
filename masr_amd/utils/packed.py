"""Packed weight artefact: the tensors the inference path reads from the reference's exports --
``inference.pt`` (TorchScript of the whole model, trainer.py:684-689) or ``model.pt`` (state_dict, trainer.py:308) -- written
once as ONE flat little-endian float32 file with a JSON header.  Loading it needs neither ``torch.jit`` nor the model code and
is a single ``np.memmap`` (the engine copies every tensor to the device as it is named, ``masr_load_tensor``).

    file = b'MASRPACK' | u32 version | u32 header bytes | header JSON (utf-8) | padding to 64 B | data (float32 or bfloat16)
    header = {"tensors": {name: {"shape": [...], "offset": first element index}}, "meta": {...}, "dtype": "f32" | "bf16"}

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
VERSION = 2          # 1: float32 data only; 2: header carries "dtype" ("f32" | "bf16")


def _keep(name):
    # encoder.* and the CTC head: ctc.ctc_lo.* (Conformer family) / decoder.ctc_lo.* (DeepSpeech2, deepspeech2/model.py:36)
    return (name.startswith('encoder.') or name.startswith('ctc.') or name.startswith('decoder.ctc_lo.')) \
        and not name.endswith('num_batches_tracked')


def export_packed(state_dict, path, meta=None, dtype='f32'):
    """state_dict ({name: tensor}, e.g. from ``load_state_dict(inference.pt)``) -> ``path``; returns the number of tensors.
    ``dtype='bf16'`` stores the weights rounded to bfloat16 (half the file; they are widened back to float32 at load -- the
    engine computes in fp32 either way, so this is an explicit, lossy storage option: ~3 significant digits per weight)."""
    if dtype not in ('f32', 'bf16'):
        raise ValueError("dtype must be 'f32' or 'bf16'")
    tensors, offset = {}, 0
    arrays = []
    for name, t in state_dict.items():
        if not _keep(name):
            continue
        t = torch.as_tensor(t).detach().cpu().to(torch.float32).contiguous()
        if dtype == 'bf16':
            a = t.to(torch.bfloat16).view(torch.int16).numpy()
        else:
            a = t.numpy()
        tensors[name] = {'shape': list(t.shape), 'offset': offset}
        offset += a.size
        arrays.append(np.ascontiguousarray(a).reshape(-1))
    header = json.dumps({'tensors': tensors, 'meta': meta or {}, 'dtype': dtype}, ensure_ascii=False).encode('utf-8')
    with open(path, 'wb') as f:
        f.write(MAGIC + struct.pack('<II', VERSION, len(header)) + header)
        f.write(b'\0' * (-f.tell() % 64))
        for a in arrays:
            f.write(a.astype('<i2' if dtype == 'bf16' else '<f4', copy=False).tobytes())
    return len(tensors)


def is_packed(path):
    with open(path, 'rb') as f:
        return f.read(len(MAGIC)) == MAGIC


def load_packed(path):
    """``path`` -> ({name: float32 tensor}, meta); one memory map, bf16 files are widened to float32 here"""
    with open(path, 'rb') as f:
        head = f.read(len(MAGIC) + 8)
        if head[:len(MAGIC)] != MAGIC:
            raise ValueError(f'{path} is not a packed MASR weight file')
        version, hlen = struct.unpack('<II', head[len(MAGIC):])
        if version not in (1, 2):
            raise ValueError(f'unsupported packed weight version {version}')
        header = json.loads(f.read(hlen).decode('utf-8'))
        data_at = f.tell() + (-f.tell() % 64)
    bf16 = header.get('dtype', 'f32') == 'bf16'
    flat = np.memmap(path, dtype='<i2' if bf16 else '<f4', mode='r', offset=data_at)
    out = {}
    for name, info in header['tensors'].items():
        n = int(np.prod(info['shape'])) if info['shape'] else 1
        t = torch.from_numpy(np.array(flat[info['offset']:info['offset'] + n]).reshape(info['shape']))
        out[name] = t.view(torch.bfloat16).to(torch.float32) if bf16 else t
    return out, header.get('meta', {})
