"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/masr_hip.h
declares (no compute calls without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'masr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(masr_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_header(built_lib):
    import ctypes
    from masr_amd import _lib
    assert os.path.exists(built_lib)
    h = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(h, s), f'{s} declared in include/masr_hip.h but not exported'
    assert sorted(_lib.SIGNATURES) == syms, 'ctypes binding and header disagree'
    assert _lib.lib().masr_version() == 1


def test_engine_refuses_without_gpu():
    import pytest
    import torch
    from masr_amd import _lib
    from masr_amd.engine import HipEngine
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.MasrError):
        HipEngine({}, vocab_size=10)          # no CPU fallback: fails loudly
