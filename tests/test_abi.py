"""The C-ABI library loads (no GPU needed) and exports every symbol the header declares."""
import ctypes
import os
import re

from immunostruct_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "immunostruct_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(is_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_bound_symbols():
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} missing from {_lib.LIB_PATH}"


def test_version_and_scratch_sizes():
    lib = _lib.load()
    assert lib.is_version() >= 100
    assert lib.is_reduce_partials_scratch_floats(2) == 2 * lib.is_reduce_partials_scratch_floats(1)
    assert lib.is_loss_partials_floats() >= 1024


def test_cpu_tensors_are_refused_loudly():
    import pytest
    import torch
    from immunostruct_amd import functional as HF
    with pytest.raises(_lib.HipExtensionError):
        HF.segment_pool(torch.zeros(4, 64), torch.tensor([0, 4], dtype=torch.int32), "mean")
