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


def header_prototypes():
    """name -> list of parameter kinds ("p" pointer, "i" int, "f" float, "l" long long) as the header declares them"""
    text = open(os.path.join(ROOT, "include", "immunostruct_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for name, args in re.findall(r"\b(?:int|long long|void|float|const char\s*\*)\s*(is_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = " ".join(args.split())
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("hipStream_t"):
                    kinds.append("p")
                elif a.startswith("long long"):
                    kinds.append("l")
                elif a.startswith("float"):
                    kinds.append("f")
                elif a.startswith(("int", "int32_t", "unsigned")):
                    kinds.append("i")
                else:
                    raise AssertionError(f"{name}: parameter {a!r} of a kind the ABI does not use")
        out[name] = kinds
    return out


def test_header_declares_the_bound_symbols():
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_header_parameter_lists_equal_the_bound_signatures():
    """every declared entry point has the parameter COUNT and, parameter by parameter, the kind (pointer / int / float / long long)
    the ctypes binding passes -- a drift between include/immunostruct_hip.h and _lib.SIGNATURES is a silent stack mismatch"""
    kind_of = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_float: "f", ctypes.c_longlong: "l"}
    protos = header_prototypes()
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    for name, kinds in protos.items():
        bound = [kind_of[t] for t in _lib.SIGNATURES[name]]
        assert len(bound) == len(kinds), f"{name}: header declares {len(kinds)} parameters, the binding passes {len(bound)}"
        assert bound == kinds, f"{name}: header {''.join(kinds)} vs binding {''.join(bound)}"


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} missing from {_lib.LIB_PATH}"


def test_failures_carry_their_text():
    """every entry point returns a bare errno-style code; the calling thread's last failure is kept as text (no GPU needed: the
    argument checks come first) and ``_lib.check`` raises with it"""
    import pytest
    lib = _lib.load()
    assert lib.is_reduce_partials(None, 0, 0, 0, None, None, None, None) == -22
    text = lib.is_last_error_string().decode()
    assert text.startswith("is_reduce_partials: invalid argument") and text.endswith("(-22)")
    assert lib.is_version() >= 100 and lib.is_last_error_string().decode() == text      # a successful call leaves it alone
    assert lib.is_egnn_layer_bwd_paired(*([None] * 2 + [128] + [None] * 5 + [131, 64] + [None] * 10 + [128, None, None, 1, 16, 8] + [None] * 19)) == -38
    assert "is_egnn_layer_bwd_paired: not covered by this build" in lib.is_last_error_string().decode()      # Fe = 8: the 256-thread kernel's
    with pytest.raises(_lib.HipExtensionError, match=r"code -22: is_multi_copy: invalid argument"):
        _lib.check(lib.is_multi_copy(None, 0, None), "is_multi_copy")
    # thread-local: another thread starts with an empty text
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.is_last_error_string().decode()))
    t.start(); t.join()
    assert seen == [""]


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
