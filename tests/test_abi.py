"""The C-ABI library builds, loads, and exports every symbol include/cmda_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'cmda_hip.h')).read()
    return sorted(set(re.findall(r'\bint (cmda_\w+)\(', text)))


def test_header_symbols_exported_by_hip_library():
    lib_path = os.path.join(ROOT, 'cmda_amd', 'libcmda_hip.so')
    if not os.path.exists(lib_path):
        subprocess.check_call(['make', '-j8', 'hip'], cwd=ROOT, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(lib_path)  # loads without a GPU: no HIP call happens at load time
    syms = declared_symbols()
    assert len(syms) >= 33
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/cmda_hip.h but not exported'
    assert lib.cmda_abi_version() == 8


def test_every_exported_entry_point_is_declared():
    srcs = os.path.join(ROOT, 'cmda_amd', 'csrc')
    defined = set()
    for f in os.listdir(srcs):
        if f.endswith('.hip'):
            defined |= set(re.findall(r'extern "C" int (cmda_\w+)\(', open(os.path.join(srcs, f)).read()))
    defined = {d for d in defined if not d.startswith('cmda_debug_')}  # tuning-build-only hooks (-DCMDA_GEMM_TIMING)
    assert defined == set(declared_symbols())


def test_product_has_no_cpu_fallback():
    import torch
    from cmda_amd import _lib, ops
    _lib._unbind_for_tests()
    x = torch.randn(4, 64)
    try:
        ops.layernorm_fwd(x, torch.ones(64), torch.zeros(64), 1e-6)
    except _lib.CmdaError as e:
        assert 'no CPU fallback' in str(e)
    else:
        raise AssertionError('CPU tensors must be rejected by the product path')
