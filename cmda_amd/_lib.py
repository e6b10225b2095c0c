"""ctypes binding of libcmda_hip.so (the C ABI declared in include/cmda_hip.h).

The product path has exactly one backend: the gfx950 kernel library built in-tree by
``__graft_entry__.build()`` / ``make hip``.  If it is missing, or a tensor is not on the GPU,
every op raises -- there is no CPU fallback.  ``_bind_for_tests`` exists only so that the
test-suite can run the *same kernel sources* through the CPU emulator build
(tests/emu/libcmda_emu.so); nothing in the package calls it.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CMDA_HIP_LIB: tuning builds of the SAME library (e.g. `make timing`); never a different backend
_LIB_PATH = os.environ.get('CMDA_HIP_LIB') or os.path.join(_HERE, 'libcmda_hip.so')

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

F32, BF16 = 0, 1


class View(ctypes.Structure):
    _fields_ = [('ptr', c_vp), ('ld', c_i64), ('R', c_i64), ('Cc', c_i64), ('batch_stride', c_i64), ('batch2_stride', c_i64),
                ('conv', c_i32), ('H', c_i32), ('W', c_i32), ('C', c_i32), ('OH', c_i32), ('OW', c_i32),
                ('KH', c_i32), ('KW', c_i32), ('stride', c_i32), ('pad', c_i32), ('dil', c_i32),
                ('in_dil', c_i32), ('reflect', c_i32), ('vec_ok', c_i32)]


class GemmParams(ctypes.Structure):
    _fields_ = [('A', View), ('B', View), ('a_kstrided', c_i32), ('b_kstrided', c_i32), ('C', c_vp),
                ('ldc', c_i64), ('c_batch_stride', c_i64), ('c_batch2_stride', c_i64), ('M', c_i32), ('N', c_i32),
                ('K', c_i32), ('batch', c_i32), ('batch2', c_i32), ('splits', c_i32), ('alpha', c_f32), ('beta', c_f32), ('bias', c_vp),
                ('act', c_i32), ('res', c_vp), ('ldres', c_i64), ('res_batch_stride', c_i64), ('res_batch2_stride', c_i64),
                ('rowscale', c_vp), ('rows_per_scale', c_i32), ('out_f32', c_i32), ('atomic', c_i32),
                ('dtype', c_i32), ('c_vec_ok', c_i32), ('colsum', c_vp),
                ('c_patch_ow', c_i32), ('c_patch_kh', c_i32), ('c_patch_kwci', c_i32), ('tile_hint', c_i32), ('c_perm_ci', c_i32), ('c_perm_cells', c_i32),
                ('res_f32', c_i32), ('colstats_rows', c_i32), ('colstats', c_vp)]


class CmdaError(RuntimeError):
    pass


_ERR = {-1: 'bad shape', -2: 'unsupported dtype', -3: 'HIP launch error', -4: 'unsupported argument combination'}

_lib = None
_emulated = False


def _declare(lib):
    lib.cmda_abi_version.restype = ctypes.c_int
    lib.cmda_layernorm_bwd_ws_floats.restype = ctypes.c_int64
    lib.cmda_layernorm_bwd_ws_floats.argtypes = [ctypes.c_int64, ctypes.c_int]
    lib.cmda_bn_ws_floats.restype = ctypes.c_int64
    lib.cmda_bn_ws_floats.argtypes = [ctypes.c_int]
    lib.cmda_attention_bwd_ws_floats.restype = ctypes.c_int64
    lib.cmda_layernorm_slots.restype = ctypes.c_int
    lib.cmda_gemm_grouped_ws_bytes.restype = ctypes.c_int64
    return lib


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise CmdaError(
            f'{_LIB_PATH} not found: build the gfx950 kernel library first '
            '(python -c "import __graft_entry__ as g; g.build()" or `make hip`). '
            'cmda_amd has no CPU fallback.')
    _lib = _declare(ctypes.CDLL(_LIB_PATH))
    if _lib.cmda_abi_version() != 8:
        raise CmdaError('libcmda_hip.so ABI version mismatch')
    return _lib


def _bind_for_tests(path):
    """TEST ONLY: route the C ABI to the CPU emulator build of the same kernel sources."""
    global _lib, _emulated
    _lib = _declare(ctypes.CDLL(path))
    _emulated = True
    return _lib


def _unbind_for_tests():
    global _lib, _emulated
    _lib = None
    _emulated = False


def emulated():
    return _emulated


def lib():
    return _load()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_of(t):
    """the current HIP stream of t's device as a void* (raw handle: building a torch.cuda.Stream object per launch cost
    ~5 us of the ~14 us a launch spends in Python)"""
    if _emulated:
        return None
    if _raw_stream is not None:
        return c_vp(_raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device()))
    return c_vp(torch.cuda.current_stream(t.device).cuda_stream)


def check_dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if _emulated:
            if t.is_cuda:
                raise CmdaError('emulator build expects CPU tensors')
        elif not t.is_cuda:
            raise CmdaError('cmda_amd ops run only on the GPU (no CPU fallback); got a CPU tensor')
        if not t.is_contiguous():
            raise CmdaError('cmda_amd ops expect contiguous tensors')


def ptr(t):
    return c_vp(t.data_ptr()) if t is not None else None


def dtype_tag(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise CmdaError(f'unsupported activation dtype {t.dtype}')


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise CmdaError(f'{name} failed: {_ERR.get(rc, rc)}')
