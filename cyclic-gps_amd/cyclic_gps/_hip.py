"""ctypes binding of libcgps.so (C ABI: include/cgps.h).

The library is the product: if it is missing this module raises, it never falls
back to a CPU or torch implementation.
"""
import ctypes
import functools
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGPS_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libcgps.so"))

F32, F64 = 0, 1
OP_MAHAL_LOGDET, OP_DECOMPOSE, OP_HALFSOLVE, OP_BACKSOLVE, OP_SOLVE, OP_LOGDET_FACTOR, OP_INVERSE_BLOCKS, \
    OP_MAHAL_LOGDET_LEVELWISE, OP_DECOMPOSE_SOLVE = range(9)
MAX_LEVELS = 64

_lib = None

_vp, _i64, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
_SIGNATURES = {
    "cgps_version": (_int, []),
    "cgps_last_error": (ctypes.c_char_p, []),
    "cgps_profile_next_call": (_int, [_vp, _vp]),
    "cgps_reset_counters": (_int, [_vp]),
    "cgps_boundary_solve": (_int, [_vp, _sz, _i64, _int, _int, _vp, _vp, _vp]),
    "cgps_boundary_recursions": (_int, [_vp, _sz, _i64, _i64, _int, _int, _vp, _vp, _vp]),
    "cgps_level_layout": (_int, [_i64, ctypes.POINTER(_int)] + [ctypes.POINTER(_i64)] * 4),
    "cgps_workspace_bytes": (_int, [_i64, _int, _int, _int, ctypes.POINTER(_sz)]),
    "cgps_mahal_logdet": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _sz, _vp, _vp, _vp]),
    "cgps_mahal_logdet_levelwise": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _sz, _vp, _vp, _vp]),
    "cgps_decompose_step": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgps_decompose": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cgps_decompose_solve": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cgps_solve_workspace_bytes": (_int, [_i64, _int, _int, _int, _int, ctypes.POINTER(_sz)]),
    "cgps_halfsolve": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cgps_backsolve": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "cgps_solve": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "cgps_logdet_factor": (_int, [_vp, _i64, _int, _int, _vp, _sz, _vp, _vp]),
    "cgps_inverse_blocks": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "cgps_mahal_logdet_adjoint": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp]),
    "cgps_peg_precision": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp]),
    "cgps_leg_mahal_logdet": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _vp, ctypes.c_size_t, _vp, _vp, _vp]),
    "cgps_leg_mahal_logdet_pair": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _vp, ctypes.c_size_t, _vp, _vp, _vp]),
    "cgps_peg_precision_adjoint": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "cgps_leg_intercast": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgps_record_elems": (_int, [_int, _int, ctypes.POINTER(_i64)]),
    "cgps_shard_reduce": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _sz, _vp, _vp, _vp]),
    "cgps_finish_records": (_int, [_vp, _sz, _vp, _sz, _i64, _i64, _i64, _int, _int, _vp, _vp, _vp]),
}


class CgpsError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load libcgps.so once; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CgpsError(
                "libcgps.so not found at %s -- build it with `python __graft_entry__.py` "
                "(there is no CPU fallback)" % LIB_PATH)
        # libcgps needs libamdhip64.so.7; torch ships its own copy under that soname and must be
        # the one runtime in the process (its streams/pointers are what we are handed), so make
        # sure it is resident before the loader resolves our dependency.
        rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(rt):
            ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise CgpsError("libcgps error %d: %s" % (rc, lib().cgps_last_error().decode()))


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.float64:
        return F64
    raise TypeError("cyclic reduction supports float32 / float64 blocks, got %s" % dt)


def ptr(t):
    return None if t is None or t.numel() == 0 else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@functools.lru_cache(maxsize=512)
def _level_layout_cached(N):
    n = _int(0)
    arrs = [(ctypes.c_int64 * (MAX_LEVELS + 1))() for _ in range(4)]
    check(lib().cgps_level_layout(N, ctypes.byref(n), *arrs))
    L = n.value
    return (tuple(arrs[0][:L]), tuple(arrs[1][:L + 1]), tuple(arrs[2][:L + 1]), tuple(arrs[3][:L + 1]))


def level_layout(N):
    """(ms, offD, offF, offG) python lists; offsets have one extra entry (the total)."""
    return tuple(list(t) for t in _level_layout_cached(int(N)))


@functools.lru_cache(maxsize=2048)
def _workspace_bytes(N, d, dtc, op, nrhs=1):
    b = _sz(0)
    if nrhs == 1:
        check(lib().cgps_workspace_bytes(N, d, dtc, op, ctypes.byref(b)))
    else:
        check(lib().cgps_solve_workspace_bytes(N, d, dtc, op, nrhs, ctypes.byref(b)))
    return b.value


_ws_cache = {}


def workspace(N, d, dt, op, device, nrhs=1):
    """A cached scratch tensor of the size the library asks for (torch owns the memory)."""
    nbytes = _workspace_bytes(int(N), int(d), dtype_code(dt), int(op), int(nrhs))
    key = (device, torch.cuda.current_stream().cuda_stream)
    cur = _ws_cache.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = cur
    return cur, nbytes
