"""CPU-only checks of the boundary: the shared library loads, exports every symbol
include/cgps.h declares, its host-side arithmetic (level layout, workspace
sizes, argument checks) is right, and the product refuses to run without a GPU
instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

import _util
from cyclic_gps import _hip
import cyclic_gps.cyclic_reduction as cr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cgps.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cgps_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_hip.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_hip.exported_symbols()) == declared
    assert _hip.lib().cgps_version() == 320


@pytest.mark.parametrize("N", [1, 2, 3, 6, 31, 32, 33, 502, 1024, 2 ** 20, 2 ** 24 + 5])
def test_level_layout_matches_reference_level_sizes(N):
    ms, offD, offF, offG = _hip.level_layout(N)
    assert ms == _util.level_sizes(N)
    assert offD[-1] == N
    assert offF[-1] == sum(m // 2 for m in ms) and offG[-1] == sum((m - 1) // 2 for m in ms)
    for i, m in enumerate(ms):
        assert offD[i + 1] - offD[i] == (m + 1) // 2
        assert offF[i + 1] - offF[i] == m // 2
        assert offG[i + 1] - offG[i] == (m - 1) // 2


def test_workspace_and_argument_errors():
    lib = _hip.lib()
    b = ctypes.c_size_t(0)
    for op in range(8):
        assert lib.cgps_workspace_bytes(1 << 20, 4, _hip.F64, op, ctypes.byref(b)) == 0
        assert b.value > 0
    assert lib.cgps_workspace_bytes(0, 4, _hip.F64, 0, ctypes.byref(b)) == 1
    assert lib.cgps_workspace_bytes(8, 9, _hip.F64, 0, ctypes.byref(b)) == 3
    assert lib.cgps_workspace_bytes(8, 4, 7, 0, ctypes.byref(b)) == 3
    assert b"" != lib.cgps_last_error()
    # null pointers are rejected before anything is launched
    assert lib.cgps_mahal_logdet(None, None, None, 8, 4, _hip.F64, None, 0, None, None, None) == 1
    assert lib.cgps_decompose(None, None, 8, 4, _hip.F64, None, None, None, None, 0, None, None) == 1
    assert lib.cgps_solve(None, None, None, 8, 4, _hip.F64, 1, None, None, None, 0, None) == 1


def test_surface_names_match_reference_module():
    for name in ("decompose", "decompose_step", "mahal_and_det", "halfsolve", "backhalfsolve", "solve", "det",
                 "mahal", "inverse_blocks", "UU_T", "Ux", "U_Tx", "SigU", "UtV_diags", "interleave", "JITTER",
                 "np", "torch"):
        assert hasattr(cr, name), name


def test_shape_contract_errors_without_gpu():
    Rs = torch.zeros(4, 2, 2, dtype=torch.float64)
    with pytest.raises(AssertionError):
        cr.decompose(Rs, torch.zeros(2, 2, 2, dtype=torch.float64))
    with pytest.raises(TypeError):
        cr.decompose(Rs, torch.zeros(3, 3, 3, dtype=torch.float64))
    with pytest.raises(TypeError):
        cr.decompose(Rs.to(torch.float16), torch.zeros(3, 2, 2, dtype=torch.float16))


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback():
    Rs, Os, b, _, _ = _util.conditioned_system(8, 2)
    with pytest.raises(_hip.CgpsError):
        cr.mahal_and_det(Rs, Os, b)
    with pytest.raises(_hip.CgpsError):
        cr.decompose(Rs, Os)


def _check_helpers(device):
    """The banded helper products of the surface against vectors recorded from the reference."""
    import numpy as np
    for name in ("helpers_d1_n4_sq", "helpers_d1_n4_nsq", "helpers_d2_n3_sq", "helpers_d2_n3_nsq"):
        g = np.load(os.path.join(_util.GOLDEN, name + ".npz"))
        t = lambda a: torch.from_numpy(a).to(device)   # noqa: E731
        A, B = t(g["A"]), t(g["B"])
        dg, off = cr.UU_T(A, B)
        np.testing.assert_allclose(dg.cpu().numpy(), g["uut_d"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(off.cpu().numpy(), g["uut_o"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(cr.Ux(A, B, t(g["x"])).cpu().numpy(), g["ux"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(cr.U_Tx(A, B, t(g["y"])).cpu().numpy(), g["utx"], rtol=1e-12, atol=1e-12)
        mid, hi = cr.SigU(t(g["Sd"]), t(g["So"]), A, B)
        np.testing.assert_allclose(mid.cpu().numpy(), g["su_mid"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(hi.cpu().numpy(), g["su_hi"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(cr.UtV_diags(A, B, mid, hi).cpu().numpy(), g["utv"], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(cr.interleave(t(g["il_a"]), t(g["il_b"])).cpu().numpy(), g["il1"])
        np.testing.assert_array_equal(cr.interleave(t(g["il_b"]), t(g["il_a"])).cpu().numpy(), g["il2"])


def test_helpers_match_reference_golden():
    _check_helpers("cpu")


@pytest.mark.gpu
def test_helpers_on_device_tensors():
    """The same six names handed device tensors (batched device products, see the module docstring)."""
    _check_helpers("cuda")
