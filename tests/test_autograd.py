"""Gradients through the HIP path (analytic adjoints in cyclic_gps/cyclic_reduction.py) against
gradients recorded from the reference's own autograd (tests/golden/grad_d3_n37.npz from
make_golden.py; leg_co2like.npz from make_golden_leg.py).  GPU only."""
import os

import numpy as np
import pytest
import torch

import _util
import cyclic_gps.cyclic_reduction as cr
from cyclic_gps import leg

pytestmark = pytest.mark.gpu
T = dict(rtol=1e-7, atol=1e-9)


def _leaves(g, dev="cuda"):
    return [torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ("Rs", "Os", "v")]


def test_mahal_and_det_gradients_match_reference_autograd():
    g = np.load(os.path.join(_util.GOLDEN, "grad_d3_n37.npz"))
    for which, name in ((0, "mahal"), (1, "logdet")):
        R, O, v = _leaves(g)
        out = cr.mahal_and_det(R, O, v)
        assert abs(float(out[which]) - float(g[name])) <= 1e-10 * abs(float(g[name]))
        out[which].backward()
        np.testing.assert_allclose(R.grad.cpu().numpy(), g["g_%s_R" % name], **T)
        np.testing.assert_allclose(O.grad.cpu().numpy(), g["g_%s_O" % name], **T)
        gv = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(g["v"])
        np.testing.assert_allclose(gv, g["g_%s_v" % name], **T)


@pytest.mark.parametrize("d,dtype,n", [(1, torch.float64, 1), (2, torch.float64, 2), (3, torch.float64, 37),
                                       (4, torch.float64, 70001), (4, torch.float32, 5000), (8, torch.float32, 1000)])
def test_mahal_and_det_adjoint_kernel(d, dtype, n):
    """cgps_mahal_logdet_adjoint (both upstream gradients at once, CPU and GPU callers) against
    dR = gl Sigma_diag - gm w w^T, dO = 2 (gl Sigma_off - gm w[1:] w[:-1]^T), dx = 2 gm w."""
    Rs, Os, b, _, _ = _util.conditioned_system(n, d, dtype=dtype, seed=5 + n)
    tol = dict(rtol=1e-9, atol=1e-11) if dtype == torch.float64 else dict(rtol=2e-4, atol=2e-5)
    for dev in ("cuda", "cpu"):
        R, O, v = (t.to(dev).clone().requires_grad_(True) for t in (Rs, Os, b))
        m, ld = cr.mahal_and_det(R, O, v)
        (0.7 * m - 1.3 * ld).backward()
        dec = cr.decompose(Rs.cuda(), Os.cuda())
        w = cr.solve(dec, b.cuda())
        Sd, So = cr.inverse_blocks(dec)
        eR = -1.3 * Sd - 0.7 * w.unsqueeze(-1) * w.unsqueeze(-2)
        eO = 2 * (-1.3 * So - 0.7 * w[1:].unsqueeze(-1) * w[:-1].unsqueeze(-2))
        assert R.grad.device.type == dev
        np.testing.assert_allclose(R.grad.cpu().numpy(), eR.cpu().numpy(), **tol)
        np.testing.assert_allclose(O.grad.cpu().numpy(), eO.cpu().numpy(), **tol)
        np.testing.assert_allclose(v.grad.cpu().numpy(), (1.4 * w).cpu().numpy(), **tol)


def test_det_and_solve_gradients_match_reference_autograd():
    g = np.load(os.path.join(_util.GOLDEN, "grad_d3_n37.npz"))
    R, O, v = _leaves(g)
    cr.det(cr.decompose(R, O)).backward()
    np.testing.assert_allclose(R.grad.cpu().numpy(), g["g_logdet_R"], **T)
    np.testing.assert_allclose(O.grad.cpu().numpy(), g["g_logdet_O"], **T)
    R, O, v = _leaves(g)
    w = torch.from_numpy(g["w"]).cuda()
    s = (cr.solve(cr.decompose(R, O), v) * w).sum()
    assert abs(float(s) - float(g["solvedot"])) <= 1e-9 * abs(float(g["solvedot"]))
    s.backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), g["g_solvedot_v"], **T)
    np.testing.assert_allclose(O.grad.cpu().numpy(), g["g_solvedot_O"], **T)
    ref = g["g_solvedot_R"]       # the reference's autograd leaves dR unsymmetrised; J is symmetric
    np.testing.assert_allclose(R.grad.cpu().numpy(), 0.5 * (ref + ref.transpose(0, 2, 1)), **T)


def test_solve_with_constant_factor_is_differentiable_in_y():
    Rs, Os, b, x_true, _ = _util.conditioned_system(50, 4)
    dec = cr.decompose(Rs.cuda(), Os.cuda())
    y = b.cuda().requires_grad_(True)
    u = torch.randn(50, 4, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    (cr.solve(dec, y) * u).sum().backward()
    np.testing.assert_allclose(y.grad.cpu().numpy(), cr.solve(dec, u).cpu().numpy(), rtol=1e-9, atol=1e-11)


def test_leg_training_gradient_matches_reference():
    """One training-step gradient of the LEG log-likelihood (models.py:374-381) on the CO2-shaped
    workload, entirely on the GPU, against the reference's autograd through its own code."""
    g = np.load(os.path.join(_util.GOLDEN, "leg_co2like.npz"))
    t = lambda k: torch.from_numpy(g[k]).cuda()     # noqa: E731
    N, R, B, L = (t(k).requires_grad_(True) for k in ("N", "R", "B", "Lambda"))
    ll = leg.log_likelihood(leg.LEGMatrices(N, R, B, L), t("ts"), t("xs"))
    assert abs(float(ll) - float(g["grad_ll"])) <= 1e-8 * abs(float(g["grad_ll"]))
    ll.backward()
    d = N.shape[0]
    tril = np.tril(np.ones((d, d)), 0).astype(bool)
    stril = np.tril(np.ones((d, d)), -1).astype(bool)
    scale = max(1.0, np.abs(g["gB"]).max())
    np.testing.assert_allclose(N.grad.cpu().numpy()[tril], g["gN"][tril], rtol=1e-5, atol=1e-6 * scale)
    # G = N N^T + R - R^T: only the strictly lower entries of R are parameters (models.py:139-143);
    # a free full matrix R gets dG - dG^T, whose strictly lower part is what the reference reports
    np.testing.assert_allclose(R.grad.cpu().numpy()[stril], g["gR"][stril], rtol=1e-5, atol=1e-6 * scale)
    np.testing.assert_allclose(B.grad.cpu().numpy(), g["gB"], rtol=1e-5, atol=1e-6 * scale)
    np.testing.assert_allclose(L.grad.cpu().numpy(), g["gLambda"], rtol=1e-5, atol=1e-6 * scale)
