"""BASELINE config 5: LEG marginal likelihood + posterior mean on the CO2-shaped workload
(cyclic_gps/leg.py) against vectors recorded from the reference's LEGFamily
(tests/golden/make_golden_leg.py).  north_star tolerance: 1e-4; fp64 results here agree to ~1e-9.
"""
import math
import os

import numpy as np
import pytest
import torch

import _util
from oracle import cr_oracle as O
from cyclic_gps import leg

FILES = ["leg_co2like", "leg_small_regular", "leg_small_irregular"]


def _load(name, device="cpu", dtype=torch.float64):
    g = np.load(os.path.join(_util.GOLDEN, name + ".npz"))
    t = lambda k: torch.from_numpy(g[k]).to(dtype).to(device)   # noqa: E731
    m = leg.LEGMatrices(t("N"), t("R"), t("B"), t("Lambda"))
    return g, m, t("ts"), t("xs")


@pytest.mark.parametrize("name", FILES)
def test_operand_assembly_matches_reference(name):
    """The operands handed to the cyclic reduction are the reference's (CPU, no kernels involved)."""
    g, m, ts, xs = _load(name)
    np.testing.assert_allclose(m.G.numpy(), g["G"], rtol=1e-12, atol=1e-14)
    Rs, Os = leg.peg_precision(ts, m.G)
    np.testing.assert_allclose(Rs.numpy(), g["Sig_Rs"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(Os.numpy(), g["Sig_Os"], rtol=1e-9, atol=1e-11)
    K_Rs, K_Os = leg.posterior_precision(m, ts)
    np.testing.assert_allclose(K_Rs.numpy(), g["K_Rs"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(K_Os.numpy(), g["K_Os"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(leg.compute_v(m, xs).numpy(), g["v"], rtol=1e-9, atol=1e-11)
    # the likelihood formula with the oracle doing the reductions
    LLT = m.LLT
    xl = torch.linalg.solve(LLT, xs.T).T
    k_m, k_d = O.mahal_and_det(K_Rs, K_Os, leg.compute_v(m, xs))
    _, s_d = O.mahal_and_det(Rs, Os, torch.zeros_like(g_v := leg.compute_v(m, xs)))
    ll = -0.5 * (((xl * xs).sum() - k_m) + (torch.logdet(2 * math.pi * LLT) * xs.shape[0] + k_d - s_d))
    assert abs(float(ll) - float(g["ll"])) <= 1e-9 * abs(float(g["ll"]))
    if "naive_ll" in g.files:
        assert abs(float(ll) - float(g["naive_ll"])) <= 1e-8 * abs(float(g["naive_ll"]))
    del g_v


def test_co2_workload_shape():
    ts, xs, tts, txs = leg.co2_workload()
    assert ts.shape[0] == 770 and tts.shape[0] == 502 and txs.shape == (502, 1)
    assert float(tts[0]) == 0.0 and abs(float(xs.mean())) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_likelihood_and_posterior_on_gpu(name):
    g, m, ts, xs = _load(name, device="cuda")
    ll = leg.log_likelihood(m, ts, xs)
    assert ll.device.type == "cuda"
    assert abs(float(ll) - float(g["ll"])) <= 1e-8 * abs(float(g["ll"]))
    mean, (cRs, cOs) = leg.insample_posterior(m, ts, xs)
    np.testing.assert_allclose(mean.cpu().numpy(), g["post_mean"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(cRs.cpu().numpy(), g["post_cov_Rs"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(cOs.cpu().numpy(), g["post_cov_Os"], rtol=1e-7, atol=1e-9)


@pytest.mark.gpu
def test_posterior_mean_fp32_within_1e4():
    """north_star: posterior mean within 1e-4 in fp32."""
    g, m, ts, xs = _load("leg_co2like", device="cuda", dtype=torch.float32)
    mean, _ = leg.insample_posterior(m, ts, xs)
    err = np.abs(mean.cpu().double().numpy() - g["post_mean"]).max()
    assert err <= 1e-4 * max(1.0, np.abs(g["post_mean"]).max()), err
