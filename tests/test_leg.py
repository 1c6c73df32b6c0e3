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
@pytest.mark.parametrize("name", FILES)
def test_log_likelihood_replayed_from_a_hip_graph(name):
    """leg.GraphedLogLikelihood: the captured evaluation replays to the reference's value, and follows an
    in-place change of the data and of a parameter (what an optimiser loop does between replays)."""
    g, m, ts, xs = _load(name, device="cuda")
    gll = leg.GraphedLogLikelihood(m, ts, xs)
    for _ in range(3):
        assert abs(float(gll()) - float(g["ll"])) <= 1e-8 * abs(float(g["ll"]))
    xs.mul_(1.25)
    m.Lambda.mul_(1.5)
    eager = float(leg.log_likelihood(m, ts, xs))
    assert abs(eager - float(g["ll"])) > 1e-3 * abs(float(g["ll"]))        # the change matters
    assert abs(float(gll()) - eager) <= 1e-10 * abs(eager)


@pytest.mark.gpu
def test_predictions_replayed_from_a_hip_graph():
    """leg.Graphed(predict.make_predictions, ..., check_sorted=False): posterior (decompose, solve,
    inverse_blocks) and the intercast kernel in one replayable graph."""
    predict, g, m, ts, xs, target_ts, _, _ = _check_predictions("leg_co2like", "cuda")
    gp = leg.Graphed(predict.make_predictions, m, ts, xs, target_ts, check_sorted=False)
    for _ in range(2):
        pm, pv = gp()
        np.testing.assert_allclose(pm.cpu().numpy(), g["pred_mean"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(pv.cpu().numpy(), g["pred_cov"], rtol=1e-6, atol=1e-7)
    xs.mul_(0.5)                                     # new data, same shapes
    pm, _ = gp()
    em, _ = predict.make_predictions(m, ts, xs, target_ts)
    np.testing.assert_allclose(pm.cpu().numpy(), em.cpu().numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
def test_training_evaluation_replayed_from_a_hip_graph():
    """leg.GraphedValueAndGrad: forward + backward captured once; the replayed gradients are the reference's
    autograd gradients, and follow an in-place parameter update."""
    g, m, ts, xs = _load("leg_co2like", device="cuda")
    mg = leg.LEGMatrices(*(t.clone().requires_grad_(True) for t in (m.N, m.R, m.B, m.Lambda)))
    gv = leg.GraphedValueAndGrad(mg, ts, xs)
    for _ in range(2):
        ll, grads = gv()
        assert abs(float(ll.detach()) - float(g["grad_ll"])) <= 1e-8 * abs(float(g["grad_ll"]))
        # (the fixture holds the gradients of the reference's triangular parametrisation)
        for got, key in ((grads[0].tril(), "gN"), (grads[1].tril(-1), "gR"), (grads[2], "gB"), (grads[3].tril(), "gLambda")):
            np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-6, atol=1e-7, err_msg=key)
    with torch.no_grad():
        mg.N.mul_(1.1)
        mg.B.add_(0.05)
    ll, grads = gv()
    replayed = [float(ll.detach())] + [t.clone() for t in grads]
    for p_ in gv.params:
        p_.grad = None
    eager = leg.log_likelihood(mg, ts, xs)
    eager.backward()
    assert abs(replayed[0] - float(eager.detach())) <= 1e-10 * abs(float(eager.detach()))
    for got, p_ in zip(replayed[1:], gv.params):
        np.testing.assert_allclose(got.cpu().numpy(), p_.grad.cpu().numpy(), rtol=1e-9, atol=1e-10)


@pytest.mark.gpu
def test_posterior_mean_fp32_within_1e4():
    """north_star: posterior mean within 1e-4 in fp32."""
    g, m, ts, xs = _load("leg_co2like", device="cuda", dtype=torch.float32)
    mean, _ = leg.insample_posterior(m, ts, xs)
    err = np.abs(mean.cpu().double().numpy() - g["post_mean"]).max()
    assert err <= 1e-4 * max(1.0, np.abs(g["post_mean"]).max()), err


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_hip_operand_assembly_matches_reference(name):
    """cgps_peg_precision (one HIP kernel) against the blocks recorded from the reference."""
    g, m, ts, xs = _load(name, device="cuda")
    Rs, Os = leg._peg_precision_hip(ts, m.G)
    np.testing.assert_allclose(Rs.cpu().numpy(), g["Sig_Rs"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(Os.cpu().numpy(), g["Sig_Os"], rtol=1e-9, atol=1e-11)
    # the dispatcher takes the forward kernel either way; with a gradient wanted it is an autograd function whose
    # backward is the adjoint kernel (cgps_peg_precision_adjoint)
    Rs2, _ = leg.peg_precision(ts, m.G)
    assert Rs2.grad_fn is None and torch.equal(Rs2, Rs)
    Gg = m.G.clone().requires_grad_(True)
    Rs3, _ = leg.peg_precision(ts, Gg)
    assert Rs3.grad_fn is not None
    np.testing.assert_allclose(Rs3.detach().cpu().numpy(), Rs.cpu().numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_hip_operand_assembly_against_torch_ops(d, dtype):
    """Every block size, irregular gaps over three orders of magnitude (scaling and squaring of
    the exponential exercised from s = 0 to s ~ 10), against the batched torch restatement.
    Small gaps make I - E^T E nearly singular (blocks ~ 1/gap), so the two ways of solving it
    (Cholesky here, LU in torch) agree to cond * eps, not to eps: hence the tolerances."""
    gen = torch.Generator().manual_seed(40 + d)
    Nm = 0.6 * torch.randn(d, d, generator=gen, dtype=torch.float64).tril()
    Rm = 0.4 * torch.randn(d, d, generator=gen, dtype=torch.float64).tril(-1)
    G = (Nm @ Nm.T + Rm - Rm.T + 1e-5 * torch.eye(d, dtype=torch.float64)).to(dtype)
    lo = -1.0 if dtype == torch.float64 else 0.0     # fp32: I - E^T E cancels for gaps << 1, in any implementation
    gaps = 10.0 ** (torch.rand(700, generator=gen, dtype=torch.float64) * (2 - lo) + lo)   # 0.1 (1) .. 100
    ts = torch.cat([torch.zeros(1, dtype=torch.float64), gaps.cumsum(0)]).to(dtype)
    Gd = G.double().cuda().requires_grad_(True)                                         # forces the torch-op path
    rRs, rOs = leg.peg_precision(ts.double().cuda(), Gd)
    Rs, Os = leg.peg_precision(ts.cuda(), G.cuda())
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == torch.float64 else dict(rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(Rs.double().cpu().numpy(), rRs.detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(Os.double().cpu().numpy(), rOs.detach().cpu().numpy(), **tol)
    if d == 3 and dtype == torch.float64:                                               # one system of one row
        R1, O1 = leg.peg_precision(ts[:1].cuda(), G.cuda())
        assert R1.shape == (1, d, d) and O1.shape == (0, d, d)
        assert torch.equal(R1[0].cpu(), torch.eye(d, dtype=dtype))


@pytest.mark.gpu
def test_hip_operand_assembly_reports_a_zero_gap():
    g, m, ts, xs = _load("leg_small_regular", device="cuda")
    ts = ts.clone()
    ts[5] = ts[4]
    with pytest.raises(leg.cr.NotPSDError):
        leg.peg_precision(ts, m.G)


# ---- prediction glue (cyclic_gps/predict.py; reference models.py:394-546, model_utils.py:64-107) ----------
def _check_predictions(name, device):
    from cyclic_gps import predict
    g, m, ts, xs = _load(name, device=device)
    target_ts = torch.from_numpy(g["target_ts"]).to(device)
    mean = torch.from_numpy(g["post_mean"]).to(device)
    cov = {"Rs": torch.from_numpy(g["post_cov_Rs"]).to(device), "Os": torch.from_numpy(g["post_cov_Os"]).to(device)}
    return predict, g, m, ts, xs, target_ts, mean, cov


@pytest.mark.parametrize("name", FILES)
def test_intercast_matches_reference_given_its_insample_posterior(name):
    """The batched glue alone (CPU tensors, no kernels): fed the reference's own in-sample posterior it
    reproduces the reference's predictive posterior at every target (backward / forward forecasts,
    exact hits of the first, last and interior observations, interpolation inside the masked gap)."""
    predict, g, m, ts, xs, target_ts, mean, cov = _check_predictions(name, "cpu")
    pm, pv = predict.intercast(m, mean, cov, ts, target_ts)
    np.testing.assert_allclose(pm.numpy(), g["pp_mean"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pv.numpy(), g["pp_cov"], rtol=1e-6, atol=1e-8)
    # the un-batched building blocks, one target at a time like the reference calls them
    k = int(np.searchsorted(g["ts"], g["target_ts"][3]))
    if 0 < k < ts.shape[0]:
        t = target_ts[3]
        e1, e2 = predict.compute_eG(m.G, (t - ts[k - 1])[None])[0], predict.compute_eG(m.G, (ts[k] - t)[None])[0]
        m1, v1 = predict.interpolate(e1, e2, mean[k - 1], cov["Rs"][k - 1], cov["Os"][k - 1], mean[k], cov["Rs"][k])
        np.testing.assert_allclose(m1.numpy(), g["pp_mean"][3], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(v1.numpy(), g["pp_cov"][3], rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_make_predictions_on_gpu(name):
    """End to end on the device: HIP in-sample posterior + batched glue against the reference's
    make_predictions (north_star tolerance 1e-4; fp64 agrees far tighter)."""
    predict, g, m, ts, xs, target_ts, _, _ = _check_predictions(name, "cuda")
    pm, pv = predict.make_predictions(m, ts, xs, target_ts)
    assert pm.device.type == "cuda"
    np.testing.assert_allclose(pm.cpu().numpy(), g["pred_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(pv.cpu().numpy(), g["pred_cov"], rtol=1e-6, atol=1e-7)
    lm, lv = predict.predictive_posterior(m, ts, xs, target_ts)
    np.testing.assert_allclose(lm.cpu().numpy(), g["pp_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(lv.cpu().numpy(), g["pp_cov"], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_hip_intercast_matches_reference_given_its_insample_posterior(name):
    """cgps_leg_intercast alone: fed the reference's own in-sample posterior it reproduces the reference's
    predictive posterior at every target (all five branches occur in the fixtures' targets)."""
    predict, g, m, ts, xs, target_ts, mean, cov = _check_predictions(name, "cuda")
    pm, pv = predict.intercast(m, mean, cov, ts, target_ts)
    assert pm.device.type == "cuda"
    np.testing.assert_allclose(pm.cpu().numpy(), g["pp_mean"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pv.cpu().numpy(), g["pp_cov"], rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("d,dtype", [(1, torch.float64), (2, torch.float64), (3, torch.float32), (4, torch.float64),
                                     (5, torch.float64), (6, torch.float32), (7, torch.float64), (8, torch.float64)])
def test_hip_intercast_against_torch_ops(d, dtype, monkeypatch):
    """Every rank and both dtypes, irregular times, targets before / at / between / exactly on / after the
    observations, n = 1 included: the kernel against the batched torch form of the same glue."""
    from cyclic_gps import predict
    gen = torch.Generator().manual_seed(100 + d)
    Nm = torch.tril(torch.randn(d, d, generator=gen, dtype=torch.float64)) * 0.4 + torch.eye(d, dtype=torch.float64)
    Rm = torch.tril(torch.randn(d, d, generator=gen, dtype=torch.float64), -1) * 0.3
    tol = dict(rtol=1e-9, atol=1e-11) if dtype == torch.float64 else dict(rtol=2e-3, atol=2e-4)
    for n in (1, 2, 37):
        ts = torch.cumsum(0.2 + torch.rand(n, generator=gen, dtype=torch.float64), 0)
        inner = (ts[:-1] + (ts[1:] - ts[:-1]) * torch.rand(max(n - 1, 0), generator=gen, dtype=torch.float64))
        tt = torch.cat([ts[:1] - 1.3, ts[:1] - 0.1, ts[:1], inner, ts[n // 2:n // 2 + 1] if n > 2 else ts[:0],
                        ts[-1:] if n > 1 else ts[:0], ts[-1:] + 0.4, ts[-1:] + 2.0])
        tt = torch.unique(tt)                                           # sorted, strictly increasing
        A = torch.randn(n, d, d, generator=gen, dtype=torch.float64) * 0.2
        Rs = A @ A.transpose(-1, -2) + 0.5 * torch.eye(d, dtype=torch.float64)
        Os = torch.randn(max(n - 1, 0), d, d, generator=gen, dtype=torch.float64) * 0.05
        mu = torch.randn(n, d, generator=gen, dtype=torch.float64)
        m = leg.LEGMatrices(Nm.to(dtype).cuda(), Rm.to(dtype).cuda(), torch.ones(1, d, dtype=dtype).cuda(),
                            torch.ones(1, 1, dtype=dtype).cuda())
        args = (m, mu.to(dtype).cuda(), (Rs.to(dtype).cuda(), Os.to(dtype).cuda()), ts.to(dtype).cuda(), tt.to(dtype).cuda())
        hm, hv = predict.intercast(*args)
        monkeypatch.setenv("CGPS_LEG_TORCH_INTERCAST", "1")
        rm, rv = predict.intercast(*args)
        monkeypatch.delenv("CGPS_LEG_TORCH_INTERCAST")
        np.testing.assert_allclose(hm.cpu().numpy(), rm.cpu().numpy(), err_msg="n=%d" % n, **tol)
        np.testing.assert_allclose(hv.cpu().numpy(), rv.cpu().numpy(), err_msg="n=%d" % n, **tol)


@pytest.mark.gpu
def test_make_predictions_fp32_within_1e4():
    predict, g, m, ts, xs, target_ts, _, _ = _check_predictions("leg_co2like", "cuda")
    m32 = leg.LEGMatrices(*(t.float() for t in (m.N, m.R, m.B, m.Lambda)))
    pm, _ = predict.make_predictions(m32, ts.float(), xs.float(), target_ts.float())
    err = np.abs(pm.cpu().double().numpy() - g["pred_mean"]).max()
    assert err <= 1e-4 * max(1.0, np.abs(g["pred_mean"]).max()), err


# ---- adjoint of the operand assembly (cgps_peg_precision_adjoint) ------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_hip_assembly_adjoint_against_torch_autograd(d, dtype):
    """d loss / d G and d loss / d ts of the assembly kernel's analytic adjoint against autograd through
    the batched torch restatement (matrix_exp + two solves per gap), random upstream gradients."""
    gen = torch.Generator().manual_seed(90 + d)
    Nm = 0.6 * torch.randn(d, d, generator=gen, dtype=torch.float64).tril()
    Rm = 0.4 * torch.randn(d, d, generator=gen, dtype=torch.float64).tril(-1)
    G = Nm @ Nm.T + Rm - Rm.T + 1e-5 * torch.eye(d, dtype=torch.float64)
    gaps = 10.0 ** (torch.rand(300, generator=gen, dtype=torch.float64) * 1.5 - 0.3)       # 0.5 .. 16
    ts = torch.cat([torch.zeros(1, dtype=torch.float64), gaps.cumsum(0)])
    uR = torch.randn(301, d, d, generator=gen, dtype=torch.float64)
    uO = torch.randn(300, d, d, generator=gen, dtype=torch.float64)
    res = {}
    for how in ("torch", "hip"):
        os.environ["CGPS_LEG_TORCH_ASSEMBLY"] = "1" if how == "torch" else "0"
        try:
            dt_ = torch.float64 if how == "torch" else dtype
            Gd = G.to(dt_).cuda().requires_grad_(True)
            td = ts.to(dt_).cuda().requires_grad_(True)
            Rs, Os = leg.peg_precision(td, Gd)
            ((Rs * uR.to(dt_).cuda()).sum() + (Os * uO.to(dt_).cuda()).sum()).backward()
            res[how] = (Gd.grad.double().cpu().numpy(), td.grad.double().cpu().numpy())
        finally:
            os.environ.pop("CGPS_LEG_TORCH_ASSEMBLY", None)
    scale = max(1.0, np.abs(res["torch"][0]).max())
    tol = 1e-8 if dtype == torch.float64 else 5e-3
    assert np.abs(res["hip"][0] - res["torch"][0]).max() <= tol * scale
    assert np.abs(res["hip"][1] - res["torch"][1]).max() <= tol * max(1.0, np.abs(res["torch"][1]).max())


@pytest.mark.gpu
def test_log_likelihood_gradients_match_reference_autograd():
    """A training step through the whole harness on the device (assembly kernel + its adjoint, fused
    solve + log-det + its adjoint) against the parameter gradients the reference's autograd gives
    through its own cyclic reduction (tests/golden/leg_co2like.npz: gN, gR, gB, gLambda)."""
    g, m, ts, xs = _load("leg_co2like", device="cuda")
    N, R, B, Lam = (t.clone().requires_grad_(True) for t in (m.N, m.R, m.B, m.Lambda))
    ll = leg.log_likelihood(leg.LEGMatrices(N, R, B, Lam), ts, xs)
    assert abs(float(ll) - float(g["grad_ll"])) <= 1e-8 * abs(float(g["grad_ll"]))
    ll.backward()
    for got, key in ((N.grad.tril(), "gN"), (R.grad.tril(-1), "gR"), (B.grad, "gB"), (Lam.grad.tril(), "gLambda")):
        ref = g[key]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(ref).max()), err_msg=key)
