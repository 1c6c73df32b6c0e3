"""The LEG reductions with the operand assembly fused into the first pass (cgps_leg_mahal_logdet, csrc/cgps_tile_leg.h:
the block rows of the PEG precision are computed in registers by the lane that eliminates them, reference
models.py:181-239 + :349-367) against the unfused path (cgps_peg_precision -> blocks in memory -> cgps_mahal_logdet),
against the oracle on the reference's recorded operands, and against the reference's recorded log-likelihood."""
import os

import numpy as np
import pytest
import torch

import _util
from oracle import cr_oracle as O
from cyclic_gps import leg
import cyclic_gps.cyclic_reduction as cr

FILES = ["leg_co2like", "leg_small_regular", "leg_small_irregular"]


def _load(name, device="cuda", dtype=torch.float64):
    g = np.load(os.path.join(_util.GOLDEN, name + ".npz"))
    t = lambda k: torch.from_numpy(g[k]).to(dtype).to(device)   # noqa: E731
    return g, leg.LEGMatrices(t("N"), t("R"), t("B"), t("Lambda")), t("ts"), t("xs")


def _model(d, dtype, seed):
    gen = torch.Generator().manual_seed(seed)
    Nm = torch.tril(0.4 * torch.randn(d, d, generator=gen, dtype=torch.float64)) + 0.8 * torch.eye(d, dtype=torch.float64)
    Rm = torch.tril(0.3 * torch.randn(d, d, generator=gen, dtype=torch.float64), -1)
    G = Nm @ Nm.T + Rm - Rm.T + 1e-5 * torch.eye(d, dtype=torch.float64)
    A = torch.randn(d, 2, generator=gen, dtype=torch.float64)
    A = A @ A.T * 0.5
    return G.to(dtype).cuda(), A.to(dtype).cuda(), gen


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_fused_reductions_against_oracle_on_reference_operands(name):
    """K_Rs / K_Os / v / Sig_Rs / Sig_Os recorded from the reference: the oracle reduces THEM; the fused kernel is
    handed ts, G, B^T (LL^T)^-1 B and v only."""
    g, m, ts, xs = _load(name)
    t = lambda k: torch.from_numpy(g[k])   # noqa: E731
    k_m, k_d = O.mahal_and_det(t("K_Rs"), t("K_Os"), t("v"))
    _, s_d = O.mahal_and_det(t("Sig_Rs"), t("Sig_Os"), torch.zeros_like(t("v")))
    BtLB = m.B.T @ m.LLT_inv @ m.B
    fm, fd = leg.leg_mahal_and_det(ts, m.G, BtLB, leg.compute_v(m, xs))
    _, sd = leg.leg_mahal_and_det(ts, m.G)
    np.testing.assert_allclose([float(fm), float(fd), float(sd)], [float(k_m), float(k_d), float(s_d)], rtol=1e-9)
    ll = leg.log_likelihood(m, ts, xs)                       # takes the fused path on the GPU
    assert leg.fused_supported(ts, m.G)
    assert abs(float(ll) - float(g["ll"])) <= 1e-8 * abs(float(g["ll"]))


@pytest.mark.gpu
@pytest.mark.parametrize("d,dtype", [(1, torch.float64), (2, torch.float64), (3, torch.float64), (4, torch.float64),
                                     (5, torch.float64), (7, torch.float64), (2, torch.float32), (4, torch.float32),
                                     (5, torch.float32), (6, torch.float32), (7, torch.float32)],
                         ids=lambda p: str(p).replace("torch.", ""))
def test_fused_against_unfused_every_block_size(d, dtype):
    """Sizes on both sides of every switch: one tile, two tiles (hand-off inside the launch), one row per lane up to
    65 536 rows, several rows per lane beyond; irregular gaps; with and without the diagonal term and the right-hand side."""
    G, A, gen = _model(d, dtype, 100 + d)
    rtol = 1e-9 if dtype == torch.float64 else 3e-4
    for n in (1, 2, 3, 255, 256, 257, 502, 5000, 70001, 300000):
        gaps = 0.05 + 0.5 * torch.rand(n, generator=gen, dtype=torch.float64)
        ts = torch.cumsum(gaps, 0).to(dtype).cuda()
        v = torch.randn(n, d, generator=gen, dtype=torch.float64).to(dtype).cuda()
        Rs, Os = leg.peg_precision(ts, G)
        for Ad, vv in ((A, v), (None, None), (A, None)):
            Rk = Rs if Ad is None else Rs + Ad
            m0, l0 = cr.mahal_and_det(Rk.double(), Os.double(), (torch.zeros_like(v) if vv is None else vv).double())
            m1, l1 = leg.leg_mahal_and_det(ts, G, Ad, vv)
            assert abs(float(l1) - float(l0)) <= rtol * max(1.0, abs(float(l0))), (n, float(l1), float(l0))
            assert abs(float(m1) - float(m0)) <= 10 * rtol * max(1.0, abs(float(m0))), (n, float(m1), float(m0))


@pytest.mark.gpu
@pytest.mark.parametrize("d,dtype,n", [(5, torch.float64, 502), (5, torch.float64, 100), (3, torch.float64, 70001),
                                       (4, torch.float32, 3000), (7, torch.float64, 1000), (2, torch.float64, 1)],
                         ids=lambda p: str(p).replace("torch.", ""))
def test_pair_launch_equals_two_single_launches(d, dtype, n):
    """cgps_leg_mahal_logdet_pair: both reductions of a log-likelihood in one launch (gridDim.y = 2), bit for bit what
    the two single launches give."""
    G, A, gen = _model(d, dtype, 40 + d)
    ts = torch.cumsum(0.05 + torch.rand(n, generator=gen, dtype=torch.float64), 0).to(dtype).cuda()
    v = torch.randn(n, d, generator=gen, dtype=torch.float64).to(dtype).cuda()
    km, kd = leg.leg_mahal_and_det(ts, G, A, v)
    _, sd = leg.leg_mahal_and_det(ts, G)
    for _ in range(3):                                       # (the counters of both systems are back at zero after a call)
        pm, pd, ps = leg.leg_loglik_reductions(ts, G, A, v)
        assert float(pm) == float(km) and float(pd) == float(kd) and float(ps) == float(sd)


@pytest.mark.gpu
def test_fused_regular_grid_large():
    """2^20 rows on a regular grid, rank 5 (the LEG rank of BASELINE config 5): against the unfused path."""
    d, n = 5, 1 << 20
    G, A, gen = _model(d, torch.float64, 7)
    ts = (0.25 * torch.arange(n, dtype=torch.float64)).cuda()
    v = torch.randn(n, d, generator=gen, dtype=torch.float64).cuda()
    Rs, Os = leg.peg_precision(ts, G)
    m0, l0 = cr.mahal_and_det(Rs + A, Os, v)
    m1, l1 = leg.leg_mahal_and_det(ts, G, A, v)
    np.testing.assert_allclose([float(m1), float(l1)], [float(m0), float(l0)], rtol=1e-10)


@pytest.mark.gpu
def test_fused_reports_a_zero_length_gap_and_unsupported_sizes():
    G, A, gen = _model(3, torch.float64, 3)
    ts = torch.cumsum(0.1 + torch.rand(1000, generator=gen, dtype=torch.float64), 0)
    ts[600] = ts[599]
    with pytest.raises(cr.NotPSDError):
        leg.leg_mahal_and_det(ts.cuda(), G, A, None)
    G8 = torch.eye(8, dtype=torch.float64).cuda()
    assert not leg.fused_supported(ts.cuda(), G8)
    from cyclic_gps import _hip
    with pytest.raises(_hip.CgpsError):
        leg.leg_mahal_and_det(ts.cuda(), G8)


@pytest.mark.gpu
def test_fused_log_likelihood_replays_from_a_graph():
    g, m, ts, xs = _load("leg_co2like")
    gll = leg.GraphedLogLikelihood(m, ts, xs)
    for _ in range(3):
        ll = float(gll())
    assert abs(ll - float(g["ll"])) <= 1e-8 * abs(float(g["ll"]))
    xs.mul_(1.01)                                            # new data in place: the replay follows
    ll2 = float(gll())
    ref = float(leg.log_likelihood(m, ts, xs))
    assert abs(ll2 - ref) <= 1e-10 * abs(ref) and abs(ll2 - ll) > 1e-6
