"""Pins the CPU oracle (oracle/cr_oracle.py):
  * against golden vectors recorded from the unmodified reference
    (tests/golden/make_golden.py),
  * against the reference's own test oracles, restated: dense Cholesky of the
    recursively even/odd permuted matrix, dense solve / slogdet / inverse
    (reference tests/test_cyclic_reduction.py:147-223),
  * against closed-form known answers (BAB, Schur-block gram;
    reference tests/test_cyclic_reduction.py:243-291).
CPU only.
"""
import numpy as np
import pytest
import torch

import _util
from oracle import cr_oracle as O

CASES = _util.golden_cr_cases()
IDS = ["d%d_n%d" % (d, n) for d, n, _ in CASES]
TOL = dict(rtol=1e-10, atol=1e-11)


def _cat(lst, d):
    lst = [t.numpy() for t in lst]
    return np.concatenate(lst, axis=0) if lst else np.zeros((0, d, d))


@pytest.mark.parametrize("d,n,path", CASES, ids=IDS)
def test_oracle_matches_reference_golden(d, n, path):
    g = np.load(path)
    Rs, Os, v = (torch.from_numpy(g[k]) for k in ("Rs", "Os", "v"))
    ms, Ds, Fs, Gs = dec = O.decompose(Rs, Os)
    assert ms.dtype == torch.int64 and ms.tolist() == g["ms"].tolist() == _util.level_sizes(n)
    np.testing.assert_allclose(_cat(Ds, d), g["Dcat"], **TOL)
    np.testing.assert_allclose(_cat(Fs, d), g["Fcat"].reshape(-1, d, d), **TOL)
    np.testing.assert_allclose(_cat(Gs, d), g["Gcat"].reshape(-1, d, d), **TOL)
    np.testing.assert_allclose(torch.cat(O.halfsolve(dec, v)).numpy(), g["half"], **TOL)
    np.testing.assert_allclose(O.solve(dec, v).numpy(), g["solve"], **TOL)
    np.testing.assert_allclose(float(O.mahal(dec, v)), float(g["mahal"]), rtol=1e-11)
    np.testing.assert_allclose(float(O.det(dec)), float(g["det"]), rtol=1e-11, atol=1e-12)
    m, ld = O.mahal_and_det(Rs, Os, v)
    np.testing.assert_allclose([float(m), float(ld)], g["mad"], rtol=1e-11, atol=1e-12)
    vcrr = _util.split_levels(torch.from_numpy(g["vcrr"]), [(mm + 1) // 2 for mm in ms.tolist()])
    np.testing.assert_allclose(O.backhalfsolve(dec, vcrr).numpy(), g["back"], **TOL)
    Sd, So = O.inverse_blocks(dec)
    np.testing.assert_allclose(Sd.numpy(), g["Sig_diag"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(So.numpy(), g["Sig_off"].reshape(-1, d, d), rtol=1e-9, atol=1e-11)
    if n > 1:
        (nn, D, F, G), (R1, O1) = O.decompose_step(Rs, Os)
        assert nn == n
        for got, key in ((D, "step_D"), (F, "step_F"), (G, "step_G"), (R1, "step_R"), (O1, "step_O")):
            np.testing.assert_allclose(got.numpy(), g[key].reshape(got.shape), **TOL)


@pytest.mark.parametrize("d,n,path", [c for c in CASES if c[0] * c[1] <= 600],
                         ids=[i for i, c in zip(IDS, CASES) if c[0] * c[1] <= 600])
def test_oracle_matches_dense_algebra(d, n, path):
    """The reference's own dense oracle: L = chol(T J T^T) with T the recursive
    even/odd permutation; halfsolve == L^-1 T v, backhalfsolve == (L^T T)^-1 v."""
    g = np.load(path)
    Rs, Os, v = g["Rs"], g["Os"], g["v"]
    J = O.dense_from_blocks(Rs, Os)
    perm = O.crr_order(n)
    T = np.kron(np.eye(n)[perm], np.eye(d))
    L = np.linalg.cholesky(T @ J @ T.T)
    dec = O.decompose(torch.from_numpy(Rs), torch.from_numpy(Os))
    tv = torch.from_numpy(v)
    np.testing.assert_allclose(torch.cat(O.halfsolve(dec, tv)).numpy().ravel(),
                               np.linalg.solve(L, T @ v.ravel()), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(float(O.mahal(dec, tv)), v.ravel() @ np.linalg.solve(J, v.ravel()), rtol=1e-8)
    np.testing.assert_allclose(float(O.det(dec)), np.linalg.slogdet(J)[1], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(O.solve(dec, tv).numpy().ravel(), np.linalg.solve(J, v.ravel()),
                               rtol=1e-7, atol=1e-9)
    ms = dec[0].tolist()
    vcrr = _util.split_levels(torch.from_numpy(g["vcrr"]), [(m + 1) // 2 for m in ms])
    np.testing.assert_allclose(O.backhalfsolve(dec, vcrr).numpy().ravel(),
                               np.linalg.solve(L.T @ T, g["vcrr"].ravel()), rtol=1e-7, atol=1e-9)
    Sig = np.linalg.inv(J).reshape(n, d, n, d)
    Sd, So = O.inverse_blocks(dec)
    np.testing.assert_allclose(Sd.numpy(), np.array([Sig[i, :, i] for i in range(n)]), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(So.numpy(), np.array([Sig[i + 1, :, i] for i in range(n - 1)]).reshape(-1, d, d),
                               rtol=1e-7, atol=1e-9)
    # the factor is the block Cholesky of the permuted matrix: D blocks sit on L's diagonal
    Dcat = np.concatenate([D.numpy() for D in dec[1]])
    for i in range(n):
        np.testing.assert_allclose(Dcat[i], L[i * d:(i + 1) * d, i * d:(i + 1) * d], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("name", ["helpers_d1_n4_sq", "helpers_d1_n4_nsq", "helpers_d2_n3_sq", "helpers_d2_n3_nsq"])
def test_oracle_helpers_match_reference_golden(name):
    g = np.load(_util.GOLDEN + "/" + name + ".npz")
    t = torch.from_numpy
    A, B = t(g["A"]), t(g["B"])
    dg, off = O.UU_T(A, B)
    np.testing.assert_allclose(dg.numpy(), g["uut_d"], **TOL)
    np.testing.assert_allclose(off.numpy(), g["uut_o"], **TOL)
    np.testing.assert_allclose(O.Ux(A, B, t(g["x"])).numpy(), g["ux"], **TOL)
    np.testing.assert_allclose(O.U_Tx(A, B, t(g["y"])).numpy(), g["utx"], **TOL)
    mid, hi = O.SigU(t(g["Sd"]), t(g["So"]), A, B)
    np.testing.assert_allclose(mid.numpy(), g["su_mid"], **TOL)
    np.testing.assert_allclose(hi.numpy(), g["su_hi"], **TOL)
    np.testing.assert_allclose(O.UtV_diags(A, B, mid, hi).numpy(), g["utv"], **TOL)
    a, b = t(g["il_a"]), t(g["il_b"])
    np.testing.assert_array_equal(O.interleave(a, b).numpy(), g["il1"])
    np.testing.assert_array_equal(O.interleave(b, a).numpy(), g["il2"])
    np.testing.assert_array_equal(O.interleave(b, b.clone()).numpy(), g["il3"])


def test_oracle_known_answers_bab():
    """BAB(10, 5, 2) as 1x1 blocks, float32 like the reference test."""
    Rs, Os, det_true, inv_true = _util.bab_blocks(10, 5, 2, dtype=torch.float32)
    dec = O.decompose(Rs, Os)
    assert np.allclose(np.log(det_true), float(O.det(dec)))
    x = torch.rand(10, 1, generator=torch.Generator().manual_seed(3))
    m, ld = O.mahal_and_det(Rs, Os, x)
    assert np.allclose(np.log(det_true), float(ld))
    assert np.allclose(float(x[:, 0].double() @ torch.from_numpy(inv_true) @ x[:, 0].double()), float(m))
    Sd, So = O.inverse_blocks(dec)
    assert np.allclose(Sd.numpy().ravel(), np.diag(inv_true))
    assert np.allclose(So.numpy().ravel(), np.diag(inv_true, -1))


def test_oracle_known_answers_schur_gram():
    Rs, Os, logdet_true, inv_scale = _util.schur_gram_blocks(5, 1.0, 2.0, dtype=torch.float32)
    dec = O.decompose(Rs, Os)
    assert np.allclose(logdet_true, float(O.det(dec)))
    x = torch.rand(5, 2, generator=torch.Generator().manual_seed(4))
    m, ld = O.mahal_and_det(Rs, Os, x)
    assert np.allclose(logdet_true, float(ld))
    assert np.allclose(inv_scale * float((x * x).sum()), float(m))
    Sd, So = O.inverse_blocks(dec)
    assert np.allclose(Sd.numpy(), inv_scale * np.eye(2)[None].repeat(5, 0))
    assert np.allclose(So.numpy(), 0.0)


def test_oracle_gradients_match_reference_autograd():
    """Closed-form adjoints used by the product's backward,
    checked against the reference's autograd (golden grad_d3_n37.npz):
      m = v^T J^-1 v, w = J^-1 v:  dm/dv = 2w, dm/dR_i = -w_i w_i^T, dm/dO_i = -2 w_{i+1} w_i^T
      l = log|J|, Sig = J^-1:      dl/dR_i = Sig_ii,  dl/dO_i = 2 Sig_{i+1,i}
      s = u^T J^-1 v, a = J^-1 u:  ds/dv = a, ds/dR_i = -a_i w_i^T, ds/dO_i = -(a_{i+1} w_i^T + w_{i+1} a_i^T)
    """
    g = np.load(_util.GOLDEN + "/grad_d3_n37.npz")
    Rs, Os, v, u = (torch.from_numpy(g[k]) for k in ("Rs", "Os", "v", "w"))
    dec = O.decompose(Rs, Os)
    w = O.solve(dec, v)
    a = O.solve(dec, u)
    Sd, So = O.inverse_blocks(dec)
    T = dict(rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(2 * w.numpy(), g["g_mahal_v"], **T)
    np.testing.assert_allclose(-torch.einsum("ni,nj->nij", w, w).numpy(), g["g_mahal_R"], **T)
    np.testing.assert_allclose(-2 * torch.einsum("ni,nj->nij", w[1:], w[:-1]).numpy(), g["g_mahal_O"], **T)
    np.testing.assert_allclose(Sd.numpy(), g["g_logdet_R"], **T)
    np.testing.assert_allclose(2 * So.numpy(), g["g_logdet_O"], **T)
    np.testing.assert_allclose(a.numpy(), g["g_solvedot_v"], **T)
    # the reference's autograd does not symmetrise dR; compare symmetric parts
    gR = g["g_solvedot_R"]
    mine = -torch.einsum("ni,nj->nij", a, w).numpy()
    np.testing.assert_allclose(0.5 * (mine + mine.transpose(0, 2, 1)), 0.5 * (gR + gR.transpose(0, 2, 1)), **T)
    np.testing.assert_allclose(-(torch.einsum("ni,nj->nij", a[1:], w[:-1])
                                 + torch.einsum("ni,nj->nij", w[1:], a[:-1])).numpy(), g["g_solvedot_O"], **T)


def test_oracle_closed_form_large():
    """Conditioned generator: logdet and planted solution are closed-form."""
    Rs, Os, b, x_true, logdet = _util.conditioned_system(4099, 4)
    m, ld = O.mahal_and_det(Rs, Os, b)
    assert abs(float(ld) - logdet) <= 1e-12 * abs(logdet)
    assert abs(float(m) - float((x_true * b).sum())) <= 1e-11 * abs(float(m))
    x = O.solve(O.decompose(Rs, Os), b)
    assert float((x - x_true).abs().max()) < 1e-11


def test_oracle_rejects_non_pd():
    Rs = -torch.eye(2, dtype=torch.float64).repeat(3, 1, 1)
    Os = torch.zeros(2, 2, 2, dtype=torch.float64)
    with pytest.raises(O.NotPSDError):
        O.decompose(Rs, Os)
