"""cgps_decompose_solve / cyclic_reduction.decompose_solve: decompose(Rs, Os) and solve(decomp, y) in one call
(reference cyclic_reduction.py:287-309 + :441-444, as compute_insample_posterior uses them, models.py:288-292), the
forward substitution riding along in the first pass of the factorisation.  The factor must be the factor of
decompose bit for bit, the solution that of solve (and the planted one), at sizes on both sides of the switch
between the plain sequence (small systems) and the fused first pass (>= 512 tiles of 128 rows), ragged included."""
import numpy as np
import pytest
import torch

import _util
from oracle import cr_oracle as O
import cyclic_gps.cyclic_reduction as cr

CASES = [(1, torch.float64), (2, torch.float64), (3, torch.float64), (4, torch.float64), (5, torch.float64),
         (4, torch.float32), (8, torch.float32), (6, torch.float64)]


@pytest.mark.gpu
@pytest.mark.parametrize("d,dtype", CASES, ids=lambda p: str(p).replace("torch.", ""))
def test_decompose_solve_equals_decompose_then_solve(d, dtype):
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    for n in (1, 2, 5, 1000, 65535, 65536, 65536 + 77, 131072 + 8 * 13 + 5, 300001, 2 ** 20):
        Rs, Os, b, x_true, _ = _util.conditioned_system(n, d, dtype=dtype, device="cuda", seed=n % 97)
        dec0 = cr.decompose(Rs, Os)
        x0 = cr.solve(dec0, b)
        dec1, x1 = cr.decompose_solve(Rs, Os, b)
        assert [int(m) for m in dec1[0]] == [int(m) for m in dec0[0]]
        for k in (1, 2, 3):
            for a, c in zip(dec0[k], dec1[k]):
                assert torch.equal(a, c), (n, k)
        assert float((x1 - x0).abs().max()) <= tol, (n, float((x1 - x0).abs().max()))
        assert float((x1.double() - x_true.double()).abs().max()) <= (1e-9 if dtype == torch.float64 else 2e-3)
        # the factor is usable afterwards as any other
        x2 = cr.solve(dec1, 2 * b)
        assert float((x2 - 2 * x0).abs().max()) <= 2 * tol


@pytest.mark.gpu
def test_decompose_solve_against_oracle_and_golden():
    g = np.load(_util.golden_path("cr_d4_n1000.npz")) if hasattr(_util, "golden_path") else None
    for n, d in ((257, 4), (1000, 3), (70001, 2)):
        Rs, Os, b, _, _ = _util.conditioned_system(n, d, seed=3)
        x_or = O.solve(O.decompose(Rs, Os), b)
        _, x = cr.decompose_solve(Rs.cuda(), Os.cuda(), b.cuda())
        np.testing.assert_allclose(x.cpu().numpy(), x_or.numpy(), rtol=1e-9, atol=1e-11)
    del g


@pytest.mark.gpu
def test_decompose_solve_reports_a_block_that_is_not_positive_definite():
    Rs, Os, b, _, _ = _util.conditioned_system(2 ** 17, 4, device="cuda", seed=2)
    Rs[77777] = -Rs[77777]
    with pytest.raises(cr.NotPSDError):
        cr.decompose_solve(Rs, Os, b)
