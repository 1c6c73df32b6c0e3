"""Shared helpers for the tests: golden loading, synthetic systems, closed forms."""
import glob
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def golden_cr_cases():
    """[(d, n, path)] of all tests/golden/cr_d*_n*.npz, smallest first."""
    out = []
    for p in glob.glob(os.path.join(GOLDEN, "cr_d*_n*.npz")):
        b = os.path.basename(p)[:-4].split("_")
        out.append((int(b[1][1:]), int(b[2][1:]), p))
    return sorted(out, key=lambda t: (t[1], t[0]))


def level_sizes(n):
    ms = []
    while n > 1:
        ms.append(n)
        n //= 2
    ms.append(1)
    return ms


def split_levels(cat, counts):
    out, o = [], 0
    for c in counts:
        out.append(cat[o:o + c])
        o += c
    assert o == cat.shape[0], (o, cat.shape)
    return out


def golden_factor(g):
    """Rebuild (ms, Ds, Fs, Gs) torch lists from a golden npz."""
    ms = [int(m) for m in g["ms"]]
    Ds = split_levels(torch.from_numpy(g["Dcat"]), [(m + 1) // 2 for m in ms])
    Fs = split_levels(torch.from_numpy(g["Fcat"]), [m // 2 for m in ms[:-1]])
    Gs = split_levels(torch.from_numpy(g["Gcat"]), [(m - 1) // 2 for m in ms[:-1]])
    return torch.tensor(ms), Ds, Fs, Gs


def conditioned_system(n, d, dtype=torch.float64, seed=1234, device="cpu"):
    """Benchmark generator of SURVEY.md section 8(d): J = L L^T with L block
    lower bidiagonal, so logdet J and the planted solution are closed-form.
    Returns Rs, Os, b, x_true, logdet_true (python float, fp64)."""
    g = torch.Generator(device=device).manual_seed(seed)
    kw = dict(dtype=torch.float64, device=device, generator=g)
    Ld = 1.5 * torch.eye(d, dtype=torch.float64, device=device) + 0.1 * torch.randn(n, d, d, **kw)
    Lo = (0.3 / d ** 0.5) * torch.randn(max(n - 1, 0), d, d, **kw)
    Rs = Ld @ Ld.transpose(-1, -2)
    Rs[1:] += Lo @ Lo.transpose(-1, -2)
    Os = Lo @ Ld[:-1].transpose(-1, -2)
    x_true = torch.randn(n, d, **kw)
    b = torch.einsum("nij,nj->ni", Rs, x_true)
    b[1:] += torch.einsum("nij,nj->ni", Os, x_true[:-1])
    b[:-1] += torch.einsum("nji,nj->ni", Os, x_true[1:])
    logdet = 2.0 * float(_sum_log_abs_det(Ld))
    return Rs.to(dtype), Os.to(dtype), b.to(dtype), x_true.to(dtype), logdet


def _sum_log_abs_det(A):
    """sum_i log|det A_i| by unpivoted elimination vectorised over the batch (the blocks
    here are 1.5 I + small noise, so no pivoting is needed); plain tensor ops only, so it
    runs on the GPU without any solver library."""
    A = A.clone()
    d = A.shape[-1]
    total = torch.zeros((), dtype=A.dtype, device=A.device)
    for j in range(d):
        piv = A[:, j, j]
        total = total + torch.log(piv.abs()).sum()
        if j + 1 < d:
            A[:, j + 1:, :] -= (A[:, j + 1:, j:j + 1] / piv[:, None, None]) * A[:, j:j + 1, :]
    return total


def bab_blocks(n, alpha, beta, dtype=torch.float64):
    """BAB(n, alpha, beta) (tridiagonal Toeplitz) as 1x1 blocks, plus its
    closed-form determinant (three-term recurrence) and inverse
    (Chebyshev-U formula) -- the known answers the reference tests use
    (tests/test_cyclic_reduction.py:246-267)."""
    Rs = torch.full((n, 1, 1), float(alpha), dtype=dtype)
    Os = torch.full((n - 1, 1, 1), float(beta), dtype=dtype)
    dets = [1.0, float(alpha)]
    for _ in range(2, n + 1):
        dets.append(alpha * dets[-1] - beta * beta * dets[-2])
    x = 0.5 * alpha / beta
    U = [1.0, 2 * x]
    for _ in range(2, n + 1):
        U.append(2 * x * U[-1] - U[-2])
    inv = np.zeros((n, n))
    for i in range(1, n + 1):
        for j in range(1, n + 1):
            lo, hi = min(i, j), max(i, j)
            inv[i - 1, j - 1] = (-1.0) ** (i + j) * U[lo - 1] * U[n - hi] / (U[n] * beta)
    return Rs, Os, dets[n], inv


def schur_gram_blocks(nb, xv, yv, dtype=torch.float64):
    """Gram matrix S^T S of the SCHUR_BLOCK matrix with nb 2x2 blocks
    [[x, y], [-y, x]] (tests/test_cyclic_reduction.py:269-291): it equals
    (x^2+y^2) I per block, so J^-1 = I/(x^2+y^2), log det J = 2 nb log(x^2+y^2)."""
    s = float(xv * xv + yv * yv)
    Rs = (s * torch.eye(2, dtype=dtype)).repeat(nb, 1, 1)
    Os = torch.zeros(nb - 1, 2, 2, dtype=dtype)
    return Rs, Os, 2 * nb * np.log(s), 1.0 / s
