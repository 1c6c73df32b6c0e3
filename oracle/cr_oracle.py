"""CPU oracle for the block-tridiagonal cyclic-reduction hot path.

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The shipped path (``cyclic-gps_amd/``) never does;
it runs the HIP library or fails loudly.

It restates, in plain torch-on-CPU, the algorithm of the reference's
``cyclic_gps/cyclic_reduction.py`` (all line numbers below are in that file),
with the same batched op sequence the reference uses (batched Cholesky, batched
triangular solves, batched small mat-mats over strided even/odd views), so
that its wall time is also a fair stand-in for the reference's own CPU time
(``cpu_baseline.kind = "port"``).

Parity pin (see DESIGN.md "Oracle"): this restatement is checked in
``tests/test_oracle.py`` against
  * golden vectors produced by importing the unmodified reference in the build
    container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``),
  * the reference's own test oracles, restated: dense Cholesky of the
    recursively even/odd permuted matrix, dense solve / slogdet / inverse
    (``tests/test_cyclic_reduction.py:147-223``), and the closed-form BAB and
    Schur-block known answers (``tests/test_cyclic_reduction.py:243-291``).
The only third-party arithmetic the reference calls on this path that is not
under /root/reference is gpytorch's ``psd_safe_cholesky`` (unpinned version);
on positive-definite input it is ``torch.linalg.cholesky``.  Its jitter-retry
failure path is NOT restated here: "parity unpinned" for non-PD input.

Notation: J is SPD block tridiagonal with n diagonal
blocks R_i = J[i,i] (d x d) and n-1 lower off-diagonal blocks O_i = J[i+1,i].
One level eliminates the even-indexed block rows:
    D_k = chol(R_2k)                          ceil(n/2) blocks
    F_k = O_2k     D_k^-T                      floor(n/2)       (odd row 2k+1 <- even 2k)
    G_k = O_2k+1^T D_k+1^-T                    floor((n-1)/2)   (odd row 2k+1 <- even 2k+2)
    R'_k = R_2k+1 - F_k F_k^T - G_k G_k^T      floor(n/2)
    O'_k = -F_k+1 G_k^T                        floor(n/2) - 1
"""
from __future__ import annotations

import numpy as np
import torch


class NotPSDError(RuntimeError):
    """A diagonal block was not positive definite (cf. gpytorch NotPSDError)."""


def _chol(A: torch.Tensor) -> torch.Tensor:
    L, info = torch.linalg.cholesky_ex(A)
    if bool((info != 0).any()):
        raise NotPSDError("block not positive definite at batch index %d"
                          % int(torch.nonzero(info.reshape(-1))[0]))
    return L


def _rsolve_lt(D: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """X = B D^-T for lower-triangular D (batched)."""
    return torch.linalg.solve_triangular(D.transpose(-1, -2), B, upper=True, left=False)


def _lsolve(D: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x = D^-1 y for lower-triangular D, y of shape [b, d]."""
    return torch.linalg.solve_triangular(D, y.unsqueeze(-1), upper=False)[..., 0]


def _lsolve_t(D: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x = D^-T y for lower-triangular D, y of shape [b, d]."""
    return torch.linalg.solve_triangular(D.transpose(-1, -2), y.unsqueeze(-1), upper=True)[..., 0]


# ----------------------------------------------------------------------------
# banded helper products with the block bidiagonal U (diag F, super-diag G)
# reference: UU_T :15-37, Ux :40-60, U_Tx :63-87, SigU :90-136, UtV_diags :139-178,
# interleave :181-200
# ----------------------------------------------------------------------------
def UU_T(diags, offdiags):
    """Block-tridiagonal part of U U^T: (diag blocks, lower off-diag blocks)."""
    F, G = diags, offdiags
    nf, ng = F.shape[0], G.shape[0]
    dg = F @ F.transpose(-1, -2)
    dg[:ng] = dg[:ng] + G @ G.transpose(-1, -2)
    npair = min(nf - 1, ng)
    off = F[1:1 + npair] @ G[:npair].transpose(-1, -2)
    return dg, off


def Ux(diags, offdiags, x):
    """U @ x; x has one more block than F when U is 'non-square' (nf == ng)."""
    F, G = diags, offdiags
    nf, ng = F.shape[0], G.shape[0]
    out = torch.einsum("bij,bj->bi", F, x[:nf])
    out[:ng] = out[:ng] + torch.einsum("bij,bj->bi", G, x[1:1 + ng])
    return out


def U_Tx(diags, offdiags, x):
    """U^T @ x; result has nf+1 blocks when nf == ng, else nf."""
    F, G = diags, offdiags
    nf, ng = F.shape[0], G.shape[0]
    ncol = nf + 1 if nf == ng else nf
    out = x.new_zeros((ncol,) + x.shape[1:])
    out[:nf] = torch.einsum("bji,bj->bi", F, x)
    out[1:1 + ng] = out[1:1 + ng] + torch.einsum("bji,bj->bi", G, x[:ng])
    return out


def SigU(sig_dblocks, sig_offdblocks, u_dblocks, u_offdblocks):
    """Diagonal and upper-diagonal blocks of Sig @ U (Sig symmetric block
    tridiagonal given by its diagonal and LOWER off-diagonal blocks)."""
    S, So, A, B = sig_dblocks, sig_offdblocks, u_dblocks, u_offdblocks
    nf, ng = A.shape[0], B.shape[0]
    mid = S @ A                                     # (Sig U)[k,k] = S_kk A_k + S_k,k-1 B_k-1
    mid[1:] = mid[1:] + So @ B[: nf - 1]
    hi = S[:ng] @ B                                 # (Sig U)[k,k+1] = S_kk B_k + S_k+1,k^T A_k+1
    npair = min(ng, nf - 1)
    hi[:npair] = hi[:npair] + So[:npair].transpose(-1, -2) @ A[1:1 + npair]
    return mid, hi


def UtV_diags(u_dblocks, u_offdblocks, v_dblocks, v_offdblocks):
    """Diagonal blocks of U^T V for two block upper-bidiagonal matrices."""
    A, B, Va, Vb = u_dblocks, u_offdblocks, v_dblocks, v_offdblocks
    nf, ng = A.shape[0], B.shape[0]
    ncol = nf + 1 if nf == ng else nf
    out = A.new_zeros((ncol,) + A.shape[1:])
    out[:nf] = A.transpose(-1, -2) @ Va
    out[1:1 + ng] = out[1:1 + ng] + B.transpose(-1, -2) @ Vb
    return out


def interleave(a, b):
    """V[::2] = a ; V[1::2] = b, lengths differing by at most one."""
    n, m = a.shape[0], b.shape[0]
    out = a.new_empty((n + m,) + tuple(a.shape[1:]))
    k = min(n, m)
    out[0:2 * k:2] = a[:k]
    out[1:2 * k:2] = b[:k]
    if n > m:
        out[2 * k:] = a[k:]
    elif m > n:
        out[2 * k:] = b[k:]
    return out


# ----------------------------------------------------------------------------
# one level, the full factorisation and the operations on it
# ----------------------------------------------------------------------------
def decompose_step(Rs, Os):
    """One reduction level (reference :203-259)."""
    n = Rs.shape[0]
    assert n == Os.shape[0] + 1
    nf, ng = n // 2, (n - 1) // 2
    D = _chol(Rs[0::2])
    F = _rsolve_lt(D[:nf], Os[0::2])
    G = _rsolve_lt(D[1:1 + ng], Os[1::2].transpose(-1, -2))
    dg, off = UU_T(F, G)
    return (n, D, F, G), (Rs[1::2] - dg, -off)


def decompose(Rs, Os):
    """Full factorisation (reference :287-309): (ms, Ds, Fs, Gs)."""
    ms, Ds, Fs, Gs = [], [], [], []
    while Rs.shape[0] > 1:
        (n, D, F, G), (Rs, Os) = decompose_step(Rs, Os)
        ms.append(n), Ds.append(D), Fs.append(F), Gs.append(G)
    Ds.append(_chol(Rs))
    ms.append(1)
    return torch.tensor(ms), Ds, Fs, Gs


def halfsolve(decomp, y):
    """L^-1 (T y) as a per-level list (reference :312-338)."""
    ms, Ds, Fs, Gs = decomp
    out = []
    for lvl in range(len(Ds)):
        x = _lsolve(Ds[lvl], y[0::2])
        out.append(x)
        if y.shape[0] == 1:
            break
        y = y[1::2] - Ux(Fs[lvl], Gs[lvl], x)
    return out


def backhalfsolve(decomp, ycrr):
    """T^T L^-T applied to a per-level list (reference :341-377)."""
    ms, Ds, Fs, Gs = decomp
    x = _lsolve_t(Ds[-1], ycrr[-1])
    for lvl in range(len(Ds) - 2, -1, -1):
        xe = _lsolve_t(Ds[lvl], ycrr[lvl] - U_Tx(Fs[lvl], Gs[lvl], x))
        x = interleave(xe, x)
    return x


def solve(decomp, y):
    """J^-1 y (reference :441-444)."""
    return backhalfsolve(decomp, halfsolve(decomp, y))


def det(decomp):
    """log|J| = 2 sum log diag(D) over all levels (reference :447-458)."""
    Ds = decomp[1]
    return 2 * sum(torch.log(torch.diagonal(D, dim1=-2, dim2=-1)).sum() for D in Ds)


def mahal(decomp, y):
    """y^T J^-1 y (reference :461-467)."""
    return sum((x * x).sum() for x in halfsolve(decomp, y))


def mahal_and_det(Rs, Os, x):
    """(x^T J^-1 x, log|J|) in one sweep without keeping the factor
    (reference :380-438)."""
    y = x
    half_logdet = 0
    m = 0
    while True:
        n = Rs.shape[0]
        if n > 1:
            (_, D, F, G), (Rs, Os) = decompose_step(Rs, Os)
        else:
            D = _chol(Rs)
        half_logdet = half_logdet + torch.log(torch.diagonal(D, dim1=-2, dim2=-1)).sum()
        z = _lsolve(D, y[0::2])
        m = m + (z * z).sum()
        if n == 1:
            break
        y = y[1::2] - Ux(F, G, z)
    return m, 2 * half_logdet


def inverse_blocks(decomp):
    """Diagonal and lower off-diagonal blocks of J^-1 (reference :470-503).

    Bottom-up: with W = U D^-1 (diag A_k = F_k D_k^-1, super-diag B_k = G_k D_k+1^-1)
    and M = Sig~ W,
        Sig[2k+1,2k+1] = Sig~[k,k]
        Sig[2k,2k]     = D_k^-T D_k^-1 + A_k^T M[k,k] + B_k-1^T M[k-1,k]
        Sig[2k+1,2k]   = -M[k,k]
        Sig[2k+2,2k+1] = -M[k,k+1]^T
    """
    ms, Ds, Fs, Gs = decomp
    d = Ds[-1].shape[-1]
    eye = torch.eye(d, dtype=Ds[-1].dtype)
    Di = torch.linalg.solve_triangular(Ds[-1], eye.expand_as(Ds[-1]), upper=False)
    Sd = Di.transpose(-1, -2) @ Di
    So = Sd.new_zeros((0, d, d))
    for lvl in range(len(Ds) - 2, -1, -1):
        D, F, G = Ds[lvl], Fs[lvl], Gs[lvl]
        nf, ng = F.shape[0], G.shape[0]
        Di = torch.linalg.solve_triangular(D, eye.expand_as(D), upper=False)
        A = F @ Di[:nf]
        B = G @ Di[1:1 + ng]
        Mmid, Mhi = SigU(Sd, So, A, B)
        See = Di.transpose(-1, -2) @ Di + UtV_diags(A, B, Mmid, Mhi)
        Sd, So = interleave(See, Sd), interleave(-Mmid, -Mhi.transpose(-1, -2))
    return Sd, So


# ----------------------------------------------------------------------------
# dense helpers used by the tests (the reference's own test oracles, restated)
# ----------------------------------------------------------------------------
def crr_order(n: int) -> np.ndarray:
    """Recursive even/odd elimination order of n block rows
    (tests/test_cyclic_reduction.py:15-25): position -> original row."""
    idx = np.arange(n)
    out = []
    while idx.size > 0:
        out.append(idx[0::2])
        idx = idx[1::2]
    return np.concatenate(out)


def dense_from_blocks(Rs, Os) -> np.ndarray:
    Rs = np.asarray(Rs)
    Os = np.asarray(Os)
    n, d = Rs.shape[0], Rs.shape[1]
    J = np.zeros((n, d, n, d), dtype=Rs.dtype)
    for i in range(n):
        J[i, :, i, :] = Rs[i]
    for i in range(n - 1):
        J[i + 1, :, i, :] = Os[i]
        J[i, :, i + 1, :] = Os[i].T
    return J.reshape(n * d, n * d)


def level_sizes(n: int):
    """ms of the reference factor: n, n//2, ..., 1."""
    ms = []
    while n > 1:
        ms.append(n)
        n //= 2
    ms.append(1)
    return ms
