"""Regenerate tests/golden/*.npz by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
The .npz files hold inputs and the reference's outputs (data only).  Seeds are
fixed here because the reference's own tests are unseeded
(tests/test_cyclic_reduction.py:151).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

warnings.filterwarnings("ignore", category=UserWarning)


def ref_test_system(rng, n, d):
    """The construction of the reference's own random test
    (tests/test_cyclic_reduction.py:157-176): J = L L^T, L block lower
    bidiagonal with randn blocks and +3 I on the diagonal blocks."""
    Ld = rng.standard_normal((n, d, d)) + 3.0 * np.eye(d)
    Lo = rng.standard_normal((max(n - 1, 0), d, d))
    Rs = np.einsum("nij,nkj->nik", Ld, Ld)
    if n > 1:
        Rs[1:] += np.einsum("nij,nkj->nik", Lo, Lo)
    Os = np.einsum("nij,nkj->nik", Lo, Ld[:-1])
    return Rs, Os


def conditioned_system(rng, n, d):
    """Well-conditioned generator of SURVEY.md section 8(d) (benchmark recipe)."""
    Ld = 1.5 * np.eye(d) + 0.1 * rng.standard_normal((n, d, d))
    Lo = (0.3 / np.sqrt(d)) * rng.standard_normal((max(n - 1, 0), d, d))
    Rs = np.einsum("nij,nkj->nik", Ld, Ld)
    if n > 1:
        Rs[1:] += np.einsum("nij,nkj->nik", Lo, Lo)
    Os = np.einsum("nij,nkj->nik", Lo, Ld[:-1])
    return Rs, Os


def cat(lst, d2):
    lst = [np.asarray(t) for t in lst]
    if not lst:
        return np.zeros((0,) + d2)
    return np.concatenate(lst, axis=0)


def cr_case(ref, rng, n, d, conditioned):
    Rs, Os = (conditioned_system if conditioned else ref_test_system)(rng, n, d)
    v = rng.standard_normal((n, d))
    tR, tO, tv = torch.from_numpy(Rs), torch.from_numpy(Os), torch.from_numpy(v)
    decomp = ref.decompose(tR, tO)
    ms, Ds, Fs, Gs = decomp
    half = ref.halfsolve(decomp, tv)
    sol = ref.solve(decomp, tv)
    mah = ref.mahal(decomp, tv)
    det = ref.det(decomp)
    m2, d2 = ref.mahal_and_det(tR, tO, tv)
    vcrr = [torch.from_numpy(rng.standard_normal((int((m + 1) // 2), d))) for m in ms.tolist()]
    back = ref.backhalfsolve(decomp, vcrr)
    Sd, So = ref.inverse_blocks(decomp)
    out = dict(Rs=Rs, Os=Os, v=v, ms=ms.numpy(),
               Dcat=cat(Ds, (d, d)), Fcat=cat(Fs, (d, d)), Gcat=cat(Gs, (d, d)),
               half=cat(half, (d,)), solve=sol.numpy(), mahal=float(mah), det=float(det),
               mad=np.array([float(m2), float(d2)]),
               vcrr=cat(vcrr, (d,)), back=back.numpy(),
               Sig_diag=Sd.numpy(), Sig_off=So.numpy())
    if n > 1:
        (nn, K, F, G), (R1, O1) = ref.decompose_step(tR, tO)
        out.update(step_D=K.numpy(), step_F=F.numpy(), step_G=G.numpy(),
                   step_R=R1.numpy(), step_O=O1.numpy())
    return out


def helper_case(ref, rng, d, nb, square):
    """The four helper-op cases of tests/test_cyclic_reduction.py:141-144."""
    A = rng.standard_normal((nb, d, d))
    if square:   # U is nb x (nb+1) blocks
        B = rng.standard_normal((nb, d, d))
        x = rng.standard_normal((nb + 1, d))
    else:
        B = rng.standard_normal((nb - 1, d, d))
        x = rng.standard_normal((nb, d))
    y = rng.standard_normal((nb, d))
    S = rng.standard_normal((nb * d, nb * d))
    S = (S @ S.T).reshape(nb, d, nb, d)
    Sd = np.array([S[i, :, i] for i in range(nb)])
    So = np.array([S[i + 1, :, i] for i in range(nb - 1)])
    t = torch.from_numpy
    uut_d, uut_o = ref.UU_T(t(A), t(B))
    ux = ref.Ux(t(A), t(B), t(x))
    utx = ref.U_Tx(t(A), t(B), t(y))
    su_mid, su_hi = ref.SigU(t(Sd), t(So), t(A), t(B))
    utv = ref.UtV_diags(t(A), t(B), su_mid, su_hi)
    a = rng.standard_normal((nb + 1, d))
    b = rng.standard_normal((nb, d))
    il1 = ref.interleave(t(a), t(b))
    il2 = ref.interleave(t(b), t(a))
    il3 = ref.interleave(t(b), t(b.copy()))
    return dict(A=A, B=B, x=x, y=y, Sd=Sd, So=So, uut_d=uut_d.numpy(), uut_o=uut_o.numpy(),
                ux=ux.numpy(), utx=utx.numpy(), su_mid=su_mid.numpy(), su_hi=su_hi.numpy(),
                utv=utv.numpy(), il_a=a, il_b=b, il1=il1.numpy(), il2=il2.numpy(), il3=il3.numpy())


def grad_case(ref, rng, n, d):
    """Autograd of the reference through mahal_and_det and solve (G5)."""
    Rs, Os = conditioned_system(rng, n, d)
    v = rng.standard_normal((n, d))
    w = rng.standard_normal((n, d))
    out = dict(Rs=Rs, Os=Os, v=v, w=w)
    for name in ("mahal", "logdet", "solvedot"):
        tR = torch.from_numpy(Rs).requires_grad_(True)
        tO = torch.from_numpy(Os).requires_grad_(True)
        tv = torch.from_numpy(v).requires_grad_(True)
        if name == "solvedot":
            val = (ref.solve(ref.decompose(tR, tO), tv) * torch.from_numpy(w)).sum()
        else:
            m, ld = ref.mahal_and_det(tR, tO, tv)
            val = m if name == "mahal" else ld
        val.backward()
        out[name] = float(val)
        out["g_%s_R" % name] = tR.grad.numpy()
        out["g_%s_O" % name] = tO.grad.numpy()
        out["g_%s_v" % name] = (tv.grad if tv.grad is not None else torch.zeros_like(tv)).numpy()
    return out


def main():
    ref = _refload.load_reference()
    rng = np.random.default_rng(20240607)
    cases = [(d, n, False) for d in (1, 3) for n in (2, 6, 30, 31, 32, 33)]
    cases += [(1, 1, False), (2, 3, False), (8, 64, True), (2, 1024, True),
              (4, 257, True), (5, 502, True), (4, 1000, True), (3, 129, True), (7, 100, True)]
    for d, n, cond in cases:
        np.savez_compressed(os.path.join(HERE, "cr_d%d_n%d.npz" % (d, n)), **cr_case(ref, rng, n, d, cond))
    for d, nb, sq in [(1, 4, True), (1, 4, False), (2, 3, True), (2, 3, False)]:
        np.savez_compressed(os.path.join(HERE, "helpers_d%d_n%d_%s.npz" % (d, nb, "sq" if sq else "nsq")),
                            **helper_case(ref, rng, d, nb, sq))
    np.savez_compressed(os.path.join(HERE, "grad_d3_n37.npz"), **grad_case(ref, rng, 37, 3))
    print("wrote", len(os.listdir(HERE)), "files in", HERE)


if __name__ == "__main__":
    main()
