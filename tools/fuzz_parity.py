"""Randomised closed-form parity sweep on the GPU (no oracle needed: J = L L^T with L block
bidiagonal, so log|J| and the planted solution are known exactly -- tests/_util.conditioned_system).

Sizes are drawn around every switch point of the kernels (rows per lane 1 / 4 / 8 / 16, one or
two workgroups per CU, record passes, decompose / solve pass structures) and uniformly in between.
Also: decompose_solve (factor bit-identical to decompose's, planted solution) and, every third case, the LEG
reductions with the operands assembled in registers against the blocks-in-memory path.

  python tools/fuzz_parity.py --seconds 120 [--seed 0] [--d 4 8]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import _util  # noqa: E402
import cyclic_gps.cyclic_reduction as cr  # noqa: E402
from cyclic_gps import leg  # noqa: E402

EDGES = [1, 2, 3, 64, 128, 129, 255, 256, 257, 1024, 4096, 4097, 16384, 32768, 32769, 65536, 65537, 131072,
         262144, 262145, 524288, 524289, 1048576, 1048577, 1 << 21, (1 << 21) + 1]
BIG_EDGES = [1 << 22, 5600003, 1 << 23, 8400001]       # beyond one round of the chip: rows per lane chosen at launch


def draw_n(rng, cap):
    r = rng.random()
    if cap > (1 << 22) and r < 0.04:
        return min(rng.choice(BIG_EDGES) + rng.choice([-5, 0, 0, 3, 1000]), cap)
    if r < 0.5:
        e = rng.choice(EDGES)
        n = e + rng.choice([-3, -2, -1, 0, 0, 1, 2, 3, 17, -17, 255, -255])
    elif r < 0.8:
        n = int(2 ** rng.uniform(0, 20.5))
    else:
        n = rng.randrange(1, 5000)
    return max(1, min(n, cap))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--d", type=int, nargs="*", default=None, help="block sizes to draw from (default: 1..8, 4 weighted)")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    t0 = time.time()
    cases = worst64 = worst32 = 0
    while time.time() - t0 < a.seconds:
        d = rng.choice(a.d if a.d else [1, 2, 3, 4, 4, 4, 5, 6, 7, 8])
        dtype = rng.choice([torch.float64, torch.float64, torch.float32])
        n = draw_n(rng, 8500000 if d <= 5 else (1 << 19) + 300)
        Rs, Os, b, x_true, logdet = _util.conditioned_system(n, d, dtype=dtype, device="cuda", seed=rng.randrange(1 << 30))
        mahal_true = float((x_true.double() * b.double()).sum())
        tol = 1e-9 if dtype == torch.float64 else 3e-4
        tag = "n=%d d=%d %s" % (n, d, "f64" if dtype == torch.float64 else "f32")
        m, ld = cr.mahal_and_det(Rs, Os, b)
        e1 = abs(float(ld) - logdet) / max(1.0, abs(logdet))
        e2 = abs(float(m) - mahal_true) / max(1.0, abs(mahal_true))
        dec = cr.decompose(Rs, Os)
        x = cr.solve(dec, b)
        e3 = float((x.double() - x_true.double()).abs().max())
        e4 = abs(float(cr.det(dec)) - logdet) / max(1.0, abs(logdet))
        # selected inverse: the block-diagonal of J Sigma = I only involves the blocks inverse_blocks
        # returns: R_i S[i,i] + O_i-1 S[i,i-1]^T + O_i^T S[i+1,i] = I
        Sd, So = cr.inverse_blocks(dec)
        res = Rs @ Sd
        if n > 1:
            res[1:] += Os @ So.transpose(1, 2)
            res[:-1] += Os.transpose(1, 2) @ So
        res -= torch.eye(d, dtype=dtype, device="cuda")
        e5 = float(res.abs().max()) * (1.0 if dtype == torch.float64 else 0.1)
        # factor and solve in one call: the same factor bit for bit, the same solution
        dec2, x2 = cr.decompose_solve(Rs, Os, b)
        same = all(torch.equal(p_, q_) for k_ in (1, 2, 3) for p_, q_ in zip(dec[k_], dec2[k_]))
        e6 = float((x2.double() - x_true.double()).abs().max()) if same else float("inf")
        # the LEG reductions with the operands assembled in registers against the blocks-in-memory path (every third case)
        e7 = 0.0
        # (fp32 time stamps cannot resolve gaps of ~0.3 beyond a few thousand rows: the model is an fp64 one)
        if cases % 3 == 0 and leg.fused_supported(b, torch.empty(d, d, dtype=dtype, device="cuda")) and \
                (dtype == torch.float64 or n <= 4096):
            g = torch.Generator().manual_seed(rng.randrange(1 << 30))
            # (a generator whose symmetric part is well conditioned: with an eigenvalue near zero E ~ I along it, the PEG
            # blocks are ~1 / that eigenvalue and an fp32 assembly is not positive definite any more -- the model's, not
            # the kernels', conditioning)
            Nm = torch.tril(0.2 * torch.randn(d, d, generator=g, dtype=torch.float64), -1) + \
                torch.diag(0.8 + 0.2 * torch.randn(d, generator=g, dtype=torch.float64).abs())
            Rm = torch.tril(0.3 * torch.randn(d, d, generator=g, dtype=torch.float64), -1)
            G = (Nm @ Nm.T + Rm - Rm.T + 1e-5 * torch.eye(d, dtype=torch.float64)).to(dtype).cuda()
            A = torch.randn(d, 2, generator=g, dtype=torch.float64)
            A = (0.5 * A @ A.T).to(dtype).cuda()
            ts = torch.cumsum(0.05 + 0.5 * torch.rand(n, generator=g, dtype=torch.float64), 0).to(dtype).cuda()
            pRs, pOs = leg.peg_precision(ts, G)
            m0, l0 = cr.mahal_and_det((pRs + A).double(), pOs.double(), b.double())
            _, s0 = cr.mahal_and_det(pRs.double(), pOs.double(), torch.zeros_like(b).double())
            m1, l1, s1 = leg.leg_loglik_reductions(ts, G, A, b)
            sc = 1.0 if dtype == torch.float64 else 1e-5 / 3e-4 * 30      # fp32 assembly: exponentials and solves in fp32
            e7 = sc * max(abs(float(l1) - float(l0)) / max(1.0, abs(float(l0))), abs(float(s1) - float(s0)) / max(1.0, abs(float(s0))),
                          abs(float(m1) - float(m0)) / max(1.0, abs(float(m0))) / 10)
        err = max(e1, e2 / 10, e3 / 10, e4, e5 / 10, e6 / 10, e7)
        if not err <= tol:
            print("FAIL", tag, "logdet %.3e mahal %.3e solve %.3e det %.3e inverse %.3e decompose_solve %.3e leg %.3e" % (
                e1, e2, e3, e4, e5, e6, e7), flush=True)
            sys.exit(1)
        if dtype == torch.float64:
            worst64 = max(worst64, err)
        else:
            worst32 = max(worst32, err)
        cases += 1
        if cases % 50 == 0:
            print("%d cases, worst fp64 %.2e fp32 %.2e, last %s" % (cases, worst64, worst32, tag), flush=True)
    print("OK: %d cases in %.0f s, worst fp64 %.2e, worst fp32 %.2e" % (cases, time.time() - t0, worst64, worst32))


if __name__ == "__main__":
    main()
