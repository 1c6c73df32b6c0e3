"""Op B at one size: decompose, solve, decompose_solve, mahal_and_det forward + backward:  python tools/time_opb.py [ROWS D f64|f32]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _util  # noqa: E402
import cyclic_gps.cyclic_reduction as cr  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
cr.CHECK_POSITIVE_DEFINITE = False
Rs, Os, b, x_true, _ = _util.conditioned_system(rows, d, dtype=dtype, device="cuda")


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


h = {}


def dec():
    h["dec"] = cr.decompose(Rs, Os)


t_dec = timeit(dec)
t_sol = timeit(lambda: cr.solve(h["dec"], b))
t_ds = timeit(lambda: cr.decompose_solve(Rs, Os, b))
_, x = cr.decompose_solve(Rs, Os, b)
Rg, Og, bg = (t.clone().requires_grad_(True) for t in (Rs, Os, b))


def step():
    m, l = cr.mahal_and_det(Rg, Og, bg)
    (m + l).backward()
    Rg.grad = Og.grad = bg.grad = None


t_fb = timeit(step, 10)
print("N=%d d=%d %s: decompose %.1f us, solve %.1f us (sum %.1f), decompose_solve %.1f us (err %.1e), mahal_and_det fwd+bwd %.1f us" % (
    rows, d, str(dtype).replace("torch.", ""), t_dec, t_sol, t_dec + t_sol, t_ds, float((x.double() - x_true.double()).abs().max()), t_fb))
