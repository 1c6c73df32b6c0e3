"""Config-5 log-likelihood (fused vs unfused operand assembly), eager and replayed from a HIP graph, and the fused
reduction at 2^20 rows on a regular grid:  python tools/time_leg.py     (CGPS_LEG_UNFUSED=1: the unfused path)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cyclic_gps import leg  # noqa: E402
import cyclic_gps.cyclic_reduction as cr  # noqa: E402


def timeit(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


g = np.load(os.path.join(ROOT, "tests", "golden", "leg_co2like.npz"))
t = lambda k: torch.from_numpy(g[k]).to(torch.float64).cuda()   # noqa: E731
m = leg.LEGMatrices(t("N"), t("R"), t("B"), t("Lambda"))
ts, xs = t("ts"), t("xs")
cr.CHECK_POSITIVE_DEFINITE = False
tag = "unfused" if os.environ.get("CGPS_LEG_UNFUSED") == "1" else "fused"
print("config 5 log-likelihood (%s): eager %.1f us" % (tag, timeit(lambda: leg.log_likelihood(m, ts, xs), 50)), flush=True)
gll = leg.GraphedLogLikelihood(m, ts, xs)
print("config 5 log-likelihood (%s): graph replay %.1f us  (ll rel err vs reference %.1e)" % (
    tag, timeit(gll, 200), abs(float(gll()) - float(g["ll"])) / abs(float(g["ll"]))), flush=True)
if tag == "fused":
    G = m.G
    A = (m.B.T @ m.LLT_inv @ m.B).contiguous()
    v = leg.compute_v(m, xs)
    print("  one fused reduction, 502 rows: %.1f us" % timeit(lambda: leg.leg_mahal_and_det(ts, G, A, v), 100))
    n = 1 << 20
    tsl = (0.25 * torch.arange(n, dtype=torch.float64)).cuda()
    vl = torch.randn(n, 5, dtype=torch.float64, device="cuda")
    print("  fused reduction, 2^20 rows, regular grid, rank 5: %.1f us" % timeit(lambda: leg.leg_mahal_and_det(tsl, G, A, vl), 10, 2))
    Rs, Os = leg.peg_precision(tsl, G)
    KR = Rs + A
    print("  unfused: peg_precision %.1f us + mahal_and_det %.1f us" % (
        timeit(lambda: leg.peg_precision(tsl, G), 10, 2), timeit(lambda: cr.mahal_and_det(KR, Os, vl), 10, 2)))
