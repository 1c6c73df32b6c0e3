"""Forward + backward through mahal_and_det at N=2^20, d=4, fp64 (what bench.py reports as
extras.opB.mahal_and_det_fwd_bwd_us); CGPS_LIB selects an alternative build for A/B runs on one box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _util
import cyclic_gps.cyclic_reduction as cr

cr.CHECK_POSITIVE_DEFINITE = False
n, d = 1 << 20, 4
Rs, Os, b, _, _ = _util.conditioned_system(n, d, dtype=torch.float64, device="cuda", seed=3)
R, O, v = (t.clone().requires_grad_(True) for t in (Rs, Os, b))


def step():
    m, ld = cr.mahal_and_det(R, O, v)
    (m + ld).backward()
    R.grad = O.grad = v.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.time()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    best = min(best, (time.time() - t0) / 20)
print("fwd+bwd %.1f us (%s)" % (best * 1e6, os.environ.get("CGPS_LIB", "default lib")))
