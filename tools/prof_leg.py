"""Config 5 (LEG log-likelihood, N = 502, rank 5) replayed from a HIP graph: the target of
rocprofv3 --kernel-trace --stats runs (which kernels the ~190 us of one evaluation are)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cyclic_gps import leg  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "leg_co2like.npz"))
t = lambda k: torch.from_numpy(g[k]).to(torch.float64).cuda()   # noqa: E731
m = leg.LEGMatrices(t("N"), t("R"), t("B"), t("Lambda"))
ts, xs = t("ts"), t("xs")
gll = leg.GraphedLogLikelihood(m, ts, xs)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    gll()
b.record()
torch.cuda.synchronize()
print("graph replay: %.1f us per evaluation, ll = %.12g (reference %.12g)" % (a.elapsed_time(b) / reps * 1e3, float(gll.value), float(g["ll"])))
