"""Target of rocprofv3 runs: the fused LEG reduction (cgps_leg_mahal_logdet) at 2^20 rows, regular grid, rank 5, and
the unfused pair (cgps_peg_precision + cgps_mahal_logdet) on the same operands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cyclic_gps import leg  # noqa: E402
import cyclic_gps.cyclic_reduction as cr  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "leg_co2like.npz"))
t = lambda k: torch.from_numpy(g[k]).to(torch.float64).cuda()   # noqa: E731
m = leg.LEGMatrices(t("N"), t("R"), t("B"), t("Lambda"))
cr.CHECK_POSITIVE_DEFINITE = False
n = 1 << 20
G, A = m.G, (m.B.T @ m.LLT_inv @ m.B).contiguous()
ts = 0.25 * torch.arange(n, dtype=torch.float64, device="cuda")
v = torch.randn(n, 5, dtype=torch.float64, device="cuda")
for _ in range(3):
    leg.leg_mahal_and_det(ts, G, A, v)
for _ in range(3):
    Rs, Os = leg.peg_precision(ts, G)
    cr.mahal_and_det(Rs + A, Os, v)
torch.cuda.synchronize()
