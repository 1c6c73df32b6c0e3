"""How much of the 2^20-row headline time is the memory-side (Infinity) cache?  The same call in steady state on ONE
system (its 302 MB streamed again and again) against the same call alternating over K systems (K x 302 MB > 256 MB of
cache: every call streams data that is not cached):   python tools/mall_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _util  # noqa: E402
import cyclic_gps.cyclic_reduction as cr  # noqa: E402

cr.CHECK_POSITIVE_DEFINITE = False
n = 1 << 20
systems = [_util.conditioned_system(n, 4, device="cuda", seed=s)[:3] for s in range(4)]


def run(ks, reps):
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(100):
        cr.mahal_and_det(*systems[i % ks])
    a.record()
    for i in range(reps):
        cr.mahal_and_det(*systems[i % ks])
    c.record()
    torch.cuda.synchronize()
    return a.elapsed_time(c) / reps * 1e3


for trial in range(2):
    for ks in (1, 2, 4, 1):
        print("2^20 rows, d = 4 fp64, alternating over %d system(s) (%d MB): %.2f us per call" % (ks, ks * 302, run(ks, 400)), flush=True)
