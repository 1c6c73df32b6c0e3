"""How the headline op's time moves over the first launches after the GPU has been idle, and what
precedes them (nothing / 20 ms of unrelated work): per-step wall time in windows of 5 steps."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _util  # noqa: E402
from cyclic_gps import _hip  # noqa: E402

N, d = 1 << 20, 4
dev = torch.device("cuda", 0)
Rs, Os, b, _, _ = _util.conditioned_system(N, d, dtype=torch.float64, device="cuda")
ws, wsb = _hip.workspace(N + 1, d, torch.float64, _hip.OP_MAHAL_LOGDET, dev)
out = torch.zeros(2, dtype=torch.float64, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib = _hip.lib()


def step():
    lib.cgps_mahal_logdet(_hip.ptr(Rs), _hip.ptr(Os), _hip.ptr(b), N, d, _hip.F64, _hip.ptr(ws), wsb, _hip.ptr(out), _hip.ptr(info), sp)


def windows(label, nwin=16, per=5):
    ts = []
    torch.cuda.synchronize()
    for _ in range(nwin):
        t0 = time.perf_counter()
        for _ in range(per):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / per * 1e6)
    print("%-34s" % label, " ".join("%5.1f" % t for t in ts), flush=True)


step()
torch.cuda.synchronize()
for idle in (2.0, 0.2):
    time.sleep(idle)
    windows("after %.1f s idle" % idle)
    time.sleep(idle)
    a = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize()
    time.sleep(idle)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:
        a = a @ a * 1e-4
    torch.cuda.synchronize()
    windows("after %.1f s idle + 20 ms of matmul" % idle)
    time.sleep(idle)
    c = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:
        c.fill_(1)
    torch.cuda.synchronize()
    windows("after %.1f s idle + 20 ms of fills" % idle)
