"""Run one op of the hot path a few times on synthetic input: the target of rocprofv3 runs.

  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/prof_case.py --op decompose
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES -d gpurun_out/pmc -- python3 tools/prof_case.py --op solve

Same generator as bench.py (tests/_util.conditioned_system).  Prints wall time per call.
"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="mahal_and_det",
                    choices=["mahal_and_det", "decompose", "solve", "halfsolve", "det", "inverse_blocks", "decompose_solve"])
    ap.add_argument("--nrhs", type=int, default=1, help="right-hand-side columns for solve / halfsolve")
    ap.add_argument("--rows", type=int, default=2 ** 20)
    ap.add_argument("--d", type=int, default=4)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dtype = torch.float64 if a.dtype == "f64" else torch.float32
    Rs, Os, b, _, _ = _util.conditioned_system(a.rows, a.d, dtype=dtype, device="cuda")
    dec = cr.decompose(Rs, Os) if a.op != "mahal_and_det" else None
    if a.nrhs > 1:
        b = (b[:, :, None] * torch.arange(1, a.nrhs + 1, dtype=b.dtype, device=b.device)).contiguous()
    fn = {
        "mahal_and_det": lambda: cr.mahal_and_det(Rs, Os, b),
        "decompose": lambda: cr.decompose(Rs, Os),
        "solve": lambda: cr.solve(dec, b),
        "halfsolve": lambda: cr.halfsolve(dec, b),
        "det": lambda: cr.det(dec),
        "inverse_blocks": lambda: cr.inverse_blocks(dec),
        "decompose_solve": lambda: cr.decompose_solve(Rs, Os, b),
    }[a.op]
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    print("%s N=%d d=%d %s nrhs=%d: %.1f us per call" % (a.op, a.rows, a.d, a.dtype, a.nrhs,
                                                         (time.perf_counter() - t0) / a.reps * 1e6))


if __name__ == "__main__":
    main()
