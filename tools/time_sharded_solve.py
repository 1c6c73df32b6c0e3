"""Per-rank time of ShardedSolve.run / ShardedPosterior.run for one rank of an 8-shard system on one GPU (the gather
replaced by a copy of records produced beforehand), with the separator systems as one kernel launch
(cgps_boundary_solve / cgps_boundary_recursions) and as batched torch ops:   python tools/time_sharded_solve.py [ROWS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _util  # noqa: E402
from cyclic_gps import sharded  # noqa: E402
import cyclic_gps.cyclic_reduction as cr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
parts, d, rank = 8, 4, 3
cr.CHECK_POSITIVE_DEFINITE = False
Rs, Os, b, x_true, _ = _util.conditioned_system(n, d, device="cuda", seed=1)
bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
rec_bytes, msg_bytes = sharded.message_layout(d, torch.float64)
allrec = torch.zeros(parts * msg_bytes, dtype=torch.uint8, device="cuda")
for r, (lo, hi) in enumerate(bounds):
    ops = sharded.HipShardOps(hi - lo, d, torch.float64, torch.device("cuda"))
    ops.shard_reduce(Rs[lo:hi].contiguous(), Os[lo:hi - 1].contiguous(), b[lo:hi].contiguous(),
                     Os[lo - 1].contiguous() if lo else None, allrec[r * msg_bytes:(r + 1) * msg_bytes], rec_bytes)
lo, hi = bounds[rank]
args = (Rs[lo:hi].contiguous(), Os[lo:hi - 1].contiguous(), Os[lo - 1].contiguous(), n, rank, parts)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


y = b[lo:hi].contiguous()
for cls in (sharded.ShardedSolve, sharded.ShardedPosterior):
    for hipb in (True, False):
        plan = cls(*args, gather=lambda send, recv: recv.copy_(allrec))
        plan._hip_boundary = hipb
        out = plan.run(y)
        x = out if cls is sharded.ShardedSolve else out[0]
        err = float((x - x_true[lo:hi]).abs().max())
        print("%s rank %d of %d, %d rows per shard, separators by %s: %.1f us per run (err %.1e)" % (
            cls.__name__, rank, parts, hi - lo, "one kernel" if hipb else "torch ops", timeit(lambda: plan.run(y)), err), flush=True)
