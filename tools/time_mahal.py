"""HIP-event timing of cgps_mahal_logdet (or of one shard through cgps_shard_reduce + cgps_finish_records with --shard)
at one size, results checked against the closed form:   python tools/time_mahal.py ROWS [D] [f64|f32] [--shard]
The library's A/B switches are environment variables read once per process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _util  # noqa: E402
import cyclic_gps.cyclic_reduction as cr  # noqa: E402
from cyclic_gps import sharded  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows = int(args[0])
d = int(args[1]) if len(args) > 1 else 4
dtype = torch.float32 if (len(args) > 2 and args[2] == "f32") else torch.float64
shard = "--shard" in sys.argv
cr.CHECK_POSITIVE_DEFINITE = False
Rs, Os, b, x_true, logdet = _util.conditioned_system(rows, d, dtype=dtype, device="cuda")
mahal = float((x_true.double() * b.double()).sum())
out = torch.zeros(2, dtype=torch.float64, device="cuda")
if shard:
    plan = sharded.ShardedMahalLogdet(Rs, Os, b, None, rows, 0, 1)
    fn = lambda: plan.run(out)            # noqa: E731
else:
    def fn():
        m, l = cr.mahal_and_det(Rs, Os, b)
        out[0], out[1] = m, l


def run(reps):
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    c.record()
    torch.cuda.synchronize()
    return a.elapsed_time(c) / reps * 1e3


run(10)
reps = 30 if rows <= 1 << 22 else 10
ts = [run(reps) for _ in range(4)]
o = out.tolist()
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("CGPS_"))
print("rows=%d d=%d %s %s %s: %s us  (min %.1f)  rel err mahal %.1e logdet %.1e" % (
    rows, d, str(dtype).replace("torch.", ""), "shard" if shard else "whole", tag, " ".join("%.1f" % t for t in ts), min(ts),
    abs(o[0] - mahal) / abs(mahal), abs(o[1] - logdet) / abs(logdet)), flush=True)
