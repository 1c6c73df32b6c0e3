"""The persistent form of the fused solve + log-det (csrc/cgps_tile_stream.h: one workgroup per CU walks its rows
tile by tile, four waves stream, four waves reduce the previous tile and fold it into the workgroup's carry row).

It takes over above one round of the chip (N > 2^20 rows of 4 x 4 fp64 blocks).  To check it against the CPU
oracle at sizes the oracle finishes in seconds, a child process is started with CGPS_STREAM_MIN_TILES=0 and a small
CGPS_STREAM_CUS (few workgroups, several tiles each -- the same kernel, the same code paths: tiles of a few lanes,
ragged last tiles, workgroups with one tile and with many).  Full sizes: closed forms and the level-wise kernel."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import _util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, os
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "cyclic-gps_amd"), os.path.join(%(root)r, "tests")]
import torch, _util
import cyclic_gps.cyclic_reduction as cr
from cyclic_gps import sharded
from oracle import cr_oracle as O
out = []
for n in %(sizes)r:
    Rs, Os, b, x_true, logdet = _util.conditioned_system(n, 4, seed=100 + n)
    m, ld = cr.mahal_and_det(Rs.cuda(), Os.cuda(), b.cuda())
    m0, ld0 = O.mahal_and_det(Rs, Os, b)
    rec = dict(n=n, m=float(m), ld=float(ld), m0=float(m0), ld0=float(ld0))
    # the same rows as three shards (left couplings, records, finish): the shard path of the same kernel
    parts = 3
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    rec_bytes, msg_bytes = sharded.message_layout(4, torch.float64)
    recv = torch.zeros(parts * msg_bytes, dtype=torch.uint8, device="cuda")
    for r, (lo, hi) in enumerate(bounds):
        ops = sharded.HipShardOps(hi - lo, 4, torch.float64, torch.device("cuda"))
        ops.shard_reduce(Rs[lo:hi].cuda(), Os[lo:hi - 1].cuda(), b[lo:hi].cuda(), Os[lo - 1].cuda().contiguous() if lo else None,
                         recv[r * msg_bytes:(r + 1) * msg_bytes], rec_bytes)
    o2 = torch.zeros(2, dtype=torch.float64, device="cuda")
    ops.finish(recv, parts, rec_bytes, msg_bytes, bounds[0][1], n, o2)
    rec.update(ms=float(o2[0]), lds=float(o2[1]), info=int(ops.info.item()))
    out.append(rec)
# a system that is not positive definite: the failing block is reported
Rs, Os, b, _, _ = _util.conditioned_system(30000, 4, seed=9)
Rs[17003] = -Rs[17003]
cr.CHECK_POSITIVE_DEFINITE = False
m, ld = cr.mahal_and_det(Rs.cuda(), Os.cuda(), b.cuda())
out.append(dict(npd_nan=bool(torch.isnan(m)) and bool(torch.isnan(ld))))
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("cus", [1, 3, 7])
def test_persistent_form_against_the_oracle_on_small_systems(cus):
    sizes = [4097, 9000, 16384 + 17, 40000, 65536, 100003]
    env = dict(os.environ, CGPS_STREAM_MIN_TILES="0", CGPS_STREAM_CUS=str(cus))
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, sizes=sizes)], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    for r in res[:-1]:
        np.testing.assert_allclose([r["m"], r["ld"]], [r["m0"], r["ld0"]], rtol=1e-10, err_msg=str(r["n"]))
        np.testing.assert_allclose([r["ms"], r["lds"]], [r["m0"], r["ld0"]], rtol=1e-10, err_msg="shards %d" % r["n"])
        assert r["info"] == 0
    assert res[-1]["npd_nan"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2 ** 20 + 1, 2 ** 20 + 4096 + 5, 3 * 2 ** 19, 2 ** 21, 2 ** 21 + 777, 2 ** 22 + 12345])
def test_persistent_form_at_full_size(n):
    """More than one round of the chip: closed form (planted solution, log-det of the bidiagonal factor) and the
    level-wise kernel (independent code: one launch per level) on the same device data."""
    import ctypes
    from cyclic_gps import _hip
    import cyclic_gps.cyclic_reduction as cr
    Rs, Os, b, x_true, logdet = _util.conditioned_system(n, 4, device="cuda", seed=5)
    mahal_true = float((x_true * b).sum())
    m, ld = cr.mahal_and_det(Rs, Os, b)
    np.testing.assert_allclose([float(m), float(ld)], [mahal_true, logdet], rtol=1e-10)
    lib = _hip.lib()
    ws, nb = _hip.workspace(n, 4, torch.float64, _hip.OP_MAHAL_LOGDET_LEVELWISE, Rs.device)
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _hip.check(lib.cgps_mahal_logdet_levelwise(_hip.ptr(Rs), _hip.ptr(Os), _hip.ptr(b), n, 4, _hip.F64, _hip.ptr(ws), nb,
                                               _hip.ptr(out), _hip.ptr(info), _hip.stream_ptr()))
    np.testing.assert_allclose([float(m), float(ld)], out.cpu().numpy(), rtol=1e-11)
    # replayed calls give the same bits (fixed elimination order, no floating-point atomics)
    m2, ld2 = cr.mahal_and_det(Rs, Os, b)
    assert float(m2) == float(m) and float(ld2) == float(ld)
