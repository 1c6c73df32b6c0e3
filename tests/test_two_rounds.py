"""Fused solve + log-det between one and two rounds of the chip (2^20 < N <= 2^21 rows of 4 x 4 fp64 blocks: two
workgroups per CU, the size of every shard of BASELINE config 4), ragged sizes included: closed forms, the level-wise
kernel (independent code: one launch per level) on the same device data, the same rows as shards with a left coupling,
and the report of a block that is not positive definite."""
import numpy as np
import pytest
import torch

import _util
from cyclic_gps import sharded


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2 ** 20 + 1, 2 ** 20 + 4096 + 5, 3 * 2 ** 19, 2 ** 21 - 777, 2 ** 21])
def test_between_one_and_two_rounds(n):
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
    # the same rows as two shards (the second with its left coupling) through shard_reduce + finish
    parts = 2 if n < 2 ** 21 else 1
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    rec_bytes, msg_bytes = sharded.message_layout(4, torch.float64)
    recv = torch.zeros(parts * msg_bytes, dtype=torch.uint8, device="cuda")
    for r, (lo, hi) in enumerate(bounds):
        ops = sharded.HipShardOps(hi - lo, 4, torch.float64, torch.device("cuda"))
        ops.shard_reduce(Rs[lo:hi], Os[lo:hi - 1], b[lo:hi], Os[lo - 1].contiguous() if lo else None,
                         recv[r * msg_bytes:(r + 1) * msg_bytes], rec_bytes)
    o2 = torch.zeros(2, dtype=torch.float64, device="cuda")
    ops.finish(recv, parts, rec_bytes, msg_bytes, bounds[0][1], n, o2)
    assert int(ops.info.item()) == 0
    np.testing.assert_allclose(o2.cpu().numpy(), [mahal_true, logdet], rtol=1e-10)


@pytest.mark.gpu
def test_two_rounds_report_a_block_that_is_not_positive_definite():
    import cyclic_gps.cyclic_reduction as cr
    n = 2 ** 20 + 50000
    Rs, Os, b, _, _ = _util.conditioned_system(n, 4, device="cuda", seed=9)
    Rs[1000003] = -Rs[1000003]
    with pytest.raises(Exception) as ei:
        cr.mahal_and_det(Rs, Os, b)
    assert "positive" in str(ei.value).lower() or "psd" in type(ei.value).__name__.lower()
