"""Fused solve + log-det beyond one round of the chip with the rows per lane chosen at launch (cgps_tile.h:
chunk_reduce_kernel<.., C = 0, ..>, one round of at most 256 workgroups whatever the size): ragged sizes at which
that form is taken for each block size / precision, against closed forms, against the level-wise kernel (independent
code: one launch per level, the reference's even/odd order of cyclic_reduction.py:380-438) on the same device data, as
shards with a left coupling, and the report of a block that is not positive definite."""
import numpy as np
import pytest
import torch

import _util
from cyclic_gps import sharded

# (rows, d, dtype): sizes where C * block bytes <= 2 KB or >= 16 KB with C > the compiled 16 rows per lane
CASES = [
    (8400001, 4, torch.float64),      # C = 132: 16.5 KB between the lanes of a wave
    (2 ** 23, 4, torch.float64),      # C = 128
    (1500001, 2, torch.float64),      # C = 24 x 32-byte blocks
    (3000001, 1, torch.float64),      # C = 48 x 8-byte blocks
    (5600003, 5, torch.float64),      # C = 88 x 200-byte blocks (one wave per SIMD kernel, 256 threads)
    (1900001, 4, torch.float32),      # C = 32 x 64-byte blocks
    (1500001, 7, torch.float64),      # 128 streaming lanes per workgroup (C = 48 x 392 bytes)
    (2 ** 22 + 5, 3, torch.float32),  # C = 68 x 36-byte blocks: not taken (2.4 KB) -> rounds of the compiled C, same test
]


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,dtype", CASES, ids=["%s_d%d_n%d" % ("f64" if t == torch.float64 else "f32", d, n) for n, d, t in CASES])
def test_long_chunks_closed_form_and_levelwise(n, d, dtype):
    from cyclic_gps import _hip
    import cyclic_gps.cyclic_reduction as cr
    rtol = 1e-10 if dtype == torch.float64 else 2e-5
    Rs, Os, b, x_true, logdet = _util.conditioned_system(n, d, dtype=dtype, device="cuda", seed=11)
    mahal_true = float((x_true.double() * b.double()).sum())
    m, ld = cr.mahal_and_det(Rs, Os, b)
    np.testing.assert_allclose([float(m), float(ld)], [mahal_true, logdet], rtol=rtol * 10)
    lib = _hip.lib()
    ws, nb = _hip.workspace(n, d, dtype, _hip.OP_MAHAL_LOGDET_LEVELWISE, Rs.device)
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _hip.check(lib.cgps_mahal_logdet_levelwise(_hip.ptr(Rs), _hip.ptr(Os), _hip.ptr(b), n, d, _hip.dtype_code(dtype),
                                               _hip.ptr(ws), nb, _hip.ptr(out), _hip.ptr(info), _hip.stream_ptr()))
    np.testing.assert_allclose([float(m), float(ld)], out.cpu().numpy(), rtol=rtol)
    m2, ld2 = cr.mahal_and_det(Rs, Os, b)                 # fixed elimination order: the same bits again
    assert float(m2) == float(m) and float(ld2) == float(ld)
    # the same rows as two ragged shards, the second with its left coupling, through shard_reduce + finish
    cut = n // 2 + 3
    bounds = [(0, cut), (cut, n)]
    rec_bytes, msg_bytes = sharded.message_layout(d, dtype)
    recv = torch.zeros(2 * msg_bytes, dtype=torch.uint8, device="cuda")
    for r, (lo, hi) in enumerate(bounds):
        ops = sharded.HipShardOps(hi - lo, d, dtype, torch.device("cuda"))
        ops.shard_reduce(Rs[lo:hi], Os[lo:hi - 1], b[lo:hi], Os[lo - 1].contiguous() if lo else None,
                         recv[r * msg_bytes:(r + 1) * msg_bytes], rec_bytes)
    o2 = torch.zeros(2, dtype=torch.float64, device="cuda")
    ops.finish(recv, 2, rec_bytes, msg_bytes, cut, n, o2)
    assert int(ops.info.item()) == 0
    np.testing.assert_allclose(o2.cpu().numpy(), [mahal_true, logdet], rtol=rtol * 10)


@pytest.mark.gpu
def test_long_chunks_report_a_block_that_is_not_positive_definite():
    import cyclic_gps.cyclic_reduction as cr
    n = 8400001
    Rs, Os, b, _, _ = _util.conditioned_system(n, 4, device="cuda", seed=9)
    Rs[7000003] = -Rs[7000003]
    with pytest.raises(Exception) as ei:
        cr.mahal_and_det(Rs, Os, b)
    assert "positive" in str(ei.value).lower() or "psd" in type(ei.value).__name__.lower()
    import re
    row = int(re.search(r"block row (\d+)", str(ei.value)).group(1))      # the first row of the failing lane's chunk
    assert 7000003 - 132 < row <= 7000003
