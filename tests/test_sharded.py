"""Time-axis sharding (cyclic_gps/sharded.py).

CPU: shard bookkeeping, the record algebra against the oracle, and the collective path with
world_size 2 and 3 under gloo (device steps replaced by the dense stand-in of _shard_dense.py).
GPU: the HIP shard records against the dense records, and the full sharded result on one GPU.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _util
import _shard_dense as SD
from oracle import cr_oracle as O
from cyclic_gps import sharded


def _split(Rs, Os, x, bounds):
    """Shards of a whole system: (Rs, Os_inside, x, O_left) per [lo, hi)."""
    out = []
    for lo, hi in bounds:
        out.append((Rs[lo:hi], Os[lo:hi - 1], x[lo:hi], Os[lo - 1] if lo > 0 else None))
    return out


def test_shard_bounds_cover_the_axis():
    for n in (1, 2, 7, 64, 1000, 2 ** 20 + 3):
        for world in (1, 2, 3, 8):
            if world > n:
                continue
            b = [sharded.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("d,n,parts", [(1, 9, 3), (2, 17, 2), (3, 40, 5), (4, 33, 4), (3, 5, 5)])
def test_record_algebra_against_oracle(d, n, parts):
    """Reducing every shard to its record and finishing the boundary system gives exactly the
    oracle's mahal_and_det of the whole system (shards of one row included)."""
    Rs, Os, b, _, _ = _util.conditioned_system(n, d, seed=5 + n)
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    recs, parts_ = [], []
    for sR, sO, sx, Ol in _split(Rs, Os, b, bounds):
        r, p = SD.dense_shard_record(sR.numpy(), sO.numpy(), sx.numpy(), None if Ol is None else Ol.numpy())
        # through the packed layout, as it travels
        recs.append(SD.unpack_record(SD.pack_record(r, d), d))
        parts_.append(p)
    m, ld = SD.dense_finish(recs, parts_, d)
    m0, ld0 = O.mahal_and_det(Rs, Os, b)
    np.testing.assert_allclose([m, ld], [float(m0), float(ld0)], rtol=1e-10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, n_total, d, q, sub_shards=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Rs, Os, b, O_left, mahal_true, logdet_true = sharded.make_sharded_system(
            n_total, d, torch.float64, torch.device("cpu"), rank, world)
        lo, hi = sharded.shard_bounds(n_total, world, rank)
        assert Rs.shape[0] == hi - lo and Os.shape[0] == hi - lo - 1
        assert (O_left is None) == (rank == 0)
        plan = sharded.ShardedMahalLogdet(Rs, Os, b, O_left, n_total, rank, world, ops=SD.DenseShardOps(d),
                                          sub_shards=sub_shards)
        assert plan.records_per_rank == min(sub_shards, n_total // world)
        out = plan.run().clone()
        # every rank must hold the same result
        outs = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(outs, out)
        # rebuild the whole system on rank 0 and run the oracle on it
        full = [None] * world
        dist.all_gather_object(full, (Rs, Os, b, O_left))
        if rank == 0:
            R = torch.cat([f[0] for f in full])
            parts = []
            for r, f in enumerate(full):
                if r > 0:
                    parts.append(f[3][None])
                parts.append(f[1])
            Oall = torch.cat(parts)
            x = torch.cat([f[2] for f in full])
            m0, ld0 = O.mahal_and_det(R, Oall, x)
            q.put(dict(outs=[o.tolist() for o in outs], oracle=[float(m0), float(ld0)],
                       closed=[mahal_true, logdet_true]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,d,sub_shards", [(2, 257, 3, 1), (3, 100, 2, 1), (2, 2, 4, 1), (2, 257, 3, 4),
                                                         (3, 10, 2, 2), (2, 5, 2, 3)])
def test_collective_path_under_gloo(world, n_total, d, sub_shards):
    """One all-gather of the ranks' records (sub_shards of them per rank: the sub-shard layout), then the
    finish over world * sub_shards records; ragged sub-shards and sub-shards of one row included."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_gloo_worker, args=(world, _free_port(), n_total, d, q, sub_shards), nprocs=world, join=True)
    res = q.get()
    for o in res["outs"]:
        np.testing.assert_allclose(o, res["outs"][0], rtol=0, atol=0)
    np.testing.assert_allclose(res["outs"][0], res["oracle"], rtol=1e-10)
    np.testing.assert_allclose(res["outs"][0], res["closed"], rtol=1e-10)


# ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("d,dtype", [(1, torch.float64), (2, torch.float64), (3, torch.float64), (4, torch.float64),
                                     (4, torch.float32), (5, torch.float32), (5, torch.float64), (6, torch.float64),
                                     (7, torch.float64), (8, torch.float64), (8, torch.float32)])
@pytest.mark.parametrize("n,parts", [(7, 3), (64, 2), (300, 3), (1000, 3), (5000, 4), (16385, 2), (40000, 5)])
def test_hip_shard_records_match_dense(d, dtype, n, parts):
    from cyclic_gps import _hip
    import ctypes
    Rs, Os, b, _, _ = _util.conditioned_system(n, d, seed=11 + n)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == torch.float64 else dict(rtol=3e-4, atol=3e-4)
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    rec_bytes, msg_bytes = sharded.message_layout(d, dtype)
    recv = torch.zeros(parts * msg_bytes, dtype=torch.uint8, device="cuda")
    for r, (sR, sO, sx, Ol) in enumerate(_split(Rs, Os, b, bounds)):
        ops = sharded.HipShardOps(sR.shape[0], d, dtype, torch.device("cuda"))
        send = recv[r * msg_bytes:(r + 1) * msg_bytes]
        ops.shard_reduce(sR.to(dtype).cuda().contiguous(), sO.to(dtype).cuda().contiguous(),
                         sx.to(dtype).cuda().contiguous(), None if Ol is None else Ol.to(dtype).cuda().contiguous(),
                         send, rec_bytes)
        gp = send[rec_bytes:].view(torch.float64).cpu().numpy()
        assert gp[2] == 0.0
        if sR.shape[0] * d > 1500:            # dense algebra is O(n^3): element-wise record check on small shards
            continue
        rec, par = SD.dense_shard_record(sR.numpy(), sO.numpy(), sx.numpy(), None if Ol is None else Ol.numpy())
        used = 3 * d * d + 2 * d              # the rest of the stride is alignment padding
        got = send[:rec_bytes].view(dtype).cpu().double().numpy()[:used]
        np.testing.assert_allclose(got, SD.pack_record(rec, d)[:used], err_msg="shard %d" % r, **tol)
        np.testing.assert_allclose(gp[:2], par[:2], rtol=tol["rtol"], atol=1e-8 if dtype == torch.float64 else 1e-2)
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    ops.finish(recv, parts, rec_bytes, msg_bytes, bounds[0][1] - bounds[0][0], n, out)
    m0, ld0 = O.mahal_and_det(Rs, Os, b)
    np.testing.assert_allclose(out.cpu().numpy(), [float(m0), float(ld0)],
                               rtol=1e-9 if dtype == torch.float64 else 2e-4)
    assert int(ops.info.item()) == 0


@pytest.mark.gpu
def test_sharded_plan_world1_on_gpu():
    n, d = 70000, 4
    Rs, Os, b, O_left, mahal_true, logdet_true = sharded.make_sharded_system(
        n, d, torch.float64, torch.device("cuda"), 0, 1)
    plan = sharded.ShardedMahalLogdet(Rs, Os, b, O_left, n, 0, 1)
    out = plan.run().cpu().numpy()
    np.testing.assert_allclose(out, [mahal_true, logdet_true], rtol=1e-10)
    assert int(plan.ops.info.item()) == 0


@pytest.mark.gpu
def test_config4_as_eight_shards_on_one_gpu():
    """BASELINE config 4 (N = 2^24, d = 4, fp64) in the shape the 8-GPU run has it: eight shards of
    2^21 rows, each reduced by cgps_shard_reduce (with its left coupling), the eight records
    finished by cgps_finish_records -- all on one GPU, against the closed form and against the
    same system reduced whole by cgps_mahal_logdet."""
    import cyclic_gps.cyclic_reduction as cr
    n, d, parts = 2 ** 24, 4, 8
    dev = torch.device("cuda")
    Rs, Os, b, x_true, logdet = _util.conditioned_system(n, d, device="cuda")
    mahal_true = float((x_true * b).sum())
    del x_true
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    assert all(hi - lo == 2 ** 21 for lo, hi in bounds)
    rec_bytes, msg_bytes = sharded.message_layout(d, torch.float64)
    recv = torch.zeros(parts * msg_bytes, dtype=torch.uint8, device=dev)
    ops = sharded.HipShardOps(2 ** 21, d, torch.float64, dev)
    for r, (sR, sO, sx, Ol) in enumerate(_split(Rs, Os, b, bounds)):
        ops.shard_reduce(sR, sO, sx, None if Ol is None else Ol.contiguous(), recv[r * msg_bytes:(r + 1) * msg_bytes],
                         rec_bytes)
    out = torch.zeros(2, dtype=torch.float64, device=dev)
    ops.finish(recv, parts, rec_bytes, msg_bytes, 2 ** 21, n, out)
    got = out.cpu().numpy()
    assert int(ops.info.item()) == 0
    np.testing.assert_allclose(got, [mahal_true, logdet], rtol=1e-10)
    m, ld = cr.mahal_and_det(Rs, Os, b)                     # the same system as ONE shard
    np.testing.assert_allclose(got, [float(m), float(ld)], rtol=1e-12)
    # the sub-shard layout: every rank's 2^21 rows as S sub-shards on S streams, S records per rank in the
    # all-gather buffer, ONE finish over the 8 S records (ShardedMahalLogdet with the ranks played in sequence)
    for S in (2, 4):
        plans, sends = [], []
        for r, (sR, sO, sx, Ol) in enumerate(_split(Rs, Os, b, bounds)):
            pl = sharded.ShardedMahalLogdet(sR, sO, sx, None if Ol is None else Ol.contiguous(), n, r, parts, sub_shards=S,
                                            gather=lambda send, recv: recv.copy_(torch.cat(sends)))
            assert pl.records_per_rank == S
            pl.reduce_to_send()
            torch.cuda.synchronize()
            sends.append(pl.send.clone())
            plans.append(pl)
        for r in (0, parts - 1):
            got_s = plans[r].run().cpu().numpy()
            assert int(plans[r].ops.info.item()) == 0
            np.testing.assert_allclose(got_s, [mahal_true, logdet], rtol=1e-10)
            np.testing.assert_allclose(got_s, [float(m), float(ld)], rtol=1e-12)
        del plans, sends


# ---- sharded solve (ShardedSolve: records -> separator values -> local interior solves) -------------------
def _gloo_solve_worker(rank, world, port, n_total, d, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Rs, Os, b, O_left, _, _ = sharded.make_sharded_system(n_total, d, torch.float64, torch.device("cpu"), rank, world)
        x_true = sharded.make_sharded_system.last_x_true
        plan = sharded.ShardedSolve(Rs, Os, O_left, n_total, rank, world, ops=SD.DenseShardOps(d), solve_ops=SD.OracleSolveOps)
        x = plan.run(b)
        err = float((x - x_true).abs().max())
        x2 = plan.run(2.0 * b)                       # the interior factor is reused; the solve is linear
        err2 = float((x2 - 2.0 * x_true).abs().max())
        errs = [None] * world
        dist.all_gather_object(errs, (err, err2))
        if rank == 0:
            q.put(errs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,d", [(2, 257, 3), (3, 100, 2), (2, 3, 4), (3, 3, 2)])
def test_sharded_solve_under_gloo(world, n_total, d):
    """x = J^-1 y of a system split over ranks (shards of one row included) against the planted solution."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_gloo_solve_worker, args=(world, _free_port(), n_total, d, q), nprocs=world, join=True)
    for err, err2 in q.get():
        assert err < 1e-9 and err2 < 1e-9, (err, err2)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,parts,dtype", [(5000, 4, 4, torch.float64), (70001, 3, 5, torch.float64),
                                             (2 ** 20, 4, 8, torch.float64), (40000, 8, 3, torch.float32),
                                             (9, 4, 5, torch.float64), (5, 2, 5, torch.float64)])
def test_sharded_solve_on_one_gpu(n, d, parts, dtype):
    """ShardedSolve.run itself with world > 1 and the HIP kernels, the ranks played one after the other on
    one GPU through an injected gather (the left-coupling correction, the kept interior factor, shards of
    two rows -- an interior with no coupling block -- and of one row included); against the planted solution."""
    dev = torch.device("cuda")
    Rs, Os, b, x_true, _ = _util.conditioned_system(n, d, dtype=dtype, device="cuda", seed=3 + n)
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    sends = []
    plans = [sharded.ShardedSolve(sR.contiguous(), sO.contiguous(), None if Ol is None else Ol.contiguous(), n, r, parts,
                                  gather=lambda send, recv: recv.copy_(torch.cat(sends)))
             for r, (sR, sO, sx, Ol) in enumerate(_split(Rs, Os, b, bounds))]
    tol = 1e-9 if dtype == torch.float64 else 2e-3
    for scale in (1.0, -2.0):                       # second pass: the interior factor is reused, the solve is linear
        y = scale * b
        sends.clear()
        for r, (lo, hi) in enumerate(bounds):
            sends.append(plans[r].reduce_to_send(y[lo:hi].contiguous()).clone())
        for r, (lo, hi) in enumerate(bounds):
            x = plans[r].run(y[lo:hi].contiguous())
            err = float((x - scale * x_true[lo:hi]).abs().max())
            assert err <= tol * abs(scale), (r, scale, err)


@pytest.mark.gpu
def test_sharded_solve_plan_world1_on_gpu():
    n, d = 30000, 4
    Rs, Os, b, O_left, _, _ = sharded.make_sharded_system(n, d, torch.float64, torch.device("cuda"), 0, 1)
    x_true = sharded.make_sharded_system.last_x_true
    plan = sharded.ShardedSolve(Rs, Os, O_left, n, 0, 1)
    assert float((plan.run(b) - x_true).abs().max()) < 1e-9
    assert float((plan.run(3 * b) - 3 * x_true).abs().max()) < 1e-9


# ---- sharded posterior: mean + covariance blocks of a system split over ranks (ShardedPosterior) -----------------
def _run_posterior_ranks(Rs, Os, b, parts, **kw):
    """The ranks played one after the other through an injected gather; returns per-rank (mean, Sig_diag, Sig_off)."""
    n = Rs.shape[0]
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    sends, plans = [], []
    for r, (lo, hi) in enumerate(bounds):
        pl = sharded.ShardedPosterior(Rs[lo:hi].contiguous(), Os[lo:hi - 1].contiguous(), Os[lo - 1].contiguous() if lo else None,
                                      n, r, parts, gather=lambda send, recv: recv.copy_(torch.cat(sends)), **kw)
        sends.append(pl.reduce_to_send(b[lo:hi].contiguous()).clone())
        plans.append(pl)
    return bounds, [plans[r].run(b[lo:hi].contiguous()) for r, (lo, hi) in enumerate(bounds)]


@pytest.mark.parametrize("n,d,parts", [(40, 3, 4), (9, 2, 5), (5, 2, 5), (257, 4, 3), (33, 1, 2)])
def test_sharded_posterior_algebra_against_oracle(n, d, parts):
    """Mean and the diagonal / off-diagonal blocks of J^-1 of every rank's rows (the coupling block across each shard
    boundary included; shards of one row included) equal the oracle's solve / inverse_blocks of the whole system."""
    Rs, Os, b, _, _ = _util.conditioned_system(n, d, seed=7 + n)
    bounds, res = _run_posterior_ranks(Rs, Os, b, parts, ops=SD.DenseShardOps(d), solve_ops=SD.OracleSolveOps)
    dec = O.decompose(Rs, Os)
    Sd0, So0 = O.inverse_blocks(dec)
    x0 = O.solve(dec, b)
    for (lo, hi), (mean, Sd, So) in zip(bounds, res):
        np.testing.assert_allclose(mean.numpy(), x0[lo:hi].numpy(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(Sd.numpy(), Sd0[lo:hi].numpy(), rtol=0, atol=1e-12)
        ref = So0[lo - 1:hi - 1] if lo else So0[:hi - 1]
        assert So.shape == ref.shape
        np.testing.assert_allclose(So.numpy(), ref.numpy(), rtol=0, atol=1e-12)


def _gloo_posterior_worker(rank, world, port, n_total, d, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Rs, Os, b, O_left, _, _ = sharded.make_sharded_system(n_total, d, torch.float64, torch.device("cpu"), rank, world)
        x_true = sharded.make_sharded_system.last_x_true
        plan = sharded.ShardedPosterior(Rs, Os, O_left, n_total, rank, world, ops=SD.DenseShardOps(d), solve_ops=SD.OracleSolveOps)
        mean, Sd, So = plan.run(b)
        full = [None] * world
        dist.all_gather_object(full, (Rs, Os, O_left, Sd, So, float((mean - x_true).abs().max())))
        if rank == 0:
            R = torch.cat([f[0] for f in full])
            Oall = torch.cat([t for r, f in enumerate(full) for t in (([f[2][None]] if r > 0 else []) + [f[1]])])
            Sd0, So0 = O.inverse_blocks(O.decompose(R, Oall))
            q.put(dict(err_mean=max(f[5] for f in full), err_d=float((torch.cat([f[3] for f in full]) - Sd0).abs().max()),
                       err_o=float((torch.cat([f[4] for f in full]) - So0).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,d", [(2, 65, 3), (3, 10, 2)])
def test_sharded_posterior_under_gloo(world, n_total, d):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_gloo_posterior_worker, args=(world, _free_port(), n_total, d, q), nprocs=world, join=True)
    res = q.get()
    assert res["err_mean"] < 1e-9 and res["err_d"] < 1e-12 and res["err_o"] < 1e-12, res


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,parts,dtype", [(5000, 4, 4, torch.float64), (70001, 3, 5, torch.float64), (9, 4, 5, torch.float64),
                                             (2 ** 20, 4, 8, torch.float64), (40000, 8, 3, torch.float32),
                                             (2 ** 18 + 5, 5, 8, torch.float64)])
def test_sharded_posterior_on_one_gpu(n, d, parts, dtype):
    """ShardedPosterior with the HIP kernels, eight (or fewer) ranks played in sequence on one GPU: mean against the
    planted solution, covariance blocks against inverse_blocks of the whole system on the same GPU."""
    import cyclic_gps.cyclic_reduction as cr
    Rs, Os, b, x_true, _ = _util.conditioned_system(n, d, dtype=dtype, device="cuda", seed=3 + n)
    bounds, res = _run_posterior_ranks(Rs, Os, b, parts)
    Sd0, So0 = cr.inverse_blocks(cr.decompose(Rs, Os))
    tol = 1e-9 if dtype == torch.float64 else 2e-3
    for (lo, hi), (mean, Sd, So) in zip(bounds, res):
        assert float((mean - x_true[lo:hi]).abs().max()) <= tol
        assert float((Sd - Sd0[lo:hi]).abs().max()) <= tol
        ref = So0[lo - 1:hi - 1] if lo else So0[:hi - 1]
        assert So.shape == ref.shape and (ref.numel() == 0 or float((So - ref).abs().max()) <= tol)


@pytest.mark.gpu
def test_sharded_posterior_config4_shape():
    """N = 2^24 rows (BASELINE config 4) as eight shards: rank 5's blocks satisfy (J Sigma)_ii = I on its rows -- the
    block-diagonal of J Sigma needs exactly the blocks the rank returns (its coupling across the shard boundary included)."""
    n, d, parts, r = 2 ** 24, 4, 8, 5
    Rs, Os, b, x_true, _ = _util.conditioned_system(n, d, device="cuda")
    bounds = [sharded.shard_bounds(n, parts, k) for k in range(parts)]
    sends = []
    for k, (lo, hi) in enumerate(bounds):
        pl = sharded.ShardedPosterior(Rs[lo:hi], Os[lo:hi - 1], Os[lo - 1].contiguous() if lo else None, n, k, parts,
                                      gather=lambda send, recv: recv.copy_(torch.cat(sends)))
        sends.append(pl.reduce_to_send(b[lo:hi]).clone())
        if k == r:
            plan = pl
    lo, hi = bounds[r]
    mean, Sd, So = plan.run(b[lo:hi])
    assert float((mean - x_true[lo:hi]).abs().max()) < 1e-9
    # rows lo .. hi-2 (the last row's right coupling belongs to rank r + 1): R_i S_ii + O_{i-1} S_{i,i-1}^T + O_i^T S_{i+1,i}
    Rl, Ol = Rs[lo:hi - 1], Os[lo - 1:hi - 2]
    JS = Rl @ Sd[:-1] + Ol @ So[:-1].transpose(-1, -2) + Os[lo:hi - 1].transpose(-1, -2) @ So[1:]
    eye = torch.eye(d, dtype=JS.dtype, device=JS.device)
    assert float((JS - eye).abs().max()) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("d,dtype", [(1, torch.float64), (3, torch.float64), (4, torch.float64), (5, torch.float64),
                                     (8, torch.float64), (4, torch.float32), (8, torch.float32)],
                         ids=lambda p: str(p).replace("torch.", ""))
@pytest.mark.parametrize("parts", [1, 2, 3, 8, 17, 64])
def test_boundary_kernels_against_the_torch_restatement(d, dtype, parts):
    """cgps_boundary_solve / cgps_boundary_recursions (one launch each) against boundary_system + a dense solve and
    against boundary_recursions (batched torch ops), on the records of a real system cut into `parts` shards; the
    separator values also against the planted solution."""
    n = parts * 37 + 5
    Rs, Os, b, x_true, _ = _util.conditioned_system(n, d, dtype=dtype, device="cuda", seed=parts)
    bounds = [sharded.shard_bounds(n, parts, r) for r in range(parts)]
    rec_bytes, msg_bytes = sharded.message_layout(d, dtype)
    recv = torch.zeros(parts * msg_bytes, dtype=torch.uint8, device="cuda")
    for r, (lo, hi) in enumerate(bounds):
        ops = sharded.HipShardOps(hi - lo, d, dtype, torch.device("cuda"))
        ops.shard_reduce(Rs[lo:hi].contiguous(), Os[lo:hi - 1].contiguous(), b[lo:hi].contiguous(),
                         Os[lo - 1].contiguous() if lo else None, recv[r * msg_bytes:(r + 1) * msg_bytes], rec_bytes)
    tol = 1e-9 if dtype == torch.float64 else 2e-3
    x_sep = sharded.hip_boundary_solve(recv, parts, msg_bytes, d, dtype)
    last = torch.tensor([hi - 1 for _, hi in bounds], device="cuda")
    assert float((x_sep.double() - x_true[last].double()).abs().max()) <= tol
    Rb, Ob, yb = sharded.boundary_system(recv, parts, rec_bytes, msg_bytes, d, dtype)
    dense = torch.zeros(parts * d, parts * d, dtype=torch.float64, device="cuda")
    for w in range(parts):
        dense[w * d:(w + 1) * d, w * d:(w + 1) * d] = Rb[w].double()
        if w + 1 < parts:
            dense[(w + 1) * d:(w + 2) * d, w * d:(w + 1) * d] = Ob[w].double()
            dense[w * d:(w + 1) * d, (w + 1) * d:(w + 2) * d] = Ob[w].double().T
    ref = torch.linalg.solve(dense, yb.double().reshape(-1)).reshape(parts, d)
    assert float((x_sep.double() - ref).abs().max()) <= tol
    for r in sorted({0, parts // 2, parts - 1}):
        Pa, pa, dR, dy = sharded.hip_boundary_recursions(recv, parts, msg_bytes, d, dtype, r)
        Pa0, pa0, dR0, dy0 = sharded.boundary_recursions(recv, parts, rec_bytes, msg_bytes, d, dtype, r)
        for a, a0 in ((Pa, Pa0), (pa, pa0), (dR, dR0), (dy, dy0)):
            if a0 is None:
                assert a is None
            else:
                assert float((a.double() - a0.double()).abs().max()) <= tol * max(1.0, float(a0.abs().max()))
