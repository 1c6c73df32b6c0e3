#!/usr/bin/env python3
"""Benchmark of the cyclic-reduction hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one fused solve + log-det (`mahal_and_det`, reference
cyclic_reduction.py:380-438) of one synthetic SPD block-tridiagonal system whose
blocks are already resident in HBM, called through the C ABI (include/cgps.h).

N = 1 : BASELINE.json configs[1] -- N_rows = 2^20, d = 4, fp64.
N > 1 : one system of N * 2^20 block rows, time axis sharded over the ranks
        (2^20 rows per GPU: weak scaling); each rank reduces its shard locally and
        ONE all-gather of the shard-boundary blocks finishes the reduction.
`value` is the whole-job algorithmic GB/s (the "log-det GB/s" of the metric:
compulsory bytes B_A = ((2n-1) d^2 + n d) s + 2s of SURVEY.md 8(d), divided by
the wall time); solves/s is reported beside it.

Extra objects in the JSON line: `roofline` (dominant kernel against the 8 TB/s
HBM peak, timed with HIP events on the launch stream) and `cpu_baseline` (the
oracle = torch-CPU restatement of the reference, timed on the host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:36
ROWS_PER_GPU = 1 << 20
D = 4
DTYPE = torch.float64


def algorithmic_bytes(n, d, s):
    return ((2 * n - 1) * d * d + n * d) * s + 2 * s


def make_system(n, d, dtype, device, seed=1234):
    """Conditioned bidiagonal-factor generator of SURVEY.md 8(d), built on the device."""
    import _util
    return _util.conditioned_system(n, d, dtype=dtype, seed=seed, device=device)


def cpu_baseline(n, d, dtype, reps=3):
    """The oracle (port of the reference's op sequence) on the host cores, same workload."""
    from oracle import cr_oracle
    Rs, Os, b, _, _ = make_system(n, d, dtype, "cpu")
    cores = torch.get_num_threads()
    cr_oracle.mahal_and_det(Rs[: n // 8], Os[: n // 8 - 1], b[: n // 8])       # warm-up
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        cr_oracle.mahal_and_det(Rs, Os, b)
        best = min(best, time.perf_counter() - t0)
    return {"value": 1.0 / best, "unit": "solves/s", "seconds": best,
            "GBps": algorithmic_bytes(n, d, Rs.element_size()) / best / 1e9, "cores": cores, "kind": "port",
            "sample": "full workload N=%d d=%d %s, min of %d runs after a 1/8-size warm-up; torch %s"
                      % (n, d, str(dtype).replace("torch.", ""), reps, torch.__version__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--levelwise", action="store_true", help="time the one-launch-per-level form instead")
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="block rows per GPU")
    ap.add_argument("--d", type=int, default=D)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    d, rows = args.d, args.rows

    from cyclic_gps import _hip
    lib = _hip.lib()

    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        from cyclic_gps import sharded
    n_total = rows * world

    # ---- synthetic system, resident in HBM ------------------------------------------------
    if world == 1:
        Rs, Os, b, x_true, logdet_true = make_system(rows, d, dtype, dev)
        mahal_true = float((x_true.double() * b.double()).sum())
    else:
        Rs, Os, b, O_left, mahal_true, logdet_true = sharded.make_sharded_system(n_total, d, dtype, dev, rank, world)
    out = torch.zeros(2, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    ws, ws_bytes = _hip.workspace(rows + 1, d, dtype, _hip.OP_MAHAL_LOGDET, dev)
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    fn = lib.cgps_mahal_logdet_levelwise if args.levelwise else lib.cgps_mahal_logdet
    dcode = _hip.dtype_code(dtype)

    if world == 1:
        def step():
            _hip.check(fn(_hip.ptr(Rs), _hip.ptr(Os), _hip.ptr(b), rows, d, dcode, _hip.ptr(ws), ws_bytes,
                          _hip.ptr(out), _hip.ptr(info), sp))
    else:
        plan = sharded.ShardedMahalLogdet(Rs, Os, b, O_left, n_total, rank, world)
        info = plan.ops.info

        def step():
            plan.run(out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()

    # ---- timed region: exactly K steps ------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same K steps again with HIP events bracketing the dominant kernel of each step, on
    # the stream it is launched on (kept out of the region above: an event record between two
    # dependent kernels is itself a ~2 us barrier packet and would inflate ms_per_step) ----------
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, c in evs:      # force creation of the underlying hipEvent_t
        a.record(stream)
        c.record(stream)
    barrier()
    t1 = time.perf_counter()
    for a, c in evs:
        lib.cgps_profile_next_call(ctypes.c_void_p(a.cuda_event), ctypes.c_void_p(c.cuda_event))
        step()
    barrier()
    elapsed_with_events = time.perf_counter() - t1
    kernel_ms = sorted(a.elapsed_time(c) for a, c in evs)
    kernel_avg_s = sum(kernel_ms) / len(kernel_ms) / 1e3

    # ---- correctness of what was timed (closed form) ---------------------------------------
    res = out.cpu()
    rel_ld = abs(float(res[1]) - logdet_true) / abs(logdet_true)
    rel_m = abs(float(res[0]) - mahal_true) / abs(mahal_true)
    tol = 1e-5
    assert int(info.item()) == 0, "library reported a non-positive-definite block"
    assert rel_ld < tol and rel_m < tol * 10, ("result mismatch", rel_ld, rel_m)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    s = 8 if dtype == torch.float64 else 4
    t_step = elapsed / args.steps
    b_total = algorithmic_bytes(n_total, d, s)
    b_kernel = algorithmic_bytes(rows, d, s)     # what ONE launch of the dominant kernel streams (one shard)
    line = {
        "metric": "block-tridiag solve+log-det (mahal_and_det) algorithmic GB/s vs HBM roofline, N=2^20 d=4 per GPU; "
                  "solves/s alongside",
        "value": b_total / t_step / 1e9, "unit": "GB/s",
        "solves_per_s": 1.0 / t_step,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if dtype == torch.float64 else "f32", "data": "synthetic",
        "config": {"workload": "mahal_and_det on one SPD block-tridiagonal system, N=%d block rows (%d per GPU), d=%d, "
                               "conditioned bidiagonal-factor generator seed 1234" % (n_total, rows, d),
                   "rows_per_gpu": rows, "d": d, "algorithmic_bytes": b_total,
                   "parallelism": "time-axis shards x%d, one all-gather of boundary blocks" % world if world > 1
                   else "single GPU",
                   "algo": "levelwise" if args.levelwise else "tile-fused"},
        "roofline_frac_whole_op": b_total / t_step / 1e9 / (HBM_PEAK_GBPS * world),
        "roofline": {"bound": "hbm", "achieved": b_kernel / kernel_avg_s / 1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": b_kernel / kernel_avg_s / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "first-pass reduction kernel (streams Rs, Os, x once)",
                     "kernel_avg_us": kernel_avg_s * 1e6, "kernel_min_us": kernel_ms[0] * 1e3,
                     "algorithmic_bytes_per_launch": b_kernel,
                     "ms_per_step_with_event_hooks": elapsed_with_events / args.steps * 1e3},
        "check": {"logdet_rel_err": rel_ld, "mahal_rel_err": rel_m},
    }
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(rows, d, dtype)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
