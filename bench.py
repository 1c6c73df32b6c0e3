#!/usr/bin/env python3
"""Benchmark of the cyclic-reduction hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one fused solve + log-det (`mahal_and_det`, reference
cyclic_reduction.py:380-438) of one synthetic SPD block-tridiagonal system whose
blocks are already resident in HBM, called through the C ABI (include/cgps.h).

N = 1 : BASELINE.json configs[1] -- N_rows = 2^20, d = 4, fp64.
N > 1 : BASELINE.json configs[3] -- ONE system of 2^24 block rows, d = 4, fp64, time axis
        sharded over the N ranks (2^24 / N rows per GPU: STRONG scaling); each rank reduces its
        shard locally and ONE all-gather of the shard-boundary blocks finishes the reduction.
        Rank 0 also times the same 2^24-row system on its GPU alone (outside the timed
        region) so that the line carries the speed-up; the weak-scaling point (2^20 rows per
        GPU) goes to `extras`.
`value` is the whole-job algorithmic GB/s (the "log-det GB/s" of the metric:
compulsory bytes B_A = ((2n-1) d^2 + n d) s + 2s of SURVEY.md 8(d), divided by
the wall time); solves/s is reported beside it.

Extra objects in the JSON line: `roofline` (dominant kernel against the 8 TB/s
HBM peak, timed with HIP events on the launch stream) and `cpu_baseline` (the
oracle = torch-CPU restatement of the reference, timed on the host cores).

Order of the run: the secondary measurements (`extras`) first, then the headline's W warm-up and K
timed steps, then the CPU baseline.  A GPU that has been idle for a long time (a fresh box's first
process) runs its first fraction of a second of kernels in a low power state: measured 77-81 us per
step for the headline under `--steps 20 --warmup 5` as the first GPU work on a box, 73-74.5 us a few
seconds after ANY other GPU work (a previous process, unrelated kernels).  With the secondary
measurements first the headline does not depend on what ran on the box before; `--headline-first`
restores the other order.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# RCCL / CUDA-tensor sharing across processes needs dmabuf IPC on this pool (already exported on
# the GPU boxes; set before HIP initialises in case a launcher dropped it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:36
ROWS_PER_GPU = 1 << 20         # config 2 (and the weak-scaling point)
ROWS_CONFIG4 = 1 << 24         # config 4: ONE system, sharded over the GPUs
PMC_TRAFFIC_FILE = "r02_pmc_traffic.json"
KERNEL_STATS_FILE = "r02_kernel_stats_headline.csv"
D = 4
DTYPE = torch.float64


def algorithmic_bytes(n, d, s):
    return ((2 * n - 1) * d * d + n * d) * s + 2 * s


def make_system(n, d, dtype, device, seed=1234):
    """Conditioned bidiagonal-factor generator of SURVEY.md 8(d), built on the device."""
    import _util
    return _util.conditioned_system(n, d, dtype=dtype, seed=seed, device=device)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _mkl_version():
    try:
        for line in torch.__config__.show().splitlines():
            if "Math Kernel Library" in line or "MKL" in line and "Version" in line:
                return line.strip(" -")
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n, d, dtype, reps=3):
    """The oracle (port of the reference's op sequence) on the host cores, same workload
    (all cores and one thread; nproc, CPU model, torch / MKL versions)."""
    from oracle import cr_oracle
    Rs, Os, b, _, _ = make_system(n, d, dtype, "cpu")
    cores = torch.get_num_threads()

    def best_of(k):
        best = float("inf")
        for _ in range(k):
            t0 = time.perf_counter()
            cr_oracle.mahal_and_det(Rs, Os, b)
            best = min(best, time.perf_counter() - t0)
        return best
    cr_oracle.mahal_and_det(Rs[: n // 8], Os[: n // 8 - 1], b[: n // 8])       # warm-up
    best = best_of(reps)
    torch.set_num_threads(1)
    try:
        cr_oracle.mahal_and_det(Rs[: n // 8], Os[: n // 8 - 1], b[: n // 8])
        best1 = best_of(2)
    finally:
        torch.set_num_threads(cores)
    nbytes = algorithmic_bytes(n, d, Rs.element_size())
    return {"value": 1.0 / best, "unit": "solves/s", "seconds": best, "GBps": nbytes / best / 1e9, "cores": cores,
            "kind": "port",
            "one_thread": {"value": 1.0 / best1, "seconds": best1, "GBps": nbytes / best1 / 1e9, "cores": 1},
            "nproc": os.cpu_count(), "cpu_model": _cpu_model(), "torch": torch.__version__, "mkl": _mkl_version(),
            "sample": "full workload N=%d d=%d %s: min of %d runs on %d threads and min of 2 runs on 1 thread "
                      "(torch.set_num_threads), each after a 1/8-size warm-up"
                      % (n, d, str(dtype).replace("torch.", ""), reps, cores)}


def _time_cuda(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def two_systems_in_flight(dev, rows, d, dtype, steps):
    """Information only (never `value`): throughput when TWO independent systems are solved on two
    HIP streams, so that the serial tail of one (LDS levels, final workgroup: HBM idle) overlaps the
    streaming of the other.  Same library calls, separate workspaces and outputs per stream."""
    from cyclic_gps import _hip
    lib = _hip.lib()
    dcode = _hip.dtype_code(dtype)
    systems = []
    for i in range(2):
        Rs, Os, b, _, logdet_true = make_system(rows, d, dtype, dev, seed=1234 + i)
        ws, ws_bytes = _hip.workspace(rows + 1 + i, d, dtype, _hip.OP_MAHAL_LOGDET, dev)
        ws = ws.clone()                      # the cache may hand the same buffer twice
        systems.append(dict(Rs=Rs, Os=Os, b=b, ws=ws, ws_bytes=ws.numel() * ws.element_size(), logdet=logdet_true,
                            out=torch.zeros(2, dtype=torch.float64, device=dev),
                            info=torch.zeros(1, dtype=torch.int32, device=dev), stream=torch.cuda.Stream(device=dev)))

    def launch(s):
        _hip.check(lib.cgps_mahal_logdet(_hip.ptr(s["Rs"]), _hip.ptr(s["Os"]), _hip.ptr(s["b"]), rows, d, dcode,
                                         _hip.ptr(s["ws"]), s["ws_bytes"], _hip.ptr(s["out"]), _hip.ptr(s["info"]),
                                         ctypes.c_void_p(s["stream"].cuda_stream)))
    torch.cuda.synchronize()
    for _ in range(10):
        for s in systems:
            launch(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for s in systems:
            launch(s)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / (2 * steps)
    errs = [abs(float(s["out"][1]) - s["logdet"]) / abs(s["logdet"]) for s in systems]
    sz = 8 if dtype == torch.float64 else 4
    return {"us_per_solve": t * 1e6, "GBps": algorithmic_bytes(rows, d, sz) / t / 1e9,
            "frac_of_8TBps": algorithmic_bytes(rows, d, sz) / t / 1e9 / HBM_PEAK_GBPS, "logdet_rel_err_max": max(errs),
            "note": "two independent systems on two streams; the headline `value` is one system, one stream"}


def extra_measurements(dev):
    """Secondary numbers of SURVEY.md 8(d), outside the timed region of the headline metric:
    Op B (decompose + solve + det through the module surface, N=2^20 d=4 fp64), BASELINE
    config 3 (N=2^22 d=8 fp32) and config 4's single-GPU point (N=2^24 d=4 fp64)."""
    import cyclic_gps.cyclic_reduction as cr
    cr.CHECK_POSITIVE_DEFINITE = False          # no device->host sync inside the timed calls
    out = {}
    for name, n, d, dtype, reps in (("opB_N2^20_d4_f64", 1 << 20, 4, torch.float64, 10),
                                    ("c3_N2^22_d8_f32", 1 << 22, 8, torch.float32, 20),
                                    ("c4_N2^24_d4_f64_1gpu", 1 << 24, 4, torch.float64, 5)):
        try:
            Rs, Os, b, x_true, logdet_true = make_system(n, d, dtype, dev)
            s = Rs.element_size()
            res = {}
            t = _time_cuda(lambda: cr.mahal_and_det(Rs, Os, b), reps, warm=5)
            res["mahal_and_det_us"] = t * 1e6
            res["mahal_and_det_GBps"] = algorithmic_bytes(n, d, s) / t / 1e9
            m, ld = cr.mahal_and_det(Rs, Os, b)
            res["logdet_rel_err"] = abs(float(ld) - logdet_true) / abs(logdet_true)
            if not name.startswith("c4"):
                holder = {}

                def dec():
                    holder["dec"] = cr.decompose(Rs, Os)
                t_dec = _time_cuda(dec, reps)
                t_sol = _time_cuda(lambda: cr.solve(holder["dec"], b), reps)
                t_det = _time_cuda(lambda: cr.det(holder["dec"]), reps)
                t_inv = _time_cuda(lambda: cr.inverse_blocks(holder["dec"]), max(2, reps // 3))
                xs = cr.solve(holder["dec"], b)
                res.update(decompose_us=t_dec * 1e6, solve_us=t_sol * 1e6, det_us=t_det * 1e6,
                           inverse_blocks_us=t_inv * 1e6,
                           decompose_GBps=5.0 * n * d * d * s / t_dec / 1e9,
                           solve_GBps=(3.0 * n * d * d + 2.0 * n * d) * s / t_sol / 1e9,
                           factor_solves_per_s=1.0 / (t_dec + t_sol),
                           solve_max_abs_err=float((xs.double() - x_true.double()).abs().max()))
                del holder, xs
            if name.startswith("opB"):
                # a training step through the path (models.py:374-381): forward + analytic adjoint
                # (backward = decompose + solve + inverse_blocks on the same kernels)
                Rg, Og, bg = (t_.clone().requires_grad_(True) for t_ in (Rs, Os, b))

                def step():
                    mm, ll = cr.mahal_and_det(Rg, Og, bg)
                    (mm + ll).backward()
                    Rg.grad = Og.grad = bg.grad = None
                res["mahal_and_det_fwd_bwd_us"] = _time_cuda(step, reps) * 1e6
                del Rg, Og, bg
            out[name] = res
            del Rs, Os, b, x_true
            torch.cuda.empty_cache()
        except Exception as e:  # keep the headline line alive whatever happens here
            out[name] = {"error": repr(e)[:200]}
    # BASELINE config 1 (N=1024, d=2, fp64): the launch-latency end of the path
    try:
        Rs, Os, b, x_true, logdet_true = make_system(1024, 2, torch.float64, dev)
        holder = {}

        def dec1():
            holder["dec"] = cr.decompose(Rs, Os)
        out["c1_N1024_d2_f64"] = {
            "mahal_and_det_us": _time_cuda(lambda: cr.mahal_and_det(Rs, Os, b), 50) * 1e6,
            "decompose_us": _time_cuda(dec1, 50) * 1e6,
            "solve_us": _time_cuda(lambda: cr.solve(holder["dec"], b), 50) * 1e6,
            "logdet_rel_err": abs(float(cr.mahal_and_det(Rs, Os, b)[1]) - logdet_true) / abs(logdet_true),
        }
        # the same call captured in a HIP graph and replayed (what a caller's optimiser loop would do)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            cr.mahal_and_det(Rs, Os, b)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gm, gl = cr.mahal_and_det(Rs, Os, b)
        out["c1_N1024_d2_f64"]["mahal_and_det_graph_replay_us"] = _time_cuda(graph.replay, 50) * 1e6
        out["c1_N1024_d2_f64"]["graph_logdet_rel_err"] = abs(float(gl) - logdet_true) / abs(logdet_true)
        del graph
    except Exception as e:
        out["c1_N1024_d2_f64"] = {"error": repr(e)[:200]}
    try:
        out["opA_two_systems_in_flight_N2^20_d4_f64"] = two_systems_in_flight(dev, 1 << 20, 4, torch.float64, 200)
    except Exception as e:
        out["opA_two_systems_in_flight_N2^20_d4_f64"] = {"error": repr(e)[:200]}
    cr.CHECK_POSITIVE_DEFINITE = True
    # BASELINE config 5: LEG log-likelihood + posterior mean on the CO2-shaped series (N=502, rank 5),
    # parameters and expected values from the fixture recorded from the reference
    try:
        import numpy as np
        from cyclic_gps import leg
        g = np.load(os.path.join(ROOT, "tests", "golden", "leg_co2like.npz"))
        t = lambda k: torch.from_numpy(g[k]).to(torch.float64).to(dev)   # noqa: E731
        m = leg.LEGMatrices(t("N"), t("R"), t("B"), t("Lambda"))
        ts, xs = t("ts"), t("xs")
        ll = leg.log_likelihood(m, ts, xs)
        mean = leg.insample_posterior(m, ts, xs)[0]
        res = {"rows": int(ts.shape[0]), "rank": int(m.G.shape[0]),
               "log_likelihood_us": _time_cuda(lambda: leg.log_likelihood(m, ts, xs), 10) * 1e6,
               "insample_posterior_us": _time_cuda(lambda: leg.insample_posterior(m, ts, xs), 10) * 1e6,
               "ll_rel_err_vs_reference": abs(float(ll) - float(g["ll"])) / abs(float(g["ll"]))}
        # the same evaluation captured in a HIP graph and replayed (leg.GraphedLogLikelihood)
        try:
            gll = leg.GraphedLogLikelihood(m, ts, xs)
            res["log_likelihood_graph_replay_us"] = _time_cuda(gll, 20) * 1e6
            res["graph_ll_rel_err_vs_reference"] = abs(float(gll()) - float(g["ll"])) / abs(float(g["ll"]))
            del gll
        except Exception as e:
            res["log_likelihood_graph_replay_us"] = "error: " + repr(e)[:160]
        # the same evaluation with a gradient wanted (operands from cgps_peg_precision through its
        # analytic adjoint, csrc/cgps_leg.h), and a whole training step: forward + backward to the parameters
        mg = leg.LEGMatrices(*(t(k).requires_grad_(True) for k in ("N", "R", "B", "Lambda")))
        res["log_likelihood_with_grad_graph_us"] = _time_cuda(lambda: leg.log_likelihood(mg, ts, xs), 5) * 1e6

        def train_step():
            leg.log_likelihood(mg, ts, xs).backward()
            for p_ in (mg.N, mg.R, mg.B, mg.Lambda):
                p_.grad = None
        res["log_likelihood_fwd_bwd_us"] = _time_cuda(train_step, 5) * 1e6
        try:      # the same training evaluation captured in a HIP graph (leg.GraphedValueAndGrad)
            gv = leg.GraphedValueAndGrad(mg, ts, xs)
            res["log_likelihood_fwd_bwd_graph_replay_us"] = _time_cuda(gv, 20) * 1e6
            del gv
            for p_ in (mg.N, mg.R, mg.B, mg.Lambda):
                p_.grad = None
        except Exception as e:
            res["log_likelihood_fwd_bwd_graph_replay_us"] = "error: " + repr(e)[:160]
        from cyclic_gps import predict
        tt = torch.from_numpy(g["target_ts"]).to(dev) if "target_ts" in g.files else ts
        pm = predict.make_predictions(m, ts, xs, tt)[0]
        res["make_predictions_us"] = _time_cuda(lambda: predict.make_predictions(m, ts, xs, tt), 5) * 1e6
        res["make_predictions_targets"] = int(tt.shape[0])
        try:      # posterior + predictions captured in a HIP graph (leg.Graphed)
            gp = leg.Graphed(predict.make_predictions, m, ts, xs, tt, check_sorted=False)
            res["make_predictions_graph_replay_us"] = _time_cuda(gp, 20) * 1e6
            res["graph_prediction_mean_max_abs_err"] = float((gp()[0] - pm).abs().max())
            del gp
        except Exception as e:
            res["make_predictions_graph_replay_us"] = "error: " + repr(e)[:160]
        if "pred_mean" in g.files:
            res["prediction_mean_max_abs_err_vs_reference"] = float((pm.cpu() - torch.from_numpy(g["pred_mean"])).abs().max())
        if "post_mean" in g.files:
            res["posterior_mean_max_abs_err_vs_reference"] = float(
                (mean.cpu() - torch.from_numpy(g["post_mean"])).abs().max())
        out["c5_leg_co2like"] = res
    except Exception as e:
        out["c5_leg_co2like"] = {"error": repr(e)[:200]}
    return out


def _timed_steps(step, steps, warmup, barrier, region_events=None):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides.
    region_events = (start, stop): HIP events recorded on the launch stream right before the first and
    right after the last timed step (they bracket steps 2..K: nothing between those steps)."""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    step()
    if region_events is not None:
        region_events[0].record()                 # after the first launch has left the host: it is not delayed
    for _ in range(steps - 1):
        step()
    if region_events is not None:
        region_events[1].record()
    barrier()
    return time.perf_counter() - t0


def _file_sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (Op B / config 3 / 2^24) numbers")
    ap.add_argument("--headline-first", action="store_true",
                    help="time the headline before the secondary measurements (on a long-idle GPU it then runs in the device's low power state)")
    ap.add_argument("--levelwise", action="store_true", help="time the one-launch-per-level form instead")
    ap.add_argument("--rows", type=int, default=0,
                    help="block rows of the WHOLE system (default: 2^20 on one GPU = config 2; 2^24 on several = config 4)")
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
    # CGPS_BENCH_REHEARSAL_GLOO=1: several ranks on ONE GPU over gloo -- exercises the multi-process
    # flow (shard kernels with a left neighbour, the collective, the finish kernel) on a 1-GPU box;
    # its timing means nothing
    rehearsal = os.environ.get("CGPS_BENCH_REHEARSAL_GLOO") == "1"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    d = args.d

    from cyclic_gps import _hip
    lib = _hip.lib()

    # CGPS_BENCH_FORCE_SHARDED=1 runs the sharded code path with a single rank (rehearsal on a 1-GPU box)
    force_sharded = os.environ.get("CGPS_BENCH_FORCE_SHARDED") == "1"
    sharded_mode = world > 1 or force_sharded
    use_dist = world > 1 or (force_sharded and "RANK" in os.environ)
    if sharded_mode:
        import torch.distributed as dist
        from cyclic_gps import sharded
        if use_dist:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
    # the workload: config 2 on one GPU; config 4 (ONE 2^24-row system, strong scaling) on several
    n_total = args.rows if args.rows > 0 else (ROWS_CONFIG4 if world > 1 else ROWS_PER_GPU)
    if sharded_mode:
        lo, hi = sharded.shard_bounds(n_total, world, rank)
        rows = hi - lo
    else:
        rows = n_total

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic system, resident in HBM ------------------------------------------------
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    fn = lib.cgps_mahal_logdet_levelwise if args.levelwise else lib.cgps_mahal_logdet
    dcode = _hip.dtype_code(dtype)
    out = torch.zeros(2, dtype=torch.float64, device=dev)

    def whole_system_step(n):
        """(step, info, mahal_true, logdet_true) for ONE n-row system on this GPU alone."""
        Rs_, Os_, b_, x_true_, logdet_ = make_system(n, d, dtype, dev)
        mahal_ = float((x_true_.double() * b_.double()).sum())
        del x_true_
        info_ = torch.zeros(1, dtype=torch.int32, device=dev)
        ws_, ws_bytes_ = _hip.workspace(n + 1, d, dtype, _hip.OP_MAHAL_LOGDET, dev)

        # the argument objects are built once: a step is then one foreign call (the first timed launch leaves
        # the host ~8 us earlier than when every pointer is wrapped again per call; steady state is GPU-bound
        # either way)
        cargs = (_hip.ptr(Rs_), _hip.ptr(Os_), _hip.ptr(b_), n, d, dcode, _hip.ptr(ws_), ws_bytes_,
                 _hip.ptr(out), _hip.ptr(info_), sp)
        keep = (Rs_, Os_, b_, ws_)

        def step_(cargs=cargs, keep=keep):
            rc = fn(*cargs)
            if rc != 0:
                _hip.check(rc)
        return step_, info_, mahal_, logdet_

    extras_early = None
    if not sharded_mode and not args.no_extras and not args.headline_first:
        extras_early = extra_measurements(dev)          # see the module docstring: order of the run
        torch.cuda.empty_cache()
    if not sharded_mode:
        step, info, mahal_true, logdet_true = whole_system_step(rows)
    else:
        Rs, Os, b, O_left, mahal_true, logdet_true = sharded.make_sharded_system(n_total, d, dtype, dev, rank, world)
        plan = sharded.ShardedMahalLogdet(Rs, Os, b, O_left, n_total, rank, world)
        info = plan.ops.info

        def step():
            plan.run(out)


    def weak_point():
        """Weak-scaling point (information only): 2^20 rows per GPU, one system of world * 2^20 rows."""
        try:
            nw = ROWS_PER_GPU * world
            sz = 8 if dtype == torch.float64 else 4
            Rw, Ow, bw, Olw, mw, ldw = sharded.make_sharded_system(nw, d, dtype, dev, rank, world)
            planw = sharded.ShardedMahalLogdet(Rw, Ow, bw, Olw, nw, rank, world)
            ew = _timed_steps(lambda: planw.run(out), args.steps, args.warmup, barrier)
            tw = torch.tensor([ew], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            ew = float(tw.item()) / args.steps
            rw = out.cpu()
            return {"rows_total": nw, "ms_per_step": ew * 1e3, "GBps": algorithmic_bytes(nw, d, sz) / ew / 1e9,
                    "logdet_rel_err": abs(float(rw[1]) - ldw) / abs(ldw)}
        except Exception as e:
            return {"error": repr(e)[:200]}

    # several GPUs: the weak-scaling point first, for the same reason as the secondary measurements on one GPU
    # (module docstring: a long-idle GPU's first kernels run in a low power state; it also brings RCCL up)
    weak_early = weak_point() if (sharded_mode and world > 1 and not args.headline_first) else None

    # ---- timed region: W warm-up steps, then exactly K steps ----------------------------------
    region_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    elapsed = _timed_steps(step, args.steps, args.warmup, barrier, region_ev)
    region_avg_s = region_ev[0].elapsed_time(region_ev[1]) / max(args.steps - 1, 1) / 1e3     # launch-to-launch time inside the timed region (steps 2..K)
    if use_dist:
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
    bracketed_avg_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    # Two upper bounds of the dominant kernel's duration: events around each single launch (kernel + the two
    # event packets: a bubble of ~1-2 us between back-to-back kernels), and -- when a step IS one launch of
    # that kernel (the record stages run inside it) -- events around the K timed steps divided by K (kernel +
    # launch-to-launch gap).  The tighter one is reported; rocprofv3's own begin/end stamps of the same command
    # are in profiles/ (kernel_avg_us_rocprofv3).
    one_launch_per_step = ((not args.levelwise) and (not sharded_mode) and rows <= (1 << 22) and d == 4
                           and dtype == torch.float64 and os.environ.get("CGPS_NO_FOLD") != "1")
    kernel_avg_s = min(bracketed_avg_s, region_avg_s) if one_launch_per_step else bracketed_avg_s

    # ---- correctness of what was timed (closed form) ---------------------------------------
    res = out.cpu()
    rel_ld = abs(float(res[1]) - logdet_true) / abs(logdet_true)
    rel_m = abs(float(res[0]) - mahal_true) / abs(mahal_true)
    tol = 1e-5
    assert int(info.item()) == 0, "library reported a non-positive-definite block"
    assert rel_ld < tol and rel_m < tol * 10, ("result mismatch", rel_ld, rel_m)

    s = 8 if dtype == torch.float64 else 4
    t_step = elapsed / args.steps
    b_total = algorithmic_bytes(n_total, d, s)
    b_kernel = algorithmic_bytes(rows, d, s)     # what ONE launch of the dominant kernel streams (one shard)

    # ---- several GPUs: the single-GPU time of the SAME system (speed-up) and the weak-scaling point,
    # outside the timed region ----------------------------------------------------------------
    scaling_extras = None
    if sharded_mode and world > 1:
        scaling_extras = {}
        del plan, Rs, Os, b
        torch.cuda.empty_cache()
        k1 = max(3, min(10, args.steps))
        if rank == 0:
            try:
                step1, info1, m1, ld1 = whole_system_step(n_total)
                e1 = _timed_steps(step1, k1, 2, torch.cuda.synchronize) / k1
                r1 = out.cpu()
                scaling_extras["single_gpu_same_system"] = {
                    "rows": n_total, "ms_per_step": e1 * 1e3, "GBps": b_total / e1 / 1e9,
                    "logdet_rel_err": abs(float(r1[1]) - ld1) / abs(ld1), "steps": k1}
                scaling_extras["speedup_vs_single_gpu"] = e1 / t_step
                del step1
            except Exception as e:
                scaling_extras["single_gpu_same_system"] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
        barrier()
        if weak_early is not None:
            scaling_extras["weak_2^20_rows_per_gpu"] = weak_early
        else:
            scaling_extras["weak_2^20_rows_per_gpu"] = weak_point()

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    cfgname = "config 4" if n_total == ROWS_CONFIG4 else ("config 2" if n_total == ROWS_PER_GPU else "custom size")
    line = {
        "metric": "block-tridiag solve+log-det (mahal_and_det) algorithmic GB/s vs HBM roofline, N=2^20 d=4 on one GPU, "
                  "N=2^24 d=4 time-axis sharded on several; solves/s alongside",
        "value": b_total / t_step / 1e9, "unit": "GB/s",
        "solves_per_s": 1.0 / t_step,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3,
        "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": "f64" if dtype == torch.float64 else "f32", "data": "synthetic",
        "config": {"workload": "BASELINE %s: mahal_and_det on ONE SPD block-tridiagonal system, N=%d block rows (%d per "
                               "GPU), d=%d, conditioned bidiagonal-factor generator seed 1234" % (cfgname, n_total, rows, d),
                   "rows_total": n_total, "rows_per_gpu": rows, "d": d, "algorithmic_bytes": b_total,
                   "parallelism": "time-axis shards x%d, one all-gather of boundary blocks" % world if world > 1
                   else "single GPU",
                   "algo": "levelwise" if args.levelwise else "tile-fused"},
        "roofline_frac_whole_op": b_total / t_step / 1e9 / (HBM_PEAK_GBPS * world),
        "roofline": {"bound": "hbm", "achieved": b_kernel / kernel_avg_s / 1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": b_kernel / kernel_avg_s / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "first-pass reduction kernel (streams Rs, Os, x once)",
                     "kernel_avg_us": kernel_avg_s * 1e6, "kernel_min_us": kernel_ms[0] * 1e3,
                     "kernel_avg_us_events_around_each_launch": bracketed_avg_s * 1e6,
                     "kernel_avg_us_events_around_the_timed_region": region_avg_s * 1e6 if one_launch_per_step else None,
                     "algorithmic_bytes_per_launch": b_kernel,
                     "ms_per_step_with_event_hooks": elapsed_with_events / args.steps * 1e3},
        "check": {"logdet_rel_err": rel_ld, "mahal_rel_err": rel_m},
    }
    if scaling_extras is not None:
        line["extras"] = scaling_extras
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process;
    # the number comes from the committed rocprofv3 --pmc passes of this same command
    # (profiles/<round>_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, calibrated as the guide says).
    # The file names the kernel sources it was measured on; when they have changed since, the figure
    # is reported as stale instead of being passed off as current.
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
        if rows == 1 << 20 and d == 4 and dtype == torch.float64 and not args.levelwise and world == 1:
            line["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
            line["roofline"]["traffic_source"] = "profiles/" + PMC_TRAFFIC_FILE
            sha = pmc.get("kernel_source_sha16")
            if sha is not None:
                cur = {f: _file_sha16(os.path.join(ROOT, "cyclic-gps_amd", "csrc", f)) for f in sha}
                line["roofline"]["traffic_stale"] = cur != sha
            # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats
            # summary of this command (HIP events around a launch also see its dispatch latency)
            import csv
            prof = "profiles/" + KERNEL_STATS_FILE
            for r in csv.DictReader(open(os.path.join(ROOT, prof))):
                if "chunk_reduce_kernel" in r["Name"]:
                    line["roofline"]["kernel_avg_us_rocprofv3"] = float(r["AverageNs"]) / 1e3
                    line["roofline"]["kernel_avg_us_rocprofv3_source"] = prof
                    break
    except Exception:
        pass
    if not sharded_mode and not args.no_extras:
        line["extras"] = extras_early if extras_early is not None else extra_measurements(dev)
        line["extras"]["order"] = "before the headline" if extras_early is not None else "after the headline"
    if not args.no_cpu_baseline and not sharded_mode:
        line["cpu_baseline"] = cpu_baseline(rows, d, dtype)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
