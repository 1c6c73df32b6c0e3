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

import socket      # noqa: E402
import subprocess  # noqa: E402
import threading   # noqa: E402

# torch is imported by the RANK processes only (_load_torch): the launcher of a several-GPU run
# (`python bench.py --gpus N` with no WORLD_SIZE in the environment) starts N fresh children and must
# not create a GPU context -- or even load torch -- itself.
torch = None


def _load_torch():
    global torch
    if torch is None:
        import torch as _torch
        torch = _torch
    return torch


ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:36
ROWS_PER_GPU = 1 << 20         # config 2 (and the weak-scaling point)
ROWS_CONFIG4 = 1 << 24         # config 4: ONE system, sharded over the GPUs
PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"
PMC_TRAFFIC_SHARD_FILE = "r03_pmc_traffic_shards.json"     # per-shard figures, keyed by rows per rank
KERNEL_STATS_FILE = "r03_kernel_stats_headline.csv"
D = 4


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


def cpu_baseline(n, d, dtype, reps=3, sample_note=None, threads=None):
    """The oracle (port of the reference's op sequence) on the host cores, same workload
    (Op A = mahal_and_det and Op B = decompose + solve, all cores and one
    thread; nproc, CPU model, torch / MKL versions)."""
    from oracle import cr_oracle
    Rs, Os, b, _, _ = make_system(n, d, dtype, "cpu")
    if threads:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()

    def best_of(fn, k):
        best = float("inf")
        for _ in range(k):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    def op_a():
        cr_oracle.mahal_and_det(Rs, Os, b)

    def op_b():
        cr_oracle.solve(cr_oracle.decompose(Rs, Os), b)
    m = max(n // 8, 2)
    cr_oracle.mahal_and_det(Rs[:m], Os[:m - 1], b[:m])       # warm-up
    best = best_of(op_a, reps)
    best_b = best_of(op_b, 2)
    torch.set_num_threads(1)
    try:
        cr_oracle.mahal_and_det(Rs[:m], Os[:m - 1], b[:m])
        best1 = best_of(op_a, 2)
        best1_b = best_of(op_b, 1)
    finally:
        torch.set_num_threads(cores)
    s = Rs.element_size()
    nbytes = algorithmic_bytes(n, d, s)
    nbytes_b = (5.0 * n * d * d + 3.0 * n * d * d + 2.0 * n * d) * s       # B_dec + B_solve of SURVEY.md 8(d)
    return {"value": 1.0 / best, "unit": "solves/s", "seconds": best, "GBps": nbytes / best / 1e9, "cores": cores,
            "kind": "port",
            "one_thread": {"value": 1.0 / best1, "seconds": best1, "GBps": nbytes / best1 / 1e9, "cores": 1},
            "opB_decompose_plus_solve": {"value": 1.0 / best_b, "unit": "factor+solves/s", "seconds": best_b,
                                         "GBps": nbytes_b / best_b / 1e9, "cores": cores,
                                         "one_thread": {"value": 1.0 / best1_b, "seconds": best1_b, "cores": 1}},
            "nproc": os.cpu_count(), "cpu_model": _cpu_model(), "torch": torch.__version__, "mkl": _mkl_version(),
            "sample": (sample_note + ": " if sample_note else "full workload ") +
                      "N=%d d=%d %s; Op A (mahal_and_det): min of %d runs on %d threads and min of 2 runs on 1 thread "
                      "(torch.set_num_threads), each after a 1/8-size warm-up; Op B (decompose + solve): min of 2 runs / 1 run"
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
                # factor and solve in one call (cgps_decompose_solve: the forward substitution rides along in the first
                # pass of the factorisation, the factor is read once instead of twice)
                t_ds = _time_cuda(lambda: cr.decompose_solve(Rs, Os, b), reps)
                res.update(decompose_solve_us=t_ds * 1e6, factor_solves_per_s_one_call=1.0 / t_ds,
                           decompose_solve_max_abs_err=float((cr.decompose_solve(Rs, Os, b)[1].double() - x_true.double()).abs().max()))
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
    # How much of the steady-state headline is the 256 MB memory-side cache?  The same call alternating over TWO systems
    # (604 MB: every call streams operands that are not cached) against one system streamed again and again.
    try:
        sysA = make_system(1 << 20, 4, torch.float64, dev, seed=1234)[:3]
        sysB = make_system(1 << 20, 4, torch.float64, dev, seed=4321)[:3]
        pair = (sysA, sysB)

        def alt(k, reps):
            for i in range(60):
                cr.mahal_and_det(*pair[i % k])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                cr.mahal_and_det(*pair[i % k])
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6
        one, two = alt(1, 200), alt(2, 200)
        out["opA_memory_side_cache_N2^20_d4_f64"] = {
            "same_system_every_call_us": one, "two_systems_alternating_us": two,
            "frac_of_8TBps_without_cache_reuse": algorithmic_bytes(1 << 20, 4, 8) / (two * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "note": "through the Python surface; the headline streams ONE system again and again, as the bench contract asks: "
                    "part of its 302 MB is still in the 256 MB Infinity Cache at the next call; operands that change "
                    "address or exceed the cache get the second figure"}
        del sysA, sysB, pair
        torch.cuda.empty_cache()
    except Exception as e:
        out["opA_memory_side_cache_N2^20_d4_f64"] = {"error": repr(e)[:200]}
    # the per-rank part of the 8-GPU run of config 4: ONE 2^21-row shard through the sharded code path
    # (cgps_shard_reduce + cgps_finish_records, everything but the collective), and the same with 2 sub-shards
    try:
        from cyclic_gps import sharded
        n21 = 1 << 21
        Rw, Ow, bw, Olw, mw, ldw = sharded.make_sharded_system(n21, 4, torch.float64, dev, 0, 1)
        o2 = torch.zeros(2, dtype=torch.float64, device=dev)
        res = {}
        for S in (1, 2):
            planw = sharded.ShardedMahalLogdet(Rw, Ow, bw, Olw, n21, 0, 1, sub_shards=S)
            res["sub_shards_%d_us" % S] = _time_cuda(lambda: planw.run(o2), 100, warm=20) * 1e6
            res["sub_shards_%d_logdet_rel_err" % S] = abs(float(o2[1]) - ldw) / abs(ldw)
            del planw
        res["whole_system_same_rows_us"] = _time_cuda(lambda: cr.mahal_and_det(Rw, Ow, bw), 100, warm=20) * 1e6
        out["c4_per_rank_shard_2^21_d4_f64"] = res
        del Rw, Ow, bw
        torch.cuda.empty_cache()
    except Exception as e:
        out["c4_per_rank_shard_2^21_d4_f64"] = {"error": repr(e)[:200]}
    # fp64 d = 5 (the LEG rank of config 5) at 2^20 rows, and solve with eight right-hand sides at config 2
    try:
        Rs, Os, b, x_true, logdet_true = make_system(1 << 20, 5, torch.float64, dev)
        t5 = _time_cuda(lambda: cr.mahal_and_det(Rs, Os, b), 20, warm=5)
        out["d5_N2^20_f64"] = {"mahal_and_det_us": t5 * 1e6, "frac_of_8TBps": algorithmic_bytes(1 << 20, 5, 8) / t5 / 1e9 / HBM_PEAK_GBPS,
                               "logdet_rel_err": abs(float(cr.mahal_and_det(Rs, Os, b)[1]) - logdet_true) / abs(logdet_true)}
        del Rs, Os, b, x_true
        Rs, Os, b, x_true, _ = make_system(1 << 20, 4, torch.float64, dev)
        dec = cr.decompose(Rs, Os)
        Y = (b[:, :, None] * torch.arange(1, 9, dtype=b.dtype, device=dev)).contiguous()
        t8 = _time_cuda(lambda: cr.solve(dec, Y), 10)
        X = cr.solve(dec, Y)
        out["solve_8rhs_N2^20_d4_f64"] = {"solve_us": t8 * 1e6, "us_per_column": t8 * 1e6 / 8,
                                          "max_abs_err": float((X[:, :, 2] - 3.0 * x_true).abs().max())}
        del Rs, Os, b, x_true, dec, Y, X
        torch.cuda.empty_cache()
    except Exception as e:
        out["d5_and_8rhs"] = {"error": repr(e)[:200]}
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
        # the same reductions at 2^20 rows on a regular grid, rank 5: operands assembled in registers inside the first
        # pass (cgps_leg_mahal_logdet: only ts and v are read from HBM) against cgps_peg_precision -> blocks in HBM ->
        # cgps_mahal_logdet; PMC traffic of the fused launch: profiles/r03_pmc_traffic_leg_fused.txt
        try:
            nl = 1 << 20
            G5 = m.G
            A5 = (m.B.T @ m.LLT_inv @ m.B).contiguous()
            tsl = 0.25 * torch.arange(nl, dtype=torch.float64, device=dev)
            vl = torch.randn(nl, G5.shape[0], dtype=torch.float64, device=dev)
            cr.CHECK_POSITIVE_DEFINITE = False
            t_f = _time_cuda(lambda: leg.leg_mahal_and_det(tsl, G5, A5, vl), 10)
            Rl, Ol = leg.peg_precision(tsl, G5)
            Kl = Rl + A5
            t_a = _time_cuda(lambda: leg.peg_precision(tsl, G5), 5)
            t_m = _time_cuda(lambda: cr.mahal_and_det(Kl, Ol, vl), 10)
            f_m, f_l = leg.leg_mahal_and_det(tsl, G5, A5, vl)
            u_m, u_l = cr.mahal_and_det(Kl, Ol, vl)
            out["leg_fused_N2^20_rank5_regular_grid"] = {
                "fused_us": t_f * 1e6, "unfused_assembly_us": t_a * 1e6, "unfused_mahal_and_det_us": t_m * 1e6,
                "hbm_bytes_read_by_the_fused_launch_algorithmic": nl * 8 * (1 + G5.shape[0]),
                "logdet_rel_diff_fused_vs_unfused": abs(float(f_l) - float(u_l)) / abs(float(u_l)),
                "mahal_rel_diff_fused_vs_unfused": abs(float(f_m) - float(u_m)) / abs(float(u_m))}
            del Rl, Ol, Kl, tsl, vl
            torch.cuda.empty_cache()
        except Exception as e:
            out["leg_fused_N2^20_rank5_regular_grid"] = {"error": repr(e)[:200]}
        finally:
            cr.CHECK_POSITIVE_DEFINITE = True
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


def kernel_source_sha16():
    """One digest over every file the headline translation unit is built from (csrc/cgps_mahal.hip, every
    header under csrc/ and include/cgps.h): profiles/*pmc_traffic*.json records it, and a traffic figure
    measured on other sources is reported as stale."""
    import hashlib
    csrc = os.path.join(ROOT, "cyclic-gps_amd", "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith(".h")) + ["cgps_mahal.hip"]
    h = hashlib.sha256()
    for f in names:
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "cgps.h"), "rb").read())
    return h.hexdigest()[:16]


# ---- several GPUs: the launcher --------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, child_cmd=None, timeout_s=None):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N fresh rank processes of this
    script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), relay rank 0's JSON line and
    return the first non-zero exit status (0 when every rank succeeded).  This process makes NO GPU call and
    never loads torch; it never re-execs.  A rank that dies takes the others down after a grace period (each
    by its own PID), so a failed collective cannot leave the job hanging.
    child_cmd: the command of a rank (tests pass a stand-in); default: this script with the same arguments."""
    if child_cmd is None:
        child_cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    if timeout_s is None:
        timeout_s = float(os.environ.get("CGPS_BENCH_TIMEOUT", "1500"))
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CGPS_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(child_cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True))
    lines = [[] for _ in range(n)]

    def pump(r):
        for ln in procs[r].stdout:
            lines[r].append(ln)
            if r != 0:                       # other ranks print nothing on stdout; whatever they do goes to stderr
                sys.stderr.write("[rank %d] %s" % (r, ln))
    threads = [threading.Thread(target=pump, args=(r,), daemon=True) for r in range(n)]
    for t in threads:
        t.start()
    t0 = time.monotonic()
    failed_at = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        now = time.monotonic()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = now
        if (failed_at is not None and now - failed_at > 15.0) or now - t0 > timeout_s:
            for p in procs:                  # exact PIDs of the children this process started
                if p.poll() is None:
                    p.terminate()
            time.sleep(5.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.1)
    for p in procs:
        p.wait()
    for t in threads:
        t.join(timeout=5.0)
    codes = [p.returncode for p in procs]
    # the status of a rank that failed by itself comes first; ranks this launcher ended have negative codes
    rc = next((c for c in codes if c > 0), next((c for c in codes if c != 0), 0))
    out = [ln for ln in lines[0] if ln.lstrip().startswith("{")]
    for ln in lines[0]:
        if not ln.lstrip().startswith("{"):
            sys.stderr.write("[rank 0] " + ln)
    if out:
        sys.stdout.write(out[-1] if out[-1].endswith("\n") else out[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
    if rc != 0:
        sys.stderr.write("bench.py launcher: rank exit codes %s\n" % codes)
    return rc if rc > 0 else (1 if rc != 0 else 0)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (Op B / config 3 / 2^24) numbers")
    ap.add_argument("--headline-first", action="store_true",
                    help="time ONLY the warm protocol's order reversed: headline before the secondary measurements")
    ap.add_argument("--levelwise", action="store_true", help="time the one-launch-per-level form instead")
    ap.add_argument("--rows", type=int, default=0,
                    help="block rows of the WHOLE system (default: 2^20 on one GPU = config 2; 2^24 on several = config 4)")
    ap.add_argument("--d", type=int, default=D)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--sub-shards", type=int, default=0,
                    help="several GPUs: sub-shards per rank (0 = the library's default for the shard size)")
    ap.add_argument("--weak-point", action="store_true", help="several GPUs: also time 2^20 rows per GPU (extras)")
    ap.add_argument("--prewarm", type=int, default=100,
                    help="one GPU: untimed launches of the headline step before the W warm-up steps (steady state of repeated "
                         "evaluation on the same buffers; 0: none)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank_main(args)


_JSON_FD = None      # several ranks: the descriptor the ONE JSON line goes to (see rank_main)


def _emit(line):
    data = (json.dumps(line) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def rank_main(args):
    global _JSON_FD
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # RCCL prints a banner (versions, host name, library path) on STDOUT when a communicator comes up, gloo its
        # "connected to N peer ranks" lines: with several ranks everything any library prints is sent to stderr, and
        # stdout carries nothing but rank 0's JSON line
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)
    _load_torch()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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

    # CGPS_BENCH_FORCE_SHARDED=1 runs the sharded code path with a single rank (the per-rank part of the
    # several-GPU run, measured on a 1-GPU box)
    force_sharded = os.environ.get("CGPS_BENCH_FORCE_SHARDED") == "1"
    sharded_mode = world > 1 or force_sharded
    use_dist = world > 1
    if sharded_mode:
        from cyclic_gps import sharded
    if use_dist:
        import datetime
        import torch.distributed as dist
        to = datetime.timedelta(seconds=float(os.environ.get("CGPS_BENCH_COLLECTIVE_TIMEOUT", "300")))
        if rehearsal:
            dist.init_process_group("gloo", timeout=to)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=to)
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

    plan = None
    if not sharded_mode:
        step, info, mahal_true, logdet_true = whole_system_step(rows)
    else:
        Rs, Os, b, O_left, mahal_true, logdet_true = sharded.make_sharded_system(n_total, d, dtype, dev, rank, world)
        plan = sharded.ShardedMahalLogdet(Rs, Os, b, O_left, n_total, rank, world,
                                          sub_shards=(args.sub_shards or None))
        info = plan.ops.info

        def step():
            plan.run(out)

    # ---- one GPU: the COLD protocol first.  The same W + K steps as the first GPU work of this process (on a
    # fresh box: of the box).  A GPU that has idled runs its first fraction of a second in a low power state and
    # a system that is streamed for the first time finds nothing of itself in the memory-side cache: this is
    # what a caller whose operands change every call sees.  Reported in extras.headline_cold; the headline
    # `value` below is the warm protocol (after the secondary measurements), as in rounds 1 and 2. -----------
    headline_cold = None
    s_el = 8 if dtype == torch.float64 else 4
    if not sharded_mode and not args.headline_first:
        e_cold = _timed_steps(step, args.steps, args.warmup, barrier) / args.steps
        headline_cold = {"ms_per_step": e_cold * 1e3, "GBps": algorithmic_bytes(rows, d, s_el) / e_cold / 1e9,
                         "frac_of_8TBps": algorithmic_bytes(rows, d, s_el) / e_cold / 1e9 / HBM_PEAK_GBPS,
                         "note": "the same %d warm-up + %d timed steps as the FIRST GPU work of the process, before the "
                                 "secondary measurements; `value` is the same protocol after them and after extras.prewarm_steps "
                                 "untimed launches of the same step" % (args.warmup, args.steps)}
    extras_early = None
    if not sharded_mode and not args.no_extras and not args.headline_first:
        extras_early = extra_measurements(dev)          # see the module docstring: order of the run
        torch.cuda.empty_cache()

    # several GPUs: bring the GPU out of its idle power state (and RCCL up) with untimed steps of the SAME plan
    # -- a fixed count on every rank, so the ranks stay in step -- before the W warm-up and the K timed steps
    # one GPU: the secondary measurements have just streamed ~30 GB of other systems through the GPU: the 256 MB
    # memory-side cache holds nothing of the headline system, and the first ~50-80 launches over the same 302 MB run
    # 2-4 us slower than the steady state they converge to.  A caller that evaluates again and again on
    # the same buffers (an optimiser loop) lives in that steady state, so the W + K protocol steps are preceded by
    # untimed launches of the same step; the line reports their number (extras.prewarm_steps) and, beside the value,
    # the same protocol as the first GPU work of the process (extras.headline_cold).  --prewarm 0 switches them off.
    prewarm = int(os.environ.get("CGPS_BENCH_PREWARM_STEPS", "2000" if sharded_mode else str(args.prewarm)))
    if sharded_mode or prewarm > 0:
        for _ in range(prewarm):
            step()
        barrier()

    # ---- timed region: W warm-up steps, then exactly K steps ----------------------------------
    region_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    elapsed = _timed_steps(step, args.steps, args.warmup, barrier, region_ev)
    region_avg_s = region_ev[0].elapsed_time(region_ev[1]) / max(args.steps - 1, 1) / 1e3     # launch period inside the timed region (steps 2..K)
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
    # ONE estimator for roofline.achieved: HIP events around each launch of the dominant kernel (an upper
    # bound of its duration: kernel + the two event packets); min <= avg by construction.  The launch period
    # inside the timed region (events around steps 2..K, divided by K-1) and rocprofv3's own figure from the
    # committed profile of this command are side fields.
    kernel_avg_s = sum(kernel_ms) / len(kernel_ms) / 1e3

    # ---- correctness of what was timed (closed form) ---------------------------------------
    res = out.cpu()
    rel_ld = abs(float(res[1]) - logdet_true) / abs(logdet_true)
    rel_m = abs(float(res[0]) - mahal_true) / abs(mahal_true)
    tol = 1e-5
    assert int(info.item()) == 0, "library reported a non-positive-definite block"
    assert rel_ld < tol and rel_m < tol * 10, ("result mismatch", rel_ld, rel_m)

    s = s_el
    t_step = elapsed / args.steps
    b_total = algorithmic_bytes(n_total, d, s)
    b_kernel = algorithmic_bytes(rows, d, s)     # what the dominant kernel streams per step on this rank (one shard)

    # ---- several GPUs: the single-GPU time of the SAME system (speed-up), outside the timed region ------
    scaling_extras = None
    if sharded_mode and world > 1:
        scaling_extras = {"sub_shards_per_rank": plan.sub_shards, "records_per_rank": plan.records_per_rank}
        if args.weak_point:
            scaling_extras["weak_2^20_rows_per_gpu"] = _weak_point(args, dev, dtype, d, rank, world, out, barrier)
        del plan, Rs, Os, b
        torch.cuda.empty_cache()
        barrier()
        dist.destroy_process_group()             # no collective after this point: the other ranks are done
        use_dist = False
        if rank != 0:
            return
        k1 = max(3, min(10, args.steps))
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
    elif sharded_mode:
        scaling_extras = {"sub_shards_per_rank": plan.sub_shards, "records_per_rank": plan.records_per_rank}

    if rank != 0:
        return
    cfgname = "config 4" if n_total == ROWS_CONFIG4 else ("config 2" if n_total == ROWS_PER_GPU else "custom size")
    line = {
        "metric": "block-tridiag solve+log-det (mahal_and_det) algorithmic GB/s vs HBM roofline, N=2^20 d=4 on one GPU, "
                  "N=2^24 d=4 time-axis sharded on several; solves/s alongside",
        "value": b_total / t_step / 1e9, "unit": "GB/s",
        "solves_per_s": 1.0 / t_step,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3,
        "higher_is_better": True, "scaling": "strong" if world > 1 else "none", "vs_baseline": None,
        "dtype": "f64" if dtype == torch.float64 else "f32", "data": "synthetic",
        "config": {"workload": "BASELINE %s: mahal_and_det on ONE SPD block-tridiagonal system, N=%d block rows (%d per "
                               "GPU), d=%d, conditioned bidiagonal-factor generator seed 1234" % (cfgname, n_total, rows, d),
                   "rows_total": n_total, "rows_per_gpu": rows, "d": d, "algorithmic_bytes": b_total,
                   "parallelism": "time-axis shards x%d, one all-gather of boundary records" % world if world > 1
                   else "single GPU",
                   "algo": "levelwise" if args.levelwise else "tile-fused"},
        "roofline_frac_whole_op": b_total / t_step / 1e9 / (HBM_PEAK_GBPS * world),
        "roofline": {"bound": "hbm", "achieved": b_kernel / kernel_avg_s / 1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": b_kernel / kernel_avg_s / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "first-pass reduction kernel (streams Rs, Os, x once)",
                     "kernel_avg_us": kernel_avg_s * 1e6, "kernel_min_us": kernel_ms[0] * 1e3,
                     "kernel_max_us": kernel_ms[-1] * 1e3,
                     "estimator": "HIP events around each launch of the dominant kernel, on its stream, K launches",
                     "launch_period_us_in_timed_region": region_avg_s * 1e6 if args.steps >= 2 else None,
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
        rl = line["roofline"]
        if rows == 1 << 20 and d == 4 and dtype == torch.float64 and not args.levelwise and not sharded_mode:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
            rl["traffic"] = pmc["hbm_bytes_per_launch"]
            rl["traffic_source"] = "profiles/" + PMC_TRAFFIC_FILE
            rl["traffic_stale"] = pmc.get("kernel_source_sha16") != kernel_source_sha16()
            # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats
            # summary of this command
            import csv
            prof = "profiles/" + KERNEL_STATS_FILE
            for r in csv.DictReader(open(os.path.join(ROOT, prof))):
                if "chunk_reduce" in r["Name"] or "stream_reduce" in r["Name"]:
                    rl["kernel_avg_us_rocprofv3"] = float(r["AverageNs"]) / 1e3
                    rl["kernel_avg_us_rocprofv3_source"] = prof
                    break
        elif sharded_mode and d == 4 and dtype == torch.float64:
            # per-shard figure: the shard kernels of one rank, measured on one GPU with the sharded code path
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_SHARD_FILE)))
            ent = pmc.get("rows_per_rank", {}).get(str(rows))
            if ent is not None:
                rl["traffic"] = ent["hbm_bytes_per_step"]
                rl["traffic_source"] = "profiles/" + PMC_TRAFFIC_SHARD_FILE + " (one rank's shard kernels, measured on one GPU)"
                rl["traffic_stale"] = pmc.get("kernel_source_sha16") != kernel_source_sha16()
            else:
                rl["traffic_note"] = "no PMC pass committed for %d rows per rank (see %s)" % (rows, PMC_TRAFFIC_SHARD_FILE)
    except Exception as e:
        line["roofline"]["traffic_note"] = "no committed PMC figure: " + repr(e)[:120]
    if not sharded_mode and not args.no_extras:
        line["extras"] = extras_early if extras_early is not None else extra_measurements(dev)
        line["extras"]["order"] = ("cold W+K steps, secondary measurements, then the headline" if extras_early is not None
                                   else "headline first")
    if headline_cold is not None:
        line.setdefault("extras", {})["headline_cold"] = headline_cold
    if not sharded_mode:
        line.setdefault("extras", {})["prewarm_steps"] = prewarm
    if not args.no_cpu_baseline:
        if world > 1 or sharded_mode:
            # bounded sample (the full 2^24-row system is ~25 s per run on the host cores): the 2^20-row system of
            # the same generator -- the oracle's time is linear in the rows, its GB/s is what carries over
            threads = None
            if os.environ.get("OMP_NUM_THREADS") == "1" and "TORCHELASTIC_RUN_ID" in os.environ:
                threads = max(1, (os.cpu_count() or 2) // 2)     # torchrun pins ranks to one thread: undo it for the baseline
            line["cpu_baseline"] = cpu_baseline(ROWS_PER_GPU, d, dtype, sample_note="bounded sample, 1/%d of the rows of the "
                                                "timed system (the oracle is linear in the rows)" % max(1, n_total // ROWS_PER_GPU),
                                                threads=threads)
        else:
            line["cpu_baseline"] = cpu_baseline(rows, d, dtype)
    else:
        line["cpu_baseline"] = None
    _emit(line)


def _weak_point(args, dev, dtype, d, rank, world, out, barrier):
    """Weak-scaling point (information only, --weak-point): 2^20 rows per GPU, one system of world * 2^20 rows.
    No exception is swallowed around a collective: a rank that fails here ends, and the launcher ends the job."""
    import torch.distributed as dist
    from cyclic_gps import sharded
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


if __name__ == "__main__":
    main()
