"""profiles/rNN_pmc_traffic_shards.json: HBM bytes per step of ONE rank's shard kernels (cgps_shard_reduce +
cgps_finish_records through bench.py's sharded code path on one GPU), per shard size, from the FETCH_SIZE / WRITE_SIZE
passes tools/round3_profiles.sh leaves in OUTDIR/rows_<n>/.  Bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of
the micro-architecture guide), summed over every library kernel of a step.

    python tools/pmc_traffic_shards_json.py OUTDIR ROUND
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir, rnd = sys.argv[1], int(sys.argv[2])


def per_step(d, counter):
    path = os.path.join(d, counter, "p_counter_collection.csv")
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "cgps::" in r["Kernel_Name"]]
    tot, launches = 0.0, {}
    for r in rows:
        tot += float(r["Counter_Value"])
        launches[r["Kernel_Name"]] = launches.get(r["Kernel_Name"], 0) + 1
    dom = max(launches, key=lambda k: sum(float(r["Counter_Value"]) for r in rows if r["Kernel_Name"] == k))
    return tot / launches[dom], launches[dom], dom


doc = {"round": rnd,
       "command": "CGPS_BENCH_FORCE_SHARDED=1 CGPS_BENCH_PREWARM_STEPS=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv "
                  "-- python3 bench.py --rows <rows per rank> --steps 10 --warmup 2 --no-cpu-baseline ; the same with --pmc "
                  "WRITE_SIZE (separate passes); python tools/pmc_traffic_shards_json.py OUT %d" % rnd,
       "what": "one rank's kernels of a sharded step (shard reduction with the record stages inside + finish kernel), d = 4 fp64",
       "rows_per_rank": {}}
for d in sorted(p for p in glob.glob(os.path.join(out_dir, "rows_*")) if os.path.isdir(p)):
    n = int(os.path.basename(d).split("_")[1])
    f, steps, dom = per_step(d, "FETCH_SIZE")
    w, _, _ = per_step(d, "WRITE_SIZE")
    alg = ((2 * n - 1) * 16 + n * 4) * 8 + 16
    hbm = 2 * f * 1024 + w * 1024
    doc["rows_per_rank"][str(n)] = {"hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": hbm / alg,
                                    "FETCH_SIZE_KB_per_step": f, "WRITE_SIZE_KB_per_step": w, "steps_averaged": steps,
                                    "dominant_kernel": dom.split("(")[0].replace("void ", "")[:120]}
    print("rows per rank %d: %.2f MB per step = %.4f x algorithmic (%d steps)" % (n, hbm / 1e6, hbm / alg, steps))
import bench  # noqa: E402
doc["kernel_source_sha16"] = bench.kernel_source_sha16()
json.dump(doc, open(os.path.join(ROOT, "profiles", "r%02d_pmc_traffic_shards.json" % rnd), "w"), indent=1)
