"""profiles/rNN_pmc_traffic.json from the two counter collections tools/pmc_traffic.sh leaves in OUTDIR:
HBM bytes per launch of the dominant headline kernel = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the
micro-architecture guide), averaged over its launches, with the sha of the kernel sources it was measured on
(bench.py reports `traffic_stale` when they differ).  Also writes the library-kernel rows of the FETCH pass
as rNN_pmc_fetch_counter_collection.csv.

    python tools/pmc_traffic_json.py gpurun_out/r02p/traffic_headline 2
"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, rnd = sys.argv[1], int(sys.argv[2])
CSRC = os.path.join(ROOT, "cyclic-gps_amd", "csrc")


def rows(counter):
    path = os.path.join(out_dir, counter, "p_counter_collection.csv")
    return [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "cgps::" in r["Kernel_Name"]]


fetch, write = rows("FETCH_SIZE"), rows("WRITE_SIZE")
# the dominant kernel: the library kernel with the largest total FETCH_SIZE
tot = {}
for r in fetch:
    tot[r["Kernel_Name"]] = tot.get(r["Kernel_Name"], 0.0) + float(r["Counter_Value"])
kernel = max(tot, key=tot.get)
f = [float(r["Counter_Value"]) for r in fetch if r["Kernel_Name"] == kernel]
w = [float(r["Counter_Value"]) for r in write if r["Kernel_Name"] == kernel]
f_avg, w_avg = sum(f) / len(f), sum(w) / len(w)
N, d, s = 1 << 20, 4, 8
alg = ((2 * N - 1) * d * d + N * d) * s + 2 * s
hbm = 2 * f_avg * 1024 + w_avg * 1024


def kernel_source_sha16():
    """The digest bench.py computes (bench.kernel_source_sha16): every header under csrc/, cgps_mahal.hip, include/cgps.h."""
    names = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["cgps_mahal.hip"]
    h = hashlib.sha256()
    for f in names:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "cgps.h"), "rb").read())
    return h.hexdigest()[:16]


doc = {
    "round": rnd,
    "command": "bash tools/pmc_traffic.sh OUT  (= rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 "
               "bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras ; same with --pmc WRITE_SIZE: separate passes); "
               "python tools/pmc_traffic_json.py OUT %d" % rnd,
    "workload": "N=2^20 d=4 fp64 mahal_and_det, ONE launch (stage 1 + record stages inside the launch)",
    "kernel": kernel.split("(")[0].replace("void ", ""),
    "FETCH_SIZE_KB_avg": f_avg,
    "WRITE_SIZE_KB_avg": w_avg,
    "launches_averaged": len(f),
    "calibration": {"note": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md HBM section; round-1 "
                            "calibration on a 134217728-byte coalesced read gave counter/true = 0.50004) -> read bytes = "
                            "2 x FETCH_SIZE; WRITE_SIZE is exact"},
    "hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": hbm / alg,
    "kernel_source_sha16": kernel_source_sha16(),
}
dst = os.path.join(ROOT, "profiles", "r%02d_pmc_traffic.json" % rnd)
json.dump(doc, open(dst, "w"), indent=1)
with open(os.path.join(ROOT, "profiles", "r%02d_pmc_fetch_counter_collection.csv" % rnd), "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "VGPR_Count", "Counter_Name", "Counter_Value"])
    for r in fetch:
        wr.writerow([r["Dispatch_Id"], r["Grid_Size"], r["Kernel_Name"][:160], r["Workgroup_Size"], r["VGPR_Count"],
                     r["Counter_Name"], r["Counter_Value"]])
print("%s: %.2f MB per launch = %.4f x algorithmic (%d launches)" % (dst, hbm / 1e6, hbm / alg, len(f)))
