"""Per-kernel averages of the counters that tools/pmc_mem.sh collected:  python tools/pmc_mem_summary.py DIR [DIR ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "cgps" not in r["Kernel_Name"]:
                continue
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("cgps::", "")
            acc[(k, r["Grid_Size"], r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("##", d)
    for (k, g, wg), cs in sorted(acc.items()):
        print("%s  grid=%s wg=%s" % (k[:90], g, wg))
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        for c in sorted(avg):
            print("    %-40s %.4g" % (c, avg[c]))
        if avg.get("TCP_TCC_READ_REQ"):
            if "TCP_TCC_READ_REQ_LATENCY" in avg:
                print("    -> average L1->L2 read latency %.0f cycles" % (avg["TCP_TCC_READ_REQ_LATENCY"] / avg["TCP_TCC_READ_REQ"]))
            print("    -> L1->L2 read requests x 128 B = %.1f MB" % (avg["TCP_TCC_READ_REQ"] * 128 / 1e6))
        if "TCC_EA0_RDREQ" in avg:
            print("    -> L2->fabric read requests x 64 B = %.1f MB (x 128 B = %.1f MB)" % (avg["TCC_EA0_RDREQ"] * 64 / 1e6, avg["TCC_EA0_RDREQ"] * 128 / 1e6))
        if "FETCH_SIZE" in avg:
            print("    -> HBM read 2 x FETCH_SIZE = %.1f MB" % (2 * avg["FETCH_SIZE"] * 1024 / 1e6))
