"""Per-kernel HBM traffic from the counter-collection CSVs of `rocprofv3 --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` passes (tools/pmc_traffic.sh, tools/pmc_traffic_op.sh).

    python tools/pmc_summary.py DIR [DIR ...]

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 a wide coalesced read is counted at half its bytes
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.
Prints, per (kernel, grid), the average over the launches found."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name.replace("cgps::", "")


def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "cgps" not in r["Kernel_Name"]:
                    continue
                acc[(r["Counter_Name"], short(r["Kernel_Name"]), r["Grid_Size"])].append(float(r["Counter_Value"]))
            print("==", os.path.relpath(f, d))
            for (cn, k, g), v in sorted(acc.items()):
                kb = sum(v) / len(v)
                mb = kb * 1024 / 1e6 * (2.0 if cn == "FETCH_SIZE" else 1.0)
                print("  %-5s %-62s grid=%-8s launches=%-3d %10.2f MB%s" % (
                    "read" if cn == "FETCH_SIZE" else "write", k[:62], g, len(v), mb, "  (2 x FETCH_SIZE)" if cn == "FETCH_SIZE" else ""))


if __name__ == "__main__":
    main()
