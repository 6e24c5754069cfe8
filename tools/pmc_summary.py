"""Summarises rocprofv3 --pmc output (counter_collection CSVs) per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py <dir with *counter_collection.csv> [kernel-name regex] [--per-wave]

Kernel names are shortened to the function name + template arguments.  Dev helper for profiles/.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(dvmvs::CostVolumeArgs\)", "", name)
    name = name.replace("void dvmvs::", "").replace("dvmvs::", "")
    return name[:110]


def main():
    root = sys.argv[1]
    rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name") or row.get("Kernel Name")
                if rx and not rx.search(k):
                    continue
                acc[short(k)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        c = acc[k]
        waves = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"]) if "SQ_WAVES" in c else None
        for name in sorted(c):
            mean = sum(c[name]) / len(c[name])
            extra = f"   {mean / waves:10.1f} / wave" if waves and name.startswith("SQ_") and name != "SQ_WAVES" else ""
            print(f"    {name:28s} {mean:14.4e}  (n={len(c[name])}){extra}")


if __name__ == "__main__":
    main()
