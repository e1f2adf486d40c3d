#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter); average per dispatch."""
import csv, glob, re, sys
from collections import defaultdict

def main(root, filt="forward"):
    acc = defaultdict(lambda: [0.0, set()])
    for f in sorted(glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
            if filt not in k:
                continue
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1].add((f, r["Dispatch_Id"]))
    print("kernel,counter,avg_per_dispatch,dispatches")
    for (k, c), (v, d) in sorted(acc.items()):
        print(f"{k[-40:]},{c},{v/len(d):.6g},{len(d)}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "forward")
