#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (kernel names trimmed)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)  # drop the argument list
    return name[-90:]


def main(path: str, top: int = 12):
    rows = list(csv.DictReader(open(path)))
    print(f"# source: {path}")
    print("kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent")
    for r in rows[:top]:
        print(",".join([short(r["Name"]), r["Calls"], f"{int(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e6:.4f}",
                        f"{int(r['MinNs'])/1e6:.4f}", f"{int(r['MaxNs'])/1e6:.4f}", r["Percentage"]]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
