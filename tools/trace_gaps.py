"""Dev tool: idle time between consecutive kernels in a rocprofv3 kernel_trace.csv (stream-ordered workloads).
Only the last `tail` kernels are analysed (steady state)."""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows = rows[-tail:]
gap_after = defaultdict(list)
busy = 0
for a, b in zip(rows, rows[1:]):
    busy += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    name = a["Kernel_Name"]
    key = name.split("(")[0].split("::")[-1][:40] if "mcmc" in name else name[:60]
    gap_after[key].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %)")
for k, g in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:14]:
    g.sort()
    print(f"{k:60s} n={len(g):6d}  gap after: p10 {g[len(g) // 10] / 1e3:6.2f}  median {g[len(g) // 2] / 1e3:6.2f}  p90 {g[len(g) * 9 // 10] / 1e3:6.2f}  mean {sum(g) / len(g) / 1e3:6.2f} us  total {sum(g) / 1e6:7.2f} ms")
