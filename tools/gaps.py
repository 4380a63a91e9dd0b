"""Kernel statistics and inter-kernel gaps from a rocprofv3 --kernel-trace --stats --output-format csv directory."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(stats)))[:18]:
    print("%-64s %7s %10.1f us %6s%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 50000:
        gap[(a["Kernel_Name"][:34], b["Kernel_Name"][:34])].append(g)
print("--- gaps (ns): predecessor -> successor, count, mean")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print("%-36s -> %-36s %6d %8.0f" % (k[0], k[1], len(v), sum(v) / len(v)))
