"""Summary of a rocprofv3 --kernel-trace CSV for runs with many LPs in flight (tools/batch_concurrency.py): how long the kernels of
each kind ran, how many ran at the same time, and what the hardware queues did between them.

    python3 tools/trace_concurrency.py DIR_OR_CSV [label]

Prints one JSON object: per kernel (by count and by time) the mean / median duration; the wall span; the time-weighted number of
kernels in flight; per queue the busy fraction and the gaps between one kernel's end and the next one's start.
"""
import csv
import json
import os
import sys
from collections import defaultdict


def find_csv(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for name in files:
            if name.endswith("kernel_trace.csv"):
                return os.path.join(root, name)
    raise SystemExit("no kernel_trace.csv under " + path)


def main():
    path = find_csv(sys.argv[1])
    rows = []
    with open(path) as handle:
        for row in csv.DictReader(handle):
            name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("relp::", "").replace("(anonymous namespace)::", "")
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name, row.get("Queue_Id", "0"), int(row.get("Workgroup_Size", 0) or 0),
                         int(row.get("Grid_Size", 0) or 0)))
    rows.sort()
    span = rows[-1][1] - rows[0][0]
    kinds = defaultdict(list)
    for start, end, name, _, _, _ in rows:
        kinds[name].append(end - start)
    # time-weighted concurrency
    events = []
    for start, end, *_ in rows:
        events.append((start, 1))
        events.append((end, -1))
    events.sort()
    level, last, weighted, histogram = 0, events[0][0], 0, defaultdict(int)
    for t, delta in events:
        weighted += level * (t - last)
        histogram[min(level, 64)] += t - last
        last = t
        level += delta
    queues = defaultdict(list)
    for start, end, name, queue, _, _ in rows:
        queues[queue].append((start, end))
    queue_summary = {}
    for queue, items in queues.items():
        items.sort()
        busy = sum(e - s for s, e in items)
        gaps = sorted(max(0, items[k + 1][0] - items[k][1]) for k in range(len(items) - 1))
        queue_summary[queue] = {"kernels": len(items), "busy_fraction": busy / span,
                                "gap_us_median": gaps[len(gaps) // 2] / 1e3 if gaps else None,
                                "gap_us_p90": gaps[int(0.9 * len(gaps))] / 1e3 if gaps else None,
                                "gap_us_mean": sum(gaps) / len(gaps) / 1e3 if gaps else None}

    def stats(values):
        values = sorted(values)
        return {"count": len(values), "mean_us": sum(values) / len(values) / 1e3, "median_us": values[len(values) // 2] / 1e3,
                "p90_us": values[int(0.9 * len(values))] / 1e3, "total_ms": sum(values) / 1e6}

    top = sorted(kinds.items(), key=lambda kv: -sum(kv[1]))[:8]
    print(json.dumps({"label": sys.argv[2] if len(sys.argv) > 2 else path, "kernels": len(rows), "span_ms": span / 1e6,
                      "sum_of_durations_ms": sum(e - s for s, e, *_ in rows) / 1e6, "mean_kernels_in_flight": weighted / span,
                      "time_share_by_kernels_in_flight": {str(k): round(v / span, 4) for k, v in sorted(histogram.items()) if v / span > 0.005},
                      "queues": len(queues), "per_queue": dict(sorted(queue_summary.items(), key=lambda kv: -kv[1]["kernels"])[:8]),
                      "by_kernel": {name: stats(values) for name, values in top}}, indent=1))


if __name__ == "__main__":
    main()
