#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 --kernel-trace CSV with the warm-up taken out.

rocprofv3's own *_kernel_stats.csv averages EVERY launch of the process, the first (code-load) launch and the bench's
warm-up steps included, so its average sits above the driver-timed step (round-3 verdict: 4.449 ms against 4.361).
This reads the per-dispatch trace of the same run and prints, per kernel: calls, the all-launch average (= the stats
file's), and over the launches left after dropping the first `skip` of that kernel: average, median, min, max (ns).

    tools/warm_kernel_stats.py <dir-or-kernel_trace.csv> <skip> [> warm_kernel_stats.csv]
"""
import csv
import glob
import statistics
import sys
from collections import defaultdict


def main():
    src, skip = sys.argv[1], int(sys.argv[2])
    files = [src] if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {src}")
    runs = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            runs[row["Kernel_Name"]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "AllAverageNs", "SkippedFirst", "WarmCalls", "WarmAverageNs", "WarmMedianNs", "WarmMinNs", "WarmMaxNs"])
    for name, launches in sorted(runs.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        d_all = [d for _, d in sorted(launches)]
        warm = d_all[skip:] if len(d_all) > skip else d_all
        w.writerow([name, len(d_all), round(statistics.mean(d_all)), skip if len(d_all) > skip else 0,
                    len(warm), round(statistics.mean(warm)), round(statistics.median(warm)), min(warm), max(warm)])


if __name__ == "__main__":
    main()
