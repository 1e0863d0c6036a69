#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short, readable table (kernel names truncated)."""
import csv
import sys


def short(name):
    import re
    m = re.match(r"_ZN5qspec(\d+)", name)
    if m:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        nums = re.findall(r"L[ib](\d+)E", rest.split("Ev")[0]) if rest.startswith("I") else []
        return "qspec::" + base + ("<" + ",".join(nums) + ">" if nums else "")
    if name.startswith("qspec::"):
        return name.split("(")[0]
    return name[:60]


def main(path, out=sys.stdout):
    rows = list(csv.DictReader(open(path)))
    print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}", file=out)
    for r in rows:
        print(f"{short(r['Name']):62s} {int(r['Calls']):7d} {int(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:9.2f} {int(r['MinNs'])/1e3:8.2f} {int(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}", file=out)


if __name__ == "__main__":
    main(sys.argv[1])
