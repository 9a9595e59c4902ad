"""Readable per-kernel totals of a rocprofv3 --kernel-trace --stats run (library kernels get short names).
usage: python tools/kernel_stats_short.py DIR [TOP]"""
import csv
import glob
import re
import sys


def short(n):
    m = re.search(r"(k_[a-z0-9_]+)", n)
    if m and "rocprim" not in n:
        return m.group(1)
    t = re.search(r"onesweep_config<[^,]+, ([a-z ]+), ([a-z ]+)>", n)
    if "onesweep_iteration" in n:
        return "rocPRIM radix pass <%s, %s>" % (t.group(1), t.group(2))
    if "onesweep_global_offsets" in n:
        return "rocPRIM radix histogram <%s, %s>" % (t.group(1), t.group(2))
    if "scan_impl" in n or "lookback_scan" in n:
        return "rocPRIM scan"
    return n[:60]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    agg = {}
    for r in csv.DictReader(open(f)):
        a = agg.setdefault(short(r["Name"]), [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"]) / 1e6
    total = sum(t for _, t in agg.values())
    for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print("%-52s calls %5d  total %9.1f ms  avg %9.3f ms  %5.1f %%" % (k, c, t, t / c, 100 * t / total))


if __name__ == "__main__":
    main()
