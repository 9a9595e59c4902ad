"""Table of the per-experiment rocprofv3 evidence collected by tools/rocprof_experiments.sh."""
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    rows = []
    for arm in sorted(d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d))):
        rec = {"arm": arm}
        for path in glob.glob(os.path.join(out, arm, "trace", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if "k_arcte_seeds" in r["Name"] or "k_arcte_lines" in r["Name"]:
                    rec["kernel"] = r["Name"].replace("void (anonymous namespace)::", "").split("((anonymous")[0]
                    rec["ms_per_launch"] = float(r["AverageNs"]) / 1e6
                    rec["launches"] = int(r["Calls"])
        for path in glob.glob(os.path.join(out, arm, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for r in csv.DictReader(open(path)):
                if "k_arcte_seeds" in r["Kernel_Name"] or "k_arcte_lines" in r["Kernel_Name"]:
                    a = acc.setdefault(r["Counter_Name"], [0.0, 0])
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
                    rec["LDS_Block_Size"] = r["LDS_Block_Size"]
                    rec["VGPR_Count"] = r["VGPR_Count"]
            for k, (v, c) in acc.items():
                rec[k] = v / max(c, 1)
        try:
            line = [l for l in open(os.path.join(out, arm + ".bench.json")) if l.startswith("{")][-1]
            b = json.loads(line)
            rec["edges_per_launch"] = b["config"]["per_seed"]["edges"] * b["config"]["seeds_per_step"]
            rec["algorithmic_bytes"] = b["roofline"]["algorithmic_bytes_per_launch"]
            rec["hot_values_per_wave"] = b["config"]["hot_values_per_wave"]
            rec["narrow_rows"] = b["config"].get("narrow_rows")
            rec["warm_end_rank"] = b["config"].get("warm_end_rank")
            rec["slots"] = b["config"]["slots_per_gpu"]
        except Exception as e:
            rec["bench_error"] = str(e)
        rows.append(rec)
    print(json.dumps(rows, indent=1))
    print()
    print("| arm | kernel | slots | LDS values/wave | warm end rank | ms/launch | alg GB/s | frac of 8 TB/s | EA rd req/edge | EA wr req/edge | TCC hit rate | FETCH+WRITE GB | wait share |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        e = r.get("edges_per_launch", 0) or 1
        ms = r.get("ms_per_launch", float("nan"))
        alg = r.get("algorithmic_bytes", 0)
        hit = r.get("TCC_HIT_sum", 0.0)
        miss = r.get("TCC_MISS_sum", 0.0)
        traffic = (r.get("FETCH_SIZE", 0.0) + r.get("WRITE_SIZE", 0.0)) * 1024 / 1e9
        wait = r.get("SQ_WAIT_ANY", 0.0) / max(r.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        print("| %s | %s | %s | %s | %s | %.1f | %.0f | %.3f | %.2f | %.2f | %.3f | %.0f | %.2f |" % (
            r["arm"], r.get("kernel", "?"), r.get("slots", "?"), r.get("hot_values_per_wave", "?"), r.get("warm_end_rank", "?"), ms, alg / ms / 1e6 if ms == ms else 0,
            alg / ms / 1e6 / 8000 if ms == ms else 0, r.get("TCC_EA0_RDREQ_sum", 0) / e, r.get("TCC_EA0_WRREQ_sum", 0) / e,
            hit / max(hit + miss, 1.0), traffic, wait))


if __name__ == "__main__":
    main()
