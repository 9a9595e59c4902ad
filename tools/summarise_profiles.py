"""Turn the rocprofv3 outputs of tools/rocprof_passes.sh (gpurun_out/prof/) into the tracked summaries under
profiles/rNN/ and refresh profiles/pmc_traffic.json, stamped with the kernel source it was measured on.

usage: python tools/summarise_profiles.py ROUND_DIR WORKLOAD_KEY [COMMIT]
   e.g. python tools/summarise_profiles.py profiles/r02 n1000000_m50000000_shards1_of_1 $(git rev-parse --short HEAD)
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "gpurun_out", "prof")


def main():
    out_dir, key = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else "uncommitted"
    os.makedirs(out_dir, exist_ok=True)
    # the kernel source the measurement belongs to: what bench.py printed while it ran under the profiler
    def kernel_source_id():
        with open(os.path.join(PROF, "trace.json")) as f:
            line = [l for l in f if l.startswith("{")][-1]
        return json.loads(line)["config"]["kernel_source_id"]
    stats = glob.glob(os.path.join(PROF, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, "bench_n1_kernel_stats.csv"))
    counters = {}
    for path in glob.glob(os.path.join(PROF, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "k_arcte_seeds" not in row["Kernel_Name"] and "k_arcte_lines" not in row["Kernel_Name"]:
                    continue
                c = counters.setdefault(row["Counter_Name"], {"launches": 0, "sum": 0.0, "kernel": row["Kernel_Name"],
                                                              "lds": row["LDS_Block_Size"], "vgpr": row["VGPR_Count"]})
                c["launches"] += 1
                c["sum"] += float(row["Counter_Value"])
    summary = {name: {"launches": c["launches"], "mean_per_launch": c["sum"] / max(c["launches"], 1)} for name, c in counters.items()}
    any_c = next(iter(counters.values()), None)
    meta = {"kernel": any_c["kernel"] if any_c else None, "LDS_Block_Size": any_c["lds"] if any_c else None,
            "VGPR_Count": any_c["vgpr"] if any_c else None, "kernel_source_id": kernel_source_id(), "commit": commit, "workload": key}
    with open(os.path.join(out_dir, "bench_n1_pmc_propagation_kernel.json"), "w") as f:
        json.dump({"meta": meta, "counters": summary}, f, indent=1, sort_keys=True)
    for name in ("trace.json",):
        src = os.path.join(PROF, name)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(out_dir, "bench_line_under_rocprof.json"))
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        fetch = summary["FETCH_SIZE"]["mean_per_launch"] * 1024
        write = summary["WRITE_SIZE"]["mean_per_launch"] * 1024
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        table = json.load(open(path)) if os.path.exists(path) else {}
        table[key] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "kernel": meta["kernel"],
                      "kernel_source_id": meta["kernel_source_id"], "commit": commit,
                      "source": os.path.relpath(os.path.join(out_dir, "bench_n1_pmc_propagation_kernel.json"), ROOT),
                      "note": "FETCH_SIZE/WRITE_SIZE in KiB from separate rocprofv3 --pmc passes, x1024 (random 8-byte state reads are "
                              "one 64-byte request each and are counted as they are: profiles/r01/calibration_fetch_write_size.txt); "
                              "bench.py adds the half of the coalesced row stream that FETCH_SIZE under-reports on gfx950 "
                              "(/opt/skills/guides/MI355X_MICROARCH.md) and says so in roofline.traffic_note"}
        with open(path, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
    print(json.dumps({"meta": meta, "counters": summary}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
