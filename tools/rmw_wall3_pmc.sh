#!/bin/bash
# Memory-side requests of the random-access micro-benchmark by mode (does a write-only update fetch the line it
# allocates?).  One rocprofv3 --pmc pass per mode, nothing else traced.  Writes gpurun_out/rmw_wall3_pmc.txt.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rmw_wall3_pmc
rm -rf $OUT && mkdir -p $OUT
: > $R/gpurun_out/rmw_wall3_pmc.txt
for em in "32 0" "32 1" "32 3" "64 0" "64 1" "64 3" "16 1"; do
  name=$(echo $em | tr ' ' '_')
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/$name -- $R/tools/rmw_wall2 $em 32768 6 1 0 > $OUT/$name.txt 2> $OUT/$name.err || exit 1
  cat $OUT/$name.txt >> $R/gpurun_out/rmw_wall3_pmc.txt
  python3 - $OUT/$name >> $R/gpurun_out/rmw_wall3_pmc.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k<" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
accesses = 1536 * 64 * 256
for k, v in sorted(acc.items()):
    print("    %-22s per launch %.4g  per access %.3f  (%d launches)" % (k, sum(v) / len(v), sum(v) / len(v) / accesses, len(v)))
PY
done
cat $R/gpurun_out/rmw_wall3_pmc.txt
