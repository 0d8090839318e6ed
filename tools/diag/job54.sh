#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/enc_trace
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/enc_bench.py > $O/run.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r2/enc_trace/**/*kernel_stats.csv"), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
grep dict_encode $O/run.log
find $O -name "*.csv" -size +2M -delete
