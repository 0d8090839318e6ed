#!/bin/bash
# dev tool: kernel trace of 12 nullable-leaf calls (w=12, 10 % NULL, 2^28 rows)
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab/leaf_prof -o leaf -- python3 $GRAFT_REPO_ROOT/tools/ab/leaf_only.py > $GRAFT_REPO_ROOT/gpurun_out/ab/leaf_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/ab/leaf_prof/**/*kernel_stats.csv', recursive=True):
    for row in list(csv.DictReader(open(f)))[:8]:
        print(row['Name'][:70], row['Calls'], row['AverageNs'], row['MinNs'])
PY
