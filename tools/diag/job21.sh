#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/leaf_trace
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/diag/leaf_only.py > $O/run.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r2/leaf_trace/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-36:]
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    print(f'{r["Kernel_Name"][:60]:60s} dur {(e-s)/1000:8.1f} us  gap {gap/1000:6.1f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size",""))} wg {r.get("Workgroup_Size_X", "")} vgpr {r.get("VGPR_Count","")} lds {r.get("LDS_Block_Size","")}')
    prev_end = e
PY
find $O -name "*.csv" -size +2M -delete
