#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/leaf_sq
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/diag/leaf_only.py > $O/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/diag/leaf_only.py > $O/b.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, os, collections
for sub in ("a", "b"):
    fs = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], f"gpurun_out/r2/leaf_sq/{sub}/**/*counter_collection.csv"), recursive=True)
    if not fs:
        print("no counter file for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "rank" in k or "expand" in k or "fle_pred" in k:
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:24s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
find $O -name "*.csv" -size +2M -delete
