#!/bin/bash
# dev tool: SQ counters of the one-pass select kernel
mkdir -p gpurun_out/ab
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ab/selnull_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/ab/selnull_sq -- python3 $R/tools/ab/selnull.py > $R/gpurun_out/ab/selnull_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/ab/selnull_sq2 -- python3 $R/tools/ab/selnull.py > $R/gpurun_out/ab/selnull_sq2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for pat in ('gpurun_out/ab/selnull_sq/**/*counter_collection.csv', 'gpurun_out/ab/selnull_sq2/**/*counter_collection.csv'):
    for f in glob.glob(pat, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'select_nullable' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for c, v in sorted(acc.items()):
            t = len(v) // 3
            print(c, len(v), [round(sum(v[i*t:(i+1)*t]) / max(t, 1)) for i in range(3)])
PY
