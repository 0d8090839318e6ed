#!/bin/bash
# dev tool: kernels of ips_dict_select_nullable under rocprofv3
mkdir -p gpurun_out/ab
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ab/selnull_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab/selnull_prof -o sn -- python3 $R/tools/ab/selnull.py > $R/gpurun_out/ab/selnull_prof.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/ab/selnull_prof/**/*kernel_trace.csv', recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'ips::' in n and 'synth' not in n and 'encode' not in n:
            d[n[:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in d.items():
        # three selectivities x 11 calls each, in order
        t = len(v) // 3
        print(k, len(v), [round(sum(v[i*t:(i+1)*t]) / max(t, 1), 1) for i in range(3)])
PY
