#!/bin/bash
# dev tool: the one-pass chain kernel -- parity tests, Q6 timing against the per-operand plan, SQ counters
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_gpu_misc.py -m gpu -x -q -k "fused_program or random_predicate_trees" > gpurun_out/ab/chain_tests.log 2>&1 || { tail -30 gpurun_out/ab/chain_tests.log; exit 1; }
tail -1 gpurun_out/ab/chain_tests.log
timeout -k 10 300 python tools/q6_bench.py 2>&1 | grep -v amdgpu.ids
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ab/chain_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/ab/chain_sq -- python3 $R/tools/q6_bench.py > $R/gpurun_out/ab/chain_sq.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/ab/chain_sq/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'chain' in k or 'fle_pred_w' in k:
            acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print('   ', c, len(v), sum(v) / len(v))
PY
