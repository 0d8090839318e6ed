set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_nullable.py tests/test_gpu_misc.py -m gpu -x -q > $O/gputests_3.log 2>&1; echo "pytest rc=$?" >> $O/gputests_3.log
python tools/nullable_bench.py --bw 12,8,16 --nulls 0.1,0.5 > $O/nullable_bench.txt 2>&1
python tools/aux_bench.py > $O/aux_3.txt 2>&1
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/tools/kbench.py --bw 32,16,12,8 --what scan --sel 0.1 --reps 3 > $O/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/tools/kbench.py --bw 32,16,12,8 --what scan --sel 0.1 --reps 3 > $O/pmc_b.log 2>&1
cd $R
for d in pmc_a pmc_b; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python tools/pmc_summary.py $f > $O/$d.summary.txt 2>&1; done
find $O/pmc_a $O/pmc_b -name "*.csv" -size +1M -delete
echo done
