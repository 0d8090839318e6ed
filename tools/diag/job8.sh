set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
for v in "" _Q; do
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_hip$v.so python tools/kbench.py --bw 16,12,10 --what scan --sel 0.1,0.03 --reps 20 > $O/kbench_8$v.txt 2>&1
done
for v in "" _Q; do
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_hip$v.so python tools/kbench.py --bw 16,12 --what scan --sel 0.1 --reps 20 >> $O/kbench_8$v.txt 2>&1
done
IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_hip_Q.so timeout 300 python -m pytest tests/test_gpu_fle.py -m gpu -x -q > $O/gputests_8Q.log 2>&1
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_c -- python3 $R/tools/kbench.py --bw 32,16,12,8 --what scan --sel 0.1 --reps 3 > $O/pmc_c.log 2>&1
cd $R
f=$(find $O/pmc_c -name "*counter_collection.csv" | head -1); python tools/pmc_summary.py $f > $O/pmc_c.summary.txt 2>&1
find $O/pmc_c -name "*.csv" -size +1M -delete
echo done
