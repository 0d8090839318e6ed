# dev: SQ counters of the Q6 plan's kernels (per-operand plan and one-pass chain), summed per kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/chain_pmc
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/chain_pmc -- python3 $R/tools/q6_bench.py > $R/gpurun_out/c4_pmc.log 2>&1
cd $R
python3 tools/pmc_sum.py gpurun_out/chain_pmc "chain" > gpurun_out/c4_sum.log 2>&1
find gpurun_out/chain_pmc -name "*.csv" -size +1M -delete
cat gpurun_out/c4_sum.log
