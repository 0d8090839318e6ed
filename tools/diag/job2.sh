set -x
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2/gputests_2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/gputests_2.log
for v in "" _B _C _D; do
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_hip$v.so python tools/kbench.py --bw 32,16,12,8,4 --what scan --sel 0.1 --reps 20 > gpurun_out/r2/kbench_v$v.txt 2>&1
done
python tools/kbench.py --bw 32,16,8 --what scan --sel 0.01,0.03,0.3,1.0 --reps 10 > gpurun_out/r2/kbench_sel.txt 2>&1
IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_hip_B.so python tools/kbench.py --bw 32,16,8 --what scan --sel 0.01,0.03,0.3,1.0 --reps 10 > gpurun_out/r2/kbench_sel_B.txt 2>&1
echo done
