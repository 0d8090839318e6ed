set -x
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_q6.py tests/test_gpu_misc.py tests/test_gpu_nullable.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/gputests_13.log 2>&1; echo "pytest rc=$?" >> $O/gputests_13.log
python tools/q6_bench.py > $O/q6_13.txt 2>&1
echo done
