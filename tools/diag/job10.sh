set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_10.log 2>&1; echo "pytest rc=$?" >> $O/gputests_10.log
timeout -k 10 300 python tools/nullable_bench.py --bw 12,8 --nulls 0.1,0.5 > $O/nullable_10.txt 2>&1
IPS_EXPAND_TWO_PASS=1 timeout -k 10 300 python tools/nullable_bench.py --bw 12 --nulls 0.1 > $O/nullable_10_2pass.txt 2>&1
timeout -k 10 300 python tools/aux_bench.py > $O/aux_10.txt 2>&1
IPS_EXPAND_TWO_PASS=1 timeout -k 10 200 python -m pytest tests/test_gpu_nullable.py -m gpu -x -q > $O/gputests_10_2pass.log 2>&1
echo done
