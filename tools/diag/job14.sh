set -x
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_14.log 2>&1; echo "pytest rc=$?" >> $O/gputests_14.log
timeout -k 10 300 python tools/aux_bench.py > $O/aux_14.txt 2>&1
echo done
