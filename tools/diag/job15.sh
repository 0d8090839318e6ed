set -x
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_15.log 2>&1; echo "pytest rc=$?" >> $O/gputests_15.log
echo done
