set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_9.log 2>&1; echo "pytest rc=$?" >> $O/gputests_9.log
python tools/kbench.py --bw 32,24,16,12,8,4 --what scan,pred --sel 0.1 --reps 20 > $O/kbench_9.txt 2>&1
timeout -k 10 600 python bench.py > $O/bench_9.json 2> $O/bench_9.err; echo "bench rc=$?" >> $O/bench_9.err
echo done
