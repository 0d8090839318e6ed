set -x
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_17.log 2>&1; echo "pytest rc=$?" >> $O/gputests_17.log
python tools/kbench.py --bw 32,16,12,8 --what scan --sel 0.1 --reps 20 > $O/kbench_17.txt 2>&1
python tools/kbench.py --bw 32,16,12,8 --what scan --sel 0.1 --reps 20 >> $O/kbench_17.txt 2>&1
python bench.py --no-extra --no-cpu > $O/bench_17.json 2>$O/bench_17.err
echo done
