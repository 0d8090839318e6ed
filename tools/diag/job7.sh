set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_7.log 2>&1; echo "pytest rc=$?" >> $O/gputests_7.log
rm -f $O/bench_7g.rc
for ch in 8 1; do
IPS_BENCH_CHUNKS=$ch IPS_BENCH_GATHER=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 2 --no-extra > $O/bench_7g_$ch.json 2> $O/bench_7g_$ch.err; echo "bench gather $ch rc=$?" >> $O/bench_7g.rc
done
python tools/kbench.py --bw 32,24,16 --what scan,select --sel 0.1 --reps 20 > $O/kbench_7.txt 2>&1
python tools/kbench.py --bw 32 --what scan --sel 0.01,0.03,0.2,0.3 --reps 10 >> $O/kbench_7.txt 2>&1
python bench.py --no-extra --no-cpu > $O/bench_7.json 2> $O/bench_7.err
echo done
