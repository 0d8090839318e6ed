set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_4.log 2>&1; echo "pytest rc=$?" >> $O/gputests_4.log
python tools/kbench.py --bw 32,16,12,8,4 --what scan --sel 0.1 --reps 20 > $O/kbench_4.txt 2>&1
python tools/nullable_bench.py --bw 12 --nulls 0.1 > $O/nullable_4.txt 2>&1
timeout -k 10 600 python bench.py > $O/bench_4.json 2> $O/bench_4.err; echo "bench rc=$?" >> $O/bench_4.err
IPS_BENCH_GATHER=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 2 > $O/bench_4g.json 2> $O/bench_4g.err; echo "bench gather rc=$?" >> $O/bench_4g.err
echo done
