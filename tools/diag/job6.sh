set -x
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_6.log 2>&1; echo "pytest rc=$?" >> $O/gputests_6.log
rm -f $O/bench_6g.rc
for cfg in "8 2" "8 1" "1 1"; do set -- $cfg
IPS_BENCH_CHUNKS=$1 IPS_BENCH_SCAN_STREAMS=$2 IPS_BENCH_GATHER=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 2 --no-extra > $O/bench_6g_$1_$2.json 2> $O/bench_6g_$1_$2.err; echo "bench gather $1 $2 rc=$?" >> $O/bench_6g.rc
done
python tools/kbench.py --bw 32,16,12,8,4 --what scan,pred --sel 0.1 --reps 20 > $O/kbench_6.txt 2>&1
python tools/kbench.py --rows 33554432 --bw 32 --what scan --sel 0.1 --reps 20 > $O/kbench_6_small.txt 2>&1
python tools/kbench.py --rows 1048576 --bw 32 --what scan --sel 0.1 --reps 20 >> $O/kbench_6_small.txt 2>&1
echo done
