#!/bin/bash
O=gpurun_out/r2
mkdir -p $O
IPS_BENCH_GATHER=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench_gather.json 2> $O/bench_gather.err
echo "rc=$?"
tail -c 300 $O/bench_gather.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2/bench_gather.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["parallelism"][:80])
print(d["extra"]["exchange"])
print(d["extra"]["q6_sharded"])
PY
