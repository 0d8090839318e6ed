#!/bin/bash
O=gpurun_out/r2
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_q6.py tests/test_gpu_misc.py -m gpu -x -q -k "allgather or rccl" 2>&1 | tail -2
for i in 1 2; do
IPS_BENCH_GATHER=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra > $O/bench_gather.json 2> $O/bench_gather.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2/bench_gather.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["extra"]["exchange"]["scan_only_rows_per_s"])
PY
done
