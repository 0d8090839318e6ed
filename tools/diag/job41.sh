#!/bin/bash
O=gpurun_out/r2
mkdir -p $O
python bench.py > $O/bench_full.json 2> $O/bench_full.err
echo "rc=$?"
tail -c 400 $O/bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2/bench_full.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["frac_read_only"], d["ms_per_step"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for c in d["extra"]["configs"]:
    print(str(c.get("config", "?"))[:100], c.get("us_med"), c.get("frac"), c.get("check"))
print(d["extra"]["h2d"]); print(d["extra"]["latency"]); print(d["extra"]["separate_pages"])
PY
