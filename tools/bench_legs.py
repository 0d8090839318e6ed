#!/usr/bin/env python3
"""Dev tool: one line per leg of a bench.py JSON line (file argument): time, fraction of 8 TB/s, check."""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(f"headline: {j['ms_per_step']*1e3:7.1f} us  frac {r['frac']:.3f}  read-only {r.get('frac_read_only', 0):.3f}")
for v in j.get("extra", {}).get("configs", []):
    print(f"{v.get('us_med', 0):8.1f} us  frac {v.get('frac', 0):.3f}  check {v.get('check')}  {v.get('vs_contiguous', '')}  {v['config'][:110]}")
