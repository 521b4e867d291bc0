"""stdin: a BENCH_DETAIL line of bench.py (or gpurun_out/bench_legs.json's content); prints the named legs, a cell per line.
    python bench.py --legs density_sweep ... | grep BENCH_DETAIL | python tools/print_leg.py density_sweep"""
import json, sys
line = sys.stdin.readline()
d = json.loads(line.split(" ", 1)[1] if line.startswith("BENCH_DETAIL") else line)
for name in sys.argv[1:]:
    leg = d[name]
    for c in leg.get("cells", []):
        print({k: v for k, v in c.items() if k not in ("algorithmic_bytes",)})
    print(name, {k: v for k, v in leg.items() if k != "cells"})
