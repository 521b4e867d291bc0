#!/bin/bash
# Sample clocks and power while the scan kernel loops (analysis only).
cd "$(dirname "$0")/.."
for args in "--no-plant" "--no-plant --kernel 1" "" ; do
  python bench.py --no-cpu --steps 8000 --warmup 5 $args > /tmp/b.json 2>/dev/null &
  pid=$!
  : > /tmp/smi.log
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|mclk|Power \(W\)" | sed -E 's/.*(sclk|fclk|mclk)[^(]*\(([0-9]+)Mhz\).*/\1=\2/; s/.*Power \(W\): ([0-9.]+).*/W=\1/' | tr '\n' ' ' >> /tmp/smi.log
    echo >> /tmp/smi.log
    sleep 0.3
  done
  sort -t= -k5 -n /tmp/smi.log | awk -F'W=' '{print $2, $0}' | sort -n | tail -4 | cut -d' ' -f2-
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('args [$args]', d['roofline']['kernel'], d['roofline']['kernel_ms'], d['ms_per_step'])"
done
