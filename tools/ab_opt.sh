#!/bin/bash
# A/B of one option of the headline handle inside one gpurun call (boxes differ by several % between calls):
#   tools/ab_opt.sh scan_verify 0 1 [bench.py args]        alternates the two values, three rounds
cd "$(dirname "$0")/.."
key=$1; a=$2; b=$3; shift 3
for i in 1 2 3; do
  for v in $a $b; do
    python bench.py --no-cpu --no-legs --no-dense --option $key=$v "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$key=$v', 'kernel_ms', r['kernel_ms'], 'walk_ms', r['walk_kernel_ms'], 'step_ms', d['ms_per_step'], 'value', d['value'], 'whole', r['frac_whole_step'], 'check', d.get('check_ok'))"
  done
done
