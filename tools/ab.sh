#!/bin/bash
# A/B builds of libacmi inside one gpurun call (boxes differ by several % between calls):
#   LIBS="libacmi_base.so libacmi.so" tools/ab.sh [bench.py args]
cd "$(dirname "$0")/.."
LIBS=${LIBS:-"libacmi_base.so libacmi.so"}
for i in 1 2 3; do
  for lib in $LIBS; do
    ACMI_LIBRARY=$PWD/php_aho_corasick_amd/$lib python bench.py --no-cpu "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$lib', r['kernel'], 'kernel_ms', r['kernel_ms'], 'walk_ms', r['walk_kernel_ms'], 'step_ms', d['ms_per_step'], 'value', d['value'], 'check', d.get('check_ok'))"
  done
done
