#!/bin/bash
# A/B builds of libacmi on secondary legs of bench.py inside one gpurun call:
#   LIBS="libacmi.so libacmi_x.so" LEGS=keyword_set,keyword_words tools/ab_legs.sh
cd "$(dirname "$0")/.."
LIBS=${LIBS:-"libacmi_base.so libacmi.so"}
LEGS=${LEGS:-keyword_set,keyword_words}
for i in 1 2 3; do
  for lib in $LIBS; do
    ACMI_LIBRARY=$PWD/php_aho_corasick_amd/$lib python bench.py --no-cpu --no-dense --steps 2 --warmup 1 --legs $LEGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
for leg in '$LEGS'.split(','):
    c = d[leg]
    print('$lib', leg, 'scan_ms', c['scan_kernel_ms'], 'walk_ms', c['walk_kernel_ms'], 'call_ms', c['ms_per_call'], 'frac', c['frac_whole_call'], 'ok', c['check_ok'])"
  done
done
