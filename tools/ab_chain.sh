#!/bin/bash
# A/B of libacmi builds on the chain walk's cells (kernel 6 forced, 1 GiB) inside one gpurun call:
#   LIBS="libacmi_base.so libacmi.so" tools/ab_chain.sh [option=value ...]
cd "$(dirname "$0")/.."
LIBS=${LIBS:-"libacmi_base.so libacmi.so"}
for cell in "headline 64" "headline 256" "words 64" "words 256"; do
  for lib in $LIBS; do
    echo "== $lib $cell $*"
    ACMI_LIBRARY=$PWD/php_aho_corasick_amd/$lib python tools/bench_density.py $cell 8192 mib=1024 calls=3 kernel=6 "$@" 2>&1 | tail -3
  done
done
