#!/bin/bash
# tools/kregs.sh <file.hip> -- compile one kernel file for gfx950 keeping the assembly, print registers / spills / scratch per kernel
set -e
f=$(realpath "$1"); d=/tmp/kregs_$(basename "$f" .hip); mkdir -p "$d"
( cd "$d" && time hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c "$f" -o "$d/out.o" -save-temps=obj )
s=$(ls "$d"/*gfx950.s)
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" "$s" | paste - - - - - - | sed 's/\.private_segment_fixed_size/scratch/; s/_count//g; s/\s\+/ /g' | c++filt | cut -c1-220
echo "asm: $s"
