#!/bin/bash
# Round 6's capture, one gpurun call: the default bench un-profiled, the same command under rocprofv3 (tools/prof.sh: kernel trace +
# PMC passes), the dense cells of config 3's set (kernel tables, one PMC pass of L2 requests), the cliff matrix, the LDS sweep.
#   bash tools/prof_r06.sh        -> gpurun_out/r06*/ ; then python tools/collect_profiles.py gpurun_out/r06_prof profiles/r06
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=$R/gpurun_out/r06
mkdir -p "$O"
python bench.py > "$O/bench_unprofiled.log" 2>&1
grep '^{"metric' "$O/bench_unprofiled.log" | tail -1 > "$O/bench_unprofiled.json"
cp gpurun_out/bench_legs.json "$O/bench_legs_unprofiled.json"
bash tools/prof.sh r06_prof > "$O/prof_sh.log" 2>&1
for d in 32 128 512; do
  bash tools/prof_cmd.sh r06_dens_$d python $R/tools/bench_density.py cfg3 $d 8192 kernel=2 > "$O/dens_${d}_kernels.txt" 2>&1
done
bash tools/prof_cmd.sh r06_densg_512 python $R/tools/bench_density.py cfg3 512 8192 kernel=2 mib=1024 calls=6 > "$O/dens_512_1GiB_kernels.txt" 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$R/gpurun_out/r06_prof/pmc_dens128_TCC" -o pmc -- python $R/tools/bench_density.py cfg3 128 8192 kernel=2 mib=1024 calls=5 > "$O/dens_128_1GiB_pmc.log" 2>&1 )
for i in 1 2; do python tests/test_gpu_cliffs.py > "$O/cliffs_$i.txt" 2>&1; done
python tools/sweep_lds.py cfg3 > "$O/sweep_lds_cfg3.txt" 2>&1
python tools/exp_win.py > "$O/exp_win.txt" 2>&1
ls "$O"
