#!/bin/bash
# PMC passes of tools/exp_auto.py <kind> <level> [kernel]  ->  gpurun_out/<tag>/pmc_*; prints per-dispatch means of the chain kernels
TAG=$1; shift
R=$(pwd); mkdir -p $R/gpurun_out/$TAG; export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "WRITE_SIZE TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$TAG/pmc_$n -o pmc -- python $R/tools/exp_auto.py "$@" > $R/gpurun_out/$TAG/log_$n.txt 2>&1
done
cd $R/gpurun_out/$TAG
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("pmc_*/pmc_counter_collection.csv")):
    agg=collections.defaultdict(lambda:[0,0])
    for r in csv.DictReader(open(f)):
        for k in ("k_walk_chain","k_chain_emit"):
            if k in r["Kernel_Name"]:
                a=agg[(k,r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1]+=1
    for (k,c),(v,n) in sorted(agg.items()): print(k, c, round(v/n/1e6,3), "M per dispatch (", n, ")")
PY
