#!/bin/bash
# Experiment script (GPU): all PMC passes (SQ, L2, FETCH/WRITE, TA/TCP, GRBM), one rocprofv3 run each, for ONE bench command,
# summarised per kernel.  usage: tools/pmc_full.sh <name> "<kernel substring>[;<kernel substring>...]" [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
name=$1; kerns=$2; shift 2
OUT=gpurun_out/pmc_$name
mkdir -p $OUT
CMD="python bench.py --steps ${PMC_STEPS:-1} --warmup ${PMC_WARMUP:-0} --no-cpu-baseline --no-north-star --no-same-view $*"     # 1 step = one pass over the orbit: every view once
run() { n=$1; shift; timeout -s KILL 170 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $CMD > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/$n.log; exit 1; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run ta1 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
IFS=';' read -ra KS <<< "$kerns"
for k in "${KS[@]}"; do
  echo "==== $k" >> $OUT/summary.txt
  python profiles/pmc_summary.py $OUT "$k" >> $OUT/summary.txt
done
cat $OUT/summary.txt
