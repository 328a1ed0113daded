#!/bin/bash
# Experiment script (GPU): the SQ / GRBM counter passes of profiles/pmc_passes.sh for ONE bench command, summarised.
# usage: tools/pmc_quick.sh <name> <kernel substring> [bench args...]   (RT_TUNE_* from the environment)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; kern=$2; shift 2
OUT=gpurun_out/pmc_$name
mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipelined --no-north-star $*"
run() { n=$1; shift; timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $CMD > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/$n.log; exit 1; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python profiles/pmc_summary.py $OUT "$kern" > $OUT/summary.txt
cat $OUT/summary.txt
