#!/bin/bash
# SQ / cache counter passes over one forward-only run (gpurun): bash tools/pmc_pass.sh <outdir>
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --inflight 1 --steps 3 --warmup 2 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum MeanOccupancyPerCU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > /dev/null 2> $OUT/p$i.err; echo "pass $i rc=$?"
done
find $OUT -name "*kernel_trace.csv" -delete
python tools/pmc_sq.py $OUT > $OUT/summary.txt 2>&1; wc -l $OUT/summary.txt
