#!/bin/bash
# where a sweep's wave cycles go: SQ counters per launch of a 32-column (SWEEP_K) sweep on the C3 factor, in passes of 8
root=$(pwd); out=$root/gpurun_out/pmc_sq_${TAG:-k32}
rm -rf $out; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $root/tools/pmc_sweep.py > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 $root/tools/pmc_levels.py $out > $root/gpurun_out/pmc_sq_${TAG:-k32}.txt 2>&1
find $out -name "*.csv" -size +1M -delete
cat $root/gpurun_out/pmc_sq_${TAG:-k32}.txt
