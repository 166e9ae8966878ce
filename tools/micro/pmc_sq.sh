#!/bin/bash
# SQ / LDS counters of every kernel a command launches (three separate --pmc passes of <= 8 SQ counters), per-launch means.
# Run through gpurun from the repo root: bash tools/micro/pmc_sq.sh <tag> <kernel name filter> <command ...>
export TMPDIR=/tmp
TAG=$1; FILTER=$2; shift 2
OUT=$PWD/gpurun_out/${TAG}_pmc; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"
P3="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P -d $OUT/p$i -o w --output-format csv -- "$@" > $OUT/p$i.out 2> $OUT/p$i.err)
  tail -2 $OUT/p$i.out
done
python tools/pmc_summarize.py $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3 2>/dev/null | grep -i "$FILTER" | tr ' ' '\n' | grep -v "^$"
