#!/bin/bash
# PMC passes over the attention dev harness (run on the GPU box from the repo root): tools/attn_dev/pmc.sh <binary> <outdir> [S] [B]
BIN=$1; OUT=$2; S=${3:-4608}; B=${4:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $R/$OUT/p$i -- $R/$BIN $S $B 3 > $R/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $R/$OUT > $R/$OUT/summary.json
cat $R/$OUT/summary.json
