#!/bin/bash
# Instruction-count PMC passes over the region-growing kernel for several wave counts: bash tools/mw_pmc.sh <images> "<nw list>"
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-64}; NWS=${2:-"0 1 4"}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/mw_pmc; rm -rf $O; mkdir -p $O
for nw in $NWS; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_WAVES"; do
    i=$((i+1))
    OLF_LSD_NW=$nw rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $O/nw$nw/g$i -o run -- python $R/tools/prof_lines.py $N > $O/nw$nw.g$i.log 2>&1 || echo "nw $nw group $i failed"
  done
  echo "== waves per image $nw" >> $O/summary.txt
  python $R/tools/pmc_sum.py $O/nw$nw k_lsd_grow >> $O/summary.txt 2>&1
  tail -3 $O/nw$nw.g1.log >> $O/summary.txt; du -sh $O/nw$nw >> $O/summary.txt; rm -rf $O/nw$nw
done
cat $O/summary.txt
