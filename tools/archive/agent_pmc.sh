#!/bin/bash
# PMC passes over k_lsd_grow (one counter group per run): bash tools/agent_pmc.sh <images>
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-6144}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/agent_pmc; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_ITEMS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $O/g$i -o run -- python $R/tools/prof_lines.py $N > $O/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python $R/tools/pmc_sum.py $O k_lsd_grow > $O/summary.txt 2>&1; rm -rf $O/g[0-9]
cat $O/summary.txt
