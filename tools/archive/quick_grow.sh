#!/bin/bash
# quick A/B of a growth-kernel change on the GPU box: parity tests, one-stream growth time at 3072 pairs, HBM traffic per image
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/quick; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python -m pytest $R/tests/test_line_gpu.py $R/tests/test_lsd_grow_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
OLF_ONE_STREAM=1 OLF_SWEEP_NW=${NWLIST:-0} python $R/tools/grow_sweep.py ${SIZES:-3072} 2>&1 | tail -3
rm -rf /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pf -o run -- python $R/tools/prof_lines.py 4096 > /tmp/pf.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pw -o run -- python $R/tools/prof_lines.py 4096 > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw 4096 $O/quick_pmc_hbm_traffic 2>&1 | grep -i "grow\|rect\|total" | head
python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-400
