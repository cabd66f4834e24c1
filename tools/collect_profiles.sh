#!/bin/bash
# Everything profiles/ holds for one snapshot, collected on the GPU box: bash tools/collect_profiles.sh <tag> [soak pairs per config]
# Summaries go to gpurun_out/<tag>/ (copied to profiles/ by hand); the raw rocprofv3 directories stay in /tmp.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-snap}; SOAK=${2:-60}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests -q -m gpu -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; tail -1 $O/${T}_pytest_gpu.txt
rm -rf /tmp/pf /tmp/pw
# HBM traffic of EVERY kernel of the step (ORB, stereo and matcher kernels included), at the bench's own batch: one step of bench.py under each counter
BENCH1="python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 1 --warmup 0"
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pf -o run -- $BENCH1 > /tmp/pf.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pw -o run -- $BENCH1 > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw 6144 $O/${T}_pmc_hbm_traffic > $O/${T}_pmc_hbm_traffic_step6144.txt 2>&1
cp $O/${T}_pmc_hbm_traffic.json $R/profiles/      # bench.py reads roofline.traffic from the summary whose source hash matches the sources it runs
rm -rf /tmp/pb
# instruction budget at the bench's batch (3072 pairs: the kernel variants the step really launches -- VERDICT r4 weak 7a)
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d /tmp/pb -o run -- $BENCH1 > /tmp/pb.log 2>&1
python $R/tools/pmc_budget.py /tmp/pb 6144 $O/${T}_valu_budget.json > $O/${T}_valu_budget_per_kernel.txt 2>&1; cp $O/${T}_valu_budget.json $R/profiles/
timeout 900 python $R/bench.py 2>/dev/null | tail -1 > $O/${T}_bench_C3.json; cut -c1-160 $O/${T}_bench_C3.json
OLF_ONE_STREAM=1 timeout 600 python $R/bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${T}_bench_C3_one_stream.json
# the other configurations: the bench line AND the rocprofv3 kernel summary of the same command
for c in C2 C4 C5; do
  rm -rf /tmp/ks_$c; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -o run -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/${T}_bench_$c.json
  cp $(ls /tmp/ks_$c/*kernel_stats.csv | head -1) $O/${T}_bench_${c}_kernel_stats.csv
done
timeout 600 python $R/bench.py --sequence 6 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${T}_bench_C3_sequence6.json
timeout 600 python $R/bench.py --scene bars --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${T}_bench_C3_long_scene.json
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o run -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /tmp/ks.log 2>&1
cp $(ls /tmp/ks/*kernel_stats.csv | head -1) $O/${T}_bench_C3_kernel_stats.csv
python $R/tools/timeline.py /tmp/ks > $O/${T}_timeline_two_streams.txt 2>&1
rm -rf /tmp/ks1; OLF_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o run -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /tmp/ks1.log 2>&1
cp $(ls /tmp/ks1/*kernel_stats.csv | head -1) $O/${T}_bench_C3_one_stream_kernel_stats.csv
OLF_ONE_STREAM=1 timeout 600 python $R/tools/grow_sweep.py 1 8 128 512 1024 2048 3072 > $O/${T}_grow_sweep.txt 2>&1
timeout 900 python $R/tools/soak.py $SOAK > $O/${T}_soak.txt 2>&1; tail -2 $O/${T}_soak.txt
timeout 900 python $R/tools/fuzz.py 500 11 2>/dev/null | tail -8 > $O/${T}_fuzz_500configs.txt; tail -1 $O/${T}_fuzz_500configs.txt
timeout 600 python $R/tools/fuzz_match.py 300 5 2>/dev/null | tail -2 > $O/${T}_fuzz_match_300cases.txt
timeout 600 python $R/tools/pcie_rate.py 2048 8 > $O/${T}_pcie_rate.txt 2>&1; tail -2 $O/${T}_pcie_rate.txt
timeout 300 python $R/tools/pair_latency.py > $O/${T}_pair_latency.txt 2>&1
# the one-pair shape: latency by growth workgroups per image, the kernel / copy timeline of one call, the cross-CU hand-over under uneven load
OLF_AB_SETTINGS="1:512,2:512,4:512" timeout 300 python $R/tools/ab_groups.py 1 8 32 2>&1 | grep -v amdgpu.ids > $O/${T}_growth_groups.txt
timeout 300 bash $R/tools/pair_timeline_full.sh > $O/${T}_pair_timeline.txt 2>&1
timeout 300 python $R/tools/stress_mg.py 60 2>&1 | grep -v amdgpu.ids > $O/${T}_stress_uneven_load.txt
timeout 300 python $R/tools/search_latency.py 2>/dev/null | tail -1 > $O/${T}_search_latency.txt
timeout 300 python $R/tools/wide_latency.py 2>&1 | grep -v amdgpu.ids > $O/${T}_wide_latency.txt
ls -la $O
