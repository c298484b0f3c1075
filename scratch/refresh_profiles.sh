#!/bin/bash
# Round profile refresh on the GPU box: kernel stats of the bench command, PMC traffic passes, bench line.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf $R/gpurun_out/prof_cur
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -o r1 -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-scenes 0 > $R/gpurun_out/prof_cur.log 2>&1
cd $R
bash scratch/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/r1_pmc_traffic.json 2>/dev/null
python3 bench.py > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err
tail -1 gpurun_out/bench_cur.json | cut -c1-400
head -25 gpurun_out/prof_cur/r1_kernel_stats.csv | cut -c1-150
python3 scratch/profile_table.py > /dev/null
