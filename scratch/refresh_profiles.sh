#!/bin/bash
# Round profile refresh on the GPU box: kernel stats of the bench command, PMC traffic passes, bench line.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; RND=${1:-r3}
cd /tmp; rm -rf $R/gpurun_out/prof_cur
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -o $RND -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-scenes 0 --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-two-streams > $R/gpurun_out/prof_cur.log 2>&1
cd $R
bash scratch/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
mkdir -p gpurun_out/profiles_new
cp gpurun_out/pmc_traffic.json profiles/${RND}_pmc_traffic.json 2>/dev/null
cp gpurun_out/prof_cur/${RND}_kernel_stats.csv profiles/${RND}_kernel_stats.csv 2>/dev/null
python3 bench.py --steps 30 > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err
tail -1 gpurun_out/bench_cur.json > profiles/${RND}_bench_n1.json
tail -1 gpurun_out/bench_cur.json | cut -c1-400
head -25 gpurun_out/prof_cur/${RND}_kernel_stats.csv | cut -c1-150
python3 scratch/step_trace.py gpurun_out/prof_cur/${RND}_kernel_trace.csv > profiles/${RND}_step_trace.txt
python3 scratch/profile_table.py $RND > /dev/null
# only gpurun_out/ travels back from the GPU box: leave copies there (copy them into profiles/ and commit)
cp profiles/${RND}_* gpurun_out/profiles_new/
