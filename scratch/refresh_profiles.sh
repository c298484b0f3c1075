#!/bin/bash
# Round profile refresh on the GPU box (headline workload): kernel stats + launch order of the default command, the two-stream
# trace, the bench line (which runs its own counter passes).  usage: bash scratch/refresh_profiles.sh [r4]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; RND=${1:-r4}
mkdir -p $R/gpurun_out/profiles_new
cd /tmp; rm -rf $R/gpurun_out/prof_cur $R/gpurun_out/prof_2s
# single-stream: per-kernel durations add up to the step
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -o $RND -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-scenes 0 --single-stream --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-counters --no-voxel-density --no-prior-variants --no-launch-power > $R/gpurun_out/prof_cur.log 2>&1
# two streams (the headline mode): the overlap
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_2s -o $RND -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-scenes 0 --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-counters --no-voxel-density --no-prior-variants --no-launch-power > $R/gpurun_out/prof_2s.log 2>&1
# the same step on clouds at voxel-grid density (bench.py --cloud voxel2mm; value_voxel_density of the default run)
rm -rf $R/gpurun_out/prof_vox
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vox -o ${RND}_voxel2mm -- python3 $R/bench.py --cloud voxel2mm --steps 5 --warmup 2 --cpu-scenes 0 --single-stream --no-reference-order --no-native-arith --no-evidence --no-f16x2 --no-counters --no-prior-variants --no-launch-power > $R/gpurun_out/prof_vox.log 2>&1
cd $R
cp gpurun_out/prof_vox/${RND}_voxel2mm_kernel_stats.csv gpurun_out/profiles_new/${RND}_voxel2mm_kernel_stats.csv
python3 scratch/step_trace.py gpurun_out/prof_vox/${RND}_voxel2mm_kernel_trace.csv > gpurun_out/profiles_new/${RND}_voxel2mm_step_trace.txt
cp gpurun_out/prof_cur/${RND}_kernel_stats.csv gpurun_out/profiles_new/${RND}_kernel_stats.csv
python3 scratch/step_trace.py gpurun_out/prof_cur/${RND}_kernel_trace.csv > gpurun_out/profiles_new/${RND}_step_trace.txt
python3 scratch/two_stream_trace.py gpurun_out/prof_2s/${RND}_kernel_trace.csv > gpurun_out/profiles_new/${RND}_two_stream_trace.txt
[ -n "$SKIP_BENCH" ] || python3 bench.py > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err
tail -1 gpurun_out/bench_cur.json > gpurun_out/profiles_new/${RND}_bench_n1.json
tail -1 gpurun_out/bench_cur.json | cut -c1-300
tail -2 gpurun_out/profiles_new/${RND}_step_trace.txt; tail -1 gpurun_out/profiles_new/${RND}_two_stream_trace.txt
tail -3 gpurun_out/bench_cur.err
