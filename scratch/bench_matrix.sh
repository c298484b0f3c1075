#!/bin/bash
# every bench.py mode once, small: catches report / workload regressions (rc and `ok` per mode)
S="--steps 2 --warmup 1 --scenes-per-gpu 4 --cpu-scenes 0 --no-counters"
i=0
for F in "" "--single-stream" "--mlp-arith native --no-f16x2" "--mlp-arith split16" "--materialize-tuples" "--eager-scale-head" "--separate-encode" "--cloud voxel2mm" "--workload ensemble" "--workload ensemble --single-stream" "--workload dense64k" "--breakdown --no-voxel-density --no-evidence" "--cpu-scenes 1 --no-voxel-density" "--workload ensemble --cpu-scenes 1"; do
  i=$((i+1))
  python bench.py $S $F > gpurun_out/bm_$i.json 2> gpurun_out/bm_$i.err; rc=$?
  python - "$F" $rc gpurun_out/bm_$i.json <<'PY'
import json, sys
f, rc, path = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print("rc %s ok %s value %.0f %s frac %.4f | %s" % (rc, d.get("ok"), d["value"], d["unit"], d["roofline"]["frac"], f or "(default)"))
except Exception as e:
    print("rc %s NO LINE (%r) | %s" % (rc, e, f))
PY
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/bm_$i.err; fi
done
