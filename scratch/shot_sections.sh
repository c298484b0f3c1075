#!/bin/bash
# Builds libcppf_hip variants with sections of shot_hist_kernel disabled (SHOT_DBG) into scratch/shotdbg/ (run here, not on the GPU box).
cd "$(dirname "$0")/.."; mkdir -p scratch/shotdbg
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -fPIC -I include -I cppf2_amd/csrc -fno-slp-vectorize"
for d in ${@:-1 2 4 7}; do
  /opt/rocm/bin/hipcc $F -DSHOT_DBG=$d $SHOT_EXTRA -c cppf2_amd/csrc/cppf_shot.hip -o scratch/shotdbg/shot_$d.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/shotdbg/lib_$d.so scratch/shotdbg/shot_$d.o $(ls cppf2_amd/csrc/*.o | grep -v cppf_shot.o) && echo built $d
done
