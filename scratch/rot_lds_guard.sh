#!/bin/bash
# Builds libcppf_hip variants whose rotation-vote kernel keeps guard bytes around its LDS data (run here, not on the GPU box).
cd "$(dirname "$0")/.."; mkdir -p scratch/rotdbg
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -fPIC -I include -I cppf2_amd/csrc"
build() {  # name flags
  /opt/rocm/bin/hipcc $F $2 -c cppf2_amd/csrc/cppf_vote.hip -o scratch/rotdbg/vote_$1.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/rotdbg/lib_$1.so scratch/rotdbg/vote_$1.o $(ls cppf2_amd/csrc/*.o | grep -v cppf_vote.o) && echo built $1
}
build none "-DROT_LDS_FLOOR=0" &
build tail "-DROT_LDS_FLOOR=0 -DROT_LDS_TAIL=1536" &
build head "-DROT_LDS_FLOOR=0 -DROT_LDS_HEAD=1536" &
build both "-DROT_LDS_FLOOR=0 -DROT_LDS_HEAD=1280 -DROT_LDS_TAIL=1280" &
wait
