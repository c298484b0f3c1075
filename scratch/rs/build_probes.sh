#!/bin/bash
# builds one probe binary per RS_DBG mask (and any extra -D given as NAME=FLAGS pairs) into scratch/rs/bin/
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -fno-slp-vectorize -I include -I cppf2_amd/csrc"
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  /opt/rocm/bin/hipcc $FL $flags scratch/rs/rs_probe.hip -o scratch/rs/bin/probe_$name &
done
wait
ls -la scratch/rs/bin
