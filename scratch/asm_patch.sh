#!/bin/bash
# usage: scratch/asm_patch.sh <name> <python expression editing list `lines` of the device assembly of cppf_vote.hip>
# Builds scratch/rotdbg/lib_<name>.so from hand-edited device assembly (the host half is compiled normally, LDS floor off).
set -e
cd "$(dirname "$0")/.."; mkdir -p scratch/rotdbg /tmp/isa
NAME=$1; EDIT=$2
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -fPIC -I include -I cppf2_amd/csrc -fno-slp-vectorize"
LL=/opt/rocm/lib/llvm/bin
[ -f /tmp/isa/vote_base.s ] || /opt/rocm/bin/hipcc $F --cuda-device-only -S cppf2_amd/csrc/cppf_vote.hip -o /tmp/isa/vote_base.s 2>/dev/null
python3 - "$NAME" "$EDIT" <<'PY'
import sys, re
name, edit = sys.argv[1], sys.argv[2]
lines = open("/tmp/isa/vote_base.s").read().split("\n")
beg = next(i for i, l in enumerate(lines) if l.startswith("_Z19rot_bins_lut_kernelILi2E"))
end = next(i for i in range(beg, len(lines)) if "s_endpgm" in lines[i])
body = lines[beg:end]
ns = {"body": body, "re": re}
exec(edit, ns)
lines[beg:end] = ns["body"]
open("/tmp/isa/vote_%s.s" % name, "w").write("\n".join(lines))
print("edited", name, "kernel lines", len(body), "->", len(ns["body"]))
PY
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/isa/vote_$NAME.s -o /tmp/isa/vote_$NAME.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o /tmp/isa/vote_$NAME.out /tmp/isa/vote_$NAME.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=/tmp/isa/vote_$NAME.out -output=/tmp/isa/vote_$NAME.hipfb
/opt/rocm/bin/hipcc $F --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang /tmp/isa/vote_$NAME.hipfb -c cppf2_amd/csrc/cppf_vote.hip -o scratch/rotdbg/vote_$NAME.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/rotdbg/lib_$NAME.so scratch/rotdbg/vote_$NAME.o $(ls cppf2_amd/csrc/*.o | grep -v cppf_vote.o) && echo built $NAME
