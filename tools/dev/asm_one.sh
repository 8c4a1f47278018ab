#!/bin/bash
# usage: [KARGS="3, false, true, 2, true, true"] asm_one.sh [extra hipcc flags]
# Compiles ONE instantiation of k_transport<NDIM, DDMC, TALLY, GRAY, EXACT, LEAN> (default: the
# headline kernel of BASELINE configs[1]) to ISA in /tmp/asm/one.s and prints the instruction
# count of its larger basic blocks, its register count and scratch use.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/asm
cat > /tmp/asm/one.hip <<HIP
#include "jb_kernels.hpp"
template __global__ void jb::k_transport<${KARGS:-3, false, true, 2, true, true}>(jb::DevMesh, jb::DevParams, jb::DevSwarm, double, double, long long, long long, unsigned long long *, const int *);
HIP
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
  -Wno-unused-function -I$ROOT/jaybenne_amd/csrc -S --cuda-device-only "$@" /tmp/asm/one.hip -o /tmp/asm/one.s 2>&1 | grep -v "hip-link" || true
python3 $ROOT/tools/dev/asm_count.py /tmp/asm/one.s k_transport | awk '$5>40 || /total/'
grep -E "NumVgprs:|ScratchSize" /tmp/asm/one.s | tail -2
