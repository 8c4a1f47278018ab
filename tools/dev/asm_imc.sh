#!/bin/bash
# usage: [KARGS="3, true, true, true"] asm_imc.sh [extra hipcc flags]
# Compiles ONE instantiation of k_imc_cell<NDIM, TALLY, NOABS, UNIFORM> (default: the headline kernel of
# BASELINE configs[1]) to ISA in /tmp/asm/imc.s and prints per-basic-block instruction counts.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/asm
cat > /tmp/asm/imc.hip <<HIP
#include "jb_kernel_imc.hpp"
template __global__ void jb::k_imc_cell<${KARGS:-3, true, true, true}>(const jb::DevMesh *, jb::DevParams, jb::DevSwarm, double, double, long long, long long, unsigned long long *, const int *);
HIP
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
  -Wno-unused-function -I$ROOT/jaybenne_amd/csrc -S --cuda-device-only "$@" /tmp/asm/imc.hip -o /tmp/asm/imc.s 2>&1 | grep -v "hip-link" || true
python3 $ROOT/tools/dev/asm_count.py /tmp/asm/imc.s k_imc_cell | awk -v m=${MINV:-12} '$5>m || /total/'
grep -E "NumVgprs:|ScratchSize|NumSgprs|sgpr_spill|vgpr_spill" /tmp/asm/imc.s | tail -6
