#!/bin/bash
# usage: asm_one.sh [extra hipcc flags]  -> /tmp/asm/one.s + per-block instruction counts of the one kernel
# instantiated by /tmp/asm/one.hip (default k_transport<3,false,true,2>, the C2 kernel)
set -e
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
  -Wno-unused-function -I/root/repo/jaybenne_amd/csrc -S --cuda-device-only "$@" /tmp/asm/one.hip -o /tmp/asm/one.s 2>&1 | grep -v "hip-link" || true
python3 /root/repo/tools/dev/asm_count.py /tmp/asm/one.s k_transport | awk '$5>60 || /total/'
grep -E "NumVgprs:|ScratchSize|Occupancy" /tmp/asm/one.s | head -3
