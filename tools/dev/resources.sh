#!/bin/bash
# usage: resources.sh [extra hipcc flags]
# Compiles the device code of the library to ISA (/tmp/asm/lib.s) and lists every kernel's vector /
# scalar register count, scratch bytes per lane and LDS bytes (what tests/test_cabi.py guards).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/asm
[ -n "$JB_RES_REUSE" ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
  -Wno-unused-function -S --cuda-device-only "$@" $ROOT/jaybenne_amd/csrc/jb_api.hip -o /tmp/asm/lib.s
python3 - <<'P'
import re, subprocess
text = open("/tmp/asm/lib.s").read()
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
    rows.append((name, g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for (n, v, s, sc, lds), d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void jb::", "")
    flag = " <-- SPILL" if sc else (" <-- >168" if v > 168 and ("k_transport" in n or "k_ddmc" in n or "k_hybrid" in n) else "")
    print(f"{d:58s} vgpr {v:4d} sgpr {s:4d} scratch {sc:5d} lds {lds:6d}{flag}")
P
