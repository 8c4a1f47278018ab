#!/bin/bash
# tolerance tests + A/B + PMC of the in-tree library in one box: r04_abpmc.sh <workload> <particles> <item> ...
set -e
bash tools/dev/ab_checked.sh "$@"
bash tools/dev/pmc2.sh $1 $2 cur 2>&1 | tail -1 | tee gpurun_out/r04_pmc_last.txt
