#!/bin/bash
# the side workloads with the in-tree library; the IMC ones also with the x-space lean kernel of round 3
set -e
bash tools/dev/ab2.sh c4 10000000 cur cur@JB_NO_IMC_CELL=1
bash tools/dev/ab2.sh c1 100000 cur cur@JB_NO_IMC_CELL=1
bash tools/dev/ab2.sh c5 10000000 cur
bash tools/dev/ab2.sh c3 100000000 cur
bash tools/dev/ab2.sh c3-1d 100000000 cur
