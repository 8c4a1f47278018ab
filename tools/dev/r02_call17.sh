set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c3 100000000 w3 xnostore xnotally xnone | tee gpurun_out/r02_c17_ab.txt
