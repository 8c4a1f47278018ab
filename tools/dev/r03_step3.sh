set -e
mkdir -p gpurun_out
bash tools/dev/pmc2.sh c2 10000000 base cur > gpurun_out/s3_pmc.txt 2>&1 || true
grep -v "^pmc pass" gpurun_out/s3_pmc.txt | tail -8
