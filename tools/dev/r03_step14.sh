set -e
mkdir -p gpurun_out
for m in 32,32,3 64,32,3 256,128,2 1024,256,2; do
  echo "mesh $m"
  JB_BENCH_C3_MESH=$m bash tools/dev/ab2.sh c3 50000000 cur cur@JB_COOP_GATHER=0
done
