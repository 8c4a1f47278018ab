set -e
mkdir -p gpurun_out
JAYBENNE_AMD_LIB=$PWD/variants/libjb_hstats.so timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/hstats.json 2> gpurun_out/hstats_err.txt
grep JB_HYB_STATS gpurun_out/hstats_err.txt | tail -3
