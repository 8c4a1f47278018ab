set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export JAYBENNE_AMD_LIB=$PWD/variants/libjb_w3.so
for c in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/tcc_c3_$tag -o runc -- \
    python3 bench.py --workload c3 --particles-per-gpu 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/tcc_c3_$tag.json 2> gpurun_out/tcc_err_$tag.txt || { echo "pass $tag failed"; tail -3 gpurun_out/tcc_err_$tag.txt; continue; }
  python3 - gpurun_out/tcc_c3_$tag <<'P'
import csv, glob, os, sys
tot = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ddmc_all" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(tot)
P
done
