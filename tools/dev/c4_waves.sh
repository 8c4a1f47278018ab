set -e
bash tools/dev/ab2.sh c4 10000000 cur lowd4
bash tools/dev/ab2.sh c1 100000 cur lowd4
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in cur lowd4; do
  if [ "$tag" = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$tag.so; fi
  export JAYBENNE_AMD_LIB=$L
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcw_c4_${tag}_$c
    timeout -k 10 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcw_c4_${tag}_$c -o runc -- python3 bench.py --workload c4 --particles-per-gpu 10000000 --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant > gpurun_out/pmcw_c4.json 2> gpurun_out/pmcw_err.txt
    python3 - gpurun_out/pmcw_c4_${tag}_$c $tag $c <<'P'
import csv, glob, os, sys
tot = 0.0
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_imc_cell" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]: tot += float(r["Counter_Value"])
print(sys.argv[2], sys.argv[3], "%.3f GB" % (tot * 1024 / 1e9))
P
  done
done
