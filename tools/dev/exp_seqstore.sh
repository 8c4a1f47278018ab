bash tools/dev/ab1.sh c3-1d 100000000 cur seqstore nostore
bash tools/dev/ab1.sh c3 100000000 cur seqstore nostore
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c3 c3-1d; do
  export JAYBENNE_AMD_LIB=$PWD/variants/libjb_seqstore.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcw_${w}_seq_$c
    timeout -k 10 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcw_${w}_seq_$c -o runc -- python3 bench.py --workload $w --particles-per-gpu 100000000 --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant > gpurun_out/pmcw_c3.json 2> gpurun_out/pmcw_err.txt
    python3 - gpurun_out/pmcw_${w}_seq_$c $w/seqstore $c <<'P'
import csv, glob, os, sys
tot = 0.0
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ddmc" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]: tot += float(r["Counter_Value"])
print(sys.argv[2], sys.argv[3], "%.3f GB" % (tot * 1024 / 1e9))
P
  done
done
