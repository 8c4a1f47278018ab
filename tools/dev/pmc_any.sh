# usage: pmc_any.sh <workload> <particles> "<counters of pass 1>" ["<counters of pass 2>" ...]
# one `--pmc` pass of `bench.py --steps 1 --warmup 0` per counter list (in-tree library); prints the
# counters summed over the tracking kernel's launches
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=$1; n=$2; shift 2
i=0
for CN in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcany_${w}_$i
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d gpurun_out/pmcany_${w}_$i -o runc -- \
      python3 bench.py --workload $w --particles-per-gpu $n --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant \
      > gpurun_out/pmcany_${w}_$i.json 2> gpurun_out/pmcany_err.txt
  python3 - gpurun_out/pmcany_${w}_$i <<'P'
import csv, glob, os, sys, collections
tot = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("k_transport", "k_ddmc_all", "k_ddmc_q", "k_imc_cell", "k_hybrid")):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(f"  {k:32s} {tot[k]:.5g}")
P
done
