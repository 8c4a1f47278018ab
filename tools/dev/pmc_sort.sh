# memory-side counters of the DefragParticles kernels (one --pmc pass each)
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf gpurun_out/pmcsort_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcsort_$tag -o s -- python3 tools/dev/steps.py c3 100000000 4 4 > /dev/null 2> gpurun_out/pmcsort_err.txt || echo "pass $tag failed"
done
python3 - <<'P'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmcsort_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sort" in r["Kernel_Name"] or "k_scan" in r["Kernel_Name"]:
            tot[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
for k in sorted(tot): print(k, "%.4e" % tot[k])
P
