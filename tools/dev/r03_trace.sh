# usage: r03_trace.sh <workload> <particles>  -- kernel trace of one bench step
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=$1; n=$2
rm -rf gpurun_out/trace_$w
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$w -o t -- \
  python3 bench.py --workload $w --particles-per-gpu $n --steps 2 --warmup 1 --no-cpu-baseline --no-other-variant > gpurun_out/trace_$w.json 2> gpurun_out/trace_err.txt
python3 - "$w" <<'P'
import csv, glob, sys
w = sys.argv[1]
for f in glob.glob(f"gpurun_out/trace_{w}/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:90], r["Calls"], "calls avg(us)", float(r["AverageNs"]) / 1e3)
P
