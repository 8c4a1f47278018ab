# usage: r03_kt.sh <workload> <particles>: kernel trace + stats of 3 steps
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --workload $1 --particles-per-gpu $2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-variant > gpurun_out/kt_bench.json 2> gpurun_out/kt_err.txt
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
P
