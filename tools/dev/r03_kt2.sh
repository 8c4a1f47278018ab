set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt2 -o kt -- python3 tools/dev/steps.py c3 100000000 8 4 > gpurun_out/kt2_out.txt 2> gpurun_out/kt2_err.txt
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/kt2/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
P
