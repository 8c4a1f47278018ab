# longer runs of every workload (many cycles each): nothing hangs, nothing is lost
set -e
mkdir -p gpurun_out
for w in "c2 10000000 12" "c3 100000000 6" "c4 10000000 12" "c5 10000000 10" "c1 100000 40" "c3-1d 100000000 6"; do
  set -- $w
  timeout -k 10 400 python bench.py --workload $1 --particles-per-gpu $2 --steps $3 --warmup 1 --no-cpu-baseline --no-other-variant > gpurun_out/soak_$1.json 2> gpurun_out/soak_$1.err
  python - $1 <<'P'
import json, sys
d = json.load(open(f"gpurun_out/soak_{sys.argv[1]}.json"))
k = d["kernel_diagnostics"]
print(sys.argv[1], "steps", d["steps"], "ms/step %.2f" % d["ms_per_step"], "value %.4e" % d["value"], {a: k[a] for a in ("n_census", "n_absorbed", "n_escaped", "n_outgoing")})
P
done
