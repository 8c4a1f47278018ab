# the bench line of every workload into gpurun_out/<tag>prof/bench_<w>.json (collect_profiles.py copies them to
# profiles/): run AFTER the round's pmc summaries are in profiles/, so that roofline.traffic and the counter
# fractions quoted in the lines are the round's own
TAG=${1:-r06}
O=gpurun_out/${TAG}prof
mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc $?"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_20steps.json 2> $O/bench_c2_20steps.err; echo "bench c2 20 steps rc $?"
timeout -k 10 400 python3 bench.py --particles-per-gpu 12500000 --no-cpu-baseline > $O/bench_c2_12500k.json 2> $O/bench_c2_12500k.err; echo "bench c2 1.25e7 rc $?"
for w in "c1 100000" "c3 100000000" "c3-1d 100000000" "c4 10000000" "c5 10000000"; do
  set -- $w
  timeout -k 10 300 python3 bench.py --workload $1 --particles-per-gpu $2 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  echo "bench $1 rc $?"
done
