# the bench lines of the two all-DDMC workloads into gpurun_out/r06prof/ (after a change that touches only k_ddmc_q)
O=gpurun_out/r06prof
mkdir -p $O
for w in "c3 100000000" "c3-1d 100000000"; do set -- $w; timeout -k 10 300 python3 bench.py --workload $1 --particles-per-gpu $2 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc $?"; done
