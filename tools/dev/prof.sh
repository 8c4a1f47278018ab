# usage: prof.sh <tag>   (e.g. r04)
# The command list behind profiles/<tag>_*: kernel stats of the default bench under rocprofv3, one
# --pmc pass per counter group for the headline (c2), DDMC (c3), SMR (c4) and hybrid (c5) workloads,
# and the bench lines of every workload.  `python tools/collect_profiles.py <tag>` turns
# gpurun_out/<tag>prof/ into profiles/<tag>_*.
# PART=1: kernel stats + the --pmc passes of c2, c3, c3-1d; PART=2: the --pmc passes of c5, c4, c2 exact, the marker
# trace and the bench lines; unset: everything (longer than one gpurun call allows)
set -e
TAG=${1:-r06}
PART=${PART:-all}
O=gpurun_out/${TAG}prof
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C2="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant"
C3="python3 bench.py --workload c3 --particles-per-gpu 100000000 --steps 1 --warmup 0 --no-cpu-baseline"
C31="python3 bench.py --workload c3-1d --particles-per-gpu 100000000 --steps 1 --warmup 0 --no-cpu-baseline"
C4="python3 bench.py --workload c4 --particles-per-gpu 10000000 --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant"
C5="python3 bench.py --workload c5 --particles-per-gpu 10000000 --steps 1 --warmup 0 --no-cpu-baseline"
C2X="python3 bench.py --arithmetic exact --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant"
if [ "$PART" != 2 ] && [ "$PART" != pmc ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o runc -- python3 bench.py > $O/bench_c2_under_rocprof.json 2> $O/stats_c2.err
echo "stats c2 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o runc -- python3 bench.py --workload c3 --particles-per-gpu 100000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2> $O/stats_c3.err
echo "stats c3 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o runc -- python3 bench.py --workload c5 --particles-per-gpu 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_under_rocprof.json 2> $O/stats_c5.err
echo "stats c5 done"
fi
PA="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
PB="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"
PC="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
PD="FETCH_SIZE"
PE="WRITE_SIZE"
PF="TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"
PG="TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"
PH="TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum"
PI="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
# PART=pmc WLS="c2 c4": only the --pmc passes of the workloads named
case $PART in 1) WLS="c2 c3 c3-1d";; 2) WLS="c5 c4 c2x";; pmc) ;; *) WLS="c2 c3 c3-1d c5 c4 c2x";; esac
for wl in $WLS; do
  case $wl in c2) CMD=$C2;; c3) CMD=$C3;; c3-1d) CMD=$C31;; c4) CMD=$C4;; c5) CMD=$C5;; c2x) CMD=$C2X;; esac
  for p in A B C D E F G H I; do
    eval "CN=\$P$p"
    timeout -k 5 100 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d $O/pmc_${wl}_$p -o runc -- $CMD > $O/pmc_${wl}_$p.json 2> $O/pmc_${wl}_$p.err || echo "pass $wl $p FAILED"
    echo "pmc $wl $p done"
  done
done
[ "$PART" = 1 ] && { echo "part 1 done"; exit 0; }
[ "$PART" = pmc ] && { echo "pmc passes done"; exit 0; }
# the reference's trace ranges (Jaybenne::Timestep, Jaybenne::TransportLoop, one per task) beside the kernels of a
# configs[4] cycle
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/markers_c5 -o runc -- python3 bench.py --workload c5 --particles-per-gpu 10000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_markers.json 2> $O/markers_c5.err || echo "marker trace FAILED"
echo "marker trace done"
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for w in "c1 100000" "c3 100000000" "c3-1d 100000000" "c4 10000000" "c5 10000000"; do
  set -- $w
  timeout -k 10 300 python3 bench.py --workload $1 --particles-per-gpu $2 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  echo "bench $1 done"
done
echo all done
