# The --pmc passes of prof_r02.sh for two more workloads: c4 (2-D SMR, pure IMC) and c5 (3-level
# IMC / DDMC hybrid); tools/collect_profiles2.py picks them up as profiles/r02_pmc_summary_c{4,5}.json
set -e
O=gpurun_out/r02prof
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PA="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
PB="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"
PC="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
PD="FETCH_SIZE"
PE="WRITE_SIZE"
PF="TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"
for wl in c4 c5; do
  CMD="python3 bench.py --workload $wl --particles-per-gpu 10000000 --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant"
  for p in A B C D E F; do
    eval "CN=\$P$p"
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d $O/pmc_${wl}_$p -o runc -- $CMD > $O/pmc_${wl}_$p.json 2> $O/pmc_${wl}_$p.err || echo "pass $wl $p FAILED"
    echo "pmc $wl $p done"
  done
done
echo all done
