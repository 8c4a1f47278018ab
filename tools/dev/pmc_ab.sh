# usage: pmc_ab.sh "<lib tags>" <workload> <particles>
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $1; do
  if [ $v = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$v.so; fi
  export JAYBENNE_AMD_LIB=$L
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmcab_${2}_$v -o runc -- python3 bench.py --workload $2 --particles-per-gpu $3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/pmcab_err.txt
done
echo pmc ab done
