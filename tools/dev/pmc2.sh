# usage: pmc2.sh <workload> <particles> <item> ...   (items as in ab2.sh)
# three --pmc passes per item over `bench.py --steps 1 --warmup 0`; summaries by tools/pmc_summary.py
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=$1; n=$2; shift 2
PA="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
PB="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
PC="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU"
for item in "$@"; do
  tag=${item%%@*}
  envs=""
  if [ "$tag" != "$item" ]; then envs=$(echo "${item#*@}" | tr ',' ' '); fi
  if [ "$tag" = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$tag.so; fi
  name=$(echo "$item" | tr '@=,' '___')
  export JAYBENNE_AMD_LIB=$L
  for e in $envs; do export $e; done
  for p in A B C; do
    eval "CN=\$P$p"
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d gpurun_out/pmc2_${w}_${name}_$p -o runc -- \
        python3 bench.py --workload $w --particles-per-gpu $n --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant \
        > gpurun_out/pmc2_${w}_${name}_$p.json 2> gpurun_out/pmc2_err.txt
    echo "pmc pass $p of $item done"
  done
  for e in $envs; do unset ${e%%=*}; done
  python3 tools/pmc_summary.py $w $name
done
