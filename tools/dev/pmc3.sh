# usage: pmc3.sh <workload> <particles>   -- texture-path counters (TA / TCP / UTCL1) of the in-tree library
set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=$1; n=$2
G1="TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
G2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
G3="TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G4="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
G5="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G6="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT"
for p in 1 2 3 4 5 6; do
  eval "CN=\$G$p"
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d gpurun_out/pmc3_${w}_$p -o runc -- \
     python3 bench.py --workload $w --particles-per-gpu $n --steps 1 --warmup 0 --no-cpu-baseline --no-other-variant \
     > gpurun_out/pmc3_${w}_$p.json 2> gpurun_out/pmc3_err.txt || echo "pass $p FAILED"
  echo "pass $p done"
done
python3 - $w <<'P'
import csv, glob, sys, collections
w = sys.argv[1]
tot = collections.defaultdict(float); dur = {}
for p in "123456":
    d = f"gpurun_out/pmc3_{w}_{p}"
    names = collections.Counter()
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_transport" in r["Kernel_Name"] or "k_ddmc_all" in r["Kernel_Name"] or "k_ddmc_q" in r["Kernel_Name"]:
                names[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not names: continue
    main = names.most_common(1)[0][0]; dur[p] = names[main] * 1e-6
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == main: tot[r["Counter_Name"]] += float(r["Counter_Value"])
print("kernel ms by pass", dur)
for k, v in sorted(tot.items()): print(f"{k:44s} {v:.6e}")
P
