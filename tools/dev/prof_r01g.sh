set -e
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1g -o runc -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_prof_g.json 2> gpurun_out/bench_prof_g.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_r1g_SQ_WAVES -o runc -- $B > /dev/null 2> gpurun_out/pmc_g1.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_r1g_SQ_INSTS_VMEM_RD -o runc -- $B > /dev/null 2> gpurun_out/pmc_g2.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r1g_FETCH_SIZE -o runc -- $B > /dev/null 2> gpurun_out/pmc_g3.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r1g_WRITE_SIZE -o runc -- $B > /dev/null 2> gpurun_out/pmc_g4.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/pmc_r1g_TCC_HIT_sum -o runc -- $B > /dev/null 2> gpurun_out/pmc_g5.err
C3="python3 bench.py --workload c3 --particles-per-gpu 100000000 --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1g_c3 -o runc -- python3 bench.py --workload c3 --particles-per-gpu 100000000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof_g_c3.json 2> gpurun_out/bench_prof_g_c3.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_r1g_c3_SQ -o runc -- $C3 > /dev/null 2> gpurun_out/pmc_g6.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r1g_c3_FETCH_SIZE -o runc -- $C3 > /dev/null 2> gpurun_out/pmc_g7.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r1g_c3_WRITE_SIZE -o runc -- $C3 > /dev/null 2> gpurun_out/pmc_g8.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_r1g_c3_TCC -o runc -- $C3 > /dev/null 2> gpurun_out/pmc_g9.err
timeout -k 10 300 python bench.py > gpurun_out/bench_final_g.json 2> gpurun_out/bench_final_g.err
echo all done
