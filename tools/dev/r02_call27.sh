set -e
mkdir -p gpurun_out
bash tools/dev/ab2.sh c3 100000000 timing | tee gpurun_out/r02_c27_ab.txt
grep JB_TIMING gpurun_out/ab_err.txt | tail -3
python - <<'P'
import json
d=json.load(open('gpurun_out/ab_c3_timing.json'))['kernel_diagnostics']
print('cyc_ev>>10', d['n_wave_passes'], 'cyc_sv>>10', d['n_wave_services'])
P
