# per-cycle wall time over 24 cycles: DefragParticles on the library's schedule (-1, the default), never (0),
# every 8th cycle (profiles/r04_long_run.json)
set -e
mkdir -p gpurun_out
: > gpurun_out/long_run.jsonl
for d in -1 0 8; do timeout -k 10 500 python tools/dev/steps.py c3 100000000 24 $d 2>/dev/null | tail -1 >> gpurun_out/long_run.jsonl; echo "c3 $d done"; done
for d in -1 0; do timeout -k 10 500 python tools/dev/steps.py c2 10000000 24 $d 2>/dev/null | tail -1 >> gpurun_out/long_run.jsonl; echo "c2 $d done"; done
for d in -1 0; do timeout -k 10 500 python tools/dev/steps.py c5 10000000 24 $d 2>/dev/null | tail -1 >> gpurun_out/long_run.jsonl; echo "c5 $d done"; done
cat gpurun_out/long_run.jsonl
