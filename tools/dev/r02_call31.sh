mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|Counter_Name)?\s*:?\s*(TCP_|TA_|TD_)" | head -120 > gpurun_out/avail_tcp.txt
rocprofv3 --list-avail > gpurun_out/avail_all.txt 2>&1
wc -l gpurun_out/avail_all.txt
grep -c TCP_ gpurun_out/avail_all.txt
