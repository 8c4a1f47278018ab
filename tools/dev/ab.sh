# usage: ab.sh "<lib tags>" "<workload:particles ...>"  -> gpurun_out/ab_<wl>_<tag>.json
set -e
mkdir -p gpurun_out
for wp in $2; do
  w=${wp%%:*}; n=${wp##*:}
  for v in $1; do
    if [ $v = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$v.so; fi
    JAYBENNE_AMD_LIB=$L timeout -k 10 300 python bench.py --workload $w --particles-per-gpu $n --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${w}_$v.json 2> gpurun_out/ab_err.txt
  done
done
echo ab done
