# usage: longrun_ab.sh <cycles> <tag> ...   (tag "cur" or variants/libjb_<tag>.so): per-cycle wall time of
# c2, c3 and c5 on the library's default sort schedule, one line per workload and library
mkdir -p gpurun_out
c=$1; shift
for tag in "$@"; do
  if [ "$tag" = cur ]; then L=$PWD/jaybenne_amd/libjaybenne_amd.so; else L=$PWD/variants/libjb_$tag.so; fi
  for w in "c2 10000000" "c3 100000000" "c5 10000000"; do
    set -- $w
    env JAYBENNE_AMD_LIB=$L timeout -k 10 500 python tools/dev/steps.py $1 $2 $c -1 2>/dev/null | tail -1 > gpurun_out/lr_$1_$tag.json
    python - gpurun_out/lr_$1_$tag.json $1 $tag <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
ms = d.get("ms_by_cycle") or d.get("ms") or []
k = [key for key in d if isinstance(d[key], list)]
print(sys.argv[2], sys.argv[3], {key: d[key] for key in d if not isinstance(d[key], list)}, "mean of cycles 5..:", round(sum(d[k[0]][4:]) / max(len(d[k[0]][4:]), 1), 3) if k else None)
print("   ", [round(x, 1) for x in d[k[0]]] if k else d)
P
  done
done
