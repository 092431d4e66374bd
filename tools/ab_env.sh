#!/bin/bash
# same-box interleaved A/B of an environment switch over bench.py (headline arithmetic only): tools/ab_env.sh VAR=1 [cfg2|cfg3] [rounds] [kernel filter regex]
# prints ms/step and the matching per-kernel rows with the switch off (A) and on (B)
V=$1; WL=${2:-cfg3}; N=${3:-2}; F=${4:-.}
for i in $(seq 1 $N); do
  for arm in A B; do
    if [ $arm = B ]; then export $V; else unset ${V%%=*}; fi
    python bench.py --workload $WL --modes f16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-search --no-gan --no-sustained --traffic off --detail /tmp/ab_$arm.json > /dev/null 2>&1
    python - $arm "$F" <<'PY'
import json, re, sys
d = json.load(open(f"/tmp/ab_{sys.argv[1]}.json"))
rows = [f"{k.split('_kernel')[0][8:] + k.split('_kernel')[1]}={v['ms_per_step']:.4f}" for k, v in d["kernels"].items() if re.search(sys.argv[2], k)][:8]
print(f"  {sys.argv[1]} {d['ms_per_step']:.4f} ms/step  ew {d['elementwise']['ms_per_step']:.3f}  " + "  ".join(rows))
PY
  done
done
