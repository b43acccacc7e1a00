#!/bin/bash
# A/B of the decode variants at full bench size on one box: tools/ab_impls.sh [impl ...]
for i in "${@:-1 2 3 4 5 6}"; do
  for j in $i; do
    timeout -k 5 200 python bench.py --cpu-seconds 0 --steps 4 --warmup 1 --decode-impl $j > /tmp/ab_$j.json || exit 1
    python - $j <<'PY'
import sys, json
d = json.loads(open(f"/tmp/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k = d["kernel_ms"]
print("impl", sys.argv[1], "enc %.3f" % k["encode_kernel"], "walk %.3f" % k["decode_prepare"], "dec %.3f" % k["decode_kernel"], "frac %.3f" % d["roofline"]["frac"], "dec GB/s %.0f" % d["decode_GBps"], flush=True)
PY
  done
done
