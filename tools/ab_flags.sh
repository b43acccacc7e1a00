#!/bin/bash
# A/B of decode variants and debug flags at full bench size on one box: tools/ab_flags.sh "impl:flags" ...
for spec in "$@"; do
  j=${spec%%:*}; f=${spec##*:}
  timeout -k 5 200 python bench.py --cpu-seconds 0 --steps 4 --warmup 1 --decode-impl $j --debug-flags $f > /tmp/ab_$j.json || exit 1
  python - $j $f <<'PY'
import sys, json
d = json.loads(open(f"/tmp/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k = d["kernel_ms"]
print("impl", sys.argv[1], "flags", sys.argv[2], "enc %.3f" % k["encode_kernel"], "walk %.3f" % k["decode_prepare"], "dec %.3f" % k["decode_kernel"], "frac %.3f" % d["roofline"]["frac"], "dec GB/s %.0f" % d["decode_GBps"], flush=True)
PY
done
