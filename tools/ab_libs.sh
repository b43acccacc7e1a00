#!/bin/bash
# Same-box A/B of two builds of libdeltarice_hip.so (DRX_LIB_PATH): tools/ab_libs.sh base.so new.so [rounds]
A=$1; B=$2; N=${3:-3}
for r in $(seq 1 $N); do
  for L in "$A" "$B"; do
    DRX_LIB_PATH=$PWD/$L timeout -k 5 200 python bench.py --cpu-seconds 0 --steps 4 --warmup 1 > /tmp/abl.json || exit 1
    python - "$L" <<'PY'
import sys, json
d = json.loads(open("/tmp/abl.json").read().strip().splitlines()[-1])
k = d["kernel_ms"]
print("%-32s enc %.3f dec %.3f  value %.0f" % (sys.argv[1], k["encode_kernel"], k["decode_kernel"], d["value"]), flush=True)
PY
  done
done
