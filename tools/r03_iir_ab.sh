#!/bin/bash
# inverse-filter kernel A/B on one box: tools/r03_iir_ab.sh variant... ("prod" = the library in the tree)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for w in noptrex_fir4 nedm_fir4 raglong_fir4; do
  for v in "$@"; do
    if [ $v = prod ]; then unset DRX_LIB_PATH; else export DRX_LIB_PATH=$PWD/deltarice_amd/variants/lib_$v.so; fi
    timeout -k 10 200 python3 tools/workload.py $w --steps 10 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', '$v', 'decode_ms', round(d['decode_ms']['total'], 4))"
  done
done
done
