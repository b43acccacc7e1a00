#!/bin/bash
# encode / decode against the noise level, RiceParameter chosen for it: tools/r03_noise.sh [workload]
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
w=${1:-nab100}
for sm in "10 8" "40 32" "80 64" "160 128" "320 256" "1000 1024" "3000 2048"; do
  set -- $sm
  timeout -k 10 200 python3 tools/workload.py $w --sigma $1 --m $2 --steps 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w sigma', d['sigma'], 'm', d['m'], 'bits/sample', round(d['ratio'] * 16, 2), 'encode_ms', round(d['encode_ms']['total'], 3), 'frac', round(d['encode_frac_of_8TBps'], 3), 'decode_ms', round(d['decode_ms']['total'], 3), 'frac', round(d['decode_frac_of_8TBps'], 3))"
done
