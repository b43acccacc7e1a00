#!/bin/bash
# detector-like data (quiet baseline + pulses) through every workload: tools/r03_pulses.sh
cd ${GRAFT_REPO_ROOT:-$PWD}
for w in nab100 noptrex nedm long25 config5; do for sg in 2 10; do
timeout -k 10 200 python3 tools/workload.py $w --pulses --sigma $sg --m 8 --steps 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w pulses on sigma', d['sigma'], 'm 8 bits/sample', round(d['ratio'] * 16, 2), 'encode_ms', round(d['encode_ms']['total'], 3), 'frac', round(d['encode_frac_of_8TBps'], 3), 'decode_ms', round(d['decode_ms']['total'], 3), 'frac', round(d['decode_frac_of_8TBps'], 3))"
done; done
