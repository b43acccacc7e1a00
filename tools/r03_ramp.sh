cd $GRAFT_REPO_ROOT
for w in noptrex nedm long25 nab100; do timeout -k 10 300 python3 tools/workload.py $w --ramp --steps 3 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w ramp: path', d['decode_path'], 'bits/sample', round(d['ratio']*16,2), 'encode_ms', round(d['encode_ms']['total'],3), 'decode_ms', round(d['decode_ms']['total'], 3), 'frac', round(d['decode_frac_of_8TBps'], 4))"; done
