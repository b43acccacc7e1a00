#!/bin/bash
# A/B of library variants (deltarice_amd/variants/lib_*.so) on one box: tools/r02_variants.sh "workloads" variant...
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02v; mkdir -p $O; cd $R
WL=$1; shift
for rep in 1 2; do
for v in "$@"; do
  for w in $WL; do
    echo "== $v $w" >> $O/variants.txt
    DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$v.so timeout -k 10 90 python3 tools/workload.py $w --steps 7 2>&1 | grep -v amdgpu.ids >> $O/variants.txt || echo "FAILED" >> $O/variants.txt
  done
done
done
python3 - <<'PY'
import json,re,os,collections
R=os.environ.get("GRAFT_REPO_ROOT",os.getcwd())
cur=None;res=collections.defaultdict(list)
for ln in open(f"{R}/gpurun_out/r02v/variants.txt"):
    if ln.startswith("== "): cur=tuple(ln.split()[1:3])
    elif ln.startswith("{"):
        d=json.loads(ln); res[cur].append((d["decode_ms"]["decode"], d["encode_ms"]["total"]))
    elif "FAILED" in ln: res[cur].append((float("nan"),)*2)
for k,v in res.items(): print(k, "decode kernel ms:", " ".join(f"{a:.3f}" for a,_ in v), " encode ms:", " ".join(f"{b:.3f}" for _,b in v))
PY
