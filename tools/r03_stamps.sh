set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03s
for w in long25 nedm noptrex noptrex_fir4; do
  echo "== $w"; DRX_LIB_PATH=$PWD/deltarice_amd/variants/lib_stamps.so timeout -k 10 200 python3 tools/workload.py $w --steps 3 2>&1 | grep -E "blk stamps|decode_ms" | tail -4 | cut -c1-600
done
