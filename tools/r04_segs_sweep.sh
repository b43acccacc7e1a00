# encode ms versus WaveformLength (uniform plans, 50 chunks of 14 M samples): k_encode_stream_segs from every length on
# (DRX_SEGS_MIN_LEN=0, an -DDRX_ABLATION build) against the encoders it replaces (encode_impl 1: pieces / fused)
set -o pipefail
cd $GRAFT_REPO_ROOT
export DRX_LIB_PATH=$PWD/deltarice_amd/variants/lib_abl.so DRX_SWEEP_CHUNKS=50 DRX_NO_VERIFY=1
LENS="7000 9000 10240 12000 16384 24000 32768 65536 131072 1000000 14000000"
echo "== segs everywhere"; DRX_SEGS_MIN_LEN=0 timeout -k 10 300 python3 tools/len_sweep.py $LENS
echo "== encode_impl 1 (pieces / fused)"; DRX_ENCODE_IMPL=1 timeout -k 10 300 python3 tools/len_sweep.py $LENS
echo "== default dispatch"; unset DRX_LIB_PATH; timeout -k 10 300 python3 tools/len_sweep.py $LENS
