#!/bin/bash
# SQ counters of the headline kernels (bench.py workload): tools/r03_sq.sh tag [lib]   -> gpurun_out/r03_sq_<tag>/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp; O=$R/gpurun_out/r03_sq_$1; rm -rf $O; mkdir -p $O; cd $R
[ -n "$2" ] && export DRX_LIB_PATH=$R/deltarice_amd/variants/lib_$2.so
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/a -o p --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 2 --warmup 1 --no-collect $EXTRA > $O/a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR -d $O/b -o p --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 2 --warmup 1 --no-collect $EXTRA > $O/b.log 2>&1
python3 profiles/pmc_summary.py $O/a/p_counter_collection.csv $O/b/p_counter_collection.csv > $O/summary.txt 2>&1
grep -A18 "k_decode_lanes\|k_encode_fused" $O/summary.txt | head -60
rm -rf $O/a $O/b
