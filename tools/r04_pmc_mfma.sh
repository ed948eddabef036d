# MFMA-busy and LDS-bank-conflict counters per kernel of the serial Whisper step (two --pmc passes, kernel-trace only).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4m; rm -rf $O; mkdir -p $O
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/_a -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/a.log 2>&1
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/_b -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/b.log 2>&1
python3 tools/pmc_kernel_counters.py $(find $O/_a -name '*counter_collection.csv' | head -1) > $O/pmc_mfma.txt 2>&1
python3 tools/pmc_kernel_counters.py $(find $O/_b -name '*counter_collection.csv' | head -1) > $O/pmc_lds.txt 2>&1
rm -rf $O/_a $O/_b
head -30 $O/pmc_mfma.txt; head -30 $O/pmc_lds.txt
