#!/bin/bash
# Round 6: counters of the two covariance kernels on the C5 cloud.  Separate rocprofv3 passes (--pmc with --kernel-trace only), then a timing pass.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06; mkdir -p $O
R=$PWD
timeout 300 python3 scripts/r06/c5_run.py 9 > $O/c5_run.json 2> $O/c5_run.err; echo "exit $?"; cat $O/c5_run.json
i=0
for ctrs in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1)); rm -rf /tmp/c5pmc$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/c5pmc$i -o p -- python3 $R/scripts/r06/c5_run.py 5 src-only > /dev/null 2> $O/c5_pmc_pass$i.err)
  f=$(find /tmp/c5pmc$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_pmc_pass$i.csv || echo "pass $i: no counters ($(tail -2 $O/c5_pmc_pass$i.err))"
done
rm -rf /tmp/c5st; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5st -o s -- python3 $R/scripts/r06/c5_run.py 9 src-only > /dev/null 2> $O/c5_stats.err)
f=$(find /tmp/c5st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv
f=$(find /tmp/c5st -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && grep -E "Kernel_Name|covariance" $f > $O/c5_kernel_trace.csv
python3 scripts/r06/c5_pmc_summary.py $O > $O/c5_pmc.txt; cat $O/c5_pmc.txt
