#!/bin/bash
# what the kernels of a SYNCHRONOUS batched call take (poses read zero-copy from host memory) vs the device-pose timing loop: C4 shard and C3
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; rm -rf $O; mkdir -p $O
for wl in c4 c3; do
  C4_CONFIGS=0:0:0 WORKLOAD=$wl timeout 300 python scripts/r03_c4_traffic.py 2>/dev/null | grep "^{" | head -1 | cut -c1-300 | tee -a $O/times.jsonl
  rm -rf /tmp/pb && PMC=1 C4_CONFIGS=0:0:0 WORKLOAD=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python scripts/r03_c4_traffic.py > /tmp/pb.log 2>&1
  f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "vgicp_stream_kernel|finalize" $f | cut -c1-60,170-260 | tee -a $O/sync_kernels_$wl.txt
done
