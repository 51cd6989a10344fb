#!/bin/bash
# round 2, GPU call Z2: CLAHE LUT kernel at 1920x1080 in isolation: workgroups per tile x load width
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
for S in 1 2 4 8; do for V in 0 1; do
  n=S${S}_V$V
  LVK_CLAHE_S=$S LVK_CLAHE_VEC=$V timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$n -- python tools/gpu/img_probe.py 1920 1080 > $O/$n.log 2>&1
  for db in $(find $O/p_$n -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/$n.csv; done
  echo "== $n $(grep k_clahe $O/$n.csv)"
done; done
find $O -name "*.db" -delete
