#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
d() { name=$1; shift; env "$@" timeout 300 python tools/gpu/diag_pipe.py 5 60 4 > $O/$name.txt 2>&1; echo "== $name: $(grep -m1 '^frame' $O/$name.txt | cut -c1-60) :: $(tail -1 $O/$name.txt)"; }
for k in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do d fix$k X=1; done
for k in 1 2 3; do d nofuse$k LVK_CHOL_FUSED=0; done
