#!/bin/bash
# round 2, GPU call N: same-box A/B of three library builds (J: image overlap, K: + fused Cholesky / host IMU, L: + image kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2n; mkdir -p $O
B="timeout 600 python bench.py --no-cpu-baseline --no-shard-probe --no-device-pass"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['config']['backend'], d['config']['timed_region'], d['roofline']['avg_launch_us'], d['roofline_mfma']['avg_launch_us'])
except Exception as e:
    print('$name', 'ERR', open('$O/$name.err').read()[-300:].replace(chr(10),' | '))
PY
}
run J1 LVK_LIB=$PWD/build_variants/liblvk_J.so
run K1 LVK_LIB=$PWD/build_variants/liblvk_K.so
run L1 X=1
run J2 LVK_LIB=$PWD/build_variants/liblvk_J.so
run K2 LVK_LIB=$PWD/build_variants/liblvk_K.so
run K_nofuse LVK_LIB=$PWD/build_variants/liblvk_K.so LVK_CHOL_FUSED=0
run L2 X=1
run L_q4 GPU_MAX_HW_QUEUES=4
rocm-smi --showclocks 2>/dev/null | head -20
