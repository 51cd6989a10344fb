#!/bin/bash
# round 2, GPU call P: config-5 divergence of the host-image pass: A/B switches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
B="timeout 600 python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline --no-shard-probe --no-device-pass"
run() { name=$1; shift; env "$@" $B $EXTRA > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['config']['backend'], d['config']['timed_region'], 'be', d.get('back_end_ms_per_message'))
except Exception as e:
    print('$name', 'ERR', open('$O/$name.err').read()[-300:].replace(chr(10),' | '))
PY
}
run c5_default X=1
run c5_legacy_img LVK_FE_LEGACY_IMAGE_KERNELS=1
run c5_nofuse LVK_CHOL_FUSED=0
run c5_libK LVK_LIB=$PWD/build_variants/liblvk_K.so
run c5_libJ LVK_LIB=$PWD/build_variants/liblvk_J.so
EXTRA=--sequential run c5_seq X=1
run c5_default2 X=1
