#!/bin/bash
# round 2, GPU call W2: fused level-0 + ORB mosaic kernel: front-end/driver tests, then same-box A/B against the committed library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frontend_sequence.py tests/test_gpu_frontend_edge.py tests/test_gpu_frontend_stages.py tests/test_gpu_vio_driver.py tests/test_gpu_zz_golden.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
B="timeout 600 python bench.py --no-cpu-baseline --no-shard-probe --no-device-pass"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], 'be', d.get('back_end_ms_per_message'), 'fe', d.get('front_end_ms_per_frame'), 'wait', d.get('caller_wait_ms_per_frame'), 'idle', d.get('worker_idle_ms_per_message'), d['config']['backend']['gated_in'])
except Exception as e:
    print('$name', 'ERR', open('$O/$name.err').read()[-300:].replace(chr(10),' | '))
PY
}
for k in 1 2 3 4; do run head$k LVK_LIB=$PWD/build_variants/liblvk_head.so; run cur$k X=1; done
