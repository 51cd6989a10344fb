#!/bin/bash
# round 2, GPU call Q: config-5 pipelined nondeterminism: dependency switches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2q; mkdir -p $O
B="timeout 600 python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline --no-shard-probe"
run() { name=$1; shift; env "$@" $B $EXTRA > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['config']['backend'], d['config']['timed_region'], 'be', d.get('back_end_ms_per_message'))
except Exception as e:
    print('$name', 'ERR', open('$O/$name.err').read()[-300:].replace(chr(10),' | '))
PY
}
EXTRA=--no-device-pass
run dep1 LVK_FE_DEBUG_DEP=1
run dep1b LVK_FE_DEBUG_DEP=1
run dep2 LVK_FE_DEBUG_DEP=2
run dep2b LVK_FE_DEBUG_DEP=2
run q4 GPU_MAX_HW_QUEUES=4
run q4b GPU_MAX_HW_QUEUES=4
run zc LVK_FE_ZEROCOPY=1
run zcb LVK_FE_ZEROCOPY=1
