#!/bin/bash
# round 2, GPU call O: stream-ordered memsets at create time; repeat the pipelined bench (config A x3, config 5)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
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
run L1 X=1
run L2 X=1
run L3 X=1
run L_legacy LVK_FE_LEGACY_IMAGE_KERNELS=1
timeout 900 python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline --no-shard-probe > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
cut -c1-200 $O/bench_c5.json; tail -2 $O/bench_c5.err
