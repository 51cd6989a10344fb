#!/bin/bash
# round 2, GPU call X: host bookkeeping of the filter (sorted-observation lookups, cursor walk of the map): configs[4] trace + A
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backend.py tests/test_gpu_vio_driver.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="timeout 900 python bench.py --no-cpu-baseline --no-shard-probe --no-device-pass"
LVK_EKF_TRACE=1 $B --config 5 --steps 100 --warmup 10 > $O/bench_c5_trace.json 2> $O/bench_c5_trace.err
$B --config 5 --steps 200 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err
$B > $O/bench_a.json 2> $O/bench_a.err
$B > $O/bench_a2.json 2> $O/bench_a2.err
tail -3 $O/pytest.log
grep -A20 "lvk_ekf trace" $O/bench_c5_trace.err | head -22
for f in bench_c5 bench_a bench_a2; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print('$f', d['value'], d['config']['backend']['gated_in'], d['config']['timed_region'], 'be', d.get('back_end_ms_per_message'), 'fe', d.get('front_end_ms_per_frame'), 'wait', d.get('caller_wait_ms_per_frame'), 'idle', d.get('worker_idle_ms_per_message'))
except Exception as e:
    print('$f', 'ERR', open('$O/$f.err').read()[-300:].replace(chr(10),' | '))
PY
done
