#!/bin/bash
# round 2, GPU call V: one image-stage event per pipelined frame, no per-slot staging events
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2v; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="timeout 900 python bench.py --no-cpu-baseline --no-shard-probe"
$B > $O/bench_a.json 2> $O/bench_a.err
$B > $O/bench_a2.json 2> $O/bench_a2.err
$B > $O/bench_a3.json 2> $O/bench_a3.err
$B --config 5 --steps 200 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_a.log 2>&1
for db in $(find $O/prof_a -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/a_kernel_stats.csv; python tools/queue_gaps.py $db > $O/a_queue_gaps.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
tail -4 $O/pytest.log
for f in bench_a bench_a2 bench_a3 bench_c5; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print('$f', d['value'], (d.get('device_resident') or {}).get('value'), d['config']['backend']['gated_in'], d['config']['timed_region'], 'be', d.get('back_end_ms_per_message'), 'fe', d.get('front_end_ms_per_frame'), 'wait', d.get('caller_wait_ms_per_frame'), 'idle', d.get('worker_idle_ms_per_message'))
except Exception as e:
    print('$f', 'ERR', open('$O/$f.err').read()[-300:].replace(chr(10),' | '))
PY
done
head -12 $O/a_queue_gaps.txt | cut -c1-160
