#!/bin/bash
# round 2, GPU call S: after the two race fixes: full tests, config A x3 (+trace, driver-style, sequential), config 5 x2, config 3/4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="timeout 900 python bench.py --no-cpu-baseline --no-shard-probe"
$B > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
$B > $O/bench_a2.json 2> $O/bench_a2.err
$B > $O/bench_a3.json 2> $O/bench_a3.err
$B --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
$B --config 5 --steps 200 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
$B --config 5 --steps 200 --warmup 10 > $O/bench_c5b.json 2> $O/bench_c5b.err
$B --config 3 > $O/bench_c3.json 2> $O/bench_c3.err
$B --config 4 > $O/bench_c4.json 2> $O/bench_c4.err
tail -4 $O/pytest.log
for f in bench_a bench_a2 bench_a3 bench_driver bench_c5 bench_c5b bench_c3 bench_c4; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print('$f', d['value'], (d.get('device_resident') or {}).get('value'), d['config']['backend'], d['config']['timed_region'], 'be', d.get('back_end_ms_per_message'), 'fe', d.get('front_end_ms_per_frame'))
except Exception as e:
    print('$f', 'ERR', open('$O/$f.err').read()[-300:].replace(chr(10),' | '))
PY
done
