#!/bin/bash
# round 2, GPU call U: wave-parallel delayed-initialisation kernels (one launch fewer); tests + config A x3 + trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2u; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="timeout 900 python bench.py --no-cpu-baseline --no-shard-probe"
$B > $O/bench_a.json 2> $O/bench_a.err
$B > $O/bench_a2.json 2> $O/bench_a2.err
$B > $O/bench_a3.json 2> $O/bench_a3.err
LVK_EKF_TRACE=1 $B --no-device-pass > $O/bench_a_trace.json 2> $O/bench_a_trace.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_a.log 2>&1
for db in $(find $O/prof_a -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/a_kernel_stats.csv; done
find $O -name "*.db" -size +20M -delete
tail -4 $O/pytest.log
for f in bench_a bench_a2 bench_a3; do python - <<PY
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], (d.get('device_resident') or {}).get('value'), d['config']['backend']['gated_in'], d['config']['timed_region'], 'be', d.get('back_end_ms_per_message'), 'fe', d.get('front_end_ms_per_frame'))
PY
done
grep -A20 "lvk_ekf trace" $O/bench_a_trace.err | head -22; grep "append\|dx_new" $O/a_kernel_stats.csv
