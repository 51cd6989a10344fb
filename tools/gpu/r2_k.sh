#!/bin/bash
# round 2, GPU call K: fused Cholesky + solve (one launch, flag hand-off), host IMU composition vectorised
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2k; mkdir -p $O
timeout 180 python -m pytest tests/test_gpu_backend.py -m gpu -q -x -k "update_matches or ekf_update" > $O/pytest_first.log 2>&1; echo "first rc $?" >> $O/pytest_first.log
if ! grep -q "first rc 0" $O/pytest_first.log; then tail -30 $O/pytest_first.log; exit 1; fi
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
LVK_CHOL_FUSED=0 timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a_nofuse.json 2> $O/bench_a_nofuse.err
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a2.json 2> $O/bench_a2.err
LVK_EKF_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-shard-probe --no-device-pass > $O/bench_a_trace.json 2> $O/bench_a_trace.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shard-probe > $O/bench_driver.json 2> $O/bench_driver.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_a.log 2>&1
for db in $(find $O/prof_a -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/a_kernel_stats.csv; python tools/queue_gaps.py $db > $O/a_queue_gaps.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
tail -4 $O/pytest.log; cut -c1-160 $O/bench_a.json; cut -c1-160 $O/bench_a_nofuse.json; cut -c1-160 $O/bench_a2.json; cut -c1-160 $O/bench_driver.json; grep -A20 "lvk_ekf trace" $O/bench_a_trace.err; head -12 $O/a_kernel_stats.csv
