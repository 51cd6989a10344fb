#!/bin/bash
# round 2, GPU call I: lazy message fetch + batched CAQR staging: tests, config A (+ filter phase trace, kernel table), CAQR timing, config 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a2.json 2> $O/bench_a2.err
LVK_EKF_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-shard-probe --no-device-pass > $O/bench_a_trace.json 2> $O/bench_a_trace.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shard-probe > $O/bench_driver.json 2> $O/bench_driver.err
timeout 1500 python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_a.log 2>&1
for db in $(find $O/prof_a -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/a_kernel_stats.csv; python tools/queue_gaps.py $db > $O/a_queue_gaps.txt 2>&1; done
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_qr -- python $GRAFT_REPO_ROOT/tools/gpu/qr_probe.py > $GRAFT_REPO_ROOT/$O/prof_qr.log 2>&1; cd $GRAFT_REPO_ROOT
for db in $(find $O/prof_qr -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/qr_kernel_stats.csv; done
find $O -name "*.db" -size +20M -delete
tail -3 $O/pytest.log; cut -c1-160 $O/bench_a.json; cut -c1-160 $O/bench_driver.json; cut -c1-160 $O/bench_c5.json; cat $O/qr_kernel_stats.csv
