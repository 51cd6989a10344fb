#!/bin/bash
# round 2, GPU call L: split CLAHE histogram, word-path mosaic blur: full tests, config A + config 5 benches and kernel tables
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_frontend_stages.py -m gpu -q -x > $O/pytest_first.log 2>&1; echo "first rc $?" >> $O/pytest_first.log
if ! grep -q "first rc 0" $O/pytest_first.log; then tail -40 $O/pytest_first.log; exit 1; fi
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
timeout 1500 python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline --no-shard-probe > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -- python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_c5.log 2>&1
for db in $(find $O/prof_c5 -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/c5_kernel_stats.csv; python tools/queue_gaps.py $db > $O/c5_queue_gaps.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
tail -4 $O/pytest.log; cut -c1-160 $O/bench_a.json; cut -c1-160 $O/bench_c5.json; head -32 $O/c5_kernel_stats.csv
