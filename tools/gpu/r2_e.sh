#!/bin/bash
# round 2, GPU call D: Gram+Cholesky compression (tests, A/B at config A, config-5 depth), adapter test, sharded test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_backend.py -m gpu -x -q -s -k "structure_aware or two_ranks or config5_depth or hybrid" ) > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
LVK_SPARSE_QR=0 timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a_nosparse.json 2> $O/bench_a_nosparse.err
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a2.json 2> $O/bench_a2.err
timeout 900 python bench.py --backend-only --steps 60 --warmup 6 > $O/bench_be.json 2> $O/bench_be.err; echo "rc $?" >> $O/bench_be.err
timeout 1500 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -- python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --no-device-pass > $O/prof_c5.log 2>&1
for db in $(find $O/prof_c5 -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/c5_kernel_stats.csv; done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_a.log 2>&1
for db in $(find $O/prof_a -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/a_kernel_stats.csv; python tools/queue_gaps.py $db > $O/a_queue_gaps.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
tail -6 $O/pytest_new.log; tail -3 $O/pytest.log; cut -c1-200 $O/bench_a.json; cut -c1-300 $O/bench_be.json; cut -c1-200 $O/bench_c5.json
