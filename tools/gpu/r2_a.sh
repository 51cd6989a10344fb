#!/bin/bash
# round 2, GPU call A: full GPU test suite, driver-style + default bench lines, zero-copy A/B, calibration microbenchmarks, rocprof passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
nproc > $O/nproc.txt; lscpu | head -20 >> $O/nproc.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc $?" >> $O/bench_driver.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
LVK_FE_ZEROCOPY=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_zerocopy.json 2> $O/bench_zerocopy.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default2.json 2> $O/bench_default2.err
timeout 600 python bench.py --no-cpu-baseline --sequential > $O/bench_sequential.json 2> $O/bench_sequential.err
timeout 300 tools/lvk_microbench all > $O/microbench.jsonl 2> $O/microbench.err
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_micro -- $GRAFT_REPO_ROOT/tools/lvk_microbench stream16 > $GRAFT_REPO_ROOT/$O/pmc_micro_s.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_micro -- $GRAFT_REPO_ROOT/tools/lvk_microbench rows24 > $GRAFT_REPO_ROOT/$O/pmc_micro_r.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_micro -- $GRAFT_REPO_ROOT/tools/lvk_microbench gather24 > $GRAFT_REPO_ROOT/$O/pmc_micro_g.log 2>&1
cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --no-cpu-baseline --no-device-pass > $O/prof.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python bench.py --no-cpu-baseline --no-device-pass > $O/pmc.log 2>&1
for db in $(find $O/prof -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/kernel_stats.csv; python tools/queue_gaps.py $db > $O/queue_gaps.txt 2>&1; done
for c in $(find $O/pmc -name "*counter_collection.csv" | head -1); do python tools/pmc_summary.py $c FETCH_SIZE > $O/pmc_fetch_size.csv; done
for c in $(find $O/pmc_micro -name "*counter_collection.csv"); do python tools/pmc_summary.py $c FETCH_SIZE; done > $O/pmc_micro_fetch_size.csv
# keep the merge small: drop raw traces
find $O -name "*.db" -size +20M -delete; find $O/pmc $O/pmc_micro -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
tail -3 $O/pytest.log; cut -c1-400 $O/bench_driver.json; cut -c1-300 $O/bench_default.json; cat $O/microbench.jsonl | tail -3
