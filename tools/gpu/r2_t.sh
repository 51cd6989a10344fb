#!/bin/bash
# round 2, GPU call T: the reference set of measurements for the committed profiles (all configurations), final code of the round
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q --durations=6 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc $?" >> $O/bench_driver.err
timeout 900 python bench.py --sequential --no-cpu-baseline > $O/bench_a_seq.json 2> $O/bench_a_seq.err
timeout 900 python bench.py --config 3 --steps 300 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "rc $?" >> $O/bench_c3.err
timeout 900 python bench.py --config 4 --steps 300 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "rc $?" >> $O/bench_c4.err
timeout 1500 python bench.py --config 5 --steps 200 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
timeout 900 python bench.py --backend-only --steps 100 --warmup 6 > $O/bench_be.json 2> $O/bench_be.err; echo "rc $?" >> $O/bench_be.err
LVK_EKF_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/bench_a_trace.json 2> $O/bench_a_trace.err
LVK_EKF_TRACE=1 timeout 900 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --no-device-pass > $O/bench_c5_trace.json 2> $O/bench_c5_trace.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_a.log 2>&1
for db in $(find $O/prof_a -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/a_kernel_stats.csv; python tools/queue_gaps.py $db > $O/a_queue_gaps.txt 2>&1; done
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/pmc_a.log 2>&1
for c in $(find $O/pmc_a -name "*counter_collection.csv" | head -1); do python tools/pmc_summary.py $c FETCH_SIZE > $O/a_pmc_fetch_size.csv; done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -- python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --no-device-pass > $O/prof_c5.log 2>&1
for db in $(find $O/prof_c5 -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/c5_kernel_stats.csv; python tools/queue_gaps.py $db > $O/c5_queue_gaps.txt 2>&1; done
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c5 -- python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --no-device-pass > $O/pmc_c5.log 2>&1
for c in $(find $O/pmc_c5 -name "*counter_collection.csv" | head -1); do python tools/pmc_summary.py $c FETCH_SIZE > $O/c5_pmc_fetch_size.csv; done
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_c5 -- python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --no-device-pass > $O/pmcw_c5.log 2>&1
for c in $(find $O/pmcw_c5 -name "*counter_collection.csv" | head -1); do python tools/pmc_summary.py $c WRITE_SIZE > $O/c5_pmc_write_size.csv; done
find $O -name "*.db" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +10M -delete; find $O -name "*counter_collection.csv" -size +20M -delete
tail -3 $O/pytest.log; tail -2 $O/smoke.log; for f in a driver a_seq c3 c4 c5 be; do cut -c1-160 $O/bench_$f.json; done
