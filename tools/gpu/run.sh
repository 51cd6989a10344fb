#!/bin/bash
# One parameterised GPU-box script (replaces the per-call r2_*.sh files):  tools/gpu/run.sh <tag> <step> [<step> ...]
#   steps: smoke | tests | tests:<pytest args> | bench:<name>:<bench.py args> | prof:<name>:<bench.py args> | pmc:<name>:<bench.py args>
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun); summaries to keep are copied into profiles/ by hand.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$O"
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}
  case $kind in
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc $?"; tail -1 "$O/smoke.log" ;;
    tests) a=""; [ "$rest" != "$step" ] && a=$rest
           timeout 1500 python -m pytest ${a:-tests} -m gpu -q -x -s > "$O/pytest.log" 2>&1; echo "pytest rc $?" | tee -a "$O/pytest.log"; grep -E "passed|failed|error" "$O/pytest.log" | tail -3 ;;
    bench) name=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
           timeout 900 python bench.py $args > "$O/bench_$name.json" 2> "$O/bench_$name.err"; echo "bench $name rc $?"; head -c 600 "$O/bench_$name.json"; echo ;;
    prof)  name=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
           timeout 900 rocprofv3 --kernel-trace --stats -d "$O/prof_$name" -- python bench.py $args > "$O/prof_$name.json" 2> "$O/prof_$name.err"; echo "prof $name rc $?"
           for db in $(find "$O/prof_$name" -name "*.db" | head -1); do
             python tools/prof_summary.py "$db" > "$O/${name}_kernel_stats.csv" 2>> "$O/prof_$name.err"
             python tools/queue_gaps.py "$db" > "$O/${name}_queue_gaps.txt" 2>> "$O/prof_$name.err"
           done
           find "$O" -name "*.db" -size +20M -delete; head -14 "$O/${name}_kernel_stats.csv" ;;
    pmc)   name=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
           for c in FETCH_SIZE WRITE_SIZE; do
             timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_${name}_$c" -- python bench.py $args > /dev/null 2> "$O/pmc_${name}_$c.err"; echo "pmc $name $c rc $?"
             for f in $(find "$O/pmc_${name}_$c" -name "*counter_collection.csv" | head -1); do
               python tools/pmc_summary.py "$f" $c > "$O/${name}_pmc_$(echo $c | tr A-Z a-z).csv" 2>> "$O/pmc_${name}_$c.err"
             done
             find "$O/pmc_${name}_$c" -name "*.csv" -size +20M -delete
           done ;;
    sh)    bash -c "$rest" > "$O/sh.log" 2>&1; echo "sh rc $?"; tail -5 "$O/sh.log" ;;
  esac
done
