#!/bin/bash
# Same-box A/B of variants, alternating:  tools/gpu/ab.sh <tag> <rounds> "<bench args>" <label>:<VAR=VAL>[,<VAR=VAL>...] ...
# (a library built elsewhere is selected with LVK_LIB=variants/x.so; "-" = no variable)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; R=$2; ARGS=$3; shift 3
O=gpurun_out/$TAG; mkdir -p $O
for r in $(seq 1 $R); do
  for spec in "$@"; do
    n=${spec%%:*}; envs=${spec#*:}
    ( IFS=','; for kv in $envs; do [ "$kv" != "-" ] && export "$kv"; done; unset IFS
      timeout 600 python ${BENCH_PY:-bench.py} $ARGS > $O/${n}_r$r.json 2> $O/${n}_r$r.err )
    python - "$O/${n}_r$r.json" "$n r$r" <<'P'
import json, sys
d = None
for l in reversed(open(sys.argv[1]).read().strip().splitlines()):
    if l.startswith("{"):
        d = json.loads(l); break
if d is None:
    print(sys.argv[2], "no line")
else:
    print(sys.argv[2], d["value"], "fe", d.get("front_end_ms_per_frame"), "be", d.get("back_end_ms_per_message"), "p50msg", d.get("p50_ms_frame_with_message"), "lk_us", d["roofline"]["avg_launch_us"],
          "adapter", (d.get("adapter_path") or {}).get("value"))
P
  done
done
