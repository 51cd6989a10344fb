#!/bin/bash
# Same-box A/B of library variants: tools/gpu/ab.sh <tag> <rounds> "<bench args>" <lib1> <lib2> ...   (libs under variants/, selected through LVK_LIB)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; R=$2; ARGS=$3; shift 3
O=gpurun_out/$TAG; mkdir -p $O
for r in $(seq 1 $R); do
  for lib in "$@"; do
    n=$(basename $lib .so)
    LVK_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python bench.py $ARGS > $O/${n}_r$r.json 2> $O/${n}_r$r.err
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
