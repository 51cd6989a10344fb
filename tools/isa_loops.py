#!/usr/bin/env python3
"""Static picture of a kernel's loops from the compiler's assembly: for every back edge, the instructions between its target and
its branch, by class.  usage:
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -S -o fe.s larvio_amd/csrc/frontend.hip
  tools/isa_loops.py fe.s k_fe_lk_bothILi21E [--dump <first label> <last label>]
(what the LK iteration costs in issue slots, next to what the event-bracketed launch time says it costs in microseconds)"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    blocks, cur = [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), []]; blocks.append(cur)
        elif cur is not None and l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    lab = {b[0]: k for k, b in enumerate(blocks)}

    def cls(ins):
        op = ins.split()[0]
        if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("scratch_load"): return "vmem load"
        if op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("scratch_store"): return "vmem store/atomic"
        if op.startswith("ds_"): return "lds"
        if "dpp" in ins or "readlane" in op or "readfirstlane" in op or "permlane" in op: return "cross-lane (dpp/readlane)"
        if op == "s_waitcnt": return "s_waitcnt"
        if op.startswith("s_barrier"): return "s_barrier"
        if op.startswith("v_"): return "valu"
        if op.startswith("s_"): return "salu"
        return "other"
    loops = set()
    for k, b in enumerate(blocks):
        for ins in b[1]:
            m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ins)
            if m and m.group(1) in lab and lab[m.group(1)] <= k:
                loops.add((lab[m.group(1)], k))
    print("%s: %d instructions, %d basic blocks, %d back edges" % (lines[start].split(":")[0][:60], sum(len(b[1]) for b in blocks), len(blocks), len(loops)))
    for lo, hi in sorted(loops):
        ins = [i for b in blocks[lo:hi + 1] for i in b[1]]
        cat = {}
        for i in ins:
            cat[cls(i)] = cat.get(cls(i), 0) + 1
        print("  %-10s .. %-10s %5d instructions  %s" % (blocks[lo][0], blocks[hi][0], len(ins), ", ".join("%s %d" % kv for kv in sorted(cat.items(), key=lambda kv: -kv[1]))))
    if "--dump" in sys.argv:
        a, b = sys.argv[sys.argv.index("--dump") + 1], sys.argv[sys.argv.index("--dump") + 2]
        print("\n---- %s .. %s" % (a, b))
        for blk in blocks[lab[a]:lab[b] + 1]:
            print(blk[0] + ":")
            for i in blk[1]:
                print("\t" + i)


if __name__ == "__main__":
    main()
