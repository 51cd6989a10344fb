"""Diagnostic: config-5 workload, sequential driver vs pipelined driver drained after every frame; reports the first frame where the
front-end's track table or the filter state differ."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
from larvio_amd import synthetic as S

cfg = sys.argv[1] if len(sys.argv) > 1 else "5"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 160
drain_every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
wl = S.workload(cfg, None, None)
first = int(2.0 * wl["img_rate"])
ts, frames = S.render_frames(first, N + 2, cam=wl["cam"], seed=S.MASTER_SEED, img_rate=wl["img_rate"], procs=16)
seq = S.imu_only_sequence(S.MASTER_SEED, cam=wl["cam"])
k_lo = max(int(ts[0] * 200) - 4, 0)
imu_all = seq.imu_array(k_lo, int(ts[-1] * 200) + 40)

class A: pass
def run(sequential):
    r = bench.Run(wl, A(), 0, imu_all, seq, ts, sequential)
    out = []
    for i in range(N):
        has = r.step(host_img=frames[i])
        if (i + 1) % drain_every == 0 or sequential:
            r.drain()
            tr = r.fe.tracks()
            c = r.be.counters()
            sd = r.be.state(); st = np.concatenate([np.atleast_1d(np.asarray(sd[k], float)).ravel() for k in ("t", "q", "v", "p", "bg", "ba", "t_c_b", "td")])
            out.append((i, bool(has), tr["ids"].copy(), tr["pts"].copy(), dict(c), len(r.be.clones()), st))
    r.close()
    return out

hf = os.environ.get("LVK_MSG_HASH")
def grab():
    if not hf or not os.path.exists(hf): return []
    l = open(hf).read().splitlines(); os.remove(hf); return l
grab()
import subprocess
def sub(mode):
    # the log is written when the library unloads: one child process per mode
    subprocess.run([sys.executable, os.path.abspath(__file__), cfg, str(N), str(drain_every), mode], check=False)
    return grab()
if len(sys.argv) > 4:
    run(sys.argv[4] == "seq"); sys.exit(0)
if hf:
    ha = sub("seq"); hb = sub("pipe")
    nd = 0
    for k, (x, y) in enumerate(zip(ha, hb)):
        if x != y:
            xs, ys = x.split(" state"), y.split(" state")
            print("update", k, "inputs", "SAME" if xs[0] == ys[0] else "DIFFER", "\n   seq ", xs[0], "\n   pipe", ys[0])
            if xs[0] == ys[0]:
                da = np.array(xs[1].split(), float); db = np.array(ys[1].split(), float); print("   max state diff", np.abs(da - db).max())
            nd += 1
            if nd > 3: break
    print("updates compared", min(len(ha), len(hb)), "differing", nd, "(counts", len(ha), len(hb), ")")
    sys.exit(0)
a = run(True); ha = []
b = run(False); hb = []
if hf:
    nd = 0
    for k, (x, y) in enumerate(zip(ha, hb)):
        if x != y:
            print("message", k, "differs:\n   seq ", x, "\n   pipe", y); nd += 1
            if nd > 3: break
    print("messages compared", min(len(ha), len(hb)), "differing", nd, "(counts", len(ha), len(hb), ")")
ia = {x[0]: x for x in a}
bad = 0
for x in b:
    y = ia[x[0]]
    same_ids = np.array_equal(x[2], y[2]); same_pts = np.array_equal(x[3], y[3])
    same_c = x[4] == y[4] and x[5] == y[5]
    same_s = (x[6] is None) or np.array_equal(x[6], y[6])
    if not (same_ids and same_pts and same_c and same_s):
        print("frame", x[0], "has", x[1], y[1], "ids", same_ids, "pts", same_pts, "counters", same_c, "state", same_s, "| n tracks", len(x[2]), len(y[2]), "| clones", x[5], y[5])
        if not same_c: print("   pipe", x[4]); print("   seq ", y[4])
        if not same_s:
            names = ["t"] + ["q"] * 4 + ["v"] * 3 + ["p"] * 3 + ["bg"] * 3 + ["ba"] * 3 + ["tcb"] * 3 + ["td"]
            d = np.abs(x[6] - y[6]); print("   state diff:", {n: float(d[[k for k, m in enumerate(names) if m == n]].max()) for n in dict.fromkeys(names)})
        bad += 1
        if bad > 6: break
print("compared", len(b), "checkpoints; mismatching:", bad)
