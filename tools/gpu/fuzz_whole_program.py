#!/usr/bin/env python3
"""Fuzz of the PRODUCT against the reference's whole program (not part of the suite): random configurations of camera model, window,
augmentation grid, tracker budget, publish rate and the FEJ / td / extrinsics / ZUPT / IMU-intrinsics switches, each written as an ASL
directory and run through the reference's main() twice - with the reference's own classes (oracle/_ref/larvio_ref_full, CPU) and on the
product (oracle/_ref/larvio_ref_main over adapter/ + liblvk_hip.so, this GPU).  Per case: the number of poses, the largest position and
rotation difference, the driver's count of stable map points on both sides, and which initialiser fired (the moving-start initialiser's
minimisers are stand-ins on the reference side: 1e-3 m is what can be asked there, 1e-6 m after a static start).
usage: tools/gpu/fuzz_whole_program.py <first seed> <count> [wide] [sizes] [params]"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))

WIDE = False
SIZES = False
PARAMS = False


def draw(k):
    """the configuration of case k: camera, number of frames, front-end options, filter options, first frame (70 = a start in motion)"""
    from larvio_amd import synthetic as S
    rng = np.random.default_rng([k, 909])
    fish = bool(rng.integers(0, 2))
    cam = dict(S.CAM_TUMVI_LIKE if fish else S.EUROC)
    n = int(rng.integers(220, 340))
    budget = int(rng.integers(100, 300 if fish else 260))
    fo = dict(max_features_num=budget, min_distance=int(rng.integers(10, 26)), pub_frequency=int(rng.choice([10, 10, 20])),
              pyramid_levels=int(rng.choice([2, 2, 3])), max_iteration=int(rng.choice([30, 30, 10])))
    bo = dict(sw_size=int(rng.integers(8, 31)), if_fej=int(rng.integers(0, 2)), estimate_td=int(rng.integers(0, 2)), estimate_extrin=int(rng.integers(0, 2)),
              if_zupt_valid=int(rng.integers(0, 2)), calib_imu_instrinsic=int(rng.random() < 0.25),
              aug_grid_rows=int(rng.integers(2, 7)), aug_grid_cols=int(rng.integers(2, 7)), max_features_in_one_grid=int(rng.integers(1, 4)),
              max_track_len=int(rng.choice([6, 6, 8, 10])), pub_frequency=fo["pub_frequency"])
    first = 70 if rng.random() < 0.2 else 0                  # one case in five starts in the moving part: the moving-start initialiser has to fire
    if WIDE:                                                    # second profile: the code paths the first one never takes - LK windows 15 / 31 (the
        fo["patch_size"] = int(rng.choice([15, 21, 31, 31]))    # generic kernel), budgets beyond 600 tracks (the two-wavefront LK kernel, the 1024-thread
        fo["max_features_num"] = budget = int(rng.integers(100, 1000))   # commit), no CLAHE, a short descriptor gate; drawn AFTER the first profile's
        fo["flag_equalize"] = int(rng.random() < 0.7)           # numbers, so a case number means the same sequence in both
        fo["min_distance"] = int(rng.integers(6, 20))
    if SIZES:                                                   # third profile: image sizes nobody chose - odd widths and heights (CLAHE tiles that do not
        w = int(rng.integers(300, 1000)); h = int(rng.integers(220, 700))      # divide the image, byte paths of the image kernels, pyramid levels of odd size), the
        sc = w / cam["width"]                                   # camera model scaled with the width, principal point off centre
        fx, fy, cx, cy = cam["intrinsics"]
        cam["width"] = w; cam["height"] = h
        cam["intrinsics"] = (fx * sc, fy * sc, w * (0.5 + 0.04 * (rng.random() - 0.5)), h * (0.5 + 0.04 * (rng.random() - 0.5)))
        n = min(n, 260)
    if PARAMS:                                                  # fourth profile: the filter's numbers - noises, the pruning and motion thresholds, track lengths, the
        bo.update(noise_feature=float(rng.uniform(0.002, 0.03)), noise_gyro=0.004 * float(rng.uniform(0.3, 3)), noise_acc=0.08 * float(rng.uniform(0.3, 3)),   # window up to 45 clones
                  rotation_threshold=float(rng.uniform(0.05, 0.5)), translation_threshold=float(rng.uniform(0.1, 1.0)), tracking_rate_threshold=float(rng.uniform(0.3, 0.8)),
                  least_observation_number=int(rng.integers(2, 5)), max_track_len=int(rng.integers(4, 16)), sw_size=int(rng.integers(6, 46)),
                  td=float(rng.uniform(-0.01, 0.01)) if bo["estimate_td"] else 0.0, static_duration=float(rng.choice([0.5, 1.0, 1.5])),
                  feature_translation_threshold=float(rng.choice([-1.0, 0.05, 0.2])), zupt_max_feature_dis=float(rng.uniform(5e-4, 5e-3)))
    return cam, n, fo, bo, first


def one(k):
    from make_euroc_dir import write_euroc_dir
    from larvio_amd import synthetic as S
    from tests.conftest import synth_frames
    from tests import test_gpu_zzz_ref_main as T
    cam, n, fo, bo, first = draw(k)
    fish = cam["distortion_model"] == 1; budget = fo["max_features_num"]
    fcfg = S.frontend_config(cam=cam, **fo); bcfg = S.backend_config(cam=cam, **bo)
    frames = synth_frames(first, n, cam=cam)
    seq = S.imu_only_sequence(cam=cam)
    ts = [f[0] for f in frames]
    imu_all = seq.imu_array(max(int(ts[0] * 200) - 4, 0), int(ts[-1] * 200) + 40)
    keep = os.environ.get("LVK_FUZZ_KEEP")                     # a directory: write case k there and stop (for a rerun by hand)
    d = keep if keep else tempfile.mkdtemp(prefix="lv", dir="/tmp")
    try:
        os.makedirs(os.path.join(d, "logs"))
        write_euroc_dir(d, frames, imu_all, fcfg, bcfg, output_dir=os.path.join(d, "logs") + "/")
        args = [d + "/mav0/imu0/data.csv", d + "/mav0/cam0/data.csv", d + "/mav0/cam0/data", d + "/config.yaml"]
        if keep:
            return "written: " + " ".join(args), 0.0
        env = {kk: v for kk, v in os.environ.items() if kk != "LVK_GRID_REFERENCE"}
        poses = os.path.join(d, "p.txt")
        rf = subprocess.run([T.FULL] + args, capture_output=True, text=True, timeout=900, env=dict(env, LVREF_MAIN_POSES=poses + ".full"))
        if os.environ.get("LVK_FUZZ_REF_ONLY"):
            return "reference only: exit %d, %d poses | %s" % (rf.returncode, len(open(poses + ".full").readlines()) if os.path.exists(poses + ".full") else -1, rf.stdout[-300:].replace("\n", " ")), 0.0
        rm = subprocess.run([T.BIN] + args, capture_output=True, text=True, timeout=600, env=dict(env, LVREF_MAIN_POSES=poses))
        tag = "case %3d%s %s %3d frames budget %3d md %2d lv %d it %2d sw %2d grid %dx%dx%d pub %2d fej %d td %d ex %d zupt %d calib %d" % (
            k, ((" wide patch %d clahe %d" % (fo["patch_size"], fo["flag_equalize"])) if WIDE else "") + ((" %dx%d" % (cam["width"], cam["height"])) if SIZES else "") + ((" params sigma %.3f len %d..%d thr %.2f/%.2f/%.2f" % (bo["noise_feature"], bo["least_observation_number"], bo["max_track_len"], bo["rotation_threshold"], bo["translation_threshold"], bo["tracking_rate_threshold"])) if PARAMS else ""), "fisheye" if fish else "radtan ", n, budget, fo["min_distance"], fo["pyramid_levels"], fo["max_iteration"], bo["sw_size"], bo["aug_grid_rows"], bo["aug_grid_cols"], bo["max_features_in_one_grid"], fo["pub_frequency"],
            bo["if_fej"], bo["estimate_td"], bo["estimate_extrin"], bo["if_zupt_valid"], bo["calib_imu_instrinsic"])
        if rf.returncode != 0 or rm.returncode != 0:
            return tag + "  EXIT CODES reference %d product %d | %s" % (rf.returncode, rm.returncode, (rm.stdout + rm.stderr)[-200:].replace("\n", " ")), None
        Mf = np.loadtxt(poses + ".full", ndmin=2) if os.path.exists(poses + ".full") and os.path.getsize(poses + ".full") else np.zeros((0, 16))
        M = np.loadtxt(poses, ndmin=2) if os.path.exists(poses) and os.path.getsize(poses) else np.zeros((0, 16))
        dyn = "Dynamic initialization success" in rf.stdout
        nf = int(rf.stdout.split("Totally")[1].split()[0]) if "Totally" in rf.stdout else -1
        nm = int(rm.stdout.split("Totally")[1].split()[0]) if "Totally" in rm.stdout else -1
        if M.shape != Mf.shape:
            # which end do they share?  (a product that stops early shares the FIRST poses, a different first message the LAST ones)
            m = min(len(M), len(Mf)); head = tail = float("nan")
            if m:
                head = float(np.linalg.norm(M[:m, 12:15] - Mf[:m, 12:15], axis=1).max()); tail = float(np.linalg.norm(M[-m:, 12:15] - Mf[-m:, 12:15], axis=1).max())
            err = [l for l in (rm.stdout + rm.stderr).splitlines() if "LarVio::" in l or "ImageProcessor::" in l]
            return tag + "  POSE COUNTS reference %d product %d (%s start); aligned at the first pose %.1e m, at the last %.1e m%s" % (
                len(Mf), len(M), "moving" if dyn else "static", head, tail, (" | product said: " + err[0][:160]) if err else ""), None
        if len(M) == 0:
            return tag + "  no pose on either side (never initialised)", 0.0
        dp = float(np.linalg.norm(M[:, 12:15] - Mf[:, 12:15], axis=1).max()); dR = float(np.abs(M[:, :12] - Mf[:, :12]).max())
        # the REFERENCE's own filter jumping by more than 1 m / 0.5 m/s in one update (larvio.cpp's "Update change is too large"): a run that is
        # diverging on its own amplifies the last digits by orders of magnitude per second - listed, not counted on either side
        wild = rf.stdout.count("Update change is too large")
        if wild:
            d = np.linalg.norm(M[:, 12:15] - Mf[:, 12:15], axis=1); first = int(np.argmax(d > 1e-6)) if (d > 1e-6).any() else len(d)
            return tag + "  %3d poses  REFERENCE DIVERGING (%d updates above 1 m / 0.5 m/s; %.0f m travelled): agreement %.1e m over the first %d poses, %.1e m at the end  %s start" % (
                len(M), wild, float(np.linalg.norm(np.diff(Mf[:, 12:15], axis=0), axis=1).sum()), float(d[:max(first, 1)].max()), first, float(d[-1]), "moving" if dyn else "static"), -1.0
        ok = dp < (1e-3 if dyn else 1e-6) and nf == nm
        return tag + "  %3d poses  position %.2e m  rotation %.2e  map points %d / %d  %s start%s" % (len(M), dp, dR, nf, nm, "moving" if dyn else "static", "" if ok else "  <-- DIFFERS"), (dp if ok else None)
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)


def main():
    global WIDE, SIZES, PARAMS
    first, count = int(sys.argv[1]), int(sys.argv[2]); WIDE = "wide" in sys.argv[3:]; SIZES = "sizes" in sys.argv[3:]; PARAMS = "params" in sys.argv[3:]
    bad = 0; wild = 0; worst_static = 0.0
    for k in range(first, first + count):
        line, dp = one(k)
        print(line, flush=True)
        if dp is None:
            bad += 1
        elif dp < 0:
            wild += 1
        elif "static" in line:
            worst_static = max(worst_static, dp)
    print("%d cases, %d differ, %d where the reference's own filter diverges; worst position difference after a static start %.2e m" % (count, bad, wild, worst_static))


if __name__ == "__main__":
    main()
