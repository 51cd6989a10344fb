"""Host-side pieces of the dataset driver (examples/larvio_euroc): the configuration-file reader, the PNG reader, the ASL/EuRoC
CSV readers and the trajectory-error tool.  None of them touches the GPU; the C++ ones are exercised through examples/host_tools."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
TOOL = os.path.join(ROOT, "examples", "host_tools")


@pytest.fixture(scope="module")
def host_tools():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "host_tools"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return TOOL


def _run(*args):
    r = subprocess.run([TOOL, *args], capture_output=True, text=True)
    return r.returncode, r.stdout, r.stderr


def _parse_fields(text):
    out = {}
    for line in text.splitlines():
        k, *v = line.split()
        out[k] = v
    return out


# ------------------------------------------------------------------ configuration file
def test_config_file_fills_both_abi_structs(host_tools, tmp_path):
    from larvio_amd import synthetic as S
    from larvio_amd.image_processor import make_fe_config
    from larvio_amd.larvio import make_ekf_config
    from make_euroc_dir import write_config_yaml
    fe = S.frontend_config(max_features_num=150, pyramid_levels=3)
    be = S.backend_config(sw_size=30, calib_imu_instrinsic=1, td=0.0123)
    path = tmp_path / "cfg.yaml"
    write_config_yaml(str(path), fe, be, output_dir="/tmp/out/")
    rc, out, err = _run("config", str(path))
    assert rc == 0, err
    got = _parse_fields(out)
    assert got["output_dir"] == ["/tmp/out/"]
    a, b = make_fe_config(fe), make_ekf_config(be)
    for name, _ in a._fields_:
        v = getattr(a, name)
        want = list(v) if hasattr(v, "__len__") else [v]
        assert [float(x) for x in got["fe." + name]] == [float(x) for x in want], name
    for name, _ in b._fields_:
        if name == "max_features":
            assert float(got["ekf.max_features"][0]) == 150          # capacity hint = max_features_num
            continue
        v = getattr(b, name)
        want = list(v) if hasattr(v, "__len__") else [v]
        assert [float(x) for x in got["ekf." + name]] == [float(x) for x in want], name


def test_config_file_syntax_variants(host_tools, tmp_path):
    text = """%YAML:1.0
---
# comment line
output_dir: "/a b/#c/"   # a '#' inside quotes is not a comment
distortion_model: 'equidistant'
resolution_width: 512
resolution_height: 512
intrinsics: # nested map
   fx: 1.9e2
   fy: 190.
   cx: 254.93170605935475
   cy: .25e3
distortion_coeffs:
   k1: 0.0034
   k2: 0.00071
   p1: -0.0020
   p2: 2e-4
T_cam_imu: !!opencv-matrix
   rows: 4
   cols: 4
   dt: d
   data: [ -1., 0., 0., 0.1, 0., 0., -1., 0.2,
       0., -1., 0., 0.3,
       0., 0., 0., 1. ]
pyramid_levels: 3
max_features_num: 300.0
feature_idp_dim: 1
"""
    p = tmp_path / "v.yaml"; p.write_text(text)
    rc, out, err = _run("config", str(p))
    assert rc == 0, err
    got = _parse_fields(out)
    assert " ".join(got["output_dir"]) == "/a b/#c/"
    assert float(got["fe.distortion_model"][0]) == 1 and float(got["fe.width"][0]) == 512
    assert [float(x) for x in got["fe.intrinsics"]] == [190.0, 190.0, 254.93170605935475, 250.0]
    assert [float(x) for x in got["fe.distortion"]] == [0.0034, 0.00071, -0.0020, 2e-4]
    T = np.array([float(x) for x in got["ekf.T_cam_imu"]]).reshape(4, 4)
    assert np.array_equal(T, np.array([[-1, 0, 0, 0.1], [0, 0, -1, 0.2], [0, -1, 0, 0.3], [0, 0, 0, 1.0]]))
    R = np.array([float(x) for x in got["fe.R_cam_imu"]]).reshape(3, 3)
    assert np.array_equal(R, T[:3, :3].T)                         # image_processor.cpp:93
    assert float(got["fe.max_features_num"][0]) == 300 and float(got["fe.pyramid_levels"][0]) == 3
    assert float(got["ekf.noise_gyro"][0]) == 0.0                 # a missing key reads as 0, as cv::FileNode does


def test_config_file_refusals(host_tools, tmp_path):
    from larvio_amd import synthetic as S
    from make_euroc_dir import write_config_yaml
    fe, be = S.frontend_config(), S.backend_config()
    p = tmp_path / "c.yaml"
    write_config_yaml(str(p), fe, be)
    good = p.read_text()
    for old, new, msg in (("feature_idp_dim: 1", "feature_idp_dim: 3", "feature_idp_dim"), ("use_schmidt: 0", "use_schmidt: 1", "use_schmidt"),
                          ('distortion_model: "radtan"', 'distortion_model: "fov"', "distortion_model"), ("   rows: 4", "   rows: 3", "T_cam_imu")):
        assert old in good
        p.write_text(good.replace(old, new))
        rc, out, err = _run("config", str(p))
        assert rc == 1 and msg in err, (old, err)
    rc, out, err = _run("config", str(tmp_path / "missing.yaml"))
    assert rc == 1 and "cannot open" in err
    p.write_text(good.replace("   data:\n    [", "   data:\n    ["))   # unchanged: sanity
    assert _run("config", str(p))[0] == 0
    p.write_text(good.replace("]", ""))
    rc, out, err = _run("config", str(p))
    assert rc == 1 and "unterminated" in err


def test_reads_the_reference_configuration_unchanged(host_tools):
    ref = "/root/reference/config/euroc.yaml"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    rc, out, err = _run("config", ref)
    assert rc == 0, err
    got = _parse_fields(out)
    f = lambda k: float(got[k][0])
    assert (f("fe.width"), f("fe.height"), f("fe.pyramid_levels"), f("fe.patch_size"), f("fe.max_features_num")) == (752, 480, 2, 21, 200)
    assert (f("ekf.sw_size"), f("ekf.imu_rate"), f("ekf.noise_gyro_bias"), f("ekf.zupt_noise_q"), f("ekf.static_duration")) == (20, 200, 2e-6, 3.4e-2, 1.0)
    assert [float(x) for x in got["fe.intrinsics"]] == [458.654, 457.296, 367.215, 248.375]
    assert float(got["fe.distortion"][3]) == 1.76187114e-05
    T = np.array([float(x) for x in got["ekf.T_cam_imu"]]).reshape(4, 4)
    assert T[0, 1] == 0.999557249008346 and T[3, 3] == 1.0 and T[2, 3] == -0.008054602460030


# ------------------------------------------------------------------ PNG reader
def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c; pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, ftype_of_row):
    """rows: list of bytes (unfiltered scanlines) -> filtered stream (an independent PNG encoder for the test)"""
    out = bytearray(); prev = bytes(len(rows[0])) if rows else b""
    for y, cur in enumerate(rows):
        ft = ftype_of_row(y); out.append(ft)
        for i in range(len(cur)):
            a = cur[i - bpp] if i >= bpp else 0; b = prev[i]; c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft] if ft < 5 else 0
            out.append((cur[i] - pred) & 255)
        prev = cur
    return bytes(out)


def _encode_png(samples, depth, colour, interlace=False, ftype=lambda y: y % 5, plte=None, idat_split=3):
    """samples: (H, W, C) integer array of raw sample values (C per the colour type)"""
    H, W, C = samples.shape

    def pack_rows(sub):
        rows = []
        for r in sub:
            flat = r.reshape(-1)
            if depth == 16:
                rows.append(b"".join(struct.pack(">H", int(v)) for v in flat))
            elif depth == 8:
                rows.append(bytes(int(v) for v in flat))
            else:
                bits = "".join(format(int(v), f"0{depth}b") for v in flat); bits += "0" * (-len(bits) % 8)
                rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        return rows
    bpp = max(1, depth * C // 8)
    if interlace:
        passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
        stream = b""
        for x0, y0, dx, dy in passes:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                stream += _filter_rows(pack_rows(sub), bpp, ftype)
    else:
        stream = _filter_rows(pack_rows(samples), bpp, ftype)
    z = zlib.compress(stream, 6)
    cut = [len(z) * k // idat_split for k in range(idat_split + 1)]
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, colour, 0, 0, 1 if interlace else 0))
    png += _chunk(b"tEXt", b"Comment\0ancillary chunks are skipped")
    if plte is not None:
        png += _chunk(b"PLTE", bytes(plte))
    for k in range(idat_split):
        png += _chunk(b"IDAT", z[cut[k]:cut[k + 1]])
    return png + _chunk(b"IEND", b"")


def _decode(tmp_path, png_bytes, name="t.png"):
    p = tmp_path / name; p.write_bytes(png_bytes); raw = tmp_path / (name + ".raw")
    rc, out, err = _run("png", str(p), str(raw))
    if rc != 0:
        return None, err
    w, h = [int(x) for x in out.split()]
    return np.frombuffer(raw.read_bytes(), np.uint8).reshape(h, w), ""


def _grey_of_rgb(rgb8):
    r, g, b = [rgb8[..., k].astype(np.int64) for k in range(3)]
    return ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8)


@pytest.mark.parametrize("interlace", [False, True])
def test_png_reader_all_filters_depths_and_colour_types(host_tools, tmp_path, interlace):
    rng = np.random.default_rng(5)
    H, W = 37, 53                                                  # not multiples of 8: ragged Adam7 passes, ragged bit packing
    smooth = (np.add.outer(np.arange(H) * 3, np.arange(W) * 2) % 256)
    for depth in (1, 2, 4, 8, 16):                                 # grey
        hi = (1 << depth) - 1
        s = rng.integers(0, hi + 1, (H, W, 1)) if depth != 8 else ((smooth + rng.integers(0, 9, (H, W))) % 256)[..., None]
        img, err = _decode(tmp_path, _encode_png(s, depth, 0, interlace))
        assert img is not None, err
        want = (s[..., 0] >> 8) if depth == 16 else (s[..., 0] if depth == 8 else s[..., 0] * 255 // hi)
        assert np.array_equal(img, want.astype(np.uint8)), depth
    for depth in (8, 16):                                          # grey+alpha, RGB, RGBA
        hi = (1 << depth) - 1
        ga = rng.integers(0, hi + 1, (H, W, 2))
        img, err = _decode(tmp_path, _encode_png(ga, depth, 4, interlace)); assert img is not None, err
        assert np.array_equal(img, (ga[..., 0] >> (depth - 8)).astype(np.uint8))
        for colour, C in ((2, 3), (6, 4)):
            s = rng.integers(0, hi + 1, (H, W, C))
            img, err = _decode(tmp_path, _encode_png(s, depth, colour, interlace)); assert img is not None, err
            assert np.array_equal(img, _grey_of_rgb((s[..., :3] >> (depth - 8)).astype(np.uint8))), (depth, colour)
    for depth in (1, 2, 4, 8):                                     # palette
        n = 1 << depth if depth < 8 else 200
        pal = rng.integers(0, 256, (n, 3)).astype(np.uint8)
        s = rng.integers(0, n, (H, W, 1))
        img, err = _decode(tmp_path, _encode_png(s, depth, 3, interlace, plte=pal.reshape(-1))); assert img is not None, err
        assert np.array_equal(img, _grey_of_rgb(pal)[s[..., 0]]), depth


def test_png_reader_matches_pillow_on_camera_sized_images(host_tools, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:480, 0:752]
    img8 = ((np.sin(xx / 17.0) * np.cos(yy / 23.0) * 90 + 128) + rng.integers(-6, 7, (480, 752))).clip(0, 255).astype(np.uint8)
    for level in (0, 1, 9):
        p = tmp_path / f"g{level}.png"; Image.fromarray(img8).save(p, compress_level=level)
        got, err = _decode(tmp_path, p.read_bytes(), f"g{level}b.png"); assert got is not None, err
        assert np.array_equal(got, img8)
    img16 = (img8.astype(np.uint16) << 8) | rng.integers(0, 256, img8.shape).astype(np.uint16)     # the TUM-VI files are 16-bit
    p = tmp_path / "g16.png"; Image.fromarray(img16).save(p)
    got, err = _decode(tmp_path, p.read_bytes(), "g16b.png"); assert got is not None, err
    assert np.array_equal(got, img8)
    rgb = rng.integers(0, 256, (120, 160, 3)).astype(np.uint8)
    p = tmp_path / "rgb.png"; Image.fromarray(rgb).save(p)
    got, err = _decode(tmp_path, p.read_bytes(), "rgbb.png"); assert got is not None, err
    assert np.array_equal(got, _grey_of_rgb(rgb))
    assert np.abs(got.astype(int) - np.asarray(Image.fromarray(rgb).convert("L")).astype(int)).max() <= 1


def test_png_reader_rejects_damaged_files(host_tools, tmp_path):
    s = np.arange(12 * 9).reshape(12, 9, 1) % 256
    good = _encode_png(s, 8, 0)
    img, err = _decode(tmp_path, good); assert img is not None
    bad = bytearray(good); bad[60] ^= 0x40                          # flip a bit inside a chunk: CRC
    img, err = _decode(tmp_path, bytes(bad)); assert img is None and "CRC" in err
    img, err = _decode(tmp_path, good[:len(good) // 2]); assert img is None
    img, err = _decode(tmp_path, b"JFIF" + good); assert img is None and "not a PNG" in err
    img, err = _decode(tmp_path, _encode_png(s, 8, 0, ftype=lambda y: 7)); assert img is None and "filter" in err
    short = _encode_png(s[:6], 8, 0)                               # IHDR says 12 rows, data for 6
    short = short.replace(struct.pack(">II", 9, 6), struct.pack(">II", 9, 12), 1)
    ihdr = short[12:12 + 4 + 13]; short = short[:12 + 4 + 13] + struct.pack(">I", zlib.crc32(ihdr) & 0xFFFFFFFF) + short[12 + 4 + 13 + 4:]
    img, err = _decode(tmp_path, short); assert img is None and "zlib" in err


# ------------------------------------------------------------------ CSV readers
def test_asl_csv_readers(host_tools, tmp_path):
    imu = tmp_path / "imu.csv"
    imu.write_bytes(b"#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z\r\n"
                    b"1403636579758555392,-0.099134701513277898,0.14730578886832138,0.02722713633111154,8.1476917083333333,-0.37592158333333331,-2.4026292499999999\r\n"
                    b"1403636579763555584,-1e-3,2.5E-2,3,4.5,-6,7e0\r\n\r\n")
    rc, out, err = _run("imu", str(imu)); assert rc == 0, err
    rows = [[float(x) for x in l.split()] for l in out.splitlines()]
    assert len(rows) == 2                                           # the header and the blank tail are not records
    assert rows[0] == [1e-9 * 1403636579758555392, -0.099134701513277898, 0.14730578886832138, 0.02722713633111154,
                       8.1476917083333333, -0.37592158333333331, -2.4026292499999999]
    assert rows[1] == [1e-9 * 1403636579763555584, -1e-3, 2.5e-2, 3.0, 4.5, -6.0, 7.0]
    cam = tmp_path / "cam.csv"
    cam.write_bytes(b"#timestamp [ns],filename\r\n1403636579763555584,1403636579763555584.png\r\n1403636579813555456,1403636579813555456.png\n")
    rc, out, err = _run("images", str(cam)); assert rc == 0, err
    lines = [l.split() for l in out.splitlines()]
    assert [l[1] for l in lines] == ["1403636579763555584.png", "1403636579813555456.png"]      # CR stripped
    assert [float(l[0]) for l in lines] == [1e-9 * 1403636579763555584, 1e-9 * 1403636579813555456]
    assert _run("imu", str(tmp_path / "nope.csv"))[0] == 1


def test_synthetic_euroc_directory_round_trips_through_the_readers(host_tools, tmp_path):
    from larvio_amd import synthetic as S
    from make_euroc_dir import write_euroc_dir
    seq = S.imu_only_sequence()
    imu = seq.imu_array(0, 50)
    rng = np.random.default_rng(3)
    frames = [(seq.frame_time(i), rng.integers(0, 256, (48, 64)).astype(np.uint8)) for i in range(3)]
    t_img, t_imu = write_euroc_dir(str(tmp_path / "d"), frames, imu, S.frontend_config(), S.backend_config())
    mav = tmp_path / "d" / "mav0"
    rc, out, err = _run("imu", str(mav / "imu0" / "data.csv")); assert rc == 0, err
    rows = np.array([[float(x) for x in l.split()] for l in out.splitlines()])
    assert np.array_equal(rows[:, 0], t_imu) and np.array_equal(rows[:, 1:4], imu["gyro"]) and np.array_equal(rows[:, 4:7], imu["acc"])
    rc, out, err = _run("images", str(mav / "cam0" / "data.csv")); assert rc == 0, err
    for (t, img), line, ts in zip(frames, out.splitlines(), t_img):
        stamp, name = line.split()
        assert float(stamp) == ts and abs(ts - t) < 1e-9
        got, err = _decode(tmp_path, (mav / "cam0" / "data" / name).read_bytes(), "rt.png"); assert got is not None, err
        assert np.array_equal(got, img)
    assert _run("config", str(tmp_path / "d" / "config.yaml"))[0] == 0


# ------------------------------------------------------------------ trajectory error
def test_traj_rmse_tool(tmp_path, capsys):
    import traj_rmse as T
    rng = np.random.default_rng(11)
    t = np.arange(0, 20, 0.005)
    p_gt = np.stack([np.sin(t), np.cos(0.7 * t) * 2, 0.3 * t], 1)
    th = 0.8; Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    te = t[::20][5:-5] + 0.0013
    p_true = np.stack([np.interp(te, t, p_gt[:, k]) for k in range(3)], 1)
    noise = rng.normal(0, 0.01, p_true.shape)
    p_est = (p_true + noise - np.array([1.0, -2.0, 0.5])) @ Rz                  # a yaw + shift away from the truth
    want = np.sqrt((noise ** 2).sum(1).mean())
    for mode in ("se3", "yaw"):
        rmse, n = T.ate_rmse(te, p_est, t, p_gt, mode)
        assert n == len(te) and abs(rmse - want) < 1.5e-3, (mode, rmse, want)
    assert T.ate_rmse(te, p_est, t, p_gt, "none")[0] > 1.0
    # file formats: TUM estimate, EuRoC ground truth (ns stamps), the reference's state log + take-off stamp
    gt = tmp_path / "gt.csv"
    with open(gt, "w") as f:
        f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z []\n")
        for ti, p in zip(t, p_gt):
            f.write(f"{int(round((ti + 1403636579.0) * 1e9))},{float(p[0])!r},{float(p[1])!r},{float(p[2])!r},1,0,0,0\n")
    tum = tmp_path / "est.txt"
    with open(tum, "w") as f:
        for ti, p in zip(te, p_est):
            f.write(f"{ti + 1403636579.0:.9f} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r} 0 0 0 1\n")
    log = tmp_path / "msckf_2_state.txt"
    (tmp_path / "msckf_2_takeoff.txt").write_text(f"{1403636579.0 + te[0]:.9f}\n")
    with open(log, "w") as f:
        for ti, p in zip(te, p_est):
            f.write(" ".join(f"{x:.10g}" for x in [ti - te[0], 1, 0, 0, 0, 0, 0, 0, p[0], p[1], p[2]] + [0] * 13) + "\n")
    assert T.main([str(tum), str(gt), "--ref", str(log)]) == 0
    out = capsys.readouterr().out
    lines = out.splitlines()
    assert abs(float(lines[0].split()[4]) - want) < 1.5e-3 and abs(float(lines[1].split()[4]) - want) < 1.5e-3
    assert abs(float(lines[2].split()[1])) < 0.01                                # the two files hold the same trajectory: < 0.01 mm apart


# ------------------------------------------------------------------ the Python twin of the configuration reader
def test_python_config_reader_agrees_with_the_cpp_reader(host_tools, tmp_path):
    from larvio_amd import synthetic as S
    from larvio_amd.config import load_config, parse
    from larvio_amd.image_processor import make_fe_config
    from larvio_amd.larvio import make_ekf_config
    from make_euroc_dir import write_config_yaml
    paths = []
    p = tmp_path / "a.yaml"
    write_config_yaml(str(p), S.frontend_config(max_features_num=150, pyramid_levels=3), S.backend_config(sw_size=30, calib_imu_instrinsic=1, td=-0.004),
                      output_dir="/tmp/o/")
    paths.append(str(p))
    for ref in ("/root/reference/config/euroc.yaml", "/root/reference/config/mynteye.yaml"):
        if os.path.exists(ref):
            paths.append(ref)
    for path in paths:
        rc, out, err = _run("config", path); assert rc == 0, err
        got = _parse_fields(out)
        fe, be, out_dir = load_config(path)
        assert " ".join(got["output_dir"]) == out_dir
        a, b = make_fe_config(fe), make_ekf_config(be)
        for struct, prefix in ((a, "fe."), (b, "ekf.")):
            for name, _ in struct._fields_:
                v = getattr(struct, name)
                want = list(v) if hasattr(v, "__len__") else [v]
                assert [float(x) for x in got[prefix + name]] == [float(x) for x in want], (path, name)
    d = parse("a: 1.\nb: .5e1\nc: 'x # y'   # c\nm: !!opencv-matrix\n   rows: 1\n   cols: 2\n   data: [ 1,\n      2 ]\nn:\n   k: v\n")
    assert d["a"] == 1.0 and d["b"] == 5.0 and d["c"] == "x # y" and d["m.data"] == [1.0, 2.0] and d["n.k"] == "v"
    bad = tmp_path / "bad.yaml"; bad.write_text(p.read_text().replace("feature_idp_dim: 1", "feature_idp_dim: 3"))
    with pytest.raises(ValueError):
        load_config(str(bad))


def test_mirror_classes_accept_a_configuration_file_path(tmp_path):
    import larvio_amd
    from larvio_amd import synthetic as S
    from make_euroc_dir import write_config_yaml
    p = tmp_path / "c.yaml"
    write_config_yaml(str(p), S.frontend_config(max_features_num=123), S.backend_config(sw_size=17))
    fe = larvio_amd.ImageProcessor(str(p)); be = larvio_amd.LarVio(str(p))
    assert fe.config["max_features_num"] == 123 and be.config["sw_size"] == 17 and be.config["max_features"] == 123


def test_dataset_driver_fails_loudly_without_a_gpu_and_on_bad_input(host_tools, tmp_path):
    """examples/larvio_euroc: usage error = 1 (as the reference's driver), unreadable inputs = 1 with a message, and on a machine
    without a gfx950 device status 3 'no usable device' — never a silent CPU run"""
    from larvio_amd import synthetic as S
    from make_euroc_dir import write_euroc_dir
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "larvio_euroc"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(ROOT, "examples", "larvio_euroc")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr
    r = subprocess.run([exe, "/nonexistent/imu.csv", "/nonexistent/cam.csv", "/nonexistent", "/nonexistent.yaml"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot read IMU samples" in r.stderr
    seq = S.imu_only_sequence()
    frames = [(seq.frame_time(i), np.zeros((48, 64), np.uint8)) for i in range(2)]
    write_euroc_dir(str(tmp_path / "d"), frames, seq.imu_array(0, 30), S.frontend_config(), S.backend_config())
    d = str(tmp_path / "d")
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present: the no-device path cannot be shown here")
    r = subprocess.run([exe, d + "/mav0/imu0/data.csv", d + "/mav0/cam0/data.csv", d + "/mav0/cam0/data", d + "/config.yaml"], capture_output=True, text=True)
    assert r.returncode == 3 and "no usable gfx950 device" in r.stderr, r.stderr
