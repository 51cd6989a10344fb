"""The product's ASL / EuRoC CSV readers (examples/lvk_dataset.hpp, row N3) beside the REFERENCE'S OWN - /root/reference/include/utils/
DataReader.hpp (loadImuFile, loadImageList: what app/larvioMain.cpp:35-36 calls) compiled in place into oracle/_ref/lvref_reader
(oracle/Makefile target `ref`).  Every record both produce is identical down to the double's bits; the two differences are the ones
lvk_dataset.hpp states: the reference turns the file's trailing newline into one more record (stamp 0, the previous line's values:
`while (!inf.eof())`), and it keeps the line's CR in the image name (its driver strips exactly one character)."""
import os
import subprocess

import numpy as np
import pytest

from tests.test_host_tools import host_tools, _run  # noqa: F401  (host_tools: builds examples/host_tools)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "lvref_reader")


def _ref_tool(*args):
    if not os.path.exists(REF):
        if not os.path.isdir("/root/reference/include"):
            pytest.skip("oracle/_ref/lvref_reader not built and /root/reference absent")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    return subprocess.run([REF, *args], capture_output=True, check=True).stdout.decode().replace("\r", "<CR>")     # (bytes: text mode would turn a lone CR into a newline)


def _rows(text):
    return [[float(x) for x in l.split()] for l in text.splitlines()]


@pytest.mark.parametrize("eol,tail", [("\r\n", True), ("\n", True), ("\n", False)])
def test_readers_against_the_references_own(host_tools, tmp_path, eol, tail):
    rng = np.random.default_rng(4)
    n = 200
    stamps = 1403636579758555392 + np.arange(n) * 5000000 + rng.integers(0, 900, n)
    vals = rng.normal(0, 3, (n, 6))
    imu = tmp_path / "imu.csv"
    body = eol.join(["#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y,w_RS_S_z,a_RS_S_x [m s^-2],a_RS_S_y,a_RS_S_z"] +
                    ["%d,%s" % (s, ",".join(repr(float(x)) for x in v)) for s, v in zip(stamps, vals)]) + (eol if tail else "")
    imu.write_bytes(body.encode())
    rc, out, err = _run("imu", str(imu)); assert rc == 0, err
    mine, ref = _rows(out), _rows(_ref_tool("imu", str(imu)))
    assert len(mine) == n and mine == ref[:n]                                 # bit-identical doubles (printed with %.17g)
    assert np.array_equal(np.array(mine)[:, 1:], vals) and np.array_equal(np.array(mine)[:, 0], 1e-9 * stamps.astype(np.float64))
    if tail:
        assert len(ref) == n + 1 and ref[n][0] == 0.0 and ref[n][1:] == ref[n - 1][1:]      # the reference's record for the trailing newline
    else:
        assert len(ref) == n
    cam = tmp_path / "cam.csv"
    names = ["%d.png" % s for s in stamps[::10]]
    cam.write_bytes((eol.join(["#timestamp [ns],filename"] + ["%d,%s" % (s, nm) for s, nm in zip(stamps[::10], names)]) + (eol if tail else "")).encode())
    rc, out, err = _run("images", str(cam)); assert rc == 0, err
    mine = [l.split() for l in out.splitlines()]
    ref = [(float(l.split(" ", 1)[0]), l.split(" ", 1)[1]) for l in _ref_tool("images", str(cam)).split("\n") if l]
    assert [m[1] for m in mine] == names and [float(m[0]) for m in mine] == [r[0] for r in ref[:len(names)]]
    cr = "<CR>" if eol == "\r\n" else ""
    assert [r[1] for r in ref[:len(names)]] == ["[%s%s]" % (nm, cr) for nm in names]        # the reference keeps the CR in the name
    assert len(ref) == len(names) + (1 if tail else 0)
