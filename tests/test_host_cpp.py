"""The C++ host side (host/): rotation math KAT (CPU) and the stitch_main driver over the C ABI (GPU)."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST = os.path.join(ROOT, "host")


def _build():
    subprocess.run(["make", "-C", HOST], check=True, capture_output=True)


def test_rotation_header_matches_reference_kat():
    _build()
    r = subprocess.run([os.path.join(HOST, "rotation_kat")], capture_output=True, text=True)
    assert r.returncode == 0 and "rotation KAT OK" in r.stdout, r.stdout


def _desc(cam, R_sensor):
    """EXIF ImageDescription string of the phone app (image_stitching.cpp:413-445), full double precision."""
    m16 = "[" + ",".join(["0"] * 16) + "]"
    T = np.eye(4)
    T[:3, :3] = R_sensor
    cam_t = "[" + ",".join(repr(float(v)) for v in T.reshape(-1)) + "]"
    K = "[" + ",".join(repr(float(v)) for v in cam["K"].reshape(-1)) + "]"
    return "0;0.0;%s;%s;%s;%s" % (m16, m16, cam_t, K)


def _write_job(tmp, oracle_mod, n=3, w=320, h=180):
    import synth
    cams, frames = [], []
    for i in range(n):
        c = synth.make_camera(w, h, 60.0, 14.0 * i - 10.0, 0.5 * (i - 1), -0.3 * i)
        # the driver re-hands the sensor rotation (quaternion flip); feed it the flipped matrix of the wanted R
        R_sensor = oracle_mod.camera_rehand(c["R"], False)
        c_eff = dict(c)
        c_eff["R"] = oracle_mod.camera_rehand(R_sensor, False)     # what the driver will use
        f = synth.render_frame(c_eff)
        with open(os.path.join(tmp, "%d.ppm" % (i + 1)), "wb") as fh:
            fh.write(b"P6\n%d %d\n255\n" % (w, h))
            fh.write(f[:, :, ::-1].tobytes())
        with open(os.path.join(tmp, "%d.txt" % (i + 1)), "w") as fh:
            fh.write(_desc(c, R_sensor))
        cams.append(c_eff)
        frames.append(f)
    return cams, frames


def test_stitch_main_fails_loudly_without_gpu(tmp_path, oracle_mod):
    """No CPU fallback: on a box without a HIP device the driver reports the error and exits non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _build()
    _write_job(str(tmp_path), oracle_mod, n=2, w=160, h=96)
    r = subprocess.run([os.path.join(HOST, "stitch_main"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_stitch_main_matches_python_pipeline(tmp_path, ctx, oracle_mod):
    import torch
    import image_stitching_amd as isa
    _build()
    cams, frames = _write_job(str(tmp_path), oracle_mod)
    r = subprocess.run([os.path.join(HOST, "stitch_main"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Features in image #3" in r.stdout and "Multi-band blender, number of bands" in r.stdout
    raw = open(os.path.join(str(tmp_path), "result.ppm"), "rb").read()
    hdr, rest = raw.split(b"\n255\n", 1)
    pw, ph = [int(v) for v in hdr.split(b"\n")[1].split()]
    got = np.frombuffer(rest, np.uint8).reshape(ph, pw, 3)[:, :, ::-1]
    st = isa.Stitcher(ctx, (frames[0].shape[1], frames[0].shape[0]), isa.StitchConfig(compose_megapix=-1))
    res, mask, feats, pm, idx = st.stitch([torch.from_numpy(f).cuda() for f in frames], cams)
    assert list(idx) == [0, 1, 2]
    exp = np.clip(res.cpu().numpy(), 0, 255).astype(np.uint8)
    assert exp.shape == got.shape and np.array_equal(exp, got)


# ---- cams.data / indices.data checkpoint (row N3; image_stitching/serializer.cpp) ---------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tool():
    host = os.path.join(ROOT, "host")
    subprocess.check_call(["make", "-s", "-C", host, "serializer_tool"])
    return os.path.join(host, "serializer_tool")


def test_checkpoint_text_format_roundtrips_in_cpp_and_python(tmp_path):
    """tests/golden/cams.data follows the reference's format by construction (serializer.cpp:113-127: fields joined
    by '@', matrices as "[a,b;c,d;]", ostream default %g formatting).  Both implementations must reproduce the file
    byte for byte from its parsed content, and agree with each other on the parsed values."""
    from image_stitching_amd import serializer as ser
    tool = _tool()
    src = os.path.join(GOLDEN, "cams.data")
    out_cpp, out_py = str(tmp_path / "cams_cpp.data"), str(tmp_path / "cams_py.data")
    subprocess.check_call([tool, "copy-cams", src, out_cpp])
    cams = ser.deserializeCameraParams(src)
    ser.serializeCameraParams(cams, out_py)
    want = open(src).read()
    assert open(out_cpp).read() == want
    assert open(out_py).read() == want
    assert len(cams) == 3 and cams[1]["focal"] == 3325.54 and cams[2]["aspect"] == 0.999
    assert cams[1]["R"].dtype == np.float32 and cams[1]["R"][0, 2] == np.float32(0.258819)    # read back as CV_32F
    assert list(cams[2]["t"]) == [0.5, -2.0, float(np.float32(3e10))]
    # indices
    isrc, iout = os.path.join(GOLDEN, "indices.data"), str(tmp_path / "idx.data")
    subprocess.check_call([tool, "copy-indices", isrc, iout])
    assert open(iout).read() == open(isrc).read()
    assert ser.deserializeIndices(isrc) == [0, 2, 3, 7]
    ser.serializeIndices([0, 2, 3, 7], iout)
    assert open(iout).read() == open(isrc).read()


def test_matrix_text_parsing_matches_between_cpp_and_python():
    from image_stitching_amd import serializer as ser
    tool = _tool()
    for text in ["[1,0.5,-2e-3;4,5,6;]", "[0;0;0;]", "[3.14159;]", "[1e+06,2.5e-07;-0,7;]"]:
        out = subprocess.check_output([tool, "matrix", text]).decode().split()
        rows, cols, vals = int(out[0]), int(out[1]), np.array([float(v) for v in out[2:]], np.float32)
        m = ser.deserializeMatrix(text)
        assert m.shape == (rows, cols) and np.array_equal(m.reshape(-1), vals)
        assert ser.serializeMatrix(m.astype(np.float64)) == ser.serializeMatrix(ser.deserializeMatrix(ser.serializeMatrix(m.astype(np.float64))).astype(np.float64))
    assert subprocess.call([tool, "matrix", "1,2,3"], stderr=subprocess.DEVNULL) == 2          # no opening bracket
    with pytest.raises(ValueError):
        ser.deserializeMatrix("1,2,3")


def test_checkpoint_feeds_the_camera_dicts(tmp_path):
    """cameras -> cams.data -> cameras keeps K and R to the 6 significant digits of the format."""
    import synth
    from image_stitching_amd import serializer as ser
    cams = synth.workload("config3")[:4]
    path = str(tmp_path / "cams.data")
    ser.serializeCameraParams([dict(aspect=1.0, focal=c["K"][0, 0], ppx=c["K"][0, 2], ppy=c["K"][1, 2], t=np.zeros(3), R=c["R"]) for c in cams], path)
    back = [ser.camera_from_checkpoint(c, 3840, 2160) for c in ser.deserializeCameraParams(path)]
    for a, b in zip(cams, back):
        assert np.allclose(a["K"], b["K"], rtol=1e-5) and np.allclose(a["R"], b["R"], atol=1e-6)
