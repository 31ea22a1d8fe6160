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
