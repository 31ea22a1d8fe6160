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
    st = isa.Stitcher(ctx, (frames[0].shape[1], frames[0].shape[0]), isa.StitchConfig.hot_path(compose_megapix=-1))
    res, mask, feats, pm, idx = st.stitch([torch.from_numpy(f).cuda() for f in frames], cams)
    assert list(idx) == [0, 1, 2]
    exp = np.clip(res.cpu().numpy(), 0, 255).astype(np.uint8)
    assert exp.shape == got.shape and np.array_equal(exp, got)


def _read_ppm(path):
    raw = open(path, "rb").read()
    hdr, rest = raw.split(b"\n255\n", 1)
    pw, ph = [int(v) for v in hdr.split(b"\n")[1].split()]
    return np.frombuffer(rest, np.uint8).reshape(ph, pw, 3)[:, :, ::-1]


@pytest.mark.gpu
def test_stitch_main_seam_step_and_sift_match_python_pipeline(tmp_path, ctx, oracle_mod):
    """The optional stages of the C++ driver (exposure compensation + Voronoi seams; SIFT features) against the Python
    mirror of the same C ABI sequence."""
    import torch
    import image_stitching_amd as isa
    _build()
    cams, frames = _write_job(str(tmp_path), oracle_mod, n=3, w=480, h=270)
    size = (frames[0].shape[1], frames[0].shape[0])
    dev = [torch.from_numpy(f).cuda() for f in frames]
    exe = os.path.join(HOST, "stitch_main")
    r = subprocess.run([exe, str(tmp_path), "--expos_comp", "gain_blocks", "--seam", "voronoi"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = _read_ppm(os.path.join(str(tmp_path), "result.ppm"))
    st = isa.Stitcher(ctx, size, isa.StitchConfig.hot_path(compose_megapix=-1, expos_comp_type="gain_blocks", seam_find_type="voronoi"))
    res, _ = st.compose(dev, cams)
    assert np.array_equal(np.clip(res.cpu().numpy(), 0, 255).astype(np.uint8), got)
    plain, _ = isa.Stitcher(ctx, size, isa.StitchConfig.hot_path(compose_megapix=-1)).compose(dev, cams)
    assert not np.array_equal(np.clip(plain.cpu().numpy(), 0, 255).astype(np.uint8), got)
    # SIFT features: same panorama (the cameras are inputs), all frames kept, SIFT-sized feature counts in the log
    r = subprocess.run([exe, str(tmp_path), "--features", "sift"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    counts = [int(l.rsplit(":", 1)[1]) for l in r.stdout.splitlines() if l.startswith("Features in image #")]
    f = isa.SiftFeatureFinder(ctx, size)
    assert counts == [len(f.detect(d)) for d in dev]
    assert "kept 3 of 3" in r.stdout
    assert np.array_equal(_read_ppm(os.path.join(str(tmp_path), "result.ppm")), np.clip(plain.cpu().numpy(), 0, 255).astype(np.uint8))
    # the reference's default seam finder through the C++ driver = through the Python mirror
    r = subprocess.run([exe, str(tmp_path), "--expos_comp", "gain_blocks", "--seam", "dp_color"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    ref_default, _ = isa.Stitcher(ctx, size, isa.StitchConfig(compose_megapix=-1)).compose(dev, cams)      # gain_blocks + dp_color
    assert np.array_equal(np.clip(ref_default.cpu().numpy(), 0, 255).astype(np.uint8), _read_ppm(os.path.join(str(tmp_path), "result.ppm")))
    # the compositing loop at compose scale (image_stitching.cpp:1105-1146: frames resized, intrinsics and warper scale times
    # compose_work_aspect) with the reference's seam step, C++ driver = Python mirror
    r = subprocess.run([exe, str(tmp_path), "--expos_comp", "gain_blocks", "--seam", "dp_color", "--compose_megapix", "0.05", "--seam_megapix", "0.02"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    small, _ = isa.Stitcher(ctx, size, isa.StitchConfig(compose_megapix=0.05, seam_megapix=0.02)).compose(dev, cams)
    got_small = _read_ppm(os.path.join(str(tmp_path), "result.ppm"))
    assert got_small.shape[1] < 0.75 * got.shape[1]
    assert np.array_equal(np.clip(small.cpu().numpy(), 0, 255).astype(np.uint8), got_small)
    # options the library does not implement are refused, not ignored
    r = subprocess.run([exe, str(tmp_path), "--seam", "gc_color"], capture_output=True, text=True)
    assert r.returncode == 1 and "not implemented" in r.stdout


@pytest.mark.gpu
def test_stitch_main_bundle_adjustment_recovers_perturbed_cameras(tmp_path, ctx, oracle_mod):
    """--ba reproj: camera descriptions with a wrong yaw come back close to the truth (panorama close to the one from
    exact cameras), while without refinement the same descriptions give a visibly different panorama."""
    import synth
    _build()
    tmp = str(tmp_path)
    cams, frames = _write_job(tmp, oracle_mod, n=3, w=640, h=360)
    exe = os.path.join(HOST, "stitch_main")
    r = subprocess.run([exe, tmp], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    exact = _read_ppm(os.path.join(tmp, "result.ppm")).astype(np.int32)
    # perturb the middle camera's description by 0.6 degrees of yaw (the frames stay as rendered)
    c = synth.make_camera(640, 360, 60.0, 14.0 * 1 - 10.0 + 0.6, 0.0, -0.3)
    with open(os.path.join(tmp, "2.txt"), "w") as fh:
        fh.write(_desc(c, oracle_mod.camera_rehand(c["R"], False)))
    r = subprocess.run([exe, tmp], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    bad = _read_ppm(os.path.join(tmp, "result.ppm")).astype(np.int32)
    from image_stitching_amd import serializer as ser
    before = ser.deserializeCameraParams(os.path.join(tmp, "cams.data"))      # the perturbed description, unrefined
    r = subprocess.run([exe, tmp, "--ba", "reproj", "--wave_correct", "no"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    after = ser.deserializeCameraParams(os.path.join(tmp, "cams.data"))
    assert ser.deserializeIndices(os.path.join(tmp, "indices.data")) == [0, 1, 2]

    def rel_angle_deg(cs, a, b, truth):
        """angle between the checkpointed relative rotation a -> b and the true one"""
        Ra, Rb = np.asarray(cs[a]["R"], np.float64), np.asarray(cs[b]["R"], np.float64)
        d = (Ra.T @ Rb) @ truth.T
        return float(np.degrees(np.arccos(np.clip((np.trace(d) - 1) / 2, -1, 1))))

    truth = cams[0]["R"].T @ cams[1]["R"]
    e_before, e_after = rel_angle_deg(before, 0, 1, truth), rel_angle_deg(after, 0, 1, truth)
    assert e_before > 0.5, e_before                 # the 0.6 degree error is in the unrefined checkpoint
    assert e_after < 0.15, (e_before, e_after)      # and mostly gone after the bundle adjustment (rotations only)
    fixed = _read_ppm(os.path.join(tmp, "result.ppm")).astype(np.int32)
    assert fixed.shape[0] > 0 and not np.array_equal(fixed.shape, (0, 0, 3))
    del bad, exact


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


# ---- cropper (row N2; image_stitching/cropper.cpp) ------------------------------------------------------------
def _cropper_tool():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "cropper_tool"])
    return os.path.join(ROOT, "host", "cropper_tool")


def _write_pgm(path, a):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a, np.uint8).tobytes())


def _read_pnm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        w, h = map(int, f.readline().split())
        f.readline()
        data = np.frombuffer(f.read(), np.uint8)
    return data.reshape(h, w, 3)[..., ::-1] if magic == b"P6" else data.reshape(h, w)


def _blob(seed, h=90, w=140):
    """a smooth random blob with a hole and a detached speck"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    ang = np.arctan2(yy - h / 2, xx - w / 2)
    rad = np.hypot((yy - h / 2) / (0.42 * h), (xx - w / 2) / (0.42 * w))
    wob = 1 + 0.12 * np.sin(3 * ang + rng.uniform(0, 6)) + 0.07 * np.sin(7 * ang + rng.uniform(0, 6))
    m = (rad < wob).astype(np.uint8) * 255
    m[h // 2 - 5:h // 2 + 5, w // 2 + 8:w // 2 + 20] = 0     # a hole
    m[2:5, 3:7] = 255                                        # a speck
    return m


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_external_contour_is_the_outer_border_and_fill_closes_holes(tmp_path, seed):
    """Independent definitions (scipy.ndimage): the traced points are exactly the component's pixels with a 4-neighbour
    in the outer background, consecutive points are 8-adjacent, and the fill equals the hole-filled component.
    The C++ and Python implementations agree point for point."""
    from scipy import ndimage as ndi
    from image_stitching_amd import cropper as crp
    tool = _cropper_tool()
    m = _blob(seed)
    src = str(tmp_path / "m.pgm")
    _write_pgm(src, m)
    lines = subprocess.check_output([tool, "contour", src]).decode().strip().split("\n")
    cpp = [np.array(l.split()[1:], np.int64).reshape(-1, 2) for l in lines]
    py = crp.findExternalContours(m)
    assert len(cpp) == len(py) == 2
    for a, b in zip(cpp, py):
        assert np.array_equal(a, np.array(b))
    big = max(cpp, key=len)
    lab, _ = ndi.label(m > 0, structure=np.ones((3, 3)))
    comp = lab == lab[big[0][1], big[0][0]]
    bg, _ = ndi.label(np.pad(~(m > 0), 1, constant_values=True))          # 4-connected background, frame included
    outer = (bg == bg[0, 0])[1:-1, 1:-1]
    near = np.zeros_like(outer)
    near[1:, :] |= outer[:-1, :]; near[:-1, :] |= outer[1:, :]; near[:, 1:] |= outer[:, :-1]; near[:, :-1] |= outer[:, 1:]
    near[0, :] = near[-1, :] = True; near[:, 0] = near[:, -1] = True      # the frame counts as background
    want = {(x, y) for y, x in zip(*np.nonzero(comp & near))}
    assert {tuple(p) for p in big} == want
    d = np.abs(np.diff(np.vstack([big, big[:1]]), axis=0))
    assert d.max() == 1 and (d.sum(1) > 0).all()                          # closed 8-connected chain
    out = str(tmp_path / "f.pgm")
    subprocess.check_call([tool, "fill", src, out])
    filled = _read_pnm(out)
    assert np.array_equal(filled > 0, ndi.binary_fill_holes(comp))
    assert np.array_equal(crp.fillContour(py[int(np.argmax([len(c) for c in py]))], m.shape[1], m.shape[0]), filled)


def test_crop_known_answers_and_cross_language(tmp_path):
    from image_stitching_amd import cropper as crp
    tool = _cropper_tool()
    # a solid rectangle: the first candidate (min/max of the contour) is accepted: x0, y0, (x1 - x0), (y1 - y0)
    img = np.zeros((60, 100, 3), np.uint8)
    img[10:50, 20:90] = (30, 200, 90)
    out, rect = crp.crop(img)
    assert rect == (20, 10, 69, 39) and out.shape == (39, 69, 3)
    # a panorama-like outline (slanted sides): both implementations pick the same rectangle, its border is interior
    yy, xx = np.mgrid[0:120, 0:300]
    pano = ((xx > 20 + 0.25 * yy) & (xx < 280 - 0.15 * (120 - yy)) & (yy > 8 + 0.03 * xx) & (yy < 112 - 0.02 * xx))
    img = np.zeros((120, 300, 3), np.uint8)
    img[pano] = (90, 120, 200)
    src, dst = str(tmp_path / "p.ppm"), str(tmp_path / "c.ppm")
    with open(src, "wb") as f:
        f.write(b"P6\n300 120\n255\n")
        f.write(img[..., ::-1].tobytes())                                  # PPM is RGB; HostImage / the arrays are BGR
    r = tuple(int(v) for v in subprocess.check_output([tool, "crop", src, dst]).split())
    out, rect = crp.crop(img)
    assert r == rect and np.array_equal(_read_pnm(dst), out)
    x, y, w, h = rect
    sub = pano[y:y + h, x:x + w]
    assert sub[0].all() and sub[-1].all() and sub[:, 0].all() and sub[:, -1].all()
    assert w * h > 0.6 * pano.sum()                                        # the heuristic keeps most of the area
    # checkInteriorExterior's out-codes: most exterior pixels on top -> top
    m = np.full((20, 30), 255, np.uint8)
    m[0, 5:25] = 0
    codes = dict(top=0, bottom=0, left=0, right=0)
    assert crp.checkInteriorExterior(m, (0, 0, 30, 20), codes) is False and codes == dict(top=1, bottom=0, left=0, right=0)
    assert crp.compareX((1, 9), (2, 0)) and not crp.compareY((1, 9), (2, 0))


def _jpeg_with_description(desc, order="II", inline_pad=False, in_exif_ifd=False):
    """A JPEG header: SOI, APP0 (JFIF), APP1 "Exif" with a TIFF block whose IFD0 (or Exif sub-IFD) carries ImageDescription, then
    a quantisation-table segment and SOS -- no image data is needed to read the tag."""
    import struct
    E = "<" if order == "II" else ">"
    val = desc.encode("latin-1") + b"\0"
    def ifd(entries, next_off=0):
        return struct.pack(E + "H", len(entries)) + b"".join(entries) + struct.pack(E + "I", next_off)
    def entry(tag, typ, count, value_or_off):
        return struct.pack(E + "HHI", tag, typ, count) + value_or_off
    head = order.encode() + struct.pack(E + "HI", 42, 8)
    if in_exif_ifd:
        ifd0_len = 2 + 12 * 2 + 4
        exif_off = 8 + ifd0_len
        sub_len = 2 + 12 + 4
        data_off = exif_off + sub_len
        ifd0 = ifd([entry(0x0112, 3, 1, struct.pack(E + "HH", 1, 0)), entry(0x8769, 4, 1, struct.pack(E + "I", exif_off))])
        sub = ifd([entry(0x010E, 2, len(val), struct.pack(E + "I", data_off))])
        tiff = head + ifd0 + sub + val
    else:
        ifd0_len = 2 + 12 * 2 + 4
        data_off = 8 + ifd0_len
        v = (val + b"\0\0\0\0")[:4] if len(val) <= 4 else struct.pack(E + "I", data_off)
        ifd0 = ifd([entry(0x010E, 2, len(val), v), entry(0x0112, 3, 1, struct.pack(E + "HH", 1, 0))])
        tiff = head + ifd0 + (val if len(val) > 4 else b"")
    app1 = b"Exif\0\0" + tiff
    app0 = b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"
    seg = lambda m, body: b"\xff" + bytes([m]) + struct.pack(">H", len(body) + 2) + body
    return b"\xff\xd8" + seg(0xE0, app0) + seg(0xE1, app1) + seg(0xDB, b"\0" * 65) + seg(0xDA, b"\0" * 6) + b"\x12\x34\xff\xd9"


def test_exif_image_description_cpp_and_python_twins(tmp_path):
    """Row N4's tag walk (the reference reads the camera string from EXIF ImageDescription through libexif): both byte orders, a
    value stored inline (<= 4 bytes), the tag in the Exif sub-IFD, truncation to 1022 characters as the reference's buffer does,
    files without the tag; then the string feeds cameraFromImageDescription's format."""
    from image_stitching_amd import serializer as ser
    host = os.path.join(ROOT, "host")
    subprocess.check_call(["make", "-s", "-C", host, "serializer_tool"])
    tool = os.path.join(host, "serializer_tool")
    m4 = "[" + ",".join(str(0.125 * i - 1) for i in range(16)) + "]"
    desc = "0;12.5;%s;%s;%s;[1000,0,960,0,1000,540,0,0,1]" % (m4, m4, m4)
    long_desc = desc + ";" + "x" * 1500
    cases = [(desc, "II", False), (desc, "MM", False), ("abc", "II", False), (desc, "MM", True), (long_desc, "II", False)]
    for k, (d, order, sub) in enumerate(cases):
        blob = _jpeg_with_description(d, order, in_exif_ifd=sub)
        path = tmp_path / ("c%d.jpg" % k)
        path.write_bytes(blob)
        want = d[:1022]
        assert ser.exif_image_description(blob) == want
        assert subprocess.check_output([tool, "exif", str(path)]).decode("latin-1") == want
    for bad in (b"", b"\xff\xd8\xff\xd9", b"\x89PNG\r\n", _jpeg_with_description(desc)[:40], b"\xff\xd8" + b"\xff\xe0\x00\x04\x00\x00" + b"\xff\xda\x00\x02"):
        path = tmp_path / "bad.jpg"
        path.write_bytes(bad)
        assert ser.exif_image_description(bad) is None
        assert subprocess.call([tool, "exif", str(path)]) == 3


def write_cams_file(path, cams):
    """Camera list in the text form host/stitch_bench reads (repr doubles round-trip exactly)."""
    with open(path, "w") as fh:
        fh.write("%d %d %d\n" % (len(cams), cams[0]["width"], cams[0]["height"]))
        for c in cams:
            vals = [c["f"], c["K"][0, 2], c["K"][1, 2], c.get("gain", 1.0)] + [float(v) for v in np.asarray(c["R"], np.float64).reshape(9)]
            fh.write(" ".join(repr(float(v)) for v in vals) + "\n")


def _read_dump(prefix):
    lines = open(prefix + ".txt").read().splitlines()
    pw, ph, bands = [int(v) for v in lines[0].split()]
    indices = [int(v) for v in lines[1].split()]
    nfeat = [int(v) for v in lines[2].split()]
    conf = np.array([float(v) for v in lines[3].split()])
    pano = np.fromfile(prefix + ".pano.s16", np.int16).reshape(ph, pw, 3)
    mask = np.fromfile(prefix + ".mask.u8", np.uint8).reshape(ph, pw)
    return dict(pano=pano, mask=mask, indices=indices, nfeat=nfeat, conf=conf, bands=bands)


@pytest.mark.gpu
@pytest.mark.parametrize("stray", [False, True])
def test_cpp_job_equals_python_job(tmp_path, ctx, stray):
    """host/stitch_bench (mis::StitchJob: batched ORB, composition speculated from the matcher's hook, collapse) against the
    Python job over the same C ABI on the same synthetic frames: indices, feature counts, confidences, panorama, mask --
    byte for byte; with a stray frame the speculated panorama is replaced by the kept set's in both."""
    import torch
    import synth
    import json
    from image_stitching_amd.distributed import StitchJob
    _build()
    w, h = 640, 360
    yaws = [-26.0, -13.0, 0.0, 13.0, 26.0, 150.0 if stray else 39.0]
    cams = [synth.make_camera(w, h, 60.0, y, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5), 0.95 + 0.02 * i) for i, y in enumerate(yaws)]
    cams_path, prefix = str(tmp_path / "cams.txt"), str(tmp_path / "out")
    write_cams_file(cams_path, cams)
    r = subprocess.run([os.path.join(HOST, "stitch_bench"), cams_path, "--steps", "2", "--warmup", "1", "--dump", prefix], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    got = _read_dump(prefix)
    frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
    ref = StitchJob(ctx, (w, h), cams).run(frames)
    assert got["indices"] == ref["indices"] == ([0, 1, 2, 3, 4] if stray else [0, 1, 2, 3, 4, 5])
    assert line["kept"] == len(ref["indices"]) and line["speculation_kept"] == (not stray)
    assert got["nfeat"] == [len(f) for f in ref["features"]]
    assert np.array_equal(got["conf"], np.asarray(ref["confidence"]).reshape(-1))
    assert got["bands"] == ref["num_bands"]
    assert np.array_equal(got["mask"], ref["mask"].cpu().numpy())
    assert np.array_equal(got["pano"], ref["pano"].cpu().numpy())


def _py_rank(rank, world, port, out_path, yaws, w, h):
    """One rank of the Python sharded job on the one GPU (gloo rendezvous, device tensors staged through the host)."""
    import sys
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth
        import image_stitching_amd as isa
        from image_stitching_amd.distributed import StitchJob
        cams = _sweep_cams(yaws, w, h)
        job = StitchJob(isa.Context(0), (w, h), cams, rank=rank, world_size=world, group=dist.group.WORLD)
        frames = {i: synth.render_frame_gpu(cams[i]) for i in job.my_frames}
        out = job.run(frames)
        if rank == 0:
            np.savez(out_path, pano=out["pano"].cpu().numpy(), mask=out["mask"].cpu().numpy(), conf=out["confidence"].cpu().numpy().reshape(-1),
                     indices=np.array(out["indices"]))
    finally:
        dist.destroy_process_group()


def _sweep_cams(yaws, w, h):
    import synth
    return [synth.make_camera(w, h, 60.0, y, 0.4 * ((i % 3) - 1), 0.3 * ((i % 2) - 0.5), 0.95 + 0.02 * i) for i, y in enumerate(yaws)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,stray", [(2, False), (3, False), (3, True)])
def test_cpp_sharded_job_equals_python_sharded_job_and_oracle(tmp_path, ctx, oracle_mod, world, stray):
    """host/stitch_bench --ranks N (mis::ShardedJob: frame blocks, feature all-gather, round-robin pairs, confidence sum, strip
    exchange of pyramid rectangles, per-strip collapse, strip all-gather -- all in C++ over the C ABI, exchanges through
    mis::Communicator's host-staged implementation, N child processes on the one GPU) against
      the Python sharded job at the same N     byte for byte (same plan, same order of the f32 additions),
      the oracle's single-process run           indices, confidences and mask exact, every pixel within 1 LSB."""
    import json
    import socket
    import torch.multiprocessing as mp
    import synth
    from oracle import job as ojob
    _build()
    w, h = 640, 360
    yaws = [-30.0, -18.0, -6.0, 6.0, 18.0, 150.0 if stray else 30.0]
    cams = _sweep_cams(yaws, w, h)
    cams_path, prefix = str(tmp_path / "cams.txt"), str(tmp_path / "out")
    write_cams_file(cams_path, cams)
    r = subprocess.run([os.path.join(HOST, "stitch_bench"), cams_path, "--steps", "1", "--warmup", "1", "--ranks", str(world), "--comm", "host", "--one-gpu", "--dump", prefix],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["one_gpu_rehearsal"] is True and "ShardedJob, %d ranks" % world in line["host"]
    got = _read_dump(prefix)
    # the Python sharded job at the same world size
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    npz = str(tmp_path / "py.npz")
    mp.start_processes(_py_rank, args=(world, port, npz, yaws, w, h), nprocs=world, join=True, start_method="spawn")
    py = np.load(npz)
    kept = [0, 1, 2, 3, 4] if stray else [0, 1, 2, 3, 4, 5]
    assert got["indices"] == list(py["indices"]) == kept
    assert line["kept"] == len(kept) and line["speculation_kept"] == (not stray)
    assert np.array_equal(got["conf"], py["conf"])
    assert np.array_equal(got["mask"], py["mask"]) and np.array_equal(got["pano"], py["pano"])
    # the oracle's run of the whole sequence
    frames = [synth.render_frame_gpu(c).cpu().numpy() for c in cams]
    ref = ojob.stitch_job(frames, cams)
    assert ref["indices"] == kept
    assert np.array_equal(got["conf"].reshape(len(cams), len(cams)), ref["confidence"])
    assert got["nfeat"] == [len(f["kps"]) for f in ref["features"]]
    assert np.array_equal(got["mask"], ref["mask"])
    d = np.abs(got["pano"].astype(np.int32) - ref["pano"].astype(np.int32))
    assert d.max() <= 1, d.max()


@pytest.mark.gpu
def test_cpp_sharded_flow_on_a_one_rank_rccl_communicator(tmp_path, ctx):
    """The RCCL calls of mis::ShardedJob themselves (ncclCommInitRank, ncclAllGather, grouped ncclSend / ncclRecv on the job's
    streams) on the box's one GPU: a one-rank communicator; the result is the unsharded job's, byte for byte."""
    import json
    import synth
    from image_stitching_amd.distributed import StitchJob
    _build()
    w, h = 640, 360
    yaws = [-26.0, -13.0, 0.0, 13.0, 26.0, 39.0]
    cams = _sweep_cams(yaws, w, h)
    cams_path, prefix = str(tmp_path / "cams.txt"), str(tmp_path / "out")
    write_cams_file(cams_path, cams)
    r = subprocess.run([os.path.join(HOST, "stitch_bench"), cams_path, "--steps", "2", "--warmup", "1", "--ranks", "1", "--comm", "rccl", "--dump", prefix],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert "RCCL" in line["host"] and line["n_gpus"] == 1
    got = _read_dump(prefix)
    ref = StitchJob(ctx, (w, h), cams).run({i: synth.render_frame_gpu(c) for i, c in enumerate(cams)})
    assert got["indices"] == ref["indices"] == [0, 1, 2, 3, 4, 5]
    assert np.array_equal(got["conf"], np.asarray(ref["confidence"]).reshape(-1))
    assert np.array_equal(got["mask"], ref["mask"].cpu().numpy()) and np.array_equal(got["pano"], ref["pano"].cpu().numpy())


def test_stitch_bench_launcher_ends_the_other_ranks_when_one_fails(tmp_path):
    """No GPU needed: with a camera file that does not parse every rank exits non-zero before touching a device; the parent
    (which never touches the GPU itself) reports it and removes the session's shared-memory files."""
    _build()
    bad = tmp_path / "cams.txt"
    bad.write_text("not a camera file\n")
    r = subprocess.run([os.path.join(HOST, "stitch_bench"), str(bad), "--ranks", "2", "--comm", "host", "--one-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "a rank exited with code" in r.stderr
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("mis_bench_")]
