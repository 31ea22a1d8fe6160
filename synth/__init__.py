"""Synthetic panorama frames (test / bench infrastructure, not product code).

Implements the seeded procedural scene of SURVEY.md section 8(d): frames are rendered from a virtual
equirectangular master texture through an exact pinhole model.  ``render_frame`` runs on the CPU
(gcc build of synth/scene.h), ``render_frame_gpu`` on the GPU (hipcc build of the same source) into a
torch tensor.  Ground-truth ``K`` / ``R`` stand in for the reference's EXIF sensor path
(image_stitching/image_stitching.cpp:340-528).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class SyCamera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("f", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("R", C.c_double * 9), ("gain", C.c_double)]


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


_cpu = None
_gpu = None


def _cpu_lib():
    global _cpu
    if _cpu is None:
        p = os.path.join(_HERE, "libmissynth_cpu.so")
        if not os.path.exists(p):
            build()
        _cpu = C.CDLL(p)
        _cpu.synth_render_frame_cpu.argtypes = [C.POINTER(SyCamera), C.c_void_p, C.c_size_t]
    return _cpu


def _gpu_lib():
    global _gpu
    if _gpu is None:
        p = os.path.join(_HERE, "libmissynth_gpu.so")
        if not os.path.exists(p):
            build()
        _gpu = C.CDLL(p)
        _gpu.synth_render_frame_gpu.argtypes = [C.POINTER(SyCamera), C.c_void_p, C.c_size_t, C.c_void_p]
    return _gpu


def rotation_yxz(yaw, pitch, roll):
    """R = Ry(yaw) Rx(pitch) Rz(roll): the YXZ order of the reference's euler.h:174-193 (radians)."""
    a, b = math.cos(pitch), math.sin(pitch)
    c, d = math.cos(yaw), math.sin(yaw)
    e, f = math.cos(roll), math.sin(roll)
    ce, cf, de, df = c * e, c * f, d * e, d * f
    return np.array([[ce + df * b, de * b - cf, a * d],
                     [a * f, a * e, -b],
                     [cf * b - de, df + ce * b, a * c]], np.float64)


def make_camera(width, height, hfov_deg, yaw_deg, pitch_deg=0.0, roll_deg=0.0, gain=1.0):
    f = (width / 2.0) / math.tan(math.radians(hfov_deg) / 2.0)
    R = rotation_yxz(math.radians(yaw_deg), math.radians(pitch_deg), math.radians(roll_deg))
    K = np.array([[f, 0, width * 0.5], [0, f, height * 0.5], [0, 0, 1]], np.float64)
    return {"width": width, "height": height, "f": f, "K": K, "R": R, "gain": gain}


def _sycam(cam):
    s = SyCamera()
    s.width, s.height = cam["width"], cam["height"]
    s.f, s.cx, s.cy = cam["f"], cam["K"][0, 2], cam["K"][1, 2]
    for i, v in enumerate(np.asarray(cam["R"], np.float64).reshape(9)):
        s.R[i] = v
    s.gain = cam.get("gain", 1.0)
    return s


def render_frame(cam):
    """CPU render -> (H, W, 3) uint8 BGR."""
    out = np.empty((cam["height"], cam["width"], 3), np.uint8)
    s = _sycam(cam)
    _cpu_lib().synth_render_frame_cpu(C.byref(s), out.ctypes.data_as(C.c_void_p), out.strides[0])
    return out


def render_frame_gpu(cam, device="cuda:0"):
    """GPU render -> torch.uint8 tensor (H, W, 3) resident in HBM."""
    import torch
    out = torch.empty((cam["height"], cam["width"], 3), dtype=torch.uint8, device=device)
    s = _sycam(cam)
    with torch.cuda.device(out.device):
        rc = _gpu_lib().synth_render_frame_gpu(C.byref(s), C.c_void_p(out.data_ptr()), out.stride(0),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError("synth_render_frame_gpu: hip error %d" % rc)
    return out


def _hash_unit(i, stream):
    """Counter-based jitter in [0,1) (no libc rand), stream ids as SURVEY 8(d)."""
    x = (i * 0x9E3779B1 ^ stream * 0x85EBCA77 ^ 0x5717C4) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    x = (x * 0x297A2D39) & 0xFFFFFFFF
    x ^= x >> 15
    return x / 4294967296.0


def _gauss(i, stream):
    u1 = max(_hash_unit(i, stream), 1e-12)
    u2 = _hash_unit(i, stream + 1)
    return math.sqrt(-2.0 * math.log(u1)) * math.cos(2 * math.pi * u2)


def workload(name, width=None, height=None):
    """Camera lists of the BASELINE.json configs. Returns list of cameras."""
    if name == "config2":      # 2 x 1080p, yaw 0 / 20 deg
        w, h = width or 1920, height or 1080
        return [make_camera(w, h, 60.0, 0.0), make_camera(w, h, 60.0, 20.0)]
    if name == "config3":      # 16 x 4K sweep, 15 deg steps, jitter N(0, 0.5 deg), gain U(0.95, 1.05)
        w, h = width or 3840, height or 2160
        return [make_camera(w, h, 60.0, 15.0 * i - 112.5, 0.5 * _gauss(i, 11), 0.5 * _gauss(i, 13),
                            0.95 + 0.1 * _hash_unit(i, 17)) for i in range(16)]
    if name == "config4":      # 64 x 4K, 2 rows x 32, 9 deg steps, pitch -/+ 14 deg
        w, h = width or 3840, height or 2160
        return [make_camera(w, h, 60.0, 9.0 * (i % 32) - 139.5, (-14.0 if i < 32 else 14.0) + 0.5 * _gauss(i, 11),
                            0.5 * _gauss(i, 13), 0.95 + 0.1 * _hash_unit(i, 17)) for i in range(64)]
    if name == "config5":      # 8 x 8K, 30 deg steps
        w, h = width or 7680, height or 4320
        return [make_camera(w, h, 60.0, 30.0 * i - 105.0) for i in range(8)]
    raise ValueError(name)
