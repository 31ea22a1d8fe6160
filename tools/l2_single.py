"""One L2 2-NN of two 24 k x 128 SIFT-like descriptor sets through mis_knn2, a few times (for tools/pmc_l2.sh): python tools/l2_single.py [n]"""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa
from image_stitching_amd.stitching import KP_DTYPE
ctx = isa.Context(0)
rng = np.random.default_rng(5)
n = 24000
q = rng.integers(0, 256, (n, 128)).astype(np.float32)
t = rng.integers(0, 256, (n, 128)).astype(np.float32)
fq = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(n, KP_DTYPE), q)
ft = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(n, KP_DTYPE), t)
idx = np.zeros((n, 2), np.int32)
dist = np.zeros((n, 2), np.float32)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter()
    ctx.check(ctx.lib.mis_knn2(ctx.h, C.byref(fq.raw), C.byref(ft.raw), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))
    print("knn2 L2 %d x %d: %.2f ms (results copied to the host)" % (n, n, (time.perf_counter() - t0) * 1e3))
