"""K8 (exact L2 2-NN on MFMA) alone on SIFT-like random descriptors: python tools/k8_time.py [n_query] [n_train] [reps]"""
import ctypes as C, sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa
from image_stitching_amd.stitching import KP_DTYPE
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = isa.Context(0)
rng = np.random.default_rng(0)
def sift_like(n):
    d = rng.gamma(0.6, 40.0, (n, 128))
    return np.minimum(np.floor(d), 255).astype(np.float32)
fq = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(nq, KP_DTYPE), sift_like(nq))
ft = isa.ImageFeatures.upload(ctx, (64, 64), np.zeros(nt, KP_DTYPE), sift_like(nt))
idx = np.zeros((nq, 2), np.int32); dist = np.zeros((nq, 2), np.float32)
call = lambda: ctx.check(ctx.lib.mis_knn2(ctx.h, C.byref(fq.raw), C.byref(ft.raw), idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)))
call(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("knn2 L2 %d x %d: %.1f us per call (incl. prep, merge, download) = %.0f TFLOP/s" % (nq, nt, dt * 1e6, 2.0 * nq * nt * 128 / dt / 1e12))
