"""Where does the fused warp differ from the oracle?  Per-tile (64 x 8) mismatch map of one case (diagnostics)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, oracle, synth
ctx = isa.Context(0)
case = int(sys.argv[1]) if len(sys.argv) > 1 else 0
CASES = [(640, 360, 60.0, 0.0, 0.0, 0.0), (640, 360, 60.0, 25.0, 3.0, -2.0), (333, 517, 50.0, -40.0, 10.0, 5.0), (1920, 1080, 60.0, 10.0, 0.5, -0.3)]
w, h, fov, yaw, pitch, roll = CASES[case]
cam = synth.make_camera(w, h, fov, yaw, pitch, roll)
img = synth.render_frame(cam)
K, R = cam["K"].astype(np.float32), cam["R"].astype(np.float32)
scale = float(cam["K"][0, 0])
warper = isa.SphericalWarper(ctx, scale)
tl, s16, msk = warper.warp_fused(torch.from_numpy(img).cuda(), K, R)
oi, otl = oracle.warp_spherical(img, scale, K, R)
om, _ = oracle.warp_spherical(np.full((h, w), 255, np.uint8), scale, K, R, oracle.INTER_NEAREST, oracle.BORDER_CONSTANT)
g = s16.cpu().numpy(); gm = msk.cpu().numpy()
print("tl", tl, otl, "shape", g.shape, oi.shape)
bad = (g != oi.astype(np.int16)).any(axis=2)
badm = gm != om
print("bad pixels", bad.sum(), "of", bad.size, " bad mask", badm.sum())
H, W = bad.shape
for ty in range(0, H, 8):
    print("%4d " % ty + "".join("X" if bad[ty:ty + 8, tx:tx + 64].all() else ("x" if bad[ty:ty + 8, tx:tx + 64].any() else ("m" if badm[ty:ty + 8, tx:tx + 64].any() else ".")) for tx in range(0, W, 64)))
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:12]:
    print("(%d,%d) got %s want %s" % (y, x, g[y, x], oi[y, x]))
