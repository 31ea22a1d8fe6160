"""Per-wave phase stamps of warp_fused_tile_kernel (library built with -DWV_STAMPS)."""
import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
cams = synth.workload("config3")
cam = cams[8]
frame = synth.render_frame_gpu(cam)
scale = isa.Stitcher.warped_image_scale(cams)
w = isa.SphericalWarper(ctx, scale)
roi = w.warpRoi((3840, 2160), cam["K"], cam["R"])
dst, msk = w.alloc_fused(roi)
for _ in range(5): w.warp_fused_into(frame, cam["K"], cam["R"], roi, dst, msk)
n = ((roi[2] + 31) // 32) * ((roi[3] + 15) // 16)
buf = np.zeros((n, 8), np.uint64)
ctx.lib.mis_debug_warp_stamps(buf.ctypes.data_as(C.c_void_p), n)
s = buf[:, :6].astype(np.int64)
t0 = s[:, 0].min()
names = ["trig loads", "map+reduce", "glds issue", "glds wait", "sample+store"]
d = np.diff(s, axis=1)
print("tiles", n, "kernel span (cycles)", s[:, 5].max() - t0)
for i, nm in enumerate(names):
    print("%-14s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f" % (nm, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
life = s[:, 5] - s[:, 0]
print("wave lifetime  mean %8.0f p50 %8.0f p90 %8.0f" % (life.mean(), *np.percentile(life, [50, 90])))
r = buf[:, 6:8].astype(np.int64)
span_wall = (r[:, 1].max() - r[:, 0].min()) / 100.0   # us (100 MHz)
ratio = (s[:, 5] - s[:, 0]) / np.maximum(r[:, 1] - r[:, 0], 1)
print("kernel wall span %.2f us; shader cycles per 10 ns tick: median %.2f -> clock %.2f GHz" % (span_wall, np.median(ratio), np.median(ratio) / 10.0))
start = (r[:, 0] - r[:, 0].min()) / 100.0
end = (r[:, 1] - r[:, 0].min()) / 100.0
print("wave start (us): p10 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f max %.2f" % tuple(np.percentile(start, [10, 25, 50, 75, 90, 100])))
for tq in (2, 5, 8, 11, 14, 17, 20, 23):
    print("  t=%2d us: resident waves %d" % (tq, int(((start <= tq) & (end > tq)).sum())))
st = np.sort(s[:, 0] - t0)
print("wave start times: p10 %d p25 %d p50 %d p75 %d p90 %d max %d" % tuple(np.percentile(st, [10, 25, 50, 75, 90, 100])))
# per-XCD balance (tile -> workgroup -> XCD as in warp_fused_kernel: XCD j owns the j-th contiguous eighth of the workgroups)
TW = int(os.environ.get("WV_TILE_WAVES", "2"))
nwg = (n + TW - 1) // TW
wg = np.arange(n) // TW
q, rem = nwg >> 3, nwg & 7
bounds = [j * q + min(j, rem) for j in range(9)]
xcd = np.searchsorted(np.array(bounds[1:]), wg, side="right")
for j in range(8):
    m = xcd == j
    print("  xcd %d: tiles %5d  first start %6.2f  last end %6.2f us  sum lifetime %9.0f us  mean lifetime %5.2f us  p50 start %5.2f" %
          (j, m.sum(), start[m].min(), end[m].max(), (end[m] - start[m]).sum(), (end[m] - start[m]).mean(), np.median(start[m])))
late = start > 17
print("tiles started after 17 us: %d, mean lifetime %.2f us; before: %.2f us" % (late.sum(), (end - start)[late].mean(), (end - start)[~late].mean()))
ntx = (roi[2] + 31) // 32
trow = np.arange(n) // ntx
for r0 in range(0, trow.max() + 1, max(1, (trow.max() + 1) // 12)):
    m = (trow >= r0) & (trow < r0 + max(1, (trow.max() + 1) // 12))
    print("  tile rows %3d..: mean lifetime %5.2f us  mean start %5.2f  max end %5.2f" % (r0, (end - start)[m].mean(), start[m].mean(), end[m].max()))
