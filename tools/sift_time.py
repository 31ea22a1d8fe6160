"""GPU SIFT timing at the BASELINE sizes: python tools/sift_time.py [config5]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
ctx = isa.Context(0)
for (w, h) in [(1920, 1080), (3840, 2160), (7680, 4320)]:
    cam = synth.make_camera(w, h, 60.0, 15.0)
    fr = synth.render_frame_gpu(cam)
    f = isa.SiftFeatureFinder(ctx, (w, h))
    ft = f.detect(fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ft = f.detect(fr)
    torch.cuda.synchronize()
    print("%dx%d: %d keypoints, %.1f ms per frame" % (w, h, len(ft), (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    f.close()
