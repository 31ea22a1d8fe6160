"""Where do the library's and the oracle's refined cameras part?  (diagnostics for tests/test_reference_job_gpu.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth, oracle
import image_stitching_amd as isa
from image_stitching_amd import stitching as st
ctx = isa.Context(0)
w, h, n = 640, 360, 6
exact = [synth.make_camera(w, h, 60.0, 12.0 * i - 30.0, 2.0 * ((i % 3) - 1), 1.2 * ((i % 2) - 0.5), 0.96 + 0.015 * i) for i in range(n)]
rng = np.random.default_rng(5)
noisy = []
for c in exact:
    d = dict(c); d["R"] = synth.rotation_yxz(*np.radians(rng.normal(0, 0.4, 3))) @ c["R"]; noisy.append(d)
host = [synth.render_frame(c) for c in exact]
finder = isa.OrbFeatureFinder(ctx, (w, h))
feats = finder.detect_batch([torch.from_numpy(f).cuda() for f in host])
pm = isa.BestOf2NearestMatcher(ctx, 0.32)(feats)
start = [dict(focal=float(c["K"][0, 0]), aspect=float(c["K"][1, 1] / c["K"][0, 0]), ppx=float(c["K"][0, 2]), ppy=float(c["K"][1, 2]), R=np.asarray(c["R"], np.float64)) for c in noisy]
got = isa.bundle_adjust_reproj(ctx, feats, pm, start, 0.95, "_____")
dl = [f.download() for f in feats]
ofe = [dict(img_w=w, img_h=h, xy=np.stack([k["x"], k["y"]], 1), desc=d) for k, d in dl]
want_a, it_a = oracle.bundle_adjust_reproj(ofe, [pm[k] for k in range(n * n)], start, 0.95, "_____")
# the oracle's own features and matches
orb = oracle.Orb(w, h)
of = []
for f in host:
    k, d = orb.run(np.ascontiguousarray(f)); of.append(dict(img_w=w, img_h=h, kps=k, xy=np.stack([k["x"], k["y"]], 1), desc=d))
opm = oracle.match_all_pairs(of, oracle.match_default_params(match_conf=0.32))
want_b, it_b = oracle.bundle_adjust_reproj(of, opm, start, 0.95, "_____")
bits = lambda a: np.asarray(a, np.float64).view(np.uint64)
print("iters", it_a, it_b)
for i in range(n):
    print(i, "lib vs oracle(gpu inputs):", np.array_equal(bits(got[i]["R"]), bits(want_a[i]["R"])), "| oracle(gpu inputs) vs oracle(own inputs):", np.array_equal(bits(want_a[i]["R"]), bits(want_b[i]["R"])))
# inputs: features / matches identical?
for i in range(n):
    print("feat", i, np.array_equal(ofe[i]["xy"], of[i]["xy"]))
for k in range(n * n):
    a, b = pm[k], opm[k]
    same = (a.num_inliers == b["num_inliers"] and np.array_equal(a.matches, b["matches"]) if hasattr(a, "matches") else None)
    Ha = a.H if a.H is not None else np.zeros((3, 3)); Hb = np.asarray(b["H"], np.float64).reshape(3, 3) if b.get("has_H", b["H"] is not None) else np.zeros((3, 3))
    if not np.array_equal(bits(Ha), bits(Hb)) or a.confidence != b["confidence"] or not np.array_equal(a.inliers_mask, b["inliers_mask"]):
        print("pair", k // n, k % n, "differs: conf", a.confidence, b["confidence"], "H equal", np.array_equal(bits(Ha), bits(Hb)), "mask equal", np.array_equal(a.inliers_mask, b["inliers_mask"]), "src/dst", a.src_img_idx, a.dst_img_idx, b["src_img_idx"], b["dst_img_idx"])
gw = st.wave_correct([c["R"] for c in got], 0)
ow = oracle.wave_correct([c["R"] for c in want_a], 0)
print("wave on lib R == wave(oracle) on oracle R:", [np.array_equal(bits(a), bits(b)) for a, b in zip(gw, ow)])
ow2 = oracle.wave_correct([c["R"] for c in got], 0)
print("wave: lib vs oracle on the same (lib) R:", [np.array_equal(bits(a), bits(b)) for a, b in zip(gw, ow2)])
