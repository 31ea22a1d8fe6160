import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import image_stitching_amd as isa, oracle
from test_blend_gpu import _frames
ctx = isa.Context(0)
rng = np.random.default_rng(23)
fr = _frames(rng, 3, 180, 120)
corners = [f[2] for f in fr]; sizes = [(f[0].shape[1], f[0].shape[0]) for f in fr]
for nfeed in (1, 2, 3):
    ob = oracle.Blender(oracle.BLEND_FEATHER, 0, 0.05); ob.prepare(corners, sizes)
    gb = isa.FeatherBlender(ctx, 0.05); gb.prepare(corners, sizes)
    for img, mask, tl in fr[:nfeed]:
        ob.feed(img, mask, tl); gb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), tl)
    ol, ow = ob.level(0); gl, gw = gb.level(0)
    dw = np.argwhere(ow.view(np.uint32) != gw.view(np.uint32)); dl = np.argwhere((ol != gl).any(2))
    print(nfeed, "weight mismatches", len(dw), "lap mismatches", len(dl))
    if len(dw):
        y, x = dw[0]; print(" first w", y, x, ow[y, x], gw[y, x])
    if len(dl):
        y, x = dl[0]; print(" first l", y, x, ol[y, x], gl[y, x], ow[y, x], gw[y, x])
    oo, om = ob.blend(); go, gm = gb.blend()
    d = np.argwhere((oo != go.cpu().numpy()).any(2)); print(" final mismatches", len(d))
    if len(d):
        y, x = d[0]; print("  first", y, x, oo[y, x], go.cpu().numpy()[y, x])
