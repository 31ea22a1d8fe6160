"""Small device-to-host copies on the default stream vs a non-blocking stream of our own (diagnostics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_stitching_amd as isa
from image_stitching_amd import distributed as misdist
ctx = isa.Context(0)
eng = misdist.HipEngine(ctx, (3840, 2160))
src = [torch.randint(0, 255, (229, 384, 3), dtype=torch.uint8, device="cuda") for _ in range(32)]
dst = [torch.empty((229, 384, 3), dtype=torch.uint8).pin_memory() for _ in range(32)]
torch.cuda.synchronize()
def run(stream, name):
    with torch.cuda.stream(stream):
        for _ in range(2):
            t0 = time.perf_counter()
            for s, d in zip(src, dst):
                d.copy_(s, non_blocking=True)
            t1 = time.perf_counter()
            stream.synchronize()
            t2 = time.perf_counter()
        print("%-28s issue %.2f ms, wait %.2f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
run(torch.cuda.current_stream(), "torch current stream")
run(torch.cuda.Stream(), "torch.cuda.Stream()")
run(eng.compose_stream, "engine compose stream")
