"""Section timing of scan_tail_kernel (library built with -DMIS_TAIL_PROF; with -DMIS_PS_PROF as well: the stages of ordered_sums, whose
timers slow the passes down) and one line per tail: python tools/tail_prof.py [full]"""
import ctypes as C, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_stitching_amd as isa, synth
from image_stitching_amd.distributed import StitchJob
ctx = isa.Context(0)
cams = synth.workload("config3")
frames = {i: synth.render_frame_gpu(c) for i, c in enumerate(cams)}
job = StitchJob(ctx, (3840, 2160), cams)
full = len(sys.argv) > 1 and sys.argv[1] == "full"      # the whole step (composition speculated beside the chains), not the matcher alone
feats = job.stage_features(frames)
step = (lambda: job.run(frames)) if full else (lambda: job.stage_match(feats))
step()
out = (C.c_ulonglong * 12)()
jp = (C.c_ulonglong * 8)()
ctx.lib.mis_debug_tail_prof(out, 1)
ctx.lib.mis_debug_jac_prof(jp, 1)
hp = (C.c_ulonglong * 8)()
dp = (C.c_ulonglong * 12)()
ctx.lib.mis_debug_hyp_prof(hp, 1)
ctx.lib.mis_debug_draw_prof(dp, 1)
tl = (C.c_ulonglong * (6 * 1024))()
ctx.lib.mis_debug_tail_log(tl, 1024, 1)
step()
torch.cuda.synchronize()
ntl = ctx.lib.mis_debug_tail_log(tl, 1024, 1)
ctx.lib.mis_debug_draw_prof(dp, 1)
dd = list(dp)
if dd[11]:
    print("draw_kernel, per chunk of the parallel path (%d chunks): attempt simulation %.1f us, pointer doubling %.1f us, ranks + copy %.1f us" % (dd[11], dd[8] * 0.01 / dd[11], dd[9] * 0.01 / dd[11], dd[10] * 0.01 / dd[11]))
for ph in (0, 1):
    if dd[4 * ph + 3]:
        print("draw_kernel launches of phase %d (all estimations): %d problems, %d chunks of 4096 stream positions in all, at most %d for one problem, longest problem %.1f us"
              % (ph, dd[4 * ph + 3], dd[4 * ph], dd[4 * ph + 1], dd[4 * ph + 2] * 0.01))
ctx.lib.mis_debug_hyp_prof(hp, 1)
h = list(hp)
print("phase-1 replays: %.1f us in total, longest %.1f us" % (h[6] * 0.01, h[7] * 0.01))
if h[1]:
    print("hyp_kernel: %d 4-point solves, %.1f rotations each (max %d); per wave: %.1f rotations (its slowest lane), %.0f cycles = %.0f cycles per rotation"
          % (h[1], h[0] / h[1], h[2], h[5] / max(h[4], 1), h[3] / max(h[4], 1), h[3] / max(h[5], 1)))
ctx.lib.mis_debug_tail_prof(out, 1)
ctx.lib.mis_debug_jac_prof(jp, 1)
v = list(out)
tick = 0.01  # us (100 MHz)
print("tails %d  rotations %d  LM iterations %d" % (v[6], v[1], v[3]))
print("jacobi %.1f us total (%.2f us / rotation)   normal_eq %.1f us   dlt(incl. its jacobi) %.1f us   tail %.1f us  max tail %.1f us"
      % (v[0] * tick, v[0] * tick / max(v[1], 1), v[2] * tick, v[4] * tick, v[5] * tick, v[7] * tick))
print("DLT + LM launches of the first estimation (part 4 of the phase-0 finishers): first entry -> last exit %.1f us, longest workgroup %.1f us, %d workgroups with work"
      % ((v[9] - v[8]) * tick, v[10] * tick, v[11]))
j = list(jp)
rot = max(v[1], 1)
if j[7]:
    print("normal equations: %d passes over %.0f points on average: %.2f us per pass, %.1f ns per point" % (j[7], j[6] / j[7], v[2] * tick / j[7], 1e3 * v[2] * tick / max(j[6], 1)))
ps = j[3] and not j[2] == 0 and j[4]
if ps:      # ordered_sums (round 4): shader cycles of its stages ([4], [5]: time at the stage barrier of the accumulating / a producing wave)
    print("ordered_sums: %d stages of 16 points: accumulating lane %.0f cycles / stage (+ %.0f at the barrier), a producing lane %.0f (+ %.0f), the whole loop %.0f" % (j[3], j[0] / j[3], j[4] / j[3], j[1] / j[3], j[5] / j[3], j[2] / j[3]))
if j[5] and not ps:      # library built with -DMIS_JAC_PROF as well (the fine timers serialise the rotation: totals above are then inflated)
    print("jacobi calls %d (%.1f rotations each): set-up %.0f cycles / call, eigenvalue sort %.0f cycles / call" % (j[5], rot / j[5], j[0] / j[5], j[4] / j[5]))
    print("per rotation (shader cycles): loads + arithmetic %.0f  rotation + re-scans %.0f  pivot %.0f  sum %.0f" % (j[1] / rot, j[2] / rot, j[3] / rot, sum(j[1:4]) / rot))

# one line per tail (g_tail_log), grouped by launch kind (part * 10 + want), longest first
ent = [tuple(tl[6 * i + k] for k in range(6)) for i in range(max(ntl, 0))]
if ent:
    t_first = min(e[5] for e in ent)
    for kind in sorted(set(e[4] for e in ent)):
        grp = sorted((e for e in ent if e[4] == kind), key=lambda e: -e[3])
        print("launch kind %d: %d tails, entered %.0f .. %.0f us after the first; longest first (points, LM iterations, rotations, us):" % (kind, len(grp), (min(e[5] for e in grp) - t_first) * tick, (max(e[5] for e in grp) - t_first) * tick))
        print("   " + "  ".join("(%d, %d, %d, %.0f)" % (e[0], e[1], e[2], e[3] * tick) for e in grp[:12]))
        print("   ... median %.0f us, shortest %.0f us" % (grp[len(grp) // 2][3] * tick, grp[-1][3] * tick))
