"""Timeline of one job step from a rocprofv3 kernel trace: python tools/step_timeline.py <kernel_trace.csv> [step] [min_us]
Steps are delimited by warp_roi_kernel (the first launch of StitchJob.run).  Kernels of one name on one stream that follow each
other closely are merged into one line (count x total)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0], r['Stream_Id']) for r in rows]
ev.sort()
marks = [e[0] for e in ev if e[2] == 'warp_roi_kernel']
t0, t1 = marks[step], marks[step + 1]
cur = None
out = []
for e in ev:
    if not (t0 <= e[0] < t1):
        continue
    if cur and cur[2] == e[2] and cur[3] == e[3] and e[0] - cur[1] < 30000:
        cur = (cur[0], e[1], cur[2], cur[3], cur[4] + 1, cur[5] + e[1] - e[0])
        out[-1] = cur
    else:
        cur = (e[0], e[1], e[2], e[3], 1, e[1] - e[0])
        out.append(cur)
print("step %d: %.3f ms" % (step, (t1 - t0) / 1e6))
for s, e, name, stream, cnt, busy in out:
    if busy / 1e3 >= min_us:
        print('%-34s stream %-3s start %7.3f  end %7.3f  x%-3d busy %7.3f ms' % (name[:34], stream, (s - t0) / 1e6, (e - t0) / 1e6, cnt, busy / 1e6))
