"""Timeline of the last 8K SIFT detect in gpurun_out/sift_prof (tools/sift_prof.sh): per stream, merged runs of one kernel name.
python tools/sift_timeline.py"""
import csv, glob
f = (glob.glob('gpurun_out/sift_prof/*kernel_trace.csv') + glob.glob('gpurun_out/sift_prof/*/*kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
# the last detect: from the last sift_gray_kernel on
g = [i for i, r in enumerate(rows) if name(r).startswith('sift_gray')]
seg = rows[g[-1]:]
t0 = int(seg[0]['Start_Timestamp'])
runs = []
for r in seg:
    n, s, e, st = name(r), int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Stream_Id']
    if runs and runs[-1][0] == n and runs[-1][3] == st and s - runs[-1][2] < 20000: runs[-1][2] = e; runs[-1][4] += 1; runs[-1][5] += e - s
    else: runs.append([n, s, e, st, 1, e - s])
for n, s, e, st, k, busy in runs:
    print('%-34s stream %-3s %8.3f -> %8.3f ms  x%-3d busy %7.3f ms' % (n[:34], st, s / 1e6, e / 1e6, k, busy / 1e6))
print('detect: %.3f ms from the first kernel to the last' % ((max(int(r['End_Timestamp']) for r in seg) - t0) / 1e6))
