"""The matcher's chain kernels of one step, launch by launch (tools/step_timeline.py merges neighbours):
python tools/chain_trace.py [kernel_trace.csv] [step]"""
import csv, sys
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tl3/t_kernel_trace.csv'
step = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0], r['Stream_Id']) for r in rows)
marks = [e[0] for e in ev if e[2] == 'warp_roi_kernel']
t0, t1 = marks[step], marks[step + 1]
names = ('scan_tail_kernel', 'draw_kernel', 'hyp_quad_kernel', 'hyp_kernel', 'hyp_count_kernel', 'second_calls_kernel', 'first_calls_kernel', 'copy_segments_kernel',
         'knn2_hamming_mfma_kernel', 'warp_strip_batch_kernel', 'feed_gather_kernel', 'collapse2x2_final_kernel')
print("step %d: %.3f ms" % (step, (t1 - t0) / 1e6))
for s, e, n, st in ev:
    if t0 <= s < t1 and n in names:
        print('%-26s stream %s  %7.3f -> %7.3f  %7.3f ms' % (n, st, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
