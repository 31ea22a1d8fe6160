"""Timeline of the matcher's kernels in one step of a rocprofv3 kernel trace: python tools/match_timeline.py <trace.csv> [step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0], r['Stream_Id']) for r in rows]
ev.sort()
knn = [e for e in ev if e[2] == 'knn2_hamming_kernel']
t0, nxt = knn[step][0], knn[step + 1][0]
names = ('knn2_hamming_kernel', 'ratio_union_kernel', 'first_calls_kernel', 'second_calls_kernel', 'draw_kernel', 'hyp_kernel', 'scan_tail_kernel', 'finalize_kernel', 'collapse_kernel')
for e in ev:
    if t0 <= e[0] < nxt and e[2] in names and e[1] - e[0] > 20000:
        print('%-22s stream %-3s start %7.3f  end %7.3f  dur %6.3f ms' % (e[2], e[3], (e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6))
