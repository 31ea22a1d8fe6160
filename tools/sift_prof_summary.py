"""Per-kernel time of the last (8K) frame size in gpurun_out/sift_prof (see tools/sift_prof.sh)."""
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/sift_prof/sift_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'synth_render' in r['Kernel_Name']]
seg = rows[idx[-1] + 1:]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    agg[n][0] += 1
    agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
frames = 4
print('total kernel ms per frame %.2f' % (sum(v[1] for v in agg.values()) / frames / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-44s calls/frame %6.1f  ms/frame %.3f' % (k[:44], v[0] / frames, v[1] / frames / 1e6))
