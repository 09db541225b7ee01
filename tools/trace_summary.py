"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals (hot-path kernels only) and the
per-launch timeline of the MEDIAN benchmark step (by wall time on the GPU timeline: a single step can carry a host
hiccup of 100 us between two launches, which says nothing about the step)."""
import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
hp = [r for r in rows if 'viterbi' not in r['Kernel_Name']]
# a hot-path bench step starts at the weight re-layout launch (pack_all_kernel; pack_weights before r3) followed by the first conv of a B > 1 batch (grid y = B);
# bench.py's end-to-end leg (batch 1) is summarised separately by tools/e2e_trace_summary.py
def first_conv_after(i):      # the launch behind the weight re-packing (pack_weights, fs_pack)
    j = i + 1
    while j < len(hp) and 'pack' in hp[j]['Kernel_Name']:
        j += 1
    return hp[j] if j < len(hp) else None
idx = [i for i, r in enumerate(hp) if ('pack_all' in r['Kernel_Name'] or 'pack_weights' in r['Kernel_Name']) and first_conv_after(i) is not None
       and int(first_conv_after(i)['Grid_Size_Y']) > 1]
def wall(a, b):
    return max(int(r['End_Timestamp']) for r in hp[a:b]) - int(hp[a]['Start_Timestamp'])
cands = sorted(((wall(idx[i], idx[i + 1]), i) for i in range(max(0, len(idx) - 21), len(idx) - 1)))
s = idx[cands[len(cands) // 2][1]]
e = idx[cands[len(cands) // 2][1] + 1]
step = hp[s:e]
t0 = int(step[0]['Start_Timestamp'])
agg = collections.OrderedDict()
prev_end = t0
gap_total = 0
for r in step:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('void ', '').split('(')[0][:70]
    a = agg.setdefault(nm, [0, 0.0])
    a[0] += 1; a[1] += (en - st) / 1e3
    gap_total += max(0, st - prev_end) / 1e3
    prev_end = max(prev_end, en)
print(f"one step: {(prev_end - t0)/1e3:.1f} us wall on the GPU timeline, {len(step)} launches, gaps {gap_total:.1f} us")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t:9.1f} us  {n:4d}x  {k}")
if len(sys.argv) > 2:
    for r in step:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"{(st-t0)/1e3:9.1f} {(en-st)/1e3:8.1f}us grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']} {r['Kernel_Name'].replace('void ','')[:64]}")
