"""Summarise a rocprofv3 --kernel-trace CSV of tools/e2e_loop.py or bench.py: one training step = the launches between two
consecutive sgd_apply_kernel launches (the median of the last 20); prints the step's wall time on the GPU timeline, the kernel-time sum, the gaps,
and the per-kernel totals."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sgd_apply_kernel' in r['Kernel_Name']]
# the median (by wall time on the GPU timeline) of the last 20 steps: a single step can carry a host hiccup between two launches
def wall(k):
    return max(int(r['End_Timestamp']) for r in rows[idx[k] + 1: idx[k + 1] + 1]) - int(rows[idx[k]]['End_Timestamp'])
cands = sorted((wall(k), k) for k in range(max(0, len(idx) - 22), len(idx) - 2))
k = cands[len(cands) // 2][1]
s, e = idx[k] + 1, idx[k + 1] + 1
step = rows[s:e]
t0 = int(rows[idx[k]]['End_Timestamp'])
agg = collections.OrderedDict(); prev_end = t0; gaps = 0.0; busy = 0.0
for r in step:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('void ', '').split('(')[0][:60]
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += (en - st) / 1e3
    gaps += max(0, st - prev_end) / 1e3; busy += (en - st) / 1e3; prev_end = max(prev_end, en)
print(f"one end-to-end step: {(prev_end - t0)/1e3:.1f} us on the GPU timeline, {len(step)} launches, kernel time {busy:.1f} us, gaps {gaps:.1f} us")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {t:9.1f} us  {n:4d}x  {k}")
if len(sys.argv) > 2:
    for r in step:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"{(st-t0)/1e3:9.1f} {(en-st)/1e3:8.1f}us {r['Kernel_Name'].replace('void ','')[:70]}")
