"""List the launches of a rocprofv3 --kernel-trace CSV whose kernel name matches a regex (last N of them):
    python3 tools/trace_kernels.py <kernel_trace.csv> 'viterbi' [N]
start (us since the first listed launch), duration, grid in workgroups, workgroup size, name."""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r['Kernel_Name'])]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-int(sys.argv[3]) if len(sys.argv) > 3 else 0:]
t0 = int(rows[0]['Start_Timestamp']) if rows else 0
for r in rows:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = int(r['Workgroup_Size_X'])
    print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} us  grid {int(r['Grid_Size_X']) // wg}x{r['Grid_Size_Y']} wg {wg}  "
          f"{r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:70]}")
