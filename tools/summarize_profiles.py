"""Condense one tools/profile_round.sh output directory into the files profiles/ keeps:
  <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (as written by rocprofv3)
  <tag>_kernel_stats_hotpath.csv  the same for `bench.py --no-viterbi --no-cpu-baseline` (hot-path leg only)
  <tag>_step_timeline.txt     per-launch timeline of the last benchmark step
  <tag>_traffic.json          per-launch HBM traffic of the two tape-streaming kernels from the PMC passes
                              (FETCH_SIZE doubled for wide coalesced reads on gfx950 as MI355X_MICROARCH.md
                               prescribes; WRITE_SIZE as read), units: bytes
  <tag>_bench.json            the bench line of the un-profiled run
"""
import csv, glob, json, os, subprocess, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "gpurun_out", f"profiles_{tag}")
os.makedirs(dst, exist_ok=True)

def one(pattern):
    g = glob.glob(os.path.join(src, pattern))
    return g[0] if g else None

stats = one("trace/*/*_kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    print("kernel stats (top 12):")
    for r in rows[:12]:
        print(f"  {r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}%")
hot = one("trace_hot/*/*_kernel_stats.csv")
if hot:
    rows = list(csv.DictReader(open(hot)))
    with open(os.path.join(dst, f"{tag}_kernel_stats_hotpath.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    print("hot-path leg only (top 6):")
    for r in rows[:6]:
        print(f"  {r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}%")
trace_hot = one("trace_hot/*/*_kernel_trace.csv")      # the hot-path leg alone: its last step is a training step of the bench shape
if trace_hot:
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "trace_summary.py"), trace_hot, "v"], capture_output=True, text=True).stdout
    open(os.path.join(dst, f"{tag}_step_timeline.txt"), "w").write(out)
    print(out.split("\n")[0])
trace = one("trace/*/*_kernel_trace.csv")
if trace:
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "e2e_trace_summary.py"), trace, "v"], capture_output=True, text=True).stdout
    open(os.path.join(dst, f"{tag}_e2e_step_timeline.txt"), "w").write(out)
    print(out.split("\n")[0])

def pmc(pattern, counter):
    f = one(pattern)
    res = {}
    if not f:
        return res
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    # first_conv forward = the launch right after pack_weights; weight_gradients = the one batched TN launch of a step
    prev, fwd, wg = "", [], []
    step_tn = None
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if "nt_split_kernel<true, false, false, false>" in k or "nt_split16_kernel<true, false, false, false>" in k or ("nt_gemm_kernel" in k and "pack" in prev):
            fwd.append(float(r["Counter_Value"]))
        if "tn_batched_kernel" in k or "ts_batched_kernel" in k or "ts_persist_kernel" in k or "ts_runs_kernel" in k:
            wg.append(float(r["Counter_Value"]))
        prev = k
    return {"first_conv_fwd": fwd, "weight_gradients": wg}

fetch, write = pmc("pmc_fetch/*/*_counter_collection.csv", "FETCH_SIZE"), pmc("pmc_write/*/*_counter_collection.csv", "WRITE_SIZE")
traffic = {}
for k in ("first_conv_fwd", "weight_gradients"):
    if fetch.get(k) and write.get(k):
        fk = sum(fetch[k]) / len(fetch[k]) * 1024          # counter unit: KiB
        wk = sum(write[k]) / len(write[k]) * 1024
        traffic[k] = {"fetch_size_bytes_raw": fk, "fetch_bytes_corrected_x2": 2 * fk, "write_bytes": wk,
                      "hbm_bytes_per_launch": 2 * fk + wk, "launches_sampled": len(fetch[k])}
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print("traffic:", json.dumps(traffic))
for name in ("bench_plain.log", "bench_trace.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith('{"metric"')]
        if lines:
            open(os.path.join(dst, f"{tag}_{name.replace('.log', '.json')}"), "w").write(lines[-1])
            d = json.loads(lines[-1])
            print(name, d["value"], d["ms_per_step"], d["roofline"]["frac"])
