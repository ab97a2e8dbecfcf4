"""Per-kernel table of a fit from a rocprofv3 kernel trace: calls, total / average / minimum duration per kernel family, and for the fp64 MFMA
GEMM families (gemm_f64_kernel<128>, gemm_f64_deep_kernel = 64-tile, gemm_f64_splitk_kernel = 32-tile) the TFLOP/s of their algorithmic flops.
The flops come from the library's launch log (GPLE_GEMM_LOG, csrc/gple_gemm.hip: one line per GEMM launch with its stream, tile and k-ranges);
launches and trace rows are matched per stream / hardware queue, where both are in launch order.
usage: GPLE_GEMM_LOG=g.log rocprofv3 --kernel-trace --output-format csv -d out -- python3 probes/fit_timing.py real 4096
       python probes/fit_kernel_table.py out/*/*kernel_trace.csv g.log [> profiles/r03_fit_kernel_table_n4096.md]"""
import csv, re, sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
fam = lambda name: re.sub(r"\(anonymous namespace\)::|gple::|void ", "", name).split("(")[0]
tile_of = lambda f: 32 if "splitk" in f else (64 if "deep" in f else (128 if "gemm_f64_kernel" in f else None))
log = [l.split() for l in open(sys.argv[2]) if l.strip()]
by_stream = defaultdict(list)
for l in log:
    by_stream[l[0]].append((int(l[1]), float(l[8]), tuple(int(x) for x in l[2:8])))
by_queue = defaultdict(list)
for r in rows:
    t = tile_of(fam(r["Kernel_Name"]))
    if t:
        by_queue[r["Queue_Id"]].append((t, r))
flops_of = {}
unmatched = 0
for q, lst in by_queue.items():
    seq = [t for t, _ in lst]
    match = [s for s, ls in by_stream.items() if [t for t, _, _ in ls] == seq]
    if not match:  # several streams may share a hardware queue: fall back to the per-family order over all streams
        unmatched += len(lst)
        continue
    for (t, r), (_, fl, _) in zip(lst, by_stream[match[0]]):
        flops_of[id(r)] = fl
agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0, 0])
for r in rows:
    f = fam(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[f]
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    if id(r) in flops_of:
        a[3] += flops_of[id(r)]
        a[4] += 1
print("| kernel | calls | total us | average us | min us | algorithmic GFLOP | TFLOP/s |")
print("|---|---|---|---|---|---|---|")
for f, (n, tot, mn, fl, nm) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    rate = f"{fl / (tot * 1e-6) / 1e12:.1f}" if fl and nm == n else ("-" if not fl else f"{fl / (tot * 1e-6) / 1e12:.1f} ({nm}/{n} matched)")
    print(f"| `{f}` | {n} | {tot:.1f} | {tot / n:.2f} | {mn:.2f} | {fl / 1e9:.2f} | {rate} |" if fl else f"| `{f}` | {n} | {tot:.1f} | {tot / n:.2f} | {mn:.2f} | - | - |")
if unmatched:
    print(f"\n{unmatched} GEMM dispatches could not be matched to the launch log (streams sharing a hardware queue)")
