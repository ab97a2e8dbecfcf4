"""Condense the rocprofv3 output of probes/r04_profile.sh (gpurun_out/r04_prof) into the tracked summaries under profiles/: kernel stats
csv, per-kernel PMC means, and one key per workload in profiles/r04_traffic.json — HBM bytes per launch of the workload's dominant kernel,
corrected as MI355X_MICROARCH.md prescribes (KB = 1024 B, FETCH_SIZE doubled on gfx950), next to its algorithmic bytes.
usage: r04_summarise.py <tag> <version> [workload if the tag differs]"""
import csv, glob, hashlib, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, ver = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 else tag
base = os.path.join(ROOT, "gpurun_out", "r04_prof")
newest = lambda pat: max(glob.glob(os.path.join(base, pat)), key=os.path.getmtime)


def short(name):
    m = re.search(r"::(\w+)(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name


rows = []
for grp in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    acc = {}
    for r in csv.DictReader(open(newest(f"pmc_{grp}_{tag}/*/*counter_collection.csv"))):
        acc.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    rows += [(k, c, len(v), sum(v) / len(v), min(v), max(v)) for (k, c), v in acc.items()]
pmc_csv = f"profiles/r04_pmc_{tag.lower()}_{ver}.csv"
with open(os.path.join(ROOT, pmc_csv), "w") as f:
    f.write("kernel,counter,dispatches,mean,min,max\n")
    for k, c, n, mean, lo, hi in rows:
        f.write(f'"{k}","{c}",{n},{mean},{lo},{hi}\n')
stats_csv = f"profiles/r04_bench_{tag.lower()}_kernel_stats_{ver}.csv"
shutil.copy(newest(f"stats_{tag}/*/*kernel_stats.csv"), os.path.join(ROOT, stats_csv))
# the JSON line of the stats pass (its run is the one whose kernel durations the csv holds)
line = [l for l in open(os.path.join(base, f"stats_{tag}.log")) if l.startswith('{"metric"')]
bench = json.loads(line[-1]) if line else {}
if bench:
    json.dump(bench, open(os.path.join(ROOT, f"profiles/r04_bench_{tag.lower()}_{ver}_under_rocprof.json"), "w"), indent=1)
dom = max((r for r in rows if r[1] == "FETCH_SIZE"), key=lambda r: r[3] * r[2])[0]
get = lambda c: next((r[3] for r in rows if r[0] == dom and r[1] == c), None)
fetch, write = get("FETCH_SIZE"), get("WRITE_SIZE")
stat = next((r for r in csv.DictReader(open(os.path.join(ROOT, stats_csv))) if short(r["Name"]) == dom), None)
entry = {"kernel": dom, "fetch_size_kb": fetch, "write_size_kb": write, "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
         "rocprof_average_ms": float(stat["AverageNs"]) / 1e6 if stat else None, "rocprof_calls": int(stat["Calls"]) if stat else None}
# algorithmic bytes of one launch
from bench import WORKLOADS
N, G, kind = WORKLOADS[wl]
if kind in ("real", "complex") and (dom.startswith("rownorm") or dom.startswith("predict_fused")):
    n, rows_total = (2 * N, 2 * G * G) if kind == "complex" else (N, G * G)
    launches = bench.get("roofline", {}).get("launches_per_step") or 1
    m_chunk = rows_total / launches
    ntiles = n // 256
    entry["launch"] = f"{int(m_chunk)} typed grid rows x n={n} (one of the {int(launches)} K* chunks of a {wl} predict)"
    entry["algorithmic_bytes_per_launch"] = int(m_chunk * n * 8 * (ntiles + 1) / 2 + n * (n + 1) / 2 * 8)
    entry["algorithmic_flops_per_launch"] = m_chunk * n * (n + 1)
    entry["note"] = ("K* chunk re-read once per 256-column tile of T it meets, M_chunk * n * 8 * (ntiles + 1) / 2, + the lower triangle of T once "
                     "(the per-workgroup re-reads of T are served by L2 / MALL)")
    if dom.startswith("predict_fused"):  # K* never exists in memory: the points in, three outputs out, T once
        entry["algorithmic_bytes_per_launch"] = int(m_chunk * (16 + 24) + n * (n + 1) / 2 * 8)
        entry["note"] = "test points in (16 B), mean / variance / cut-off out (24 B), the lower triangle of T once; K* is generated in LDS"

elif kind == "opt":
    if tag.endswith("only1"):  # the complex element: two N x 2N x N block products per sub-kernel parameter (csrc/gple_capi.hip, complex_fit_derivatives)
        entry["launch"] = f"E = A M_a or F = B M_b of one sub-kernel parameter: {N} x {2 * N} x {N}"
        entry["algorithmic_bytes_per_launch"] = int(5 * N * N * 8)
        entry["algorithmic_flops_per_launch"] = 4.0 * N ** 3
        entry["note"] = ("block product N x 2N x N: A (N x N) and M's half (2N x N) read once, the result (N x 2N) written once; the XCD-aware tile order "
                         "leaves one 8 x 8-tile block of operand panels per XCD at a time: 16 panels of 4 MB per 64 tiles = 2.0 GB + 0.27 GB written expected")
    else:
        entry["launch"] = f"dK * K^-1 of one length parameter, n = {N}"
        entry["algorithmic_bytes_per_launch"] = int(3 * N * N * 8)
        entry["algorithmic_flops_per_launch"] = 2.0 * N ** 3
        entry["note"] = "dense n x n x n product: both operands read once, the result written once (tile re-reads are served by L2 / MALL)"
if entry.get("algorithmic_bytes_per_launch"):
    entry["traffic_over_algorithmic"] = round(entry["hbm_bytes_per_launch"] / entry["algorithmic_bytes_per_launch"], 3)
if entry.get("algorithmic_flops_per_launch") and entry.get("rocprof_average_ms"):
    entry["tflops_from_rocprof_average"] = round(entry["algorithmic_flops_per_launch"] / (entry["rocprof_average_ms"] * 1e-3) / 1e12, 2)
busy, act, mfma = get("SQ_VALU_MFMA_BUSY_CYCLES"), get("GRBM_GUI_ACTIVE"), get("SQ_INSTS_VALU_MFMA_MOPS_F64")
if busy and act:
    entry["mfma_busy_frac"] = round(busy / (act / 8 * 256 * 4) if act else 0.0, 4)  # GRBM_GUI_ACTIVE sums the 8 XCDs; 256 CUs x 4 SIMDs
entry["git"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
src = "gple_gemm.hip" if kind == "opt" else "gple_predict.hip"
entry["kernel_src_sha16"] = hashlib.sha256(open(os.path.join(ROOT, "gaussian_process_liouville_equation_amd", "csrc", src), "rb").read()).hexdigest()[:16]
entry["pmc_csv"], entry["kernel_stats_csv"] = pmc_csv, stats_csv
path = os.path.join(ROOT, "profiles", "r04_traffic.json")
old = json.load(open(path)) if os.path.exists(path) else {}
old[tag] = entry
json.dump(old, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
