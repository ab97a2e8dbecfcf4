"""Condense the rocprofv3 output of probes/r02_profile.sh (gpurun_out/r02_prof) into the tracked summaries under profiles/:
kernel stats csv, per-kernel PMC means, and r02_traffic.json (HBM bytes per launch of the dominant kernel, corrected as
MI355X_MICROARCH.md prescribes: KB = 1024 B, FETCH_SIZE doubled on gfx950).  usage: r02_summarise.py <workload> <version>"""
import csv, glob, hashlib, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, ver = sys.argv[1], sys.argv[2]
base = os.path.join(ROOT, "gpurun_out", "r02_prof")
newest = lambda pat: max(glob.glob(os.path.join(base, pat)), key=os.path.getmtime)


def short(name):
    m = re.search(r"::(\w+)(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name


rows = []
for grp in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    acc = {}
    for r in csv.DictReader(open(newest(f"pmc_{grp}_{wl}/*/*counter_collection.csv"))):
        acc.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    rows += [(k, c, len(v), sum(v) / len(v), min(v), max(v)) for (k, c), v in acc.items()]
with open(os.path.join(ROOT, "profiles", f"r02_rownorm_pmc_{wl.lower()}_{ver}.csv"), "w") as f:
    f.write("kernel,counter,dispatches,mean,min,max\n")
    for k, c, n, mean, lo, hi in rows:
        f.write(f'"{k}","{c}",{n},{mean},{lo},{hi}\n')
shutil.copy(newest(f"stats_{wl}/*/*kernel_stats.csv"), os.path.join(ROOT, "profiles", f"r02_bench_{wl.lower()}_kernel_stats_{ver}.csv"))
dom = max((r for r in rows if r[1] == "FETCH_SIZE"), key=lambda r: r[3])[0]
fetch = next(r[3] for r in rows if r[0] == dom and r[1] == "FETCH_SIZE")
write = next(r[3] for r in rows if r[0] == dom and r[1] == "WRITE_SIZE")
src = hashlib.sha256(open(os.path.join(ROOT, "gaussian_process_liouville_equation_amd", "csrc", "gple_predict.hip"), "rb").read()).hexdigest()[:16]
git = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
path = os.path.join(ROOT, "profiles", "r02_traffic.json")
old = json.load(open(path)) if os.path.exists(path) else {}
prev = old.get(wl, {})
old[wl] = dict(prev, kernel=dom, fetch_size_kb=fetch, write_size_kb=write, hbm_bytes_per_launch=int(fetch * 1024 * 2 + write * 1024), git=git,
               kernel_src_sha16=src, pmc_csv=f"profiles/r02_rownorm_pmc_{wl.lower()}_{ver}.csv")
json.dump(old, open(path, "w"), indent=1)
print(json.dumps(old[wl], indent=1))
for k, c, n, mean, lo, hi in rows:
    print(k, c, n, mean)
