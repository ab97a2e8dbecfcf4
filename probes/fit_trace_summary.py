"""Timeline of the LAST fit in a rocprofv3 kernel trace of probes/fit_timing.py: kernel, start offset, duration, gap to the previous end."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a fit starts with gram_train_kernel
starts = [i for i, r in enumerate(rows) if "gram_train" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
tot = {}
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gple::", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  grid {r.get('Grid_Size_X', '?'):>7} {name}")
    prev_end = max(prev_end, e)
    k = name.split("<")[0]
    tot[k] = tot.get(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += (e - s) / 1e3
print("total span us", (prev_end - t0) / 1e3)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:9.2f} us {v[0]:4d} x {k}")
