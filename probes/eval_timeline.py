"""Timeline of ONE objective evaluation with gradient from a rocprofv3 kernel trace of probes/objective_eval_timing.py: start, gap to the previous kernel's end,
duration, kernel.  usage: python probes/eval_timeline.py <kernel_trace.csv> real|complex"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda n: (lambda m: (m.group(1) + (m.group(2) or "")) if m else n[:30])(re.search(r"(\w+)(<[^>]*>)?\(", n))
key = "real_deriv_sums_kernel" if sys.argv[2] == "real" else "complex_deriv_sums_kernel"
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i1 = idx[-2]
j = i1
while j > 0 and "prep_labels_kernel" not in rows[j]["Kernel_Name"]:
    j -= 1
t0 = int(rows[j]["Start_Timestamp"]); prev = t0; k = j
while k < len(rows) and (k <= i1 or "prep_labels_kernel" not in rows[k]["Kernel_Name"]):
    r = rows[k]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:6.1f}  {nm(r['Kernel_Name'])}")
    prev = max(prev, e); k += 1
