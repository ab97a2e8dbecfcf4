"""Timeline of the LAST fit in a rocprofv3 --kernel-trace CSV: per kernel start offset, duration, gap to the previous kernel's
end and the stream (queue) it ran on.  usage: python probes/trace_timeline.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last fit: from the last prep_labels_kernel on
idx = max(i for i, r in enumerate(rows) if "prep_labels" in r["Kernel_Name"])
sel = rows[idx:]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} queue  kernel")
for r in sel:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    import re
    name = re.sub(r"\(anonymous namespace\)::|gple::|void ", "", r["Kernel_Name"]).split("(")[0][:44]
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {(st - prev_end) / 1e3:7.1f} {r.get('Queue_Id', '?'):>5}  {name} grid={r.get('Grid_Size', '')}")
    prev_end = max(prev_end, en)
print("total_us", (prev_end - t0) / 1e3)
