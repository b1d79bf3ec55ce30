import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "blind_rotate_kernel_h2" in r[2]]
i = idx[len(idx) // 2]
# one gate = from the ops before this BR to the ops before the next BR
j = idx[len(idx) // 2 + 1]
seg = rows[i - 6: j + 1]
prev = None
for s, e, n in seg:
    print(f"{'' if prev is None else f'gap {(s - prev) / 1e3:7.1f} us':>16}  {n:<52} {(e - s) / 1e3:8.1f} us")
    prev = e
