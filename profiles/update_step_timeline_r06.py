import csv, glob, os, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
ends = [int(r["End_Timestamp"]) for r in rows if "adam_all_kernel" in r["Kernel_Name"] or "adam_mlp_pack_kernel" in r["Kernel_Name"]]
# last step that contains an occ_pack kernel
occ = [int(r["Start_Timestamp"]) for r in rows if "occ_pack_kernel" in r["Kernel_Name"]]
t = occ[-1]
k0 = max(e for e in ends if e < t); k1 = min(e for e in ends if e > t)
print("update step: %.1f us long" % ((k1 - k0) / 1e3))
last_end = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    if s >= k0 and s <= k1:
        gap = (s - last_end.get(q, k0)) / 1e3
        print("%9.1f  + %6.1f us  queue %s  gap %7.1f  %s" % ((s - k0) / 1e3, (e - s) / 1e3, q, gap, r["Kernel_Name"].split("(")[0].replace("void ", "")[-60:]))
    last_end[q] = e
