"""Per-dispatch listing of one counter from a rocprofv3 --pmc counter_collection CSV:
   python tools/pmc_list.py <csv> <kernel substring> [max rows]"""
import csv, sys
sub = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = {}
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            d = rows.setdefault(int(r["Dispatch_Id"]), {"grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (k, v) in enumerate(sorted(rows.items())):
    if i >= lim:
        break
    print(k, v)
