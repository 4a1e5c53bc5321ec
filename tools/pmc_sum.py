"""Sum a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_sum.py <csv> <out.json>"""
import csv, collections, json, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:120]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
out = {k: dict(v, launches=len(calls[k])) for k, v in acc.items() if "shivae" in k}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out)[:2000])
