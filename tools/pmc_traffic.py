"""profiles/pmc_traffic.json entries from a FETCH_SIZE and a WRITE_SIZE pass (tools/pmc_pass.sh -> tools/pmc_sum.py):
   python tools/pmc_traffic.py <fetch.json> <write.json> <config> <global_batch> <seq_len> <source text>
FETCH_SIZE / WRITE_SIZE count KiB; gfx950 under-reports FETCH_SIZE by 2 (MI355X_MICROARCH.md, HBM / rocprofv3 section).
bench.py's kernel names are the keys; kernels that bench.py reports under one name are summed."""
import json, sys
fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
cfg, B, T, src = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
NAMES = {"gemm_split_nt_kernel<128x256, pre-split W>": ["shivae::gemm_split_nt_kernel"],
         "gemm_split_tn_kernel<128x256x16>": ["void shivae::gemm_split_tn_kernel<false>", "void shivae::gemm_split_tn_kernel<true>", "void shivae::gemm_split_tn_kernel<0>", "void shivae::gemm_split_tn_kernel<1>", "void shivae::gemm_split_tn_kernel<2>"],
         "gemm_kernel<NT,256x32>": ["void shivae::gemm_kernel<true, true, true, 4, 1, 2, 1, 4, false, false, false>"],
         "gemm_kernel<TN,256x32>": ["void shivae::gemm_kernel<false, false, true, 4, 1, 2, 1, 4, false, false, false>"]}
path = "profiles/pmc_traffic.json"
tab = json.load(open(path))
tab["entries"] = [e for e in tab["entries"] if not (e["config"] == cfg and e["global_batch"] == B and e["seq_len"] == T and e["kernel"] in NAMES)]
for name, keys in NAMES.items():
    keys = [k for k in keys if k in fetch and k in write]
    if not keys:
        continue
    n = sum(fetch[k]["launches"] for k in keys)
    assert n == sum(write[k]["launches"] for k in keys), "the two passes saw different launch counts"
    f = sum(fetch[k]["FETCH_SIZE"] for k in keys) * 1024 * 2 / n
    w = sum(write[k]["WRITE_SIZE"] for k in keys) * 1024 / n
    tab["entries"].append({"config": cfg, "global_batch": B, "seq_len": T, "kernel": name, "launches_in_pass": n,
                           "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w, "source": src})
    print("%-45s %5d launches  fetch %.2f GB  write %.2f GB  per launch" % (name, n, f / 1e9, w / 1e9))
json.dump(tab, open(path, "w"), indent=1)
