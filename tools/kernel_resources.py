"""Register / LDS / occupancy report of every kernel in one HIP source (compiler remarks):
   python tools/kernel_resources.py shi-vae_amd/csrc/gemm.hip [filter]"""
import re, subprocess, sys, os
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                    "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                   capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src)))
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
for b in blocks:
    name = b.split(" ")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-110s VGPR %3s AGPR %3s scratch %4s occ %s LDS %6s" % (dem[:110], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"),
          g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
