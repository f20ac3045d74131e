#!/usr/bin/env python3
"""Register / LDS / scratch usage and a memory-instruction histogram of the kernels in one object of
optimization_amd/_build (cross-compiled gfx950 code object; runs without a GPU).
Usage: python tools/kernel_info.py hot_unity.hip.o k_st_hess_fused [--asm out.s]"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj, pat = sys.argv[1], sys.argv[2]
asm_out = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None
src = obj if os.path.exists(obj) else os.path.join(ROOT, "optimization_amd", "_build", obj)
with tempfile.TemporaryDirectory() as td:
    o = os.path.join(td, "x.o")
    subprocess.run(["cp", src, o], check=True)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", o], check=True, capture_output=True, cwd=td)
    co = [f for f in os.listdir(td) if "gfx950" in f][0]
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(td, co)], capture_output=True, text=True).stdout
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", os.path.join(td, co)], capture_output=True, text=True).stdout
if asm_out:
    open(asm_out, "w").write(dis)
dem = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    g = lambda k: re.search(k + r":\s+(\d+)", blk).group(1)
    print(dem(name).split("(")[0])
    print("   vgpr", g(r"\.vgpr_count"), "sgpr", g(r"\.sgpr_count"), "lds", g(r"\.group_segment_fixed_size"),
          "scratch", g(r"\.private_segment_fixed_size"), "vgpr_spill", g(r"\.vgpr_spill_count"))
    m = re.search(r"<" + re.escape(name) + r">:\n(.*?)(?=\n\n[0-9a-f]+ <|\Z)", dis, re.S)
    if m:
        c = collections.Counter(re.findall(r"\t((?:global|buffer|scratch|flat)_(?:load|store)\w*|ds_\w+|s_barrier|s_load_\w+|s_waitcnt vmcnt\(0\))", m.group(1)))
        print("   " + ", ".join(f"{k} x{v}" for k, v in c.most_common()))
