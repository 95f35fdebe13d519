#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch of the library's kernels whose name contains any of the given substrings (from the code
object's metadata).  usage: kernel_regs.py [substr ...]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("GRAPES_LIB_PATH", os.path.join(ROOT, "grapes_amd", "libgrapes_hip.so"))
tmp = "/tmp/_grapes_co"
subprocess.run(f"rm -rf {tmp} && mkdir -p {tmp} && cd {tmp} && /opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input={so} > /dev/null 2>&1; "
               f"/opt/rocm/bin/roc-obj-ls {so} 2>/dev/null | head -1", shell=True)
out = subprocess.run(f"cd {tmp} && /opt/rocm/bin/roc-obj -o {tmp}/co {so} >/dev/null 2>&1; ls {tmp}", shell=True, capture_output=True, text=True).stdout
files = [os.path.join(tmp, f) for f in out.split() if "gfx950" in f]
for f in files:
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name: continue
        nm = name.group(1)
        if sys.argv[1:] and not any(a in nm for a in sys.argv[1:]): continue
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        print(f"{nm[:90]:90s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5}")
