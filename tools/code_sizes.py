#!/usr/bin/env python3
"""Code size of every gfx950 kernel in the built objects (csrc/build/*.o): the shared instruction cache holds 64 KiB, and a kernel
whose per-tile path streams more than that through it refetches its own main loop (the general GEMM epilogue was 156 KiB).
usage: code_sizes.py [min KiB, default 24]"""
import glob, os, struct, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 24
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal_sentiment_aanalysis_amd", "csrc", "build")
for obj in sorted(glob.glob(os.path.join(root, "*.o"))):
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fb.bin")
        if subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], capture_output=True).returncode or not os.path.exists(fb):
            continue
        data = open(fb, "rb").read()
        if not data.startswith(b"__CLANG_OFFLOAD_BUNDLE__"):
            continue
        n = struct.unpack_from("<Q", data, 24)[0]
        off = 32
        for _ in range(n):
            o, s, l = struct.unpack_from("<QQQ", data, off); off += 24
            tid = data[off:off + l].decode(); off += l
            if "gfx950" not in tid or s == 0:
                continue
            co = os.path.join(td, "co.bin")
            open(co, "wb").write(data[o:o + s])
            out = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", "--demangle", co], capture_output=True, text=True).stdout
            rows = []
            for ln in out.splitlines():
                p = ln.split(None, 7)
                if len(p) >= 8 and p[3] == "FUNC":
                    rows.append((int(p[2]), p[7]))
            for sz, nm in sorted(rows, reverse=True):
                if sz >= thr * 1024:
                    print(f"{os.path.basename(obj):22s} {sz // 1024:4d} KiB  {nm[:120]}")
