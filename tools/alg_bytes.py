#!/usr/bin/env python3
"""Algorithmic HBM bytes of the MFMA GEMM launches of one training step, from the per-launch shape table bench.py dumps
(MMSA_PROF_DUMP=<csv> python bench.py: STAMP_STEPS = 3 steps of rows): every distinct operand of a launch read once, C written
once (fp32 for weight gradients), bf16 operands. A 3x3 implicit-GEMM gather counts each source pixel once (K / 9 channels per
row), a 1x1 strided gather likewise; a grouped BERT weight-gradient launch (row M = -n) is the four problems of one layer.
usage: alg_bytes.py shapes.csv [steps=3] [hidden=768 intermediate=3072]"""
import csv, sys

def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    I = int(sys.argv[4]) if len(sys.argv) > 4 else 3072
    rows = list(csv.DictReader(open(path)))
    total, flop = 0.0, 0.0
    if rows and rows[0].get("alg_bytes"):  # newer dumps carry the library's own count per launch (side operands included)
        total = sum(float(r["alg_bytes"]) for r in rows)
        flop = sum(float(r["tflops"]) * float(r["us"]) * 1e6 for r in rows)
        rows_iter = []
    else:
        rows_iter = rows
    for r in rows_iter:
        M, N, K = int(r["M"]), int(r["N"]), int(r["K"])
        g, akm, bkm = int(r["gather"]), int(r["a_kmajor"]), int(r["b_kmajor"])
        f = float(r["tflops"]) * float(r["us"]) * 1e6
        flop += f
        if M < 0:  # grouped weight gradients of one BERT layer: K = B*S rows; operands ds2, act, dpre, h1, ds1, ctx, dqkv, x
            total += K * (5 * H + 2 * I + 3 * H) * 2 + (2 * H * I + H * H + 3 * H * H) * 4
            continue
        out_bytes = 4 if (akm and bkm) else 2  # TN = weight gradient (fp32)
        taps = 9 if (g and K % 9 == 0 and (K // 9) % 64 == 0) else 1
        if g == 1:    # A rows are gathered pixels: each source pixel once
            a = M * (K // taps) * 2
            b = N * K * 2
        elif g == 2:  # B rows are gathered pixels (weight gradient of a convolution): N = taps * Cin
            tn = 9 if (N % 9 == 0 and (N // 9) % 64 == 0) else 1
            a = K * M * 2
            b = K * (N // tn) * 2
        else:
            a, b = M * K * 2, N * K * 2
        total += a + b + M * N * out_bytes
    n = len(rows) / steps
    print(f"{len(rows)} rows = {n:.0f} launches/step; algorithmic bytes/step {total / steps / 1e9:.3f} GB = {total / len(rows) / 1e6:.2f} MB per launch; "
          f"flop/step {flop / steps / 1e12:.3f} T = {flop / len(rows) / 1e9:.2f} GFLOP per launch")

if __name__ == "__main__":
    main()
