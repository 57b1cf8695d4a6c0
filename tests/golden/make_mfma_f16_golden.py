"""Assembles tests/golden/mfma_f16_probe.npz: dot products (a[32], b[32] as f16 bit patterns in the instruction's k order, c) and the
f32 results ONE v_mfma_f32_16x16x32_f16 gave for them on an MI355X (gfx950, ROCm 7.2).

These are DEVICE-PRODUCED vectors: the inputs come from tools/ubench/f16_probe_gen.py (designed probes: exponent gaps, cancellations,
ties, zeros, subnormals, binade changes) and from the random regimes of tools/ubench/mfma_f16_split.hip; the outputs from
`tools/ubench/mfma_f16_split probe <in> <out>` on the GPU box (gpurun_out/f16split/*). Re-running this script needs those files, i.e.
a GPU run; the committed .npz is what tests/test_oracle_f16x2.py replays through oracle/nn_f16x2.hpp::mfma_f16_k32 on the CPU.
Usage: python tests/golden/make_mfma_f16_golden.py <probe_in.bin> <probe_out.bin> [more pairs ...] --regimes gpurun_out/f16split
"""
import os
import sys

import numpy as np


def diagonal_cases(in_path, out_path):
    raw = np.fromfile(in_path, np.uint8)
    nt = raw.size // 3072
    raw = raw[: nt * 3072].reshape(nt, 3072)
    A = raw[:, :1024].copy().view(np.uint16).reshape(nt, 16, 32)
    B = raw[:, 1024:2048].copy().view(np.uint16).reshape(nt, 32, 16)
    C = raw[:, 2048:].copy().view(np.float32).reshape(nt, 16, 16)
    D = np.fromfile(out_path, np.float32).reshape(nt, 16, 16)
    i = np.arange(16)
    return A[:, i, :].reshape(-1, 32), B[:, :, i].transpose(0, 2, 1).reshape(-1, 32), C[:, i, i].reshape(-1), D[:, i, i].reshape(-1)


def regime_cases(path, tiles):
    raw = open(path, "rb").read()
    nd = 512
    A = np.frombuffer(raw, np.uint16, nd * 512, 0).reshape(nd, 16, 32)
    B = np.frombuffer(raw, np.uint16, nd * 512, nd * 1024).reshape(nd, 32, 16)
    C = np.frombuffer(raw, np.float32, nd * 256, 2 * nd * 1024).reshape(nd, 16, 16)
    D = np.frombuffer(raw, np.float32, nd * 256, 3 * nd * 1024).reshape(nd, 16, 16)
    a = np.repeat(A[:tiles, :, None, :], 16, axis=2).reshape(-1, 32)                    # row i for every column j
    b = np.repeat(B[:tiles].transpose(0, 2, 1)[:, None, :, :], 16, axis=1).reshape(-1, 32)  # column j for every row i
    return a, b, C[:tiles].reshape(-1), D[:tiles].reshape(-1)


def main():
    args = sys.argv[1:]
    regimes = None
    if "--regimes" in args:
        k = args.index("--regimes")
        regimes = args[k + 1]
        args = args[:k]
    parts = [diagonal_cases(args[i], args[i + 1]) for i in range(0, len(args), 2)]
    if regimes:
        for r in range(4):
            parts.append(regime_cases(os.path.join(regimes, f"probe_regime{r}.bin"), 6))
    a = np.concatenate([p[0] for p in parts]); b = np.concatenate([p[1] for p in parts])
    c = np.concatenate([p[2] for p in parts]); d = np.concatenate([p[3] for p in parts])
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_f16_probe.npz")
    np.savez_compressed(out, a_bits=a, b_bits=b, c=c, d_device=d)
    print(len(c), "cases ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
