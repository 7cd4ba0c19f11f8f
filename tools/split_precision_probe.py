"""Accuracy of an fp32 GEMM emulated with split low-precision MFMA operands (numpy emulation, no GPU): the basis of the
"what comes next" note in DESIGN.md section 9.  A (rows x K) post-ReLU activations, W (K x N) ~ N(0, 1/K); the operands are
split into 2 or 3 bf16 / fp16 parts (round to nearest, residual split again), the listed part-products are summed exactly
(the MFMA accumulates fp32 products of 16-bit operands exactly; fp32 accumulation error comes on top as in any fp32 GEMM).
Printed: max |error| / max |result| of the truncation alone, next to a plain fp32 BLAS GEMM against float64."""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def split(x, rnd, n):
    parts, r = [], x.astype(np.float32).copy()
    for _ in range(n):
        h = rnd(r)
        parts.append(h)
        r = (r - h).astype(np.float32)
    return parts


rs = np.random.RandomState(0)
rows, K, N = 4096, 128, 128
schemes = ((2, [(0, 0), (0, 1), (1, 0)]), (2, [(0, 0), (0, 1), (1, 0), (1, 1)]), (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]))
for name, s in (("activations O(1)", 1.0), ("activations O(30)", 30.0), ("activations O(1e-2)", 1e-2)):
    A = np.maximum(rs.randn(rows, K), 0).astype(np.float32) * s
    A[:, 0] = 5 * s
    W = (rs.randn(K, N) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(ref).max()
    print("%-20s plain fp32 GEMM %.1e" % (name, np.abs((A @ W) - ref).max() / scale))
    for rn, rnd in (("bf16", bf16), ("fp16", f16)):
        for nsp, terms in schemes:
            a, w = split(A, rnd, nsp), split(W, rnd, nsp)
            acc = sum(a[i].astype(np.float64) @ w[j].astype(np.float64) for i, j in terms)
            print("    %s x%d (MFMA rate 16/%d = %.1fx fp32): %.1e" % (rn, len(terms), len(terms), 16 / len(terms), np.abs(acc - ref).max() / scale))
