#!/usr/bin/env python3
"""Golden vectors for the NVQ dequantiser: an independent numpy float32 restatement of the formulas in the reference's
JVectorIndexQuantization.java:319-361 (nvqDequantize / logisticNQT / logitNQT).  Run from the repo root:
    python tests/golden/make_nvq_golden.py  ->  tests/golden/nvq_golden.json
Inputs are small hand-picked records; the expected outputs are what the Java code computes with IEEE float32
(fma emulated through float64: the products of float32 values are exact in float64)."""
import json
import os

import numpy as np

F = np.float32


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def java_round(x):
    r = np.floor(F(x))
    return int(r) + (1 if F(x) - F(r) >= F(0.5) else 0)


def logistic_nqt(value, alpha, x0):
    temp = fma(value, alpha, F(-F(alpha) * F(x0)))
    p = java_round(F(temp + F(0.5)))
    f = fma(F(temp - F(p)), F(0.5), F(1))
    m = np.array([f], dtype=np.float32).view(np.int32)[0]
    t = np.array([np.int32(np.uint32(m) + np.uint32((p << 23) & 0xFFFFFFFF))], dtype=np.int32).view(np.float32)[0]
    return F(t / F(t + F(1)))


def logit_nqt(scaled, inv_alpha, x0):
    z = F(F(scaled) / F(F(1) - F(scaled)))
    temp = np.array([z], dtype=np.float32).view(np.int32)[0]
    e = temp & 0x7f800000
    p = F((e >> 23) - 128)
    m = np.array([np.int32((temp & 0x007fffff) + 0x3f800000)], dtype=np.int32).view(np.float32)[0]
    return F(F(F(m + p) * F(inv_alpha)) + F(x0))


def dequantize(params, codes, mean, sizes):
    out = []
    off = 0
    for s, size in enumerate(sizes):
        growth, midpoint, minv, maxv = [F(v) for v in params[s]]
        delta = F(maxv - minv)
        sg = F(growth / delta)
        sm = F(midpoint * delta)
        bias = logistic_nqt(minv, sg, sm)
        scale = F(F(logistic_nqt(maxv, sg, sm) - bias) / F(255))
        inv = F(F(1) / sg)
        for i in range(size):
            sv = fma(F(codes[off + i]), scale, bias)
            out.append(logit_nqt(sv, inv, sm))
        off += size
    return [float(F(F(o) + F(mu))) for o, mu in zip(out, mean)]


def main():
    rng = np.random.default_rng(7)
    cases = []
    for d, M in ((12, 2), (7, 3), (16, 1)):
        sizes = [d // M + (1 if m < d % M else 0) for m in range(M)]
        for _ in range(3):
            params = []
            for m in range(M):
                mn = float(F(rng.uniform(-2, -0.1)))
                mx = float(F(rng.uniform(0.1, 2)))
                params.append([float(F(rng.choice([1e-2, 1.0, 4.0]))), float(F(rng.uniform(-0.2, 0.2))), mn, mx])
            codes = [int(c) for c in rng.integers(0, 256, size=d)]
            codes[0], codes[-1] = 0, 255
            mean = [float(F(v)) for v in rng.uniform(-1, 1, size=d)]
            cases.append(dict(d=d, M=M, params=params, codes=codes, mean=mean, expected=dequantize(params, codes, mean, sizes)))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nvq_golden.json")
    json.dump(dict(source="JVectorIndexQuantization.java:319-361 restated in numpy float32", cases=cases), open(out, "w"), indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
