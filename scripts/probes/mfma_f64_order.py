"""Analyse scripts/probes/mfma_f64_order.hip output: which evaluation order reproduces D bit for bit?"""
import itertools, struct, sys
from fractions import Fraction
import numpy as np

def fma(a, b, c):            # exact fused multiply-add, one rounding
    return float(Fraction(a) * Fraction(b) + Fraction(c))

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe.bin"
raw = open(path, "rb").read()
T = struct.unpack("i", raw[:4])[0]
v = np.frombuffer(raw[4:], dtype=np.float64)
A = v[:T * 64].reshape(T, 16, 4); v = v[T * 64:]
B = v[:T * 64].reshape(T, 4, 16); v = v[T * 64:]
C = v[:T * 256].reshape(T, 16, 16); v = v[T * 256:]
D = v[:T * 256].reshape(T, 16, 16)

cands = {}
for perm in itertools.permutations(range(4)):
    cands["chain" + "".join(map(str, perm))] = ("chain", perm)
cands["exact_single_rounding"] = ("exact", None)
for perm in itertools.permutations(range(4)):
    cands["dotfirst" + "".join(map(str, perm))] = ("dotfirst", perm)   # dot = chain from 0, then + c
hits = {k: 0 for k in cands}
total = 0
for t in range(T):
    for i in range(16):
        for j in range(0, 16, 5):
            a = [float(A[t, i, k]) for k in range(4)]; b = [float(B[t, k, j]) for k in range(4)]; c = float(C[t, i, j]); d = float(D[t, i, j])
            total += 1
            for name, (kind, perm) in cands.items():
                if kind == "chain":
                    r = c
                    for k in perm: r = fma(a[k], b[k], r)
                elif kind == "exact":
                    r = float(sum(Fraction(a[k]) * Fraction(b[k]) for k in range(4)) + Fraction(c))
                else:
                    r = 0.0
                    for k in perm: r = fma(a[k], b[k], r)
                    r = r + c
                hits[name] += (r == d)
print("samples", total)
for name, h in sorted(hits.items(), key=lambda kv: -kv[1])[:8]:
    print(f"{name:28s} {h}/{total}")
