#!/usr/bin/env python3
"""Per-kernel difference of two step_profile tables measured on the same box: python scripts/ab_diff.py A.txt B.txt"""
import re, sys
def load(f):
    d = {}
    for l in open(f):
        if l.startswith('--'):
            break
        m = re.match(r'(\S.*?)\s+(\d+\.\d)\s+(\d+\.\d+)\s+(\d+\.\d+)%', l)
        if m:
            d[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[0] * a.get(k, (0, 0))[1])):
    ca, ua = a.get(k, (0, 0)); cb, ub = b.get(k, (0, 0))
    ta += ca * ua; tb += cb * ub
    if abs(cb * ub - ca * ua) >= 2.0:
        print(f"{k:50s} A {ca:5.0f} x {ua:7.2f} = {ca*ua:7.0f}   B {cb:5.0f} x {ub:7.2f} = {cb*ub:7.0f}   d {cb*ub-ca*ua:+7.0f}")
print(f"total A {ta:.0f} us, B {tb:.0f} us, d {tb-ta:+.0f}")
