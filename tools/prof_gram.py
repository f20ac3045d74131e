#!/usr/bin/env python3
"""Tiny driver for profiling the LOBPCG panel kernels at cfg5 size under rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optimization_amd import capi
m, ns = 126 ** 3, int(sys.argv[1]) if len(sys.argv) > 1 else 72
c = capi.Context(0)
rng = np.random.default_rng(0)
S = c.upload(rng.normal(size=m * ns)); AS = c.upload(rng.normal(size=m * ns))
for _ in range(3):
    c.lobpcg_gram(m, S, ns, AS, ns)
    c.lobpcg_gram(m, S, ns, S, ns)
c.close()
