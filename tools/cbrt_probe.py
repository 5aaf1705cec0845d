import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, oracle_lib as O, raytrace_amd as R
ctx = R.Context(0)
got, n_risky = ctx.selftest_cbrt_grid()
x = np.arange(1 << 24, dtype=np.float32) * np.float32(2.0 ** -24)
exp = np.empty_like(x); O.lib().orc_cbrtf_array(x.ctypes.data, exp.ctypes.data, x.size)
bad = np.flatnonzero(got.view(np.uint32) != exp.view(np.uint32))
print("mismatches", bad.size, "risky", n_risky, "of", 1 << 24, flush=True)
print(bad[:16], got[bad[:16]], exp[bad[:16]])
