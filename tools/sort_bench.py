"""micro-benchmark of the exact std::sort emulation (per-kernel HIP-event times)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import erasor_amd
g = erasor_amd.Erasor(erasor_amd.params_default())
rng = np.random.default_rng(0)
for n, r in ((512, 100), (1024, 200), (4096, 800), (4096, 1 << 30), (16000, 3000), (127000, 21000), (127000, 1 << 30), (262144, 50000)):
    k = rng.integers(0, r, n).astype(np.uint32)
    v = np.arange(n, dtype=np.uint32)
    g.exact_sort_u32(k, v)
    g.profiling(1); g.profile_reset()
    for _ in range(5):
        g.exact_sort_u32(k, v)
    p = g.profile_get(); g.profiling(0)
    print("n=%7d range=%10d : " % (n, r) + "  ".join("%s %.1f us (%d)" % (a, ms / 5 * 1e3, c // 5) for a, (ms, c) in sorted(p.items()) if "esort" in a))
