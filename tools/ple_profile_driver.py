import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import m4ri_amd
from m4ri_amd.mzd import Mzd
m4ri_amd.init(0)
n = int(sys.argv[1])
A = Mzd.random(n, n, 1)
m4ri_amd.pin(A)
t = time.perf_counter(); r, P, Q = m4ri_amd.mzd_ple(A); print("ple", n, time.perf_counter() - t, r)
m4ri_amd.unpin(A)
