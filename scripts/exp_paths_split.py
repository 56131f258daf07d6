"""Where the with-paths time goes: C2 (float profiles, 32 640 pairs, chain mode) and C3 (1024 one-hot seqs, all
1 047 552 ordered pairs, task mode, chunked) - run under rocprofv3 --kernel-trace --stats."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from bench import make_workload, synth_lengths, one_hot
nat.init(0)
w = make_workload("c2")
ii, jj = np.triu_indices(256, k=1)
pairs = np.stack([ii, jj], axis=1).astype(np.int32)
cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
ar = nat.Arena(w["profs"], w["S"])
for mode in ("global",):
    pl = nat.Plan(ar, pairs, want_paths=True)
    pl.run(mode, -11, -1); nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): pl.run(mode, -11, -1)
    nat.synchronize(); t1 = time.perf_counter()
    print("C2 float %-16s %.2f ms  %.0f GCUPS  kernel_ms(events) %.2f  %s" % (mode, (t1-t0)/5*1e3, cells*5/(t1-t0)/1e9, pl.kernel_ms(), pl.kernel_name()), flush=True)
    pl.close()
ar.close()
rng = np.random.default_rng(3)
l3 = synth_lengths(rng, 1024, 250)
a3 = nat.Arena([one_hot(rng.integers(0, 20, int(L)), 27) for L in l3], w["S"])
i3, j3 = np.divmod(np.arange(1024 * 1024, dtype=np.int64), 1024)
p3 = np.stack([i3[i3 != j3], j3[i3 != j3]], axis=1).astype(np.int32)
c3 = int((l3[p3[:, 0]].astype(np.int64) * l3[p3[:, 1]]).sum())
for mode in ("global", "local"):
    pl = nat.Plan(a3, p3, want_paths=True)
    pl.run(mode, -11, -1); nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): pl.run(mode, -11, -1)
    nat.synchronize(); t1 = time.perf_counter()
    print("C3 onehot %-16s %.2f ms  %.0f GCUPS  kernel_ms(events) %.2f" % (mode, (t1-t0)/3*1e3, c3*3/(t1-t0)/1e9, pl.kernel_ms()), flush=True)
    pl.close()
a3.close()
