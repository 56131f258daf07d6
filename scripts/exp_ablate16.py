"""Kernel time of the float scores kernel (C2 and one rank's share of C4, global) under each library build in
variants/ (scripts/build_variant.sh; ablation builds give wrong results by design - only the time is read).
  python scripts/exp_ablate16.py [name ...]     (no names: every variants/libpraline_dp_*.so, after the default build)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
out = []
for name, N, seed, shard in (("C2", 256, 2, None), ("C4/8", 4096, 4, 3)):
    if os.environ.get("ABL_SKIP_C4") == "1" and shard is not None: continue
    rng = np.random.default_rng(seed); lens = synth_lengths(rng, N, 400)
    pairs = allpairs.enumerate_pairs(N)
    if shard is not None: pairs = pairs[allpairs.shard_columns(lens, pairs, 8)[shard]]
    profs = [synth_profile(rng, int(L)) for L in lens]
    ar = nat.Arena(profs, S)
    for mode in os.environ.get("ABL_MODES", "global").split(","):
        pl = nat.Plan(ar, pairs); pl.run(mode, -11, -1)
        ms = []
        for _ in range(7 if shard is None else 3):
            pl.run(mode, -11, -1); ms.append(pl.kernel_ms())
        cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
        out.append("%%s %%s %%.3f ms %%.0f GCUPS" %% (name, mode, float(np.median(ms)), cells / float(np.median(ms)) / 1e6))
        kn = pl.kernel_name(); pl.close()
    ar.close()
print("%%-28s %%s  [%%s]" %% (os.environ.get("ABL_NAME", "default"), " | ".join(out), kn), flush=True)
''' % ROOT
names = sys.argv[1:]
libs = [("default", None)]
if names:
    libs += [(n, os.path.join(ROOT, "variants", "libpraline_dp_%s.so" % n)) for n in names if n != "default"]
    if "default" not in names: libs = libs[1:]
else:
    libs += [(os.path.basename(p)[len("libpraline_dp_"):-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "variants", "libpraline_dp_*.so")))]
for name, path in libs:
    env = dict(os.environ, ABL_NAME=name)
    if path: env["PRALINE_LIB"] = path
    subprocess.call([sys.executable, "-c", CHILD], env=env)
