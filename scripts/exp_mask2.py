"""The MFMA-fed k_dp_batch<.., LOCAL, traceback, MASK = 2> instance (wrong local scores in round 2, not
instantiated in the product).  Experiment builds (scripts/build_variant.sh <name> -DPRALINE_EXP_BATCH_MASK2 with
VARIANT_BATCH=1) send plans with more than PRALINE_MAX_RECTS rectangles per pair to that instance when
PRALINE_EXP_BATCH_MASK2=1, and count how often a lane SEES a non-zero column-mask word and whether a second, volatile,
load of the same word returns zero.  Inputs: those of tests/test_gpu_reference_order.py::test_many_rectangles_per_pair."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, ctypes, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import one_hot
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from oracle import oracle as orc
nat.init(0)
S = blosum62_matrix()
rng = np.random.default_rng(37)
lens = [60, 75, 48, 66, 90]
profs = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
floats_unused = [None for L in lens]
from conftest import synth_profile
_ = [synth_profile(rng, L)[0] for L in lens]
pairs = np.array([(i, j) for i in range(5) for j in range(5) if i != j], dtype=np.int32)
rects = []
for k, (i, j) in enumerate(pairs):
    n = [0, 3, 7, 12][k %% 4]
    rl = []
    for _ in range(n):
        y0, x0 = int(rng.integers(1, lens[i])), int(rng.integers(1, lens[j]))
        rl.append((y0, min(lens[i], y0 + int(rng.integers(0, 9))), x0, min(lens[j], x0 + int(rng.integers(0, 9)))))
    rects.append(rl)
arena = nat.Arena(profs, S)
plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
plan.run("local", -11.0, -1.0)
sc = plan.scores()
name = plan.kernel_name(); plan.close()
bad = []
for k, (i, j) in enumerate(pairs):
    zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)]
    s_or, _p = orc.pairwise_align("local", [profs[i]], [profs[j]], [S], (-11.0, -1.0), zero_idxs=zero or None)
    if sc[k] != np.float32(s_or): bad.append((k, int(i), int(j), len(rects[k]), float(sc[k]), float(s_or)))
dbg = np.zeros(128, np.uint32)
fn = getattr(nat.lib(), "praline_debug_read_10", None)
rc = fn(dbg.ctypes.data_as(ctypes.c_void_p)) if fn is not None else -1
print(os.path.basename(os.environ["PRALINE_LIB"]), name, "wrong scores:", bad)
print("   lanes that saw a non-zero mask word: %%d, of which the volatile re-load returned 0: %%d" %% (dbg[0], dbg[1]))
for n in range(min(12, int(dbg[0]))):
    p, sy, z, again, lane, L1 = dbg[8 + 6 * n: 14 + 6 * n]
    print("   pair %%d (%%d rectangles) strip %%d row %%d lane %%d L1 %%d: zrow %%08x, re-load %%08x" %% (p, len(rects[p]) if p < len(rects) else -1, sy >> 16, sy & 0xffff, lane, L1, z, again))
''' % (ROOT, ROOT)
for lib in sorted(glob.glob(os.path.join(ROOT, "variants", "libpraline_dp_mask2*.so"))):
    env = dict(os.environ, PRALINE_LIB=lib, PRALINE_EXP_BATCH_MASK2="1")
    subprocess.call([sys.executable, "-c", CHILD], env=env)
