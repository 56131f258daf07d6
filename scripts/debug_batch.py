import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_golden, one_hot
from praline_amd import native as nat
from oracle import oracle as orc
nat.init(0)
d = load_golden("synthetic_c1.npz"); S = load_golden("bba0184_inputs.npz")["blosum62"]
profs = [one_hot(d["seq%d" % i], 27) for i in range(8)]
print("lens", [p.shape[0] for p in profs])
arena = nat.Arena(profs, S)
def ref(i, j, mode="global"):
    return orc.pairwise_score_fast(mode, profs[i], profs[j], S, -11.0, -1.0)
# single pair plans
for (i, j) in [(0,1),(1,0),(0,2),(3,5)]:
    pl = nat.Plan(arena, np.array([[i, j]], np.int32)); pl.run("global", -11, -1); print("single", i, j, pl.scores()[0], ref(i, j)); pl.close()
# group plans sharing two
for two in (7, 5):
    pairs = np.array([(i, two) for i in range(8) if i != two], np.int32)
    pl = nat.Plan(arena, pairs); pl.run("global", -11, -1); sc = pl.scores(); pl.close()
    print("two", two, [(int(i), float(s), ref(i, two), profs[i].shape[0]) for (i, _), s in zip(pairs, sc)])
# same-length test: identical sequences as different ones
profs2 = [profs[0].copy() for _ in range(4)] + [profs[1]]
ar2 = nat.Arena(profs2, S)
pairs = np.array([(i, 4) for i in range(4)], np.int32)
pl = nat.Plan(ar2, pairs); pl.run("global", -11, -1); print("identical ones:", pl.scores(), ref(0, 1)); pl.close()
