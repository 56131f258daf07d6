import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, one_hot
from praline_amd import native as nat
from oracle import oracle as orc
nat.init(0)
S = load_golden("bba0184_inputs.npz")["blosum62"]
rng = np.random.default_rng(5)
for (L1, L2) in [(5,5),(32,32),(33,32),(32,33),(40,64),(64,40),(10,65),(65,10),(70,70),(100,100),(103,108)]:
    bad = 0; tot = 0
    for rep in range(6):
        a = one_hot(rng.integers(0, 20, L1), 27); b = one_hot(rng.integers(0, 20, L2), 27)
        ar = nat.Arena([a, b], S)
        pl = nat.Plan(ar, np.array([[0, 1]], np.int32)); pl.run("global", -11, -1); s = pl.scores()[0]; pl.close(); ar.close()
        r = orc.pairwise_score_fast("global", a, b, S, -11.0, -1.0)
        tot += 1; bad += (s != r)
        if s != r and rep == 0: print("   first mismatch", L1, L2, s, r)
    print(L1, L2, "bad %d/%d" % (bad, tot))
