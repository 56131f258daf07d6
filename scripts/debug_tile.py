import sys, os, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, one_hot
from praline_amd import native as nat
from oracle import oracle as orc
nat.init(0)
S = load_golden("bba0184_inputs.npz")["blosum62"]
rng = np.random.default_rng(5)
seqs = [one_hot(rng.integers(0, 20, L), 27) for L in (40, 37, 50, 45, 70)]
ar = nat.Arena(seqs, S)
L = nat.lib(); L.praline_debug_tile.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int]*5 + [ctypes.c_void_p]
for tp in (1, 2):
    lane_one = np.full(64, -1, np.int32); lane_one[:3] = [0, 1, 2]; lane_one[32:34] = [3, 1]
    out = np.zeros((64, 32), np.float32)
    two0, two1, x0, y = 4, 2, 32, 7
    rc = L.praline_debug_tile(ar._h, lane_one.ctypes.data, two0, two1, x0, y, tp, out.ctypes.data); assert rc == 0
    def expect(one, two):
        m = orc.build_scores_fma([seqs[one]], [seqs[two]], [S])
        row = np.zeros(32, np.float32); n = min(32, m.shape[1] - x0); row[:n] = m[y - 1, x0:x0 + n]; return row
    for lane, (o, t) in {0: (0, two0), 1: (1, two0), 2: (2, two0), 32: (3, two1), 33: (1, two1)}.items():
        if tp == 1 and lane >= 32: continue
        ok = np.array_equal(out[lane], expect(o, t))
        print("tp", tp, "lane", lane, "ok" if ok else "MISMATCH", out[lane][:10], expect(o, t)[:10])
