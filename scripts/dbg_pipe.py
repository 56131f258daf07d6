import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from conftest import synth_lengths, synth_profile, load_golden
from praline_amd import native as nat
nat.init(0)
S = load_golden("bba0184_inputs.npz")["blosum62"]
rng = np.random.default_rng(41); N = 70
lens = synth_lengths(rng, N, 90); lens[:6] = [1, 2, 31, 33, 35, 37]; lens[6], lens[7] = 150, 73
profs = [synth_profile(rng, int(L))[0] for L in lens]
arena = nat.Arena(profs, S)
print(arena.info())
allp = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
os.environ["PRALINE_PIPE_MIN_TASKS"] = "1"; os.environ["PRALINE_TB_PIPE"] = "1"
for pairs in (allp, allp[allp[:,0] < allp[:,1]], allp[(allp[:,0] > 7) & (allp[:,1] > 7)]):
    plan = nat.Plan(arena, pairs, want_paths=True)
    plan.run("global", -11.0, -1.0)
    print(len(pairs), plan.kernel_name())
    plan.close()
    plan = nat.Plan(arena, pairs)
    plan.run("global", -11.0, -1.0)
    print(len(pairs), 'scores-only', plan.kernel_name())
    plan.close()
