"""Randomised check of the device preprofile stage (component.build_preprofiles: all masters, paths never leave the GPU,
Waterman-Eggert masks updated on the device) against the component chain through the serial manager
(Global/LocalMasterSlaveAligner + ProfileBuilder per master, praline/component/preprofile.py:114-269, profile.py:41-74):
random sequence sets, modes, score thresholds and 1-4 Waterman-Eggert iterations; the count tracks must be equal.
usage: stress_preprofile.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import component as comp, container as ct, core, native

native.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
idx = core.TypeIndex(); idx.autoregister()
serial = core.Manager(idx)
blosum = ct.blosum62()
T_IN = [[ct.TRACK_ID_INPUT]]

def run_one(component, keys, **inputs):
    ex = core.Execution(serial, "root")
    ex.add_task(component).environment(core.Environment({}), core.Environment(dict(keys))).inputs(**inputs)
    return core.run(ex)[0]

t_end = time.time() + budget
t_print = time.time()
n_cases = n_masters = 0
while time.time() < t_end:
    n = int(rng.choice([2, 3, 6, 12, 25]))
    mu = int(rng.choice([5, 30, 70, 150]))
    base = rng.integers(0, 20, 2 * mu)
    seqs = []
    for i in range(n):
        L = int(rng.integers(max(1, mu // 2), mu * 3 // 2 + 1))
        v = base[:L].copy() if rng.random() < 0.7 else rng.integers(0, 20, L)
        flip = rng.random(L) < rng.choice([0.05, 0.3])
        v[flip] = rng.integers(0, 20, int(flip.sum()))
        seqs.append(ct.Sequence("s%d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))]))
    mode = str(rng.choice(["global", "local"]))
    kw = {}
    if rng.random() < 0.4:
        kw["score_threshold"] = float(rng.choice([0.0, 10.0, 40.0, 150.0]))
    if mode == "local":
        kw["waterman_eggert_iterations"] = int(rng.integers(1, 5))
    tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode, **kw)
    component = comp.GlobalMasterSlaveAligner if mode == "global" else comp.LocalMasterSlaveAligner
    masters = rng.permutation(n)[:6]
    for master in masters:
        slaves = [s for k, s in enumerate(seqs) if k != master]
        out = run_one(component, kw, master_sequence=seqs[master], slave_sequences=slaves, track_id_sets=T_IN, score_matrices=[blosum])
        prof = run_one(comp.ProfileBuilder, {}, alignment=out['alignment'], track_id=ct.TRACK_ID_INPUT)
        if not np.array_equal(tracks[master].counts, prof['profile_track'].counts):
            print("MISMATCH n=%d mu=%d mode=%s kw=%s master=%d" % (n, mu, mode, kw, master), flush=True)
            print("  sequences %s" % [s_.get_track(ct.TRACK_ID_INPUT).values.tolist() for s_ in seqs], flush=True)
            sys.exit(1)
        n_masters += 1
    n_cases += 1
    if time.time() - t_print > 60:
        t_print = time.time()
        print("  ... %d sets, %d masters compared" % (n_cases, n_masters), flush=True)
print("stress_preprofile ok: %d random sets, %d masters equal to the component chain" % (n_cases, n_masters))
