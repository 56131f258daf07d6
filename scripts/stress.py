"""Randomised parity stress: random batch shapes through every kernel path (scores-only with shared waves /
singles / table / staged stream, paths in task and chain mode, masks) against the oracle DP on the device's match
scores (integer scoring: also against the reference order).  usage: stress.py [seconds] [seed]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix, nucleotide_matrix
from oracle import oracle as orc
from conftest import synth_profile
nat.init(0)
MODES = ["global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_cases = n_pairs_checked = 0
t_progress = time.time()
def dp_on_m(mode, m, rects=None, gaps=None):
    g1, g2 = gaps if gaps is not None else orc.gap_arrays(m.shape[0], m.shape[1], (-11.0, -1.0))
    zero = None
    if rects:
        zero = [(y, x) for (y0, y1, x0, x1) in rects for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)]
    return orc.raw_pairwise_align(mode, np.ascontiguousarray(m), g1, g2, zero)
while time.time() < t_end:
    t_case = time.time()
    kind = rng.choice(["onehot", "profile", "dna", "wide"])
    # a quarter of the batches in the reference-order match-score mode (PRALINE_MATCH_REFERENCE); wide alphabets and
    # plans with many rectangles take that path by themselves
    ref_mode = rng.random() < 0.25
    nat.set_match_mode("ref" if ref_mode else None)
    N = int(rng.choice([64, 128, 200, 300])) if os.environ.get("STRESS_BIG") == "1" else int(rng.choice([2, 3, 5, 9, 17, 33, 48]))
    if ref_mode or kind == "wide":
        N = min(N, 64)   # (the reference-order path runs at ~20-45 GCUPS: a 300-sequence batch of it would take minutes)
    mu = int(rng.choice([3, 20, 40, 70, 130, 260, 520])) if kind != "dna" else int(rng.choice([50, 300, 900]))
    lens = np.maximum(1, rng.integers(max(1, mu // 2), mu * 3 // 2 + 1, N))
    if kind == "dna":
        S, A = nucleotide_matrix(), 15
        profs = [np.eye(A, dtype=np.float32)[rng.integers(0, 4, int(L))] for L in lens]
    elif kind == "wide":
        A = int(rng.integers(34, 70))
        S = rng.normal(0, 3, (A, A)).astype(np.float32)
        profs = []
        for L in lens:
            c = np.zeros((int(L), A), dtype=np.float32)
            for _ in range(int(rng.integers(1, 6))):
                c[np.arange(int(L)), rng.integers(0, A, int(L))] += rng.integers(1, 4, int(L))
            profs.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
    elif kind == "onehot":
        S, A = blosum62_matrix(), 27
        profs = [np.eye(A, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
    else:
        S, A = blosum62_matrix(), 27
        profs = [synth_profile(rng, int(L))[0] for L in lens]
    allp = np.array([(i, j) for i in range(N) for j in range(N)], dtype=np.int32)
    pairs = allp[rng.random(len(allp)) < rng.choice([0.15, 0.5, 1.0])]
    if len(pairs) == 0:
        continue
    mode = MODES[int(rng.integers(0, 5))]
    want_paths = bool(rng.integers(0, 2))
    rects = None
    if want_paths and mode == "local" and rng.random() < 0.5:
        rects = []
        for (i, j) in pairs:
            k = int(rng.integers(0, 3)) if rng.random() < 0.8 else int(rng.integers(0, 9))
            rr = []
            for _ in range(k):
                y0 = int(rng.integers(1, lens[i] + 1)); x0 = int(rng.integers(1, lens[j] + 1))
                rr.append((y0, min(int(lens[i]), y0 + int(rng.integers(0, 12))), x0, min(int(lens[j]), x0 + int(rng.integers(0, 12)))))
            rects.append(rr)
    # path plans: a third through the two-pass scheme (forward fill + block recompute) whatever their size
    # (of those, the float-profile global plans with their forward fill on the scores kernel, PRALINE_TB_KEEP)
    # score plans of few tasks: half on the shared-wave score kernels, half wherever the schedule's estimate sends them
    # (mostly the flag-free chain fill)
    if rng.random() < 0.5: os.environ["PRALINE_SCORES_CHAIN"] = "0"
    else: os.environ.pop("PRALINE_SCORES_CHAIN", None)
    os.environ["PRALINE_TB_TWOPASS"] = "2" if rng.random() < 0.33 else "0"
    os.environ["PRALINE_TB_KEEP"] = "1" if rng.random() < 0.5 else "0"
    if os.environ.get("STRESS_VERBOSE") == "1":
        print("batch %d: kind=%s ref=%s N=%d mu=%d pairs=%d mode=%s paths=%s rects=%s TWOPASS=%s KEEP=%s" % (
            n_cases, kind, ref_mode, N, mu, len(pairs), mode, want_paths, rects is not None, os.environ["PRALINE_TB_TWOPASS"],
            os.environ["PRALINE_TB_KEEP"]), flush=True)
    arena = nat.Arena(profs, S)
    # one batch in eight with per-position gap scores (praline_plan_run_gaps: the dense-tile instances with per-position gap scores)
    ppg = None
    if rng.random() < 0.125 and N <= 64:
        exact = kind in ("onehot", "dna")
        ppg = [np.stack([-rng.integers(8, 60, int(L)) / 4.0, -rng.integers(1, 12, int(L)) / 4.0], axis=1).astype(np.float32) if exact
               else np.stack([-rng.uniform(2.0, 15.0, int(L)), -rng.uniform(0.1, 3.0, int(L))], axis=1).astype(np.float32) for L in lens]
        arena.set_gap_scores(ppg)
    plan = nat.Plan(arena, pairs, want_paths=want_paths, rects=rects)
    mk = plan.match_kind()
    if ppg is not None: plan.run_gaps(mode)
    else: plan.run(mode, -11.0, -1.0)
    sc = plan.scores()
    paths = plan.paths() if want_paths else None
    kname = plan.kernel_name()
    packed = plan.paths_packed() if want_paths else None
    plan.close()
    if want_paths and ppg is None and rng.random() < 0.25:
        # the same plan under another configuration (other pass scheme, other scratch budget = other chunking): EVERY
        # score and path must be identical (races between chunks show up here, not in 24 sampled pairs)
        keep_env = {k: os.environ.get(k) for k in ("PRALINE_TB_TWOPASS", "PRALINE_TB_BUDGET_MB")}
        os.environ["PRALINE_TB_TWOPASS"] = "0" if os.environ.get("PRALINE_TB_TWOPASS") == "2" else "2"
        # (a 40 MB budget cuts plans of long sequences into one-task chunks - one wave per launch: minutes per batch)
        bud = rng.choice(["", "40", "500", "3000"]) if int(lens.max()) * N < 40000 else rng.choice(["", "500", "3000"])
        if bud:
            os.environ["PRALINE_TB_BUDGET_MB"] = str(bud)
        if os.environ.get("STRESS_VERBOSE") == "1":
            print("   second configuration: TWOPASS=%s budget=%r" % (os.environ["PRALINE_TB_TWOPASS"], bud), flush=True)
        plan2 = nat.Plan(arena, pairs, want_paths=True, rects=rects)
        plan2.run(mode, -11.0, -1.0)
        sc2 = plan2.scores()
        packed2 = plan2.paths_packed()
        k2 = plan2.kernel_name()
        plan2.close()
        for kk, vv in keep_env.items():
            if vv is None:
                os.environ.pop(kk, None)
            else:
                os.environ[kk] = vv
        same = np.array_equal(sc.view(np.uint32), sc2.view(np.uint32)) and np.array_equal(packed[2], packed2[2])
        if same:   # (the buffers hold one slot of capacity L1 + L2 + 2 per pair: compare the rows in use)
            for q in range(len(pairs)):
                if not np.array_equal(packed[0][packed[1][q]:packed[1][q] + packed[2][q]], packed2[0][packed2[1][q]:packed2[1][q] + packed2[2][q]]):
                    same = False
                    print("  pair %d (%d, %d): paths differ" % (q, pairs[q][0], pairs[q][1]), flush=True)
                    break
        if not same:
            print("CONFIGURATIONS DIFFER kind=%s N=%d mu=%d mode=%s pairs=%d: %s vs %s (budget %r): %d scores, %d path lengths differ" % (
                kind, N, mu, mode, len(pairs), kname, k2, bud, int((sc != sc2).sum()), int((packed[2] != packed2[2]).sum())), flush=True)
            sys.exit(1)
    check = rng.permutation(len(pairs))[:24]
    for k in check:
        i, j = pairs[k]
        m = arena.match_scores(int(i), int(j), mk)
        s_or, p_or = dp_on_m(mode, m, rects[k] if rects else None, (ppg[i], ppg[j]) if ppg is not None else None)
        if sc[k] != np.float32(s_or) or (want_paths and not np.array_equal(paths[k], p_or)):
            print("MISMATCH kind=%s N=%d mu=%d mode=%s paths=%s rects=%s pair=(%d,%d) lens=(%d,%d) dev=%r oracle=%r" % (
                kind, N, mu, mode, want_paths, rects[k] if rects else None, i, j, lens[i], lens[j], sc[k], s_or), flush=True)
            print("  kernel %s  match kind %d  ref_mode %s  TWOPASS=%s KEEP=%s  pairs in plan %d  index %d" % (
                kname, mk, ref_mode, os.environ.get("PRALINE_TB_TWOPASS"), os.environ.get("PRALINE_TB_KEEP"), len(pairs), k), flush=True)
            if want_paths:
                pd, po = np.asarray(paths[k]), np.asarray(p_or)
                print("  path rows dev %d oracle %d" % (len(pd), len(po)), flush=True)
                q = 1
                while q <= min(len(pd), len(po)) and np.array_equal(pd[-q], po[-q]):
                    q += 1
                print("  first difference from the end at row -%d: dev %s oracle %s" % (q, pd[-q] if q <= len(pd) else None, po[-q] if q <= len(po) else None), flush=True)
            out = os.path.join(ROOT, "gpurun_out", "stress_mismatch.npz")
            os.makedirs(os.path.dirname(out), exist_ok=True)
            np.savez_compressed(out, lens=lens, pairs=pairs, k=k, mode=mode, kind=kind, S=S, dev_path=np.asarray(paths[k]) if want_paths else 0,
                                or_path=np.asarray(p_or) if want_paths else 0, p_i=profs[i], p_j=profs[j], m=m,
                                profs=np.concatenate(profs, axis=0))
            sys.exit(1)
        if kind not in ("profile", "wide") and rects is None and ppg is None and k % 3 == 0:
            ref = orc.pairwise_score_fast(mode, profs[i], profs[j], S, -11.0, -1.0)
            assert sc[k] == np.float32(ref), ("reference order", kind, mode, i, j)
        if mk == 2 and k % 2 == 0:
            # the device's reference-order match scores are the reference's, bit for bit
            m_ref = np.zeros_like(m)
            orc.cext_build_scores([profs[i]], [profs[j]], [orc.build_nonzero_matrix(profs[i])], [orc.build_nonzero_matrix(profs[j])], [S], m_ref)
            assert np.array_equal(m.view(np.uint32), m_ref.view(np.uint32)), ("reference-order match scores", kind, mode, i, j)
        n_pairs_checked += 1
    arena.close()
    n_cases += 1
    if time.time() - t_case > 20:
        print("  (slow batch: %.0f s, kind=%s N=%d mu=%d mode=%s paths=%s ref_mode=%s)" % (time.time() - t_case, kind, N, mu, mode, want_paths, ref_mode), flush=True)
    if time.time() - t_progress > 60:      # (a silent GPU job is taken to be hung after a few minutes)
        t_progress = time.time()
        print("  ... %d batches, %d pairs checked" % (n_cases, n_pairs_checked), flush=True)
nat.set_match_mode(None)
print("stress ok: %d random batches, %d pairs checked against the oracle" % (n_cases, n_pairs_checked))
