"""CPU unit tests of the host scheduler (praline_amd/csrc/sched.cpp, built with g++ without HIP): the pair list ->
wavefront tasks -> launch order -> workgroup descriptors logic that praline_plan_create runs on the GPU box."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, synth_lengths

CSRC = os.path.join(ROOT, "praline_amd", "csrc")
LIB = os.path.join(ROOT, "praline_amd", "libpraline_sched_test.so")
LAG = 2   # PRALINE_MW_LAG


@pytest.fixture(scope="module")
def sched():
    src = os.path.join(CSRC, "sched.cpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", src,
                        os.path.join(CSRC, "cluster.cpp"), "-o", LIB], check=True, cwd=CSRC)
    lib = ctypes.CDLL(LIB)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.praline_sched_test.argtypes = [vp, i64, vp, ctypes.c_int, ctypes.c_int, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    lib.praline_sched_test.restype = ctypes.c_int

    def run(lens, pairs, want_paths=False, xcd_group=-1, wave_slots=2048):
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        cap_t = len(pairs) + 8 * 1024 + 64
        cap_w = cap_t
        n_t, n_w, n_s = i64(0), i64(0), i64(0)
        tf = np.zeros((cap_t, 4), np.int32)
        lp = np.zeros((cap_t, 32), np.int32)
        wf = np.zeros((cap_w, 6), np.int32)
        rc = lib.praline_sched_test(lens.ctypes.data, len(pairs), pairs.ctypes.data, int(want_paths), xcd_group, wave_slots,
                                    cap_t, cap_w, ctypes.byref(n_t), tf.ctypes.data, lp.ctypes.data, ctypes.byref(n_w),
                                    wf.ctypes.data, ctypes.byref(n_s))
        assert rc == 0
        return tf[:n_t.value], lp[:n_t.value], wf[:n_w.value], n_s.value
    return run


def all_pairs(n):
    i, j = np.triu_indices(n, k=1)
    return np.stack([i, j], axis=1).astype(np.int32)


def check_tasks(lens, pairs, tf, lp):
    seen = lp[lp >= 0]
    assert np.array_equal(np.sort(seen), np.arange(len(pairs)))            # every pair in exactly one lane
    for t in range(len(tf)):
        ps = lp[t][lp[t] >= 0]
        if len(ps) == 0:
            assert tf[t, 1] == 0                                            # placement padding
            continue
        assert np.all(pairs[ps, 1] == tf[t, 0])                             # one shared sequence two per task
        assert tf[t, 1] == lens[pairs[ps, 0]].max()                         # max_l1
        assert tf[t, 2] == (lens[tf[t, 0]] + 31) // 32                      # strips of 32 columns
        assert np.all(np.diff(lens[pairs[ps, 0]]) <= 0)                     # partners sorted by length


def test_tasks_cover_the_pair_list(sched):
    rng = np.random.default_rng(0)
    for n, mu in ((2, 40), (9, 70), (70, 150), (200, 400)):
        lens = synth_lengths(rng, n, mu)
        for pairs in (all_pairs(n), np.array([(i, j) for i in range(n) for j in range(n) if i != j and (3 * i + j) % 4 == 0], np.int32)):
            if len(pairs) == 0:
                continue
            for want_paths in (False, True):
                tf, lp, wf, ns = sched(lens, pairs, want_paths=want_paths)
                check_tasks(lens, pairs, tf, lp)
                if want_paths:
                    # path plans carry the same workgroup lists (the forward fill of their two-pass scheme can run on
                    # the scores kernel): same tasks, same lists
                    tf0, lp0, wf0, ns0 = sched(lens, pairs, want_paths=False)
                    assert np.array_equal(wf, wf0) and ns == ns0 and np.array_equal(tf[:, :3], tf0[:, :3])


def test_xcd_placement_keeps_groups_on_one_xcd(sched):
    rng = np.random.default_rng(1)
    n = 420
    lens = synth_lengths(rng, n, 300)
    pairs = all_pairs(n)
    tf0, lp0, _, _ = sched(lens, pairs, xcd_group=0)
    assert np.all(tf0[:, 1] > 0)                                            # no padding without placement
    cost = tf0[:, 2].astype(np.int64) * 100000 + tf0[:, 1]
    assert np.all(np.diff(cost) <= 0)                                       # longest first
    G = 16
    tf, lp, _, _ = sched(lens, pairs, xcd_group=G)
    check_tasks(lens, pairs, tf, lp)
    assert len(tf) % (8 * G) == 0
    # the i-th task of the unplaced order sits at block 8 q + x with g = i // G, x = g % 8, q = (g // 8) * G + i % G
    key0 = [tuple(sorted(r[r >= 0])) for r in lp0]
    key = {tuple(sorted(r[r >= 0])): b for b, r in enumerate(lp) if (r >= 0).any()}
    for i, k in enumerate(key0):
        g = i // G
        assert key[k] == 8 * ((g // 8) * G + i % G) + g % 8


def barriers(nstrips, it, W):
    return max(r * LAG + (((nstrips - r + W - 1) // W) if nstrips > r else 0) * it for r in range(W))


def test_shared_wave_descriptors(sched):
    rng = np.random.default_rng(2)
    # (3 072 slots: the plans of one-hot arenas, whose lookup kernels run three waves per SIMD)
    for n, slots in ((40, 2048), (256, 2048), (300, 2048), (128, 512), (256, 3072), (330, 3072)):
        lens = synth_lengths(rng, n, 400)
        pairs = all_pairs(n)
        tf, lp, wf, ns = sched(lens, pairs, wave_slots=slots)
        real = np.nonzero(tf[:, 1] > 0)[0]
        if len(real) >= slots:
            assert len(wf) == 0 and ns > 0
            continue
        assert len(wf) > 0 and ns == 0
        busy = [(d[:4] >= 0).any() for d in wf]                              # (idle padding workgroups exit at once)
        assert 4 * sum(busy) <= slots                                        # every working workgroup resident
        for q in range(0, len(wf), 8):                                       # padding only at the end of an XCD's list
            for x in range(8):
                if not busy[q + x]:
                    assert not any(busy[q + x::8])
        leaders = []
        for task4, share, nbar in zip(wf[:, :4], wf[:, 4], wf[:, 5]):
            assert share in (1, 2, 4)
            if share == 1:
                assert nbar == 0
                leaders += [t for t in task4 if t >= 0]
                continue
            want = 0
            for slot in range(0, 4, share):
                t = task4[slot]
                assert np.all(task4[slot + 1:slot + share] == -1)           # only the group's first slot names the task
                if t < 0:
                    continue
                leaders.append(t)
                nstrips, it = tf[t, 2], (tf[t, 1] - 1) // 12 + 1
                assert nstrips >= share and it >= share * LAG                # hand-off distances stay >= LAG iterations
                want = max(want, barriers(nstrips, it, share))
            assert nbar == want                                              # what every wave of the workgroup executes
        assert sorted(leaders) == sorted(real.tolist())                      # every task exactly once


def test_four_singles_keep_the_xcd_queues(sched):
    rng = np.random.default_rng(3)
    n = 720
    lens = synth_lengths(rng, n, 200)
    tf, lp, wf, ns = sched(lens, all_pairs(n))
    assert len(wf) == 0 and ns == (len(tf) + 31) // 32 * 8


# ---- pipeline workgroups (k_dp_pipe): sets of 32 sequences one, tasks, workgroup items ------------------------------
PIPE_MAX_TASKS, PIPE_MIN_STEPS = 4, 36   # PRALINE_PIPE_MAX_TASKS, PRALINE_PIPE_MIN_STEPS (dp_types.h)


@pytest.fixture(scope="module")
def pipe_sched(sched):
    lib = ctypes.CDLL(LIB)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.praline_sched_pipe_test.argtypes = [vp, i64, i64, vp, ctypes.c_int, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.praline_sched_pipe_test.restype = ctypes.c_int

    def run(lens, pairs, block_twos=0, wg_slots=0):
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        cap = len(pairs) + 4096
        n_i, n_t, n_s = i64(0), i64(0), i64(0)
        items = np.zeros((cap, 6), np.int32)
        tasks = np.zeros((cap, 3), np.int32)
        lp = np.zeros((cap, 32), np.int32)
        so = np.zeros((cap, 32), np.int32)
        rc = lib.praline_sched_pipe_test(lens.ctypes.data, len(lens), len(pairs), pairs.ctypes.data, block_twos, wg_slots, cap, cap,
                                         cap, ctypes.byref(n_i), ctypes.byref(n_t), ctypes.byref(n_s), items.ctypes.data,
                                         tasks.ctypes.data, lp.ctypes.data, so.ctypes.data)
        assert rc in (0, 1)
        return rc == 0, items[:n_i.value], tasks[:n_t.value], lp[:n_t.value], so[:n_s.value]
    return run


def check_pipe(lens, pairs, items, tasks, lp, so):
    seen = lp[lp >= 0]
    assert np.array_equal(np.sort(seen), np.arange(len(pairs)))            # every pair in exactly one lane of one task
    covered = np.zeros(len(tasks), bool)
    for set_id, task0, ntasks, nstrips, rsteps, nrounds in items:
        assert 1 <= ntasks <= PIPE_MAX_TASKS
        assert not covered[task0:task0 + ntasks].any()
        covered[task0:task0 + ntasks] = True
        ones = so[set_id]
        max_l1 = lens[ones[ones >= 0]].max()
        assert ones[0] >= 0 and lens[ones[0]] == max_l1                     # lane 0 holds the longest sequence of the set
        assert np.all(np.diff(lens[ones[ones >= 0]]) <= 0)                  # ... and the set is sorted by length
        assert rsteps % 12 == 0 and rsteps >= max(PIPE_MIN_STEPS, max_l1 + 1) and rsteps < max(PIPE_MIN_STEPS, max_l1 + 1) + 12
        assert nstrips == tasks[task0:task0 + ntasks, 2].sum() and nrounds == (nstrips + 3) // 4
        for t in range(task0, task0 + ntasks):
            two, t_max_l1, t_strips = tasks[t]
            assert t_max_l1 == max_l1 and t_strips == (lens[two] + 31) // 32
            lanes = np.nonzero(lp[t] >= 0)[0]
            assert len(lanes) >= 1
            assert np.all(pairs[lp[t][lanes], 1] == two)                    # one shared sequence two per task
            assert np.array_equal(pairs[lp[t][lanes], 0], ones[lanes])      # lane l of every task of the set = sequence ones[l]
    assert covered.all()


def test_pipe_schedule_covers_the_pair_list(pipe_sched):
    rng = np.random.default_rng(5)
    for n, mu in ((40, 60), (130, 200), (256, 400), (70, 20)):
        lens = synth_lengths(rng, n, mu)
        for kind in ("triangle", "ordered", "one-vs-all"):
            if kind == "triangle":
                pairs = all_pairs(n)
            elif kind == "ordered":
                pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)
            else:
                pairs = np.array([(i, 3) for i in range(n) if i != 3], dtype=np.int32)
            ok, items, tasks, lp, so = pipe_sched(lens, pairs)
            # (tiny triangles leave most lanes of their few sets empty: the schedule may decline them)
            assert ok or (kind == "triangle" and n < 100), (n, kind)
            if ok:
                check_pipe(lens, pairs, items, tasks, lp, so)
            ok, items, tasks, lp, so = pipe_sched(lens, pairs, block_twos=5, wg_slots=64)
            if ok:
                check_pipe(lens, pairs, items, tasks, lp, so)


def test_pipe_schedule_fits_the_slots_for_small_batches(pipe_sched):
    """Up to 2.5 tasks per workgroup slot every item is resident at once: at most wg_slots items, and the longest item is
    within a round of the best possible split of the tasks (the bound is found by bisection)."""
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, 256, 400)
    pairs = all_pairs(256)
    ok, items, tasks, lp, so = pipe_sched(lens, pairs)
    assert ok and len(items) <= 512
    cost = items[:, 5].astype(np.int64) * items[:, 4]
    assert cost.max() <= 1.3 * cost.sum() / 512                             # C2: longest item within 30 % of the mean load
    # large batch: short single-task items exist for the tail of the launch, the bulk sits in PIPE_MAX_TASKS-task items
    lens = synth_lengths(rng, 1024, 300)
    ok, items, tasks, lp, so = pipe_sched(lens, all_pairs(1024))
    assert ok
    counts = np.bincount(items[:, 2], minlength=PIPE_MAX_TASKS + 1)
    assert counts[PIPE_MAX_TASKS] > counts[1] >= 512
    # launch order of a large batch: by descending cost (rounds x steps), equal costs in list (= set) order
    cost = items[:, 5].astype(np.int64) * items[:, 4]
    assert np.all(np.diff(cost) <= 0)
    same = np.diff(cost) == 0
    assert np.all(np.diff(items[:, 0])[same] >= 0)


def test_pipe_schedule_declines_what_does_not_fit(pipe_sched):
    """Lists whose lanes would stay mostly empty (sparse random pairs over many sequences) and lists with a repeated pair
    keep the task schedule."""
    rng = np.random.default_rng(7)
    n = 600
    lens = synth_lengths(rng, n, 100)
    sparse = np.stack([rng.integers(0, n, 700), rng.integers(0, n, 700)], axis=1).astype(np.int32)
    sparse = np.unique(sparse[sparse[:, 0] != sparse[:, 1]], axis=0)
    ok, *_ = pipe_sched(lens, sparse)
    assert not ok
    twice = np.concatenate([all_pairs(40), all_pairs(40)[:3]])
    ok, *_ = pipe_sched(lens[:40], twice)
    assert not ok


def test_threaded_passes_give_the_one_thread_schedules(sched, pipe_sched, monkeypatch):
    """Lists of 65 536 pairs and more are scheduled with their passes (counting sorts, set / task / lane assignment, task
    records, path slots) cut into slices for a small thread pool: every schedule must come out exactly as on one thread,
    whatever the thread count - task lists, lane assignment, workgroup descriptors, pipeline items."""
    rng = np.random.default_rng(17)
    n = 300
    lens = synth_lengths(rng, n, 120)
    ordered = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)     # 89 700 pairs
    shuffled = ordered.copy()
    rng.shuffle(shuffled)
    for pairs in (ordered, shuffled[:70000]):
        out = {}
        for threads in ("1", "3", "8"):
            monkeypatch.setenv("PRALINE_SCHED_THREADS", threads)
            out[threads] = (sched(lens, pairs, want_paths=True), sched(lens, pairs, want_paths=False), pipe_sched(lens, pairs),
                            pipe_sched(lens, pairs, block_twos=-32))     # (negative: the one-thread permutation version of step 1)
        for threads in ("3", "8"):
            for a, b in zip(out["1"], out[threads]):
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), threads
        for a, b in zip(out["1"][2], out["1"][3]):
            assert np.array_equal(a, b)
    monkeypatch.delenv("PRALINE_SCHED_THREADS")
    # the same pair twice: not for the pipeline layout, in either version
    dup = np.concatenate([ordered[:70000], ordered[123:124]])
    assert not pipe_sched(lens, dup)[0] and not pipe_sched(lens, dup, block_twos=-32)[0]


def test_thread_pool_back_to_back_with_changing_thread_counts(pipe_sched, monkeypatch):
    """The scheduler's pool threads poll for the next pass before they sleep and every pool thread acknowledges every pass,
    also those beyond the pass's thread count: many schedules back to back with the thread count changing from call to
    call (and pauses long enough for the workers to fall asleep) give the one-thread schedule every time."""
    import time
    rng = np.random.default_rng(23)
    n = 380
    lens = synth_lengths(rng, n, 100)
    pairs = all_pairs(n)                                                   # 72 010 pairs: above the threading threshold
    monkeypatch.setenv("PRALINE_SCHED_THREADS", "1")
    want = pipe_sched(lens, pairs)
    assert want[0]
    for k in range(60):
        monkeypatch.setenv("PRALINE_SCHED_THREADS", str(int(rng.integers(2, 9))))
        got = pipe_sched(lens, pairs)
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), k
        if k % 20 == 19:
            time.sleep(0.02)
    monkeypatch.delenv("PRALINE_SCHED_THREADS")


def test_scheduler_threads_survive_a_fork():
    """praline_init starts the scheduler's threads; a host program that forks afterwards has none of them in the child. The
    pool forgets them there (pthread_atfork) and starts new ones: a forked child schedules a large list like its parent.
    (In a subprocess with a time limit: the failure mode is a child that waits for ever.)"""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import ctypes, os, numpy as np
        lib = ctypes.CDLL(%r)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        lib.praline_sched_pipe_test.argtypes = [vp, i64, i64, vp, ctypes.c_int, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
        def run(lens, pairs):
            cap = len(pairs) + 4096
            n_i, n_t, n_s = i64(0), i64(0), i64(0)
            items = np.zeros((cap, 6), np.int32); tasks = np.zeros((cap, 3), np.int32)
            lp = np.zeros((cap, 32), np.int32); so = np.zeros((cap, 32), np.int32)
            rc = lib.praline_sched_pipe_test(lens.ctypes.data, len(lens), len(pairs), pairs.ctypes.data, 0, 0, cap, cap, cap, ctypes.byref(n_i),
                                             ctypes.byref(n_t), ctypes.byref(n_s), items.ctypes.data, tasks.ctypes.data, lp.ctypes.data, so.ctypes.data)
            return rc, items[:n_i.value].copy()
        rng = np.random.default_rng(1); n = 400
        lens = rng.integers(50, 400, n).astype(np.int32)
        pairs = np.array([(i, j) for i in range(n) for j in range(i + 1, n)], dtype=np.int32)
        os.environ["PRALINE_SCHED_THREADS"] = "6"
        a = run(lens, pairs)
        pid = os.fork()
        if pid == 0:
            b = run(lens, pairs)
            os._exit(0 if (b[0] == a[0] and np.array_equal(a[1], b[1])) else 3)
        _, status = os.waitpid(pid, 0)
        b = run(lens, pairs)
        raise SystemExit(0 if (os.WEXITSTATUS(status) == 0 and np.array_equal(a[1], b[1])) else 4)
    ''') % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "praline_amd", "libpraline_sched_test.so")
    done = subprocess.run([sys.executable, "-c", code], timeout=120, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                          start_new_session=True)
    assert done.returncode == 0
