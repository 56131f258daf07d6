"""Operator (plugin) layer of the hot path: same type ids, ports, options, defaults and output keys
as the reference components, with the numeric work on the MI355X through native.py.

  PairwiseAligner       praline/component/align.py:37-251      -> fused device path (arena + plan)
  RawPairwiseAligner    praline/component/align.py:254-447     -> raw device path (m, g1, g2, z given)
  ProfileBuilder        praline/component/profile.py:15-74
  Global/Local/DummyMasterSlaveAligner   praline/component/preprofile.py:19-269  (callers: one
                        device submission per Waterman-Eggert iteration instead of one call per slave)
  GuideTreeBuilder      praline/component/tree.py:18-170       (caller: ONE scores-only submission for
                        all N(N-1)/2 pairs; clustering restated from praline/util/cluster.py:27-114)
  BatchManager          the Manager.execute_many seam (praline/core/manager.py:154-170): a homogeneous
                        list of PairwiseAligner requests becomes one device submission.

There is no CPU arithmetic here: every alignment goes through libpraline_dp.so.
"""
import numpy as np

from . import native
from .container import (Alignment, GapScoreModel, MatchScoreModel, PlainTrack, ProfileTrack,
                        ScoreMatrix, Sequence, SequenceTree, TRACK_ID_INPUT)
from .core import (MESSAGE_KIND_COMPLETE, BeginMessage, CompleteMessage, Component, ComponentError, DataError,
                   Environment, Execution, Manager, Port, ProgressMessage, T)
from .util import (auto_align_mode, compress_path, extend_path_local, get_frequencies,
                   zero_idxs_to_rectangles)

MODES = ("global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two")


# ---- profile preparation (praline/component/align.py:114-189) ------------------------------------
def _track_profile(track):
    """PlainTrack -> one-hot, ProfileTrack -> counts / rowsum, float32 [L, A] (align.py:163-177)."""
    if track.tid == PlainTrack.tid:
        profile = np.zeros((len(track), track.alphabet.size), dtype=np.float32)
        profile[np.arange(len(track)), track.values] = 1.0
        return profile
    if track.tid == ProfileTrack.tid:
        return track.profile.astype(np.float32)
    raise DataError("unknown track type id for this aligner: '{0}'".format(track.tid))


def _validate_track_sets(sequence_one, sequence_two, track_id_sets_one, track_id_sets_two, score_matrices):
    if len(track_id_sets_one) != len(track_id_sets_two):
        raise ComponentError("should have an identical number of track id sets for both sequences")
    for n, (ids_one, ids_two) in enumerate(zip(track_id_sets_one, track_id_sets_two)):
        sm = score_matrices[n]
        if len(ids_one) != 1 or len(ids_two) != 1:
            raise ComponentError("the fast aligner only supports single-track alignments at the moment")
        if sm.matrix.ndim != len(ids_one) + len(ids_two):
            raise ComponentError("the score matrix must consist of as many dimensions as there are "
                                 "tracks to be aligned ({0}), but it contains {1}".format(
                                     len(ids_one) + len(ids_two), sm.matrix.ndim))
        t1 = sequence_one.get_track(ids_one[0])
        t2 = sequence_two.get_track(ids_two[0])
        if sm.alphabets[0].aid != t1.alphabet.aid:
            raise DataError("track 0 for sequence one has alphabet '{0}' but the corresponding dimension "
                            "in the score matrix has alphabet '{1}'".format(t1.alphabet.aid, sm.alphabets[0].aid))
        if sm.alphabets[1].aid != t2.alphabet.aid:
            raise DataError("track 0 for sequence two has alphabet '{0}' but the corresponding dimension "
                            "in the score matrix has alphabet '{1}'".format(t2.alphabet.aid, sm.alphabets[1].aid))


def _normalise_gap_series(gap_series):
    gap_series = list(gap_series)
    if len(gap_series) == 1:
        return [gap_series[0], gap_series[0]]
    if len(gap_series) == 2:
        return gap_series
    raise ComponentError("the fast aligner only supports linear and affine gap penalties at the moment")


def _block_diagonal(score_matrices):
    """Several track sets = one concatenated alphabet with a block-diagonal matrix: the reference sums
    the per-set scores (praline/util/cext.c:389-420)."""
    mats = [np.asarray(sm.matrix, dtype=np.float32) for sm in score_matrices]
    sizes = [max(m.shape) for m in mats]
    S = np.zeros((sum(sizes), sum(sizes)), dtype=np.float32)
    off = 0
    for m, sz in zip(mats, sizes):
        S[off:off + m.shape[0], off:off + m.shape[1]] = m
        off += sz
    return S, sizes


class PairwiseBatch(object):
    """A homogeneous set of PairwiseAligner requests (same track sets, score matrices and gap series)
    executed as ONE device submission per alignment mode."""

    def __init__(self, track_id_sets_one, track_id_sets_two, score_matrices, gap_series):
        self.ids_one = [ids[0] for ids in track_id_sets_one]
        self.ids_two = [ids[0] for ids in track_id_sets_two]
        self.score_matrices = score_matrices
        self.gap_open, self.gap_extend = _normalise_gap_series(gap_series)
        self.S, self.sizes = _block_diagonal(score_matrices)
        self._index = {}      # (id(sequence), role) -> arena index
        self._profiles = []
        self._owner = []      # arena index -> id(sequence)
        self._gaps = {}       # id(sequence) -> float32 [len, 2] per-position (open, extend): set_gap_scores
        self.requests = []    # (mode, arena_one, arena_two, rects or None)

    def set_gap_scores(self, sequence, gap_score_model):
        """Per-position gap scores for one sequence: a GapScoreModel (praline/container/score.py:45-68) or a float
        [len, 2] array of (open, extend) rows.  The reference's fill reads them per position (cext.c:155-158,172-175);
        PairwiseAligner itself only ever builds constant rows (align.py:212-217).  Sequences without a model keep the
        batch's gap series.  Batches with models run on the per-position instances of the batched kernels
        (praline_plan_run_gaps)."""
        g = np.ascontiguousarray(getattr(gap_score_model, 'scores', gap_score_model), dtype=np.float32)
        if g.ndim != 2 or g.shape[1] != 2:
            raise ComponentError("gap scores must have shape (len, 2)")
        self._gaps[id(sequence)] = g

    def _gap_rows(self):
        rows = []
        for sid, prof in zip(self._owner, self._profiles):
            g = self._gaps.get(sid)
            if g is None:
                g = np.tile(np.array([[self.gap_open, self.gap_extend]], dtype=np.float32), (prof.shape[0], 1))
            if g.shape[0] != prof.shape[0]:
                raise ComponentError("gap score model of length %d for a sequence of length %d" % (g.shape[0], prof.shape[0]))
            rows.append(g)
        return rows

    def _arena_index(self, sequence, role):
        key = (id(sequence), role if self.ids_one != self.ids_two else 0)
        if key not in self._index:
            ids = self.ids_one if role == 0 else self.ids_two
            parts = []
            for tid_, sz in zip(ids, self.sizes):
                p = _track_profile(sequence.get_track(tid_))
                if p.shape[1] < sz:
                    p = np.pad(p, ((0, 0), (0, sz - p.shape[1])))
                parts.append(p)
            self._index[key] = len(self._profiles)
            self._profiles.append(np.concatenate(parts, axis=1) if len(parts) > 1 else parts[0])
            self._owner.append(id(sequence))
        return self._index[key]

    def add(self, mode, sequence_one, sequence_two, rects=None):
        if mode not in MODES:
            raise ComponentError("unknown alignment mode: '{0}'".format(mode))
        self.requests.append((mode, self._arena_index(sequence_one, 0), self._arena_index(sequence_two, 1), rects))
        return len(self.requests) - 1

    def scores_for_pairs(self, sequences, ii, jj, modes, on_device=False):
        """Scores only, for the pairs (sequences[ii[k]], sequences[jj[k]]) in mode modes[k] (array of mode names):
        the whole list as one device submission per mode, without a Python-level request per pair (an all-pairs
        stage of 400 sequences is 79 800 requests).  Both roles must use the same track sets.
        on_device: return a float32 torch tensor on the library's device instead of a numpy array - the kernels write
        straight into it (the exchange step of a multi-GPU stage then never moves its shard through host memory)."""
        if self.ids_one != self.ids_two:
            raise ComponentError("scores_for_pairs needs identical track id sets for both sequences")
        ii, jj, modes = np.asarray(ii), np.asarray(jj), np.asarray(modes)
        if on_device:
            import torch
            out = torch.zeros(max(len(ii), 1), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))[:len(ii)]
            torch.cuda.current_stream().synchronize()      # the library's stream must see the zero fill
        else:
            out = np.zeros(len(ii), dtype=np.float32)
        if len(ii) == 0:
            return out
        idx = np.array([self._arena_index(seq, 0) for seq in sequences], dtype=np.int32)
        # one mode for the whole list (the guide tree, a rescoring pass): its plan is scheduled on a second host thread
        # while the arena is created (native.prepare_schedule_async)
        prep = None
        if len(ii) >= 1024 and bool(np.all(modes == modes[0])):
            pairs0 = np.stack([idx[ii], idx[jj]], axis=1).astype(np.int32)
            prep = native.prepare_schedule_async([p.shape[0] for p in self._profiles], pairs0)
        arena = native.Arena(self._profiles, self.S, set_sizes=self.sizes)
        try:
            for mode in MODES:
                sel = np.flatnonzero(modes == mode)
                if len(sel) == 0:
                    continue
                if prep is not None:
                    plan = native.Plan(arena, pairs0, prepared=prep)
                    prep = None
                else:
                    pairs = np.stack([idx[ii[sel]], idx[jj[sel]]], axis=1).astype(np.int32)
                    plan = native.Plan(arena, pairs, want_paths=False)
                try:
                    if on_device and len(sel) == len(ii):
                        plan.run(mode, self.gap_open, self.gap_extend, d_scores=out.data_ptr())
                        native.synchronize()
                    elif on_device:
                        import torch
                        part = torch.zeros(len(sel), dtype=torch.float32, device=out.device)
                        torch.cuda.current_stream().synchronize()
                        plan.run(mode, self.gap_open, self.gap_extend, d_scores=part.data_ptr())
                        native.synchronize()
                        out[torch.as_tensor(sel, device=out.device)] = part
                    else:
                        plan.run(mode, self.gap_open, self.gap_extend)
                        out[sel] = plan.scores()
                finally:
                    plan.close()
        finally:
            arena.close()
        return out

    def run(self, want_paths=True):
        """Returns (scores float list, paths list or None) in request order."""
        n = len(self.requests)
        scores = [None] * n
        paths = [None] * n if want_paths else None
        if n == 0:
            return scores, paths
        arena = native.Arena(self._profiles, self.S, set_sizes=self.sizes)
        try:
            if self._gaps:
                arena.set_gap_scores(self._gap_rows())
            for mode in MODES:
                sel = [k for k, r in enumerate(self.requests) if r[0] == mode]
                if not sel:
                    continue
                pairs = np.array([(self.requests[k][1], self.requests[k][2]) for k in sel], dtype=np.int32)
                rects = None
                if any(self.requests[k][3] for k in sel):
                    rects = [list(self.requests[k][3] or []) for k in sel]
                plan = native.Plan(arena, pairs, want_paths=want_paths or rects is not None, rects=rects)
                try:
                    if self._gaps:
                        plan.run_gaps(mode)
                    else:
                        plan.run(mode, self.gap_open, self.gap_extend)
                    sc = plan.scores()
                    pt = plan.paths() if want_paths else None
                finally:
                    plan.close()
                for q, k in enumerate(sel):
                    scores[k] = float(sc[q])
                    if want_paths:
                        paths[k] = pt[q]
        finally:
            arena.close()
        return scores, paths


def _path_for_output(mode, path):
    """The reference hands back a list of (y, x) tuples for global / local alignments and an int array
    for semiglobal ones (praline/component/align.py:401-433)."""
    if mode.startswith("semiglobal"):
        return np.asarray(path, dtype=int)
    path = np.asarray(path).reshape(-1, 2)
    return list(zip(path[:, 0].tolist(), path[:, 1].tolist()))   # (tuples of Python ints, three times as fast as map(tuple, ...))


class PairwiseAligner(Component):
    """Profile-profile pairwise aligner (praline/component/align.py:37-251): same ports, options and
    outputs.  Match scores (P1 . S . P2^T) and the three-state affine fill, end-cell selection and
    traceback all run on the device; nothing but the score and the path comes back."""
    tid = "praline.component.PairwiseAligner"
    inputs = {'mode': Port(str),
              'sequence_one': Port(Sequence.tid),
              'sequence_two': Port(Sequence.tid),
              'track_id_sets_one': Port([[str]]),
              'track_id_sets_two': Port([[str]]),
              'zero_idxs': Port([(int, int)], optional=True),
              'score_matrices': Port([ScoreMatrix.tid])}
    outputs = {'alignment': Port(Alignment.tid), 'score': Port(float)}
    options = {'gap_series': [float], 'debug': int}
    defaults = {'gap_series': [-11.0, -1.0], 'debug': 0}

    def execute(self, mode, sequence_one, sequence_two, track_id_sets_one, track_id_sets_two, zero_idxs,
                score_matrices):
        _validate_track_sets(sequence_one, sequence_two, track_id_sets_one, track_id_sets_two, score_matrices)
        gap_series = _normalise_gap_series(self.environment['gap_series'])
        if mode not in MODES:
            raise ComponentError("unknown alignment mode: '{0}'".format(mode))
        rects = None
        if zero_idxs:
            rects = zero_idxs_to_rectangles(zero_idxs)
        if zero_idxs and rects is None:
            # arbitrary zero_idxs: dense mask through the raw path (match scores built on the device)
            outputs = self._execute_raw(mode, sequence_one, sequence_two, track_id_sets_one,
                                        track_id_sets_two, zero_idxs, score_matrices, gap_series)
        else:
            batch = PairwiseBatch(track_id_sets_one, track_id_sets_two, score_matrices, gap_series)
            batch.add(mode, sequence_one, sequence_two, rects)
            scores, paths = batch.run(want_paths=True)
            outputs = {'alignment': Alignment([sequence_one, sequence_two], _path_for_output(mode, paths[0])),
                       'score': scores[0]}
        yield CompleteMessage(outputs=outputs)

    def _execute_raw(self, mode, sequence_one, sequence_two, ids_one, ids_two, zero_idxs, score_matrices, gap_series):
        i1 = [_track_profile(sequence_one.get_track(ids[0])) for ids in ids_one]
        i2 = [_track_profile(sequence_two.get_track(ids[0])) for ids in ids_two]
        s = [sm.matrix.astype(np.float32) for sm in score_matrices]
        m = np.zeros((i1[0].shape[0], i2[0].shape[0]), dtype=np.float32)
        native.cext_build_scores(i1, i2, None, None, s, m)
        g1 = np.empty((m.shape[0], 2), dtype=np.float32)
        g2 = np.empty((m.shape[1], 2), dtype=np.float32)
        g1[:, 0], g1[:, 1] = gap_series
        g2[:, 0], g2[:, 1] = gap_series
        z = np.zeros((m.shape[0] + 1, m.shape[1] + 1), dtype=np.uint8)
        for idx in zero_idxs:
            z[idx] = 1
        score, path = native.raw_align(mode, m, g1, g2, z)
        return {'alignment': Alignment([sequence_one, sequence_two], _path_for_output(mode, path)),
                'score': float(score)}


class RawPairwiseAligner(Component):
    """Raw pairwise aligner (praline/component/align.py:254-447): caller-supplied match and gap score
    models; boundary init, fill, end cell, traceback and semiglobal extension on the device.  The
    reference's `accelerate` option is accepted; there is no non-accelerated path here."""
    tid = "praline.component.RawPairwiseAligner"
    inputs = {'mode': Port(str),
              'sequence_one': Port(Sequence.tid),
              'sequence_two': Port(Sequence.tid),
              'match_score_model': Port(MatchScoreModel.tid),
              'gap_score_model_one': Port(GapScoreModel.tid),
              'gap_score_model_two': Port(GapScoreModel.tid),
              'zero_idxs': Port([(int, int)], optional=True)}
    outputs = {'alignment': Port(Alignment.tid), 'score': Port(float)}
    options = {'debug': int, 'accelerate': bool}
    defaults = {'debug': 0, 'accelerate': True}

    def execute(self, mode, sequence_one, sequence_two, match_score_model, gap_score_model_one,
                gap_score_model_two, zero_idxs):
        if mode not in MODES:
            raise ComponentError("unknown alignment mode: '{0}'".format(mode))
        m = np.ascontiguousarray(match_score_model.scores, dtype=np.float32)
        g1 = np.ascontiguousarray(gap_score_model_one.scores, dtype=np.float32)
        g2 = np.ascontiguousarray(gap_score_model_two.scores, dtype=np.float32)
        z = None
        if zero_idxs:
            z = np.zeros((m.shape[0] + 1, m.shape[1] + 1), dtype=np.uint8)
            for idx in zero_idxs:
                z[idx] = 1
        score, path = native.raw_align(mode, m, g1, g2, z)
        alignment = Alignment([sequence_one, sequence_two], _path_for_output(mode, path))
        yield CompleteMessage(outputs={'alignment': alignment, 'score': float(score)})


class ProfileBuilder(Component):
    """Alignment -> per-column symbol counts -> ProfileTrack (praline/component/profile.py:15-74)."""
    tid = "praline.component.ProfileBuilder"
    inputs = {'alignment': Port(Alignment.tid), 'track_id': Port(str)}
    outputs = {'profile_track': Port(ProfileTrack.tid)}
    options = {'debug': int}
    defaults = {'debug': 0}

    def execute(self, alignment, track_id):
        freqs = get_frequencies(alignment, track_id)
        track = alignment.items[0].get_track(track_id)
        yield CompleteMessage(outputs={'profile_track': ProfileTrack(freqs, track.alphabet)})


# ---- the `aligner` seam of the callers ------------------------------------------------------------
def _resolve_aligner(component):
    """What `index.resolve(env['aligner'])` + `task.environment(root_env, sub_env)` give the reference's callers
    (praline/component/tree.py:115-127, preprofile.py:127-139,229-241): the aligner class named by the `aligner`
    option and its EFFECTIVE environment - the aligner's defaults, overridden by the caller's environment,
    overridden by `aligner_env` (Environment.collapse, praline/core/component.py:150-201)."""
    cls = component.manager.index.resolve(component.environment['aligner'])
    return cls, component.environment.collapse(cls, component.environment['aligner_env'])


def _is_device_aligner(cls):
    """The batched device path stands in for an aligner only if that aligner IS this package's PairwiseAligner;
    any other registered component (a user's own aligner under another or even the same type id) gets the
    reference's Execution fan-out, one task per alignment."""
    return cls is PairwiseAligner


def _run_single(component, aligner, **inputs):
    """One alignment through the configured aligner component, as the reference's callers run it; the outputs
    are left in component._last_outputs (a generator cannot return them to a for loop)."""
    execution = Execution(component.manager, component.tag)
    task = execution.add_task(aligner)
    task.environment(component.environment, component.environment['aligner_env'])
    task.inputs(**inputs)
    for message in execution.run():
        yield message
    component._last_outputs = execution.outputs[0]


# ---- callers: master-slave (preprofile) stage ----------------------------------------------------
def _identity_alignment(sequence):
    return Alignment([sequence], np.arange(len(sequence) + 1).reshape(len(sequence) + 1, 1))


def merge_master_slave(master_sequence, slave_sequences, results, threshold, local):
    """Grow the master-slave alignment (praline/component/preprofile.py:146-152,258-265): per slave and
    per (score, path) result that passes the threshold, drop the rows where the master does not
    advance, pad local paths with -1 rows, and merge the slave in."""
    alignment = _identity_alignment(master_sequence)
    for slave, res in zip(slave_sequences, results):
        for score, path in res:
            if threshold is None or score >= threshold:
                path = compress_path(np.array(path, dtype=int), 0)
                if local:
                    path = extend_path_local(path, len(master_sequence), 0)
                alignment = alignment.merge(_identity_alignment(slave), path)
    return alignment


class DummyMasterSlaveAligner(Component):
    """praline/component/preprofile.py:19-63"""
    tid = "praline.component.DummyMasterSlaveAligner"
    inputs = {'master_sequence': Port(Sequence.tid),
              'slave_sequences': Port([Sequence.tid]),
              'track_id_sets': Port([[str]]),
              'score_matrices': Port([ScoreMatrix.tid], optional=True)}
    outputs = {'alignment': Port(Alignment.tid)}
    options = {}
    defaults = {}

    def execute(self, master_sequence, slave_sequences, track_id_sets, score_matrices):
        yield CompleteMessage({'alignment': _identity_alignment(master_sequence)})


class GlobalMasterSlaveAligner(Component):
    """praline/component/preprofile.py:67-156: master vs every slave in global mode, compressed to the
    master's columns and merged.  All slaves are aligned in one device submission."""
    tid = "praline.component.GlobalMasterSlaveAligner"
    inputs = {'master_sequence': Port(Sequence.tid),
              'slave_sequences': Port([Sequence.tid]),
              'track_id_sets': Port([[str]]),
              'score_matrices': Port([ScoreMatrix.tid])}
    outputs = {'alignment': Port(Alignment.tid)}
    options = {'gap_series': [float], 'aligner': str, 'aligner_env': Environment.tid,
               'score_threshold': T(float, nullable=True)}
    defaults = {'gap_series': [-11.0, -1.0], 'aligner': PairwiseAligner.tid,
                'score_threshold': None, 'aligner_env': Environment({})}

    def execute(self, master_sequence, slave_sequences, track_id_sets, score_matrices):
        threshold = self.environment['score_threshold']
        aligner, aligner_env = _resolve_aligner(self)
        if not _is_device_aligner(aligner):
            # another aligner component: the reference's loop, one Execution per slave (preprofile.py:126-154)
            results = []
            for j, s in enumerate(slave_sequences):
                for message in _run_single(self, aligner, mode="global", sequence_one=master_sequence, sequence_two=s,
                                           track_id_sets_one=track_id_sets, track_id_sets_two=track_id_sets,
                                           score_matrices=score_matrices):
                    yield message
                out = self._last_outputs
                results.append([(out['score'], np.array(out['alignment'].path, dtype=int))])
                yield ProgressMessage((j + 1) / float(len(slave_sequences)))
            yield CompleteMessage({'alignment': merge_master_slave(master_sequence, slave_sequences, results, threshold,
                                                                    local=False)})
            return
        for s in slave_sequences:
            _validate_track_sets(master_sequence, s, track_id_sets, track_id_sets, score_matrices)
        batch = PairwiseBatch(track_id_sets, track_id_sets, score_matrices, aligner_env['gap_series'])
        for s in slave_sequences:
            batch.add("global", master_sequence, s)
        scores, paths = batch.run(want_paths=True)
        alignment = merge_master_slave(master_sequence, slave_sequences,
                                       [[(scores[j], paths[j])] for j in range(len(slave_sequences))],
                                       threshold, local=False)
        yield ProgressMessage(1.0)
        yield CompleteMessage({'alignment': alignment})


class LocalMasterSlaveAligner(Component):
    """praline/component/preprofile.py:158-269: local alignments with Waterman-Eggert re-alignment; the
    bounding rectangle of each path is masked for the next iteration (preprofile.py:247-255).  One
    device submission per iteration covers all slaves."""
    tid = "praline.component.LocalMasterSlaveAligner"
    inputs = GlobalMasterSlaveAligner.inputs
    outputs = {'alignment': Port(Alignment.tid)}
    options = {'gap_series': [float], 'aligner': str, 'aligner_env': Environment.tid,
               'waterman_eggert_iterations': int, 'score_threshold': T(float, nullable=True)}
    defaults = {'gap_series': [-11.0, -1.0], 'aligner': PairwiseAligner.tid, 'score_threshold': None,
                'aligner_env': Environment({}), 'waterman_eggert_iterations': 2}

    def execute(self, master_sequence, slave_sequences, track_id_sets, score_matrices):
        threshold = self.environment['score_threshold']
        iterations = self.environment['waterman_eggert_iterations']
        aligner, aligner_env = _resolve_aligner(self)
        if not _is_device_aligner(aligner):
            # another aligner component: the reference's loops, one Execution per slave and iteration with the
            # growing zero_idxs list (preprofile.py:226-265)
            results = []
            for j, s in enumerate(slave_sequences):
                zero_idxs = []
                results.append([])
                for _ in range(iterations):
                    for message in _run_single(self, aligner, mode="local", sequence_one=master_sequence, sequence_two=s,
                                               track_id_sets_one=track_id_sets, track_id_sets_two=track_id_sets,
                                               score_matrices=score_matrices, zero_idxs=zero_idxs):
                        yield message
                    out = self._last_outputs
                    p = np.array(out['alignment'].path, dtype=int)
                    zero_idxs = zero_idxs + [(y, x) for y in range(int(p[:, 0].min()), int(p[:, 0].max()) + 1)
                                             for x in range(int(p[:, 1].min()), int(p[:, 1].max()) + 1)]
                    results[j].append((out['score'], p))
                yield ProgressMessage((j + 1) / float(len(slave_sequences)))
            yield CompleteMessage({'alignment': merge_master_slave(master_sequence, slave_sequences, results, threshold,
                                                                    local=True)})
            return
        gap_series = aligner_env['gap_series']
        for s in slave_sequences:
            _validate_track_sets(master_sequence, s, track_id_sets, track_id_sets, score_matrices)
        rects = [[] for _ in slave_sequences]
        results = [[] for _ in slave_sequences]   # per slave: (score, path) per iteration
        for it in range(iterations):
            # one device submission per iteration, whatever the number of masked rectangles per pair
            batch = PairwiseBatch(track_id_sets, track_id_sets, score_matrices, gap_series)
            for j, s in enumerate(slave_sequences):
                batch.add("local", master_sequence, s, list(rects[j]))
            scores, paths = batch.run(want_paths=True)
            for j in range(len(slave_sequences)):
                p = np.array(paths[j], dtype=int)
                results[j].append((scores[j], p))
                rects[j].append((int(p[:, 0].min()), int(p[:, 0].max()), int(p[:, 1].min()), int(p[:, 1].max())))
        alignment = merge_master_slave(master_sequence, slave_sequences, results, threshold, local=True)
        yield ProgressMessage(1.0)
        yield CompleteMessage({'alignment': alignment})


def _preprofile_slave_counts(profiles, S, pairs, mode, gap_open, gap_extend, score_threshold, iterations, counts_out=None):
    """Device side of build_preprofiles for ONE rank's (master, slave) pair list: `iterations` passes of alignments
    with paths, the counting on the device paths, Waterman-Eggert bounds rank-local.  Returns the int32 count arena
    [sum L][A] (the slaves' share; rows of masters that are not in `pairs` stay zero).  counts_out: optional torch
    int32 CUDA tensor the counts are accumulated in directly (the multi-GPU path all-reduces it in place)."""
    arena = native.Arena(profiles, S)
    try:
        if counts_out is not None:
            arena.counts_bind(counts_out.data_ptr())
        arena.counts_reset()
        if len(pairs) and iterations <= native.MAX_RECTS + 1:
            # ONE plan for all Waterman-Eggert iterations: the path bounding boxes become the next iteration's masks on
            # the device (no second schedule, no bounds / rectangle traffic over PCIe)
            plan = native.Plan(arena, pairs, want_paths=True)
            try:
                for it in range(iterations):
                    plan.run(mode, gap_open, gap_extend)
                    plan.add_counts(score_threshold, local=(mode == "local"))
                    if it + 1 < iterations:
                        plan.mask_path_bounds()
            finally:
                plan.close()
            iterations = 0
        rects = None
        for it in range(iterations if len(pairs) else 0):     # more rectangles per pair than the slots hold
            plan = native.Plan(arena, pairs, want_paths=True, rects=rects)
            try:
                plan.run(mode, gap_open, gap_extend)
                plan.add_counts(score_threshold, local=(mode == "local"))
                if it + 1 < iterations:
                    b = plan.path_bounds().reshape(-1, 1, 4)
                    rects = b if rects is None else np.concatenate([rects, b], axis=1)
            finally:
                plan.close()
        if counts_out is not None:
            native.synchronize()
            return counts_out
        return arena.counts(staged=True)     # (a view of the reused read-back buffer: the caller copies per track)
    finally:
        arena.close()


def build_preprofiles(sequences, track_id, score_matrix, mode="global", gap_series=(-11.0, -1.0),
                      score_threshold=None, waterman_eggert_iterations=2, rank=0, world=1, group=None):
    """The whole preprofile stage in a few device submissions: for EVERY sequence as master,
    what the reference computes with one Global/LocalMasterSlaveAligner execution (preprofile.py:114-156,
    213-269) followed by ProfileBuilder (profile.py:41-74) - N(N-1) alignments with paths, the master-slave
    merge and the symbol counting - without the N(N-1) paths ever leaving the GPU: the counting runs on the
    device paths (native.Plan.add_counts), only the bounding boxes for the Waterman-Eggert masks
    (preprofile.py:247-255) and the count matrix come back.

    sequences: Sequences with a PlainTrack under track_id; mode "global" or "local".
    Returns one ProfileTrack per sequence (identical to the component chain's).

    world > 1 (one process per GPU, torch.distributed group `group`): the masters are dealt to the ranks
    (allpairs.shard_masters, balanced by DP cells), every rank runs all passes for ITS masters - the Waterman-Eggert
    bounds never leave the rank - and ONE all-reduce of the int32 count arena [sum L][A] (RCCL over xGMI; C3: 27 MB)
    gives every rank every master's counts: the workflow's per-master fan-out (praline/component/workflow.py:139-161)
    with one exchange step.  Every rank returns all N tracks."""
    if mode not in ("global", "local"):
        raise ComponentError("the preprofile stage aligns in 'global' or 'local' mode, not '{0}'".format(mode))
    tracks = [seq.get_track(track_id) for seq in sequences]
    for t in tracks:
        if t.tid != PlainTrack.tid:
            raise DataError("build_preprofiles needs plain tracks (got {0})".format(t.tid))
    alphabet = tracks[0].alphabet
    gap_open, gap_extend = _normalise_gap_series(list(gap_series))
    S = np.ascontiguousarray(score_matrix.matrix if hasattr(score_matrix, "matrix") else score_matrix, dtype=np.float32)
    n = len(sequences)
    profiles = [_track_profile(t) for t in tracks]
    lens = np.array([len(t.values) for t in tracks], dtype=np.int64)
    row_off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    # every ordered pair (master i, slave j != i), i outer, j ascending - for this rank's masters
    masters = np.arange(n, dtype=np.int32)
    iterations = waterman_eggert_iterations if mode == "local" else 1
    if world > 1:
        from . import allpairs
        masters = allpairs.shard_masters(lens, world)[rank].astype(np.int32)
    pairs = np.empty((len(masters) * max(n - 1, 0), 2), dtype=np.int32)
    if n > 1:
        pairs[:, 0] = np.repeat(masters, n - 1)
        slave = np.tile(np.arange(n - 1, dtype=np.int32), len(masters))
        pairs[:, 1] = slave + (slave >= pairs[:, 0])          # skip j == i
    counts = _preprofile_counts_exchange(profiles, S, pairs, mode, gap_open, gap_extend, score_threshold, iterations,
                                         world, group)
    # the master advances in every column: its own symbol once per position, for all sequences in one indexed add on the
    # int32 arena (a view of the read-back buffer; ProfileTrack makes its own int copy per track - one int64 copy of the
    # whole arena, 54 MB on C3, cost 25 ms of page faults)
    counts = np.asarray(counts)
    if not counts.flags.writeable:
        counts = counts.copy()
    if n:
        counts[np.arange(int(lens.sum())), np.concatenate([np.asarray(t.values, dtype=np.int64) for t in tracks])] += 1
    return [ProfileTrack(counts[row_off[i]:row_off[i] + lens[i]], alphabet) for i in range(n)]


def _preprofile_counts_exchange(profiles, S, pairs, mode, gap_open, gap_extend, score_threshold, iterations, world, group):
    """This rank's counts and, for world > 1, the exchange step: one all-reduce of the count arena."""
    if world <= 1:
        return _preprofile_slave_counts(profiles, S, pairs, mode, gap_open, gap_extend, score_threshold, iterations)
    import torch
    import torch.distributed as dist
    from . import allpairs
    if dist.get_backend(group) == "nccl":
        # accumulate straight into a torch tensor and reduce it in place: the counts never visit the host before the sum
        rows = int(sum(p.shape[0] for p in profiles))
        t = torch.zeros(rows * S.shape[0], dtype=torch.int32, device="cuda")
        torch.cuda.current_stream().synchronize()      # the library's stream must see the zero fill
        _preprofile_slave_counts(profiles, S, pairs, mode, gap_open, gap_extend, score_threshold, iterations, counts_out=t)
        allpairs.all_reduce_counts(t, group)
        return t.cpu().numpy().reshape(rows, S.shape[0])
    local = _preprofile_slave_counts(profiles, S, pairs, mode, gap_open, gap_extend, score_threshold, iterations)
    return allpairs.all_reduce_counts(local, group)


# ---- callers: guide tree (all-pairs distance stage) ----------------------------------------------
native_clustering = True   # merge_order of 64 clusters and more runs in libpraline_dp.so's host code


def merge_order(distance_matrix, linkage):
    """Agglomerative clustering merge order (praline/util/cluster.py:27-114): repeatedly merge the two
    clusters with the smallest linkage distance (first minimum in cluster-id order, both (i, j) and (j, i)
    present), the merged cluster keeps the id of the first.

    The reference rebuilds the whole cluster-by-cluster table from the N x N distances every round
    (O(N^4) element reads over a run; minutes at N = 400, days at N = 4096).  Here the table lives across
    rounds and a merge touches one row and one column: min / max of the two old rows for single / complete
    linkage (order-free, exact), and for average linkage the float64 SUMS of the member distances are kept
    and divided by the member count, which is what `a.mean()` evaluates (cluster.py:99-114).  Sums of
    integer-valued distances - integer scoring - are exact in any order, so the table is bit-identical to the
    reference's; for float scoring the sums may differ in the last bit of a float64 (from the reference's
    pairwise summation order), which matters only between candidates closer than 1e-16 relative.
    The first minimum is found from per-row (value, column) minima that are only recomputed for the rows
    whose minimum pointed at one of the two merged clusters: O(N^2) work in total for typical inputs."""
    d = np.array(distance_matrix, dtype=float)
    n = d.shape[0]
    if linkage not in ('single', 'complete', 'average'):
        raise KeyError(linkage)
    if n < 2:
        return []
    if n >= 64 and native_clustering:
        # the same algorithm in the library's host code (csrc/cluster.cpp): N = 4096 in tens of milliseconds instead
        # of 2.9 s of numpy calls; tests/test_host_logic.py compares the two
        return native.merge_order(d, linkage)
    link = d.copy()
    np.fill_diagonal(link, np.inf)
    sums, size = (d, np.ones(n)) if linkage == 'average' else (None, None)
    alive = np.ones(n, dtype=bool)
    row_val = link.min(axis=1)
    row_col = link.argmin(axis=1)              # first minimum of the row = lowest cluster id
    order = []
    for _ in range(n - 1):
        one = int(row_val.argmin())            # first row holding the smallest value
        two = int(row_col[one])
        order.append((one, two))
        if linkage == 'average':
            sums[one, :] += sums[two, :]
            sums[:, one] += sums[:, two]
            size[one] += size[two]
            with np.errstate(invalid='ignore'):
                new_row = sums[one, :] / (size[one] * size)
                new_col = sums[:, one] / (size * size[one])
        elif linkage == 'single':
            new_row, new_col = np.minimum(link[one, :], link[two, :]), np.minimum(link[:, one], link[:, two])
        else:
            new_row, new_col = np.maximum(link[one, :], link[two, :]), np.maximum(link[:, one], link[:, two])
        alive[two] = False
        dead = ~alive
        new_row[dead] = np.inf
        new_col[dead] = np.inf
        new_row[one] = new_col[one] = np.inf
        link[one, :] = new_row
        link[:, one] = new_col
        link[two, :] = np.inf
        link[:, two] = np.inf
        row_val[two] = np.inf
        # rows whose minimum sat in a column that changed or died: rescan; any other row can only have gained
        # a new minimum in column `one`
        rescan = alive & ((row_col == one) | (row_col == two))
        rescan[one] = True
        rows = np.flatnonzero(rescan)
        sub = link[rows]
        row_val[rows] = sub.min(axis=1)
        row_col[rows] = sub.argmin(axis=1)
        v = link[:, one]
        better = alive & ~rescan & ((v < row_val) | ((v == row_val) & (one < row_col)))
        row_val[better] = v[better]
        row_col[better] = one
    return order


class GuideTreeBuilder(Component):
    """praline/component/tree.py:18-170.  Only the scores of the N(N-1)/2 alignments are consumed
    (tree.py:142-145), so the whole stage is one scores-only submission (k_dp_split16)."""
    tid = "praline.component.GuideTreeBuilder"
    inputs = {'sequences': Port([Sequence.tid]),
              'track_id_sets': Port([[str]]),
              'score_matrices': Port([ScoreMatrix.tid])}
    outputs = {'guide_tree': Port(SequenceTree.tid)}
    options = {'gap_series': [float], 'aligner': str, 'aligner_env': Environment.tid,
               'linkage_method': str, 'squash_profiles': bool, 'dist_mode': str, 'debug': int}
    defaults = {'gap_series': [-11.0, -1.0], 'aligner': PairwiseAligner.tid, 'aligner_env': Environment({}),
                'linkage_method': 'average', 'squash_profiles': False, 'dist_mode': 'global', 'debug': 0}

    def execute(self, sequences, track_id_sets, score_matrices):
        linkage = self.environment['linkage_method']
        dist_mode = self.environment['dist_mode']
        if linkage not in ('single', 'complete', 'average'):
            raise ComponentError("unknown linkage method '{0}'".format(linkage))
        if dist_mode not in ('semiglobal', 'global', 'semiglobal_auto'):
            raise ComponentError("unknown alignment mode '{0}'".format(dist_mode))
        n = len(sequences)
        fixed = {"semiglobal": "semiglobal_both", "global": "global"}.get(dist_mode)
        ii, jj = np.triu_indices(n, k=1)   # tree.py:105-129: each unordered pair once, i outer, j inner
        if fixed:
            modes = np.full(len(ii), fixed)
        else:                              # auto_align_mode (util/align.py:299-305), for all pairs at once
            lens = np.array([len(s) for s in sequences])
            modes = np.where(lens[ii] > lens[jj], "semiglobal_one", "semiglobal_two")
        aligner, aligner_env = _resolve_aligner(self)
        if not _is_device_aligner(aligner):
            # another aligner component: the reference's fan-out, ONE Execution holding a task per pair
            # (tree.py:104-140); under a batching / parallel manager that is still one execute_many call
            sub_env = self.environment['aligner_env']
            if self.environment['squash_profiles']:
                sub_env.keys['squash_profiles'] = True        # tree.py:118-119
            execution = Execution(self.manager, self.tag)
            for i, j, mode in zip(ii.tolist(), jj.tolist(), modes.tolist()):
                task = execution.add_task(aligner)
                task.environment(self.environment, sub_env)
                task.inputs(mode=mode, sequence_one=sequences[i], sequence_two=sequences[j],
                            track_id_sets_one=track_id_sets, track_id_sets_two=track_id_sets,
                            score_matrices=score_matrices)
            step = 0
            for message in execution.run():
                yield message
                if message.kind == MESSAGE_KIND_COMPLETE and execution.started_task(message.tag):
                    step += 1
                    yield ProgressMessage(step / float(len(ii)))
            scores = np.array([out['score'] for out in execution.outputs], dtype=np.float32)
        else:
            # what the per-pair checks of the aligner test is per sequence (alphabets against the matrices): once each
            for s in sequences:
                _validate_track_sets(s, s, track_id_sets, track_id_sets, score_matrices)
            batch = PairwiseBatch(track_id_sets, track_id_sets, score_matrices, aligner_env['gap_series'])
            scores = self._all_pairs_scores(batch, sequences, ii, jj, modes)
        d = np.zeros((n, n), dtype=np.float32)  # tree.py:99-100,131: diagonal 0
        d[ii, jj] = d[jj, ii] = scores
        self.score_matrix = d
        dist = (-d) + d.max()  # tree.py:147
        tree = SequenceTree(sequences, merge_order(dist, linkage))
        yield CompleteMessage({'guide_tree': tree})

    def _all_pairs_scores(self, batch, sequences, ii, jj, modes):
        """Scores of the pairs (ii[k], jj[k]) in row-major order.  Under a manager that belongs to a process group
        (BatchManager(rank=, world=, group=): one process per GPU) the pair list is sharded by columns, every rank
        aligns its shard on its own device and the score shards are exchanged with ONE all-gather (allpairs.py);
        every rank returns the complete list."""
        world = getattr(self.manager, 'world', 1)
        if world <= 1 or len(ii) == 0:
            return batch.scores_for_pairs(sequences, ii, jj, modes)
        from . import allpairs
        lens = np.array([len(s) for s in sequences], dtype=np.int64)
        pairs = np.stack([ii, jj], axis=1)
        shards = allpairs.shard_columns(lens, pairs, world)
        mine = shards[self.manager.rank]
        # under RCCL the shard stays on the device: kernels -> all-gather -> reorder -> ONE copy of the complete list
        on_device = allpairs.group_is_nccl(self.manager.group)
        local = batch.scores_for_pairs(sequences, ii[mine], jj[mine], np.asarray(modes)[mine], on_device=on_device)
        return allpairs.all_gather_scores(local, shards, self.manager.rank, world, self.manager.group)


# ---- callers: progressive alignment along the guide tree -------------------------------------------
def _count_cluster(index, sequence, track_id_sets):
    """A cluster starts as its sequence with every aligned track turned into integer counts
    (praline/component/msa.py:71-98): plain tracks become one-hot counts, profile tracks keep theirs."""
    cluster = Sequence("Cluster #{0}".format(index), [])
    seen = set()
    for ids in track_id_sets:
        for trid in ids:
            if trid in seen:
                continue
            seen.add(trid)
            track = sequence.get_track(trid)
            if track.tid == PlainTrack.tid:
                counts = np.zeros((len(track), track.alphabet.size), dtype=np.int32)
                counts[np.arange(len(track)), track.values] = 1
            elif track.tid == ProfileTrack.tid:
                counts = np.array(track.counts, dtype=np.int32)
            else:
                raise DataError("unknown track type id for this aligner: '{0}'".format(track.tid))
            cluster.add_track(trid, ProfileTrack(counts, track.alphabet))
    return cluster


def _merge_clusters(one, two, track_id_sets, path):
    """Replace every aligned track of cluster `one` by its merge with `two`'s along `path`
    (container/sequence.py:205-239)."""
    merged = []
    for ids in track_id_sets:
        for trid in ids:
            merged.append((trid, one.get_track(trid).merge(two.get_track(trid), path)))
            one.del_track(trid)
    for trid, track in merged:
        one.add_track(trid, track)


class _Len(object):
    """Stands in for a cluster where only its length is asked for (auto_align_mode)."""
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


class ResidentClusters(object):
    """The growing clusters of a progressive alignment RESIDENT ON THE GPU: one arena holds every initial
    cluster - integer counts and packed profile operands - and every merge step appends the merged cluster in place
    (native.Arena.append_merged: ProfileTrack.merge, praline/container/sequence.py:205-239, on the device path).  Per
    step only the alignment path travels to the host, where Alignment.merge (praline/container/align.py:30-61) keeps
    the sequence-level bookkeeping; no count track is merged or re-uploaded on the host."""

    def __init__(self, sequences, track_id_sets, score_matrices, gap_series):
        for seq in sequences:
            _validate_track_sets(seq, seq, track_id_sets, track_id_sets, score_matrices)
        self.gap_open, self.gap_extend = _normalise_gap_series(gap_series)
        S, sizes = _block_diagonal(score_matrices)
        counts, profiles = [], []
        for i, seq in enumerate(sequences):
            cluster = _count_cluster(i, seq, track_id_sets)          # msa.py:71-98: every aligned track as integer counts
            cparts, pparts = [], []
            for ids, sz in zip(track_id_sets, sizes):
                track = cluster.get_track(ids[0])
                c = np.asarray(track.counts, dtype=np.int32)
                p = _track_profile(track)
                if c.shape[1] < sz:
                    c = np.pad(c, ((0, 0), (0, sz - c.shape[1])))
                    p = np.pad(p, ((0, 0), (0, sz - p.shape[1])))
                cparts.append(c)
                pparts.append(p)
            counts.append(np.concatenate(cparts, axis=1) if len(cparts) > 1 else cparts[0])
            profiles.append(np.concatenate(pparts, axis=1) if len(pparts) > 1 else pparts[0])
        self.arena = native.Arena(profiles, S, set_sizes=sizes)
        n, longest = len(sequences), max(len(c) for c in counts)
        self.arena.set_counts(np.concatenate(counts, axis=0), reserve_seqs=n - 1, reserve_rows=(n - 1) * 2 * longest)
        self.index = {i: i for i in range(n)}                      # cluster id -> arena sequence
        self.length = {i: len(counts[i]) for i in range(n)}

    def cluster(self, cid):
        return _Len(self.length[cid])

    def scores(self, requests):
        """Scores only of (mode, cluster a, cluster b) requests: one device submission per mode."""
        out = [None] * len(requests)
        for mode in MODES:
            sel = [k for k, r in enumerate(requests) if r[0] == mode]
            if not sel:
                continue
            pairs = np.array([(self.index[requests[k][1]], self.index[requests[k][2]]) for k in sel], dtype=np.int32)
            plan = native.Plan(self.arena, pairs, want_paths=False)
            try:
                plan.run(mode, self.gap_open, self.gap_extend)
                sc = plan.scores()
            finally:
                plan.close()
            for q, k in enumerate(sel):
                out[k] = float(sc[q])
        return out

    def align_and_merge(self, mode, a, b):
        """Align clusters a and b, merge b into a along the path on the device; returns (score, path int [rows, 2])."""
        pairs = np.array([(self.index[a], self.index[b])], dtype=np.int32)
        plan = native.Plan(self.arena, pairs, want_paths=True)
        try:
            plan.run(mode, self.gap_open, self.gap_extend)
            score = float(plan.scores()[0])
            path = np.array(plan.paths()[0], dtype=int)
            idx, ln = self.arena.append_merged(plan, 0)
        finally:
            plan.close()
        self.index[a] = idx
        self.length[a] = ln
        del self.index[b]
        del self.length[b]
        return score, path

    def align_and_merge_many(self, requests):
        """(mode, a, b) requests whose clusters are pairwise distinct - merge steps of different subtrees of the guide
        tree: one device submission per mode, then every b is merged into its a on the device.  Returns the (score, path)
        of every request, in order."""
        out = [None] * len(requests)
        merged = {}
        for mode in MODES:
            sel = [k for k, r in enumerate(requests) if r[0] == mode]
            if not sel:
                continue
            pairs = np.array([(self.index[requests[k][1]], self.index[requests[k][2]]) for k in sel], dtype=np.int32)
            plan = native.Plan(self.arena, pairs, want_paths=True)
            try:
                plan.run(mode, self.gap_open, self.gap_extend)
                sc = plan.scores()
                paths = plan.paths()
                for q, k in enumerate(sel):
                    out[k] = (float(sc[q]), np.array(paths[q], dtype=int))
                for k, m in zip(sel, self.arena.append_merged_many(plan, np.arange(len(sel)))):
                    merged[k] = m
            finally:
                plan.close()
        for k, (_, a, b) in enumerate(requests):
            self.index[a], self.length[a] = merged[k]
            del self.index[b]
            del self.length[b]
        return out

    def close(self):
        self.arena.close()


def merge_levels(steps):
    """Group the merge steps (i, j) of a guide tree (cluster j is merged into cluster i) into levels of mutually
    independent steps: a step's level is one more than the latest level that produced one of its two clusters.  Steps
    of one level touch disjoint clusters, so they can run as one batch; the result of every step - and therefore the
    final alignment - does not depend on the order in which independent steps are carried out."""
    last, levels = {}, []
    for k, (i, j) in enumerate(steps):
        lvl = max(last.get(i, 0), last.get(j, 0))
        if lvl == len(levels):
            levels.append([])
        levels[lvl].append(k)
        last[i] = lvl + 1
        last.pop(j, None)
    return levels


class TreeMultipleSequenceAligner(Component):
    """praline/component/msa.py:18-248: N-1 profile-profile alignments in guide-tree order; after each
    one the two clusters' count tracks and sub-alignments are merged along the path
    (container/sequence.py:205-239, container/align.py:30-61).  The steps depend on each other, so every
    one is its own device submission through the configured aligner component."""
    tid = "praline.component.TreeMultipleSequenceAligner"
    inputs = {'sequences': Port([Sequence.tid]),
              'guide_tree': Port(SequenceTree.tid),
              'track_id_sets': Port([[str]]),
              'score_matrices': Port([ScoreMatrix.tid])}
    outputs = {'alignment': Port(Alignment.tid)}
    options = {'gap_series': [float], 'aligner': str, 'aligner_env': Environment.tid, 'merge_mode': str,
               'debug': int, 'log_track_ids': [str]}
    defaults = {'gap_series': [-11.0, -1.0], 'aligner': PairwiseAligner.tid, 'aligner_env': Environment({}),
                'merge_mode': 'semiglobal', 'debug': 0, 'log_track_ids': [TRACK_ID_INPUT]}

    def execute(self, sequences, guide_tree, track_id_sets, score_matrices):
        merge_mode = self.environment['merge_mode']
        if merge_mode not in ("global", "semiglobal", "semiglobal_auto"):
            raise ComponentError("unknown merge mode '{0}'".format(merge_mode))
        aligner, aligner_env = _resolve_aligner(self)
        alignments = {i: _identity_alignment(seq) for i, seq in enumerate(sequences)}
        steps = list(guide_tree.merge_orders)
        self.steps = []      # (mode, score, path) of every merge step, in order (diagnostics / tests)
        if _is_device_aligner(aligner) and isinstance(self.manager, BatchManager) and steps:
            # resident path: the clusters live and grow on the GPU, a step brings back only its path
            resident = ResidentClusters(sequences, track_id_sets, score_matrices, aligner_env['gap_series'])
            # Steps of different subtrees do not depend on each other: every LEVEL of the guide tree (merge_levels) is
            # one device submission per mode instead of one per step - the same alignments, scores and paths (each step
            # sees exactly the clusters it would see in the reference's serial order, msa.py:124-237), in far fewer
            # latency-bound round trips.
            self.steps = [None] * len(steps)
            self.levels = merge_levels(steps)
            done = 0
            try:
                for level in self.levels:
                    requests = []
                    for k in level:
                        i, j = steps[k]
                        if merge_mode == "global":
                            mode = "global"
                        elif merge_mode == "semiglobal":
                            mode = "semiglobal_both"
                        else:
                            mode = auto_align_mode(resident.cluster(i), resident.cluster(j))
                        requests.append((mode, i, j))
                    if len(requests) == 1:
                        results = [resident.align_and_merge(*requests[0])]
                    else:
                        results = resident.align_and_merge_many(requests)
                    for k, (mode, i, j), (score, path) in zip(level, requests, results):
                        self.steps[k] = (mode, score, path)
                        alignments[i] = alignments[i].merge(alignments[j], path)
                        del alignments[j]
                        done += 1
                        yield ProgressMessage(done / float(len(steps)))
            finally:
                resident.close()
            yield CompleteMessage(outputs={'alignment': list(alignments.values())[0]})
            return
        clusters = {i: _count_cluster(i, seq, track_id_sets) for i, seq in enumerate(sequences)}
        for done, (i, j) in enumerate(steps):
            one, two = clusters[i], clusters[j]
            if merge_mode == "global":
                mode = "global"
            elif merge_mode == "semiglobal":
                mode = "semiglobal_both"
            else:
                mode = auto_align_mode(one, two)
            execution = Execution(self.manager, self.tag)
            task = execution.add_task(aligner)
            task.environment(self.environment, self.environment['aligner_env'])
            task.inputs(mode=mode, sequence_one=one, sequence_two=two, track_id_sets_one=track_id_sets,
                        track_id_sets_two=track_id_sets, score_matrices=score_matrices)
            for message in execution.run():
                yield message
            path = np.array(execution.outputs[0]['alignment'].path)
            self.steps.append((mode, execution.outputs[0]['score'], path))
            _merge_clusters(one, two, track_id_sets, path)
            alignments[i] = alignments[i].merge(alignments[j], path)
            del clusters[j]
            del alignments[j]
            yield ProgressMessage((done + 1) / float(len(steps)))
        yield CompleteMessage(outputs={'alignment': list(alignments.values())[0]})


class AdHocMultipleSequenceAligner(Component):
    """praline/component/msa.py:250-558: no guide tree - every round scores all pairs of current clusters
    (dist_mode), joins the best-scoring pair with a merge_mode alignment and repeats.  Scores of pairs that
    do not involve the cluster changed in the previous round are reused (msa.py:508-519), so a round is one
    fan-out of at most n-1 alignments: one device submission under BatchManager."""
    tid = "praline.component.AdHocMultipleSequenceAligner"
    inputs = {'sequences': Port([Sequence.tid]),
              'track_id_sets': Port([[str]]),
              'score_matrices': Port([ScoreMatrix.tid])}
    outputs = {'alignment': Port(Alignment.tid)}
    options = {'gap_series': [float], 'aligner': str, 'aligner_env': Environment.tid, 'merge_mode': str,
               'dist_mode': str, 'debug': int, 'log_track_ids': [str]}
    defaults = {'gap_series': [-11.0, -1.0], 'aligner': PairwiseAligner.tid, 'aligner_env': Environment({}),
                'merge_mode': 'semiglobal', 'dist_mode': 'global', 'debug': 0, 'log_track_ids': [TRACK_ID_INPUT]}

    @staticmethod
    def _align_mode(kind, one, two):
        if kind == "global":
            return "global"
        if kind == "semiglobal":
            return "semiglobal_both"
        return auto_align_mode(one, two)

    def _fan_out(self, requests, track_id_sets, score_matrices, scores_only=False):
        """Run (mode, cluster_one, cluster_two) requests as ONE Execution; leaves the outputs in
        self._last_outputs (a generator cannot return them to a for loop).  scores_only: the caller reads
        nothing but 'score' - under the batching manager and the default aligner that is one scores-only
        device submission (k_dp_split16) instead of alignments with paths."""
        aligner = self.manager.index.resolve(self.environment['aligner'])
        if scores_only and aligner is PairwiseAligner and isinstance(self.manager, BatchManager) and len(requests) > 1:
            for _, one, two in requests:
                _validate_track_sets(one, two, track_id_sets, track_id_sets, score_matrices)
            # the aligner's effective environment: ours, overridden by aligner_env (Execution.add_task / collapse)
            gap_series = self.environment['aligner_env'].get('gap_series', self.environment['gap_series'])
            batch = PairwiseBatch(track_id_sets, track_id_sets, score_matrices, gap_series)
            for mode, one, two in requests:
                batch.add(mode, one, two)
            scores, _ = batch.run(want_paths=False)
            self._last_outputs = [{'score': sc} for sc in scores]
            return
        execution = Execution(self.manager, self.tag)
        for mode, one, two in requests:
            task = execution.add_task(aligner)
            task.environment(self.environment, self.environment['aligner_env'])
            task.inputs(mode=mode, sequence_one=one, sequence_two=two, track_id_sets_one=track_id_sets,
                        track_id_sets_two=track_id_sets, score_matrices=score_matrices)
        for message in execution.run():
            yield message
        self._last_outputs = execution.outputs

    def execute(self, sequences, track_id_sets, score_matrices):
        merge_mode, dist_mode = self.environment['merge_mode'], self.environment['dist_mode']
        if merge_mode not in ("global", "semiglobal", "semiglobal_auto"):
            raise ComponentError("unknown merge mode '{0}'".format(merge_mode))
        if dist_mode not in ("global", "semiglobal", "semiglobal_auto"):
            raise ComponentError("unknown distance mode '{0}'".format(dist_mode))
        alignments = {i: _identity_alignment(seq) for i, seq in enumerate(sequences)}
        known = {}      # (cluster id a, cluster id b), a before b in cluster order -> score (float32)
        changed = None  # the cluster that grew in the previous round: its scores are stale
        aligner, aligner_env = _resolve_aligner(self)
        if _is_device_aligner(aligner) and isinstance(self.manager, BatchManager) and len(sequences) > 1:
            # resident path: clusters on the GPU; a round = one scores-only submission over the stale pairs (msa.py:
            # 504-540 with its cache) + one alignment whose merge happens on the device path
            resident = ResidentClusters(sequences, track_id_sets, score_matrices, aligner_env['gap_series'])
            try:
                ids = list(range(len(sequences)))
                total = max(len(ids) - 1, 1)
                done = 0
                while len(ids) > 1:
                    pending = [(a, b) for x, a in enumerate(ids) for b in ids[x + 1:] if changed is None or changed in (a, b)]
                    scores = resident.scores([(self._align_mode(dist_mode, resident.cluster(a), resident.cluster(b)), a, b)
                                              for a, b in pending])
                    for (a, b), sc in zip(pending, scores):
                        known[(a, b)] = np.float32(sc)
                    s = np.full((len(ids), len(ids)), -(2 ** 32), dtype=np.float32)
                    for x, a in enumerate(ids):
                        for y in range(x + 1, len(ids)):
                            s[x, y] = s[y, x] = known[(a, ids[y])]
                    x, y = np.unravel_index(s.argmax(), s.shape)     # first maximum in row-major order (msa.py:552)
                    i, j = ids[x], ids[y]
                    _, path = resident.align_and_merge(self._align_mode(merge_mode, resident.cluster(i), resident.cluster(j)), i, j)
                    alignments[i] = alignments[i].merge(alignments[j], path)
                    del alignments[j]
                    ids.remove(j)
                    changed = i
                    done += 1
                    yield ProgressMessage(done / float(total))
            finally:
                resident.close()
            yield CompleteMessage(outputs={'alignment': list(alignments.values())[0]})
            return
        clusters = {i: _count_cluster(i, seq, track_id_sets) for i, seq in enumerate(sequences)}
        total = max(len(clusters) - 1, 1)
        done = 0
        while len(clusters) > 1:
            ids = list(clusters.keys())
            pending = [(a, b) for x, a in enumerate(ids) for b in ids[x + 1:]
                       if changed is None or changed in (a, b)]
            for message in self._fan_out([(self._align_mode(dist_mode, clusters[a], clusters[b]), clusters[a], clusters[b])
                                          for a, b in pending], track_id_sets, score_matrices, scores_only=True):
                yield message
            for (a, b), out in zip(pending, self._last_outputs):
                known[(a, b)] = np.float32(out['score'])
            s = np.full((len(ids), len(ids)), -(2 ** 32), dtype=np.float32)
            for x, a in enumerate(ids):
                for y in range(x + 1, len(ids)):
                    s[x, y] = s[y, x] = known[(a, ids[y])]
            x, y = np.unravel_index(s.argmax(), s.shape)     # first maximum in row-major order (msa.py:552)
            i, j = ids[x], ids[y]
            one, two = clusters[i], clusters[j]
            for message in self._fan_out([(self._align_mode(merge_mode, one, two), one, two)], track_id_sets, score_matrices):
                yield message
            path = np.array(self._last_outputs[0]['alignment'].path)
            _merge_clusters(one, two, track_id_sets, path)
            alignments[i] = alignments[i].merge(alignments[j], path)
            del clusters[j]
            del alignments[j]
            changed = i
            done += 1
            yield ProgressMessage(done / float(total))
        yield CompleteMessage(outputs={'alignment': list(alignments.values())[0]})


COMPONENTS = [PairwiseAligner, RawPairwiseAligner, ProfileBuilder, DummyMasterSlaveAligner,
              GlobalMasterSlaveAligner, LocalMasterSlaveAligner, GuideTreeBuilder, TreeMultipleSequenceAligner,
              AdHocMultipleSequenceAligner]


# ---- the batching seam -----------------------------------------------------------------------------
class BatchManager(Manager):
    """Manager whose execute_many recognises the homogeneous request lists that Execution.run hands over for the
    reference's fan-outs (praline/core/execution.py:158-188) and runs each as a few device submissions:
      * PairwiseAligner lists (guide tree, ad-hoc rescoring): one submission per mode;
      * Global / LocalMasterSlaveAligner lists - the workflow's preprofile stage, ONE Execution with a task per master
        (praline/component/workflow.py:139-161): one plan over every (master, slave) pair, one run per Waterman-Eggert
        iteration with the path bounding boxes turned into the next iteration's masks on the device;
      * ProfileBuilder lists (workflow.py:211-224): counted on the host, one scatter-add per alignment;
      * RawPairwiseAligner lists (praline/component/align.py:254-447; caller-supplied score models): one submission, one
        launch for all requests and modes (native.RawBatch).
    Anything else falls back to the serial loop.

    rank / world / group: this process's place in a torch.distributed process group, one process per GPU - the
    counterpart of the reference's ParallelExecutionManager worker pool (praline/core/manager.py:401-463).  The
    all-pairs stages that run under this manager (GuideTreeBuilder) shard their pair list over the ranks and
    exchange the scores with one all-gather; every rank ends up with the same outputs."""

    def __init__(self, index, rank=0, world=1, group=None):
        Manager.__init__(self, index)
        if not (0 <= rank < world):
            raise ValueError("rank {0} outside a world of {1}".format(rank, world))
        self.rank, self.world, self.group = rank, world, group

    def _emit(self, requests, results, parent_tag):
        """Begin / Complete messages of the requests that were answered in bulk, the serial path for the others."""
        for k, (tid, inputs, tag, env) in enumerate(requests):
            if results[k] is None:
                for message in self._invoke(tid, inputs, tag, env, parent_tag=parent_tag):
                    yield message
                continue
            begin = BeginMessage(parent_tag)
            begin.tag = tag
            yield begin
            done = CompleteMessage(outputs=results[k])
            done.tag = tag
            yield done

    def _master_slave_batch(self, requests):
        """Outputs of a list of Global / LocalMasterSlaveAligner requests (None where a request has to take the serial
        path): requests that agree on track sets, score matrices, gap series and iteration count share ONE arena and ONE
        path plan; per Waterman-Eggert iteration one run, scores and paths back, then the bounds -> masks step on the
        device (praline/component/preprofile.py:114-156, 213-269)."""
        results = [None] * len(requests)
        groups = {}
        for k, (tid, inputs, tag, env) in enumerate(requests):
            cls = GlobalMasterSlaveAligner if tid == GlobalMasterSlaveAligner.tid else LocalMasterSlaveAligner
            component = cls(self, env, tag)
            self._check_request(component, inputs, env)
            aligner, aligner_env = _resolve_aligner(component)
            local = cls is LocalMasterSlaveAligner
            iterations = component.environment['waterman_eggert_iterations'] if local else 1
            if not _is_device_aligner(aligner) or not inputs['slave_sequences'] or iterations < 1 or iterations > native.MAX_RECTS + 1:
                continue   # another aligner component / nothing to align / more masks than the device slots hold
            for s in inputs['slave_sequences']:
                _validate_track_sets(inputs['master_sequence'], s, inputs['track_id_sets'], inputs['track_id_sets'],
                                     inputs['score_matrices'])
            key = (local, tuple(map(tuple, inputs['track_id_sets'])), tuple(id(sm) for sm in inputs['score_matrices']),
                   tuple(aligner_env['gap_series']), iterations)
            groups.setdefault(key, []).append((k, component.environment['score_threshold']))
        for (local, _, _, gap_series, iterations), members in groups.items():
            first = requests[members[0][0]][1]
            batch = PairwiseBatch(first['track_id_sets'], first['track_id_sets'], first['score_matrices'], list(gap_series))
            pairs, owner = [], []
            for k, _ in members:
                inputs = requests[k][1]
                m = batch._arena_index(inputs['master_sequence'], 0)
                for s in inputs['slave_sequences']:
                    pairs.append((m, batch._arena_index(s, 0)))
                owner.append(len(pairs))
            per_pair = [[] for _ in pairs]      # (score, path) per iteration
            arena = native.Arena(batch._profiles, batch.S, set_sizes=batch.sizes)
            try:
                plan = native.Plan(arena, np.array(pairs, dtype=np.int32), want_paths=True)
                try:
                    for it in range(iterations):
                        plan.run("local" if local else "global", batch.gap_open, batch.gap_extend)
                        sc, pt = plan.scores(), plan.paths()
                        for q in range(len(pairs)):
                            per_pair[q].append((float(sc[q]), np.array(pt[q], dtype=int)))
                        if it + 1 < iterations:
                            plan.mask_path_bounds()
                finally:
                    plan.close()
            finally:
                arena.close()
            lo = 0
            for (k, threshold), hi in zip(members, owner):
                inputs = requests[k][1]
                results[k] = {'alignment': merge_master_slave(inputs['master_sequence'], inputs['slave_sequences'],
                                                              per_pair[lo:hi], threshold, local=local)}
                lo = hi
        return results

    def _raw_batch(self, requests):
        """Outputs of a list of RawPairwiseAligner requests (praline/component/align.py:254-447): every request brings its own
        match scores, gap scores and zero cells; all of them go to the device in one submission (native.RawBatch) and
        are aligned by one launch, each in its own mode."""
        raw = []
        for tid, inputs, tag, env in requests:
            component = RawPairwiseAligner(self, env, tag)
            self._check_request(component, inputs, env)
            if inputs['mode'] not in MODES:
                raise ComponentError("unknown alignment mode: '{0}'".format(inputs['mode']))
            raw.append((inputs['match_score_model'].scores, inputs['gap_score_model_one'].scores,
                        inputs['gap_score_model_two'].scores, inputs.get('zero_idxs')))
        batch = native.RawBatch(raw)
        try:
            batch.run([inputs['mode'] for _, inputs, _, _ in requests])
            scores, paths = batch.results()
        finally:
            batch.close()
        results = []
        for (tid, inputs, tag, env), sc, pt in zip(requests, scores, paths):
            alignment = Alignment([inputs['sequence_one'], inputs['sequence_two']], _path_for_output(inputs['mode'], pt))
            results.append({'alignment': alignment, 'score': float(sc)})
        return results

    def execute_many(self, requests, parent_tag):
        self._require_open()
        requests = list(requests)
        tids = set(tid for tid, _, _, _ in requests)
        if len(requests) >= 2 and tids <= {GlobalMasterSlaveAligner.tid, LocalMasterSlaveAligner.tid}:
            for message in self._emit(requests, self._master_slave_batch(requests), parent_tag):
                yield message
            return
        if len(requests) >= 2 and tids == {RawPairwiseAligner.tid}:
            for message in self._emit(requests, self._raw_batch(requests), parent_tag):
                yield message
            return
        if len(requests) >= 2 and tids == {ProfileBuilder.tid}:
            results = []
            for tid, inputs, tag, env in requests:
                self._check_request(ProfileBuilder(self, env, tag), inputs, env)
                track = inputs['alignment'].items[0].get_track(inputs['track_id'])
                results.append({'profile_track': ProfileTrack(get_frequencies(inputs['alignment'], inputs['track_id']), track.alphabet)})
            for message in self._emit(requests, results, parent_tag):
                yield message
            return
        if len(requests) < 2 or any(tid != PairwiseAligner.tid for tid, _, _, _ in requests):
            for message in Manager.execute_many(self, requests, parent_tag):
                yield message
            return
        groups = {}
        for k, (tid, inputs, tag, env) in enumerate(requests):
            component = PairwiseAligner(self, env, tag)
            self._check_request(component, inputs, env)
            _validate_track_sets(inputs['sequence_one'], inputs['sequence_two'], inputs['track_id_sets_one'],
                                 inputs['track_id_sets_two'], inputs['score_matrices'])
            rects = None
            if inputs.get('zero_idxs'):
                rects = zero_idxs_to_rectangles(inputs['zero_idxs'])
                if rects is None:
                    groups.setdefault(('serial', k), []).append(k)
                    continue
            key = (tuple(map(tuple, inputs['track_id_sets_one'])), tuple(map(tuple, inputs['track_id_sets_two'])),
                   tuple(id(sm) for sm in inputs['score_matrices']), tuple(env['gap_series']))
            groups.setdefault(key, []).append((k, rects))
        results = [None] * len(requests)
        for key, members in groups.items():
            if key[0] == 'serial':
                continue
            first = requests[members[0][0]]
            batch = PairwiseBatch(first[1]['track_id_sets_one'], first[1]['track_id_sets_two'],
                                  first[1]['score_matrices'], first[3]['gap_series'])
            for k, rects in members:
                inputs = requests[k][1]
                batch.add(inputs['mode'], inputs['sequence_one'], inputs['sequence_two'], rects)
            scores, paths = batch.run(want_paths=True)
            for (k, _), sc, pt in zip(members, scores, paths):
                inputs = requests[k][1]
                results[k] = {'alignment': Alignment([inputs['sequence_one'], inputs['sequence_two']],
                                                     _path_for_output(inputs['mode'], pt)),
                              'score': sc}
        for message in self._emit(requests, results, parent_tag):
            yield message
