"""Data model of the hot path: input / output layouts only.

Restates the containers the pairwise aligner consumes and produces (reference:
praline/container/alphabet.py, sequence.py, score.py, align.py, tree.py).  Index conventions are
the reference's so PlainTrack index arrays, score matrices and alignment paths are interchangeable:
  * ALPHABET_AA symbol order A R N D C E Q G H I L K M F P S T W Y V U O B Z J X * (27 symbols,
    alphabet.py:100-104), ALPHABET_DNA A T G C S W R Y K M B V H D N (15), ALPHABET_RNA A U C G (4);
  * ScoreMatrix.matrix is float32 [A1, A2], row = symbol of sequence one (score.py:96-97);
  * ProfileTrack.profile = float32(counts / float32(rowsum)) evaluated in float64
    (sequence.py:200-202);
  * Alignment.path: one row per alignment column boundary, one column per sequence, entry =
    number of residues of that sequence consumed so far, -1 = not part of a local alignment.
"""
import numpy as np

from .core import AlphabetError, Container, DataError

TRACK_ID_INPUT = "praline.tracks.InputTrack"
TRACK_ID_PREPROFILE = "praline.tracks.Preprofile"
TRACK_ID_PROFILE = "praline.tracks.Profile"


class Alphabet(Container):
    """Symbol <-> index mapping; size = highest index + 1 (alphabet.py:16-83)."""
    tid = "praline.container.Alphabet"

    def __init__(self, aid, mappings):
        self.aid = aid
        self._to_index = {}
        self._to_symbol = {}
        self._max_index = 0
        for symbol, index in mappings:
            self._max_index = max(self._max_index, index)
            self._to_index[symbol] = index
            self._to_symbol[index] = symbol

    def symbol_to_index(self, symbol):
        try:
            return self._to_index[symbol]
        except KeyError:
            raise AlphabetError("symbol '{0}' not found in {1}".format(symbol, self))

    def index_to_symbol(self, index):
        try:
            return self._to_symbol[index]
        except KeyError:
            raise AlphabetError("index {0} not found in {1}".format(index, self))

    @property
    def symbols(self):
        return list(self._to_index.keys())

    @property
    def size(self):
        return self._max_index + 1

    def __repr__(self):
        return "<Alphabet aid='{0}'>".format(self.aid)


def _enumerate_symbols(symbols):
    return [(s, i) for i, s in enumerate(symbols)]


ALPHABET_AA = Alphabet("praline.alphabet.AAOneLetter", _enumerate_symbols("ARNDCEQGHILKMFPSTWYVUOBZJX*"))
ALPHABET_DNA = Alphabet("praline.alphabet.DNA", _enumerate_symbols("ATGCSWRYKMBVHDN"))
ALPHABET_RNA = Alphabet("praline.alphabet.RNA", _enumerate_symbols("AUCG"))


class Track(Container):
    tid = "praline.container.Track"

    def __len__(self):
        raise NotImplementedError("please implement __len__ in your Track subclass")


class PlainTrack(Track):
    """One symbol index per position: values int32 [L] (sequence.py:137-160)."""
    tid = "praline.container.PlainTrack"

    def __init__(self, values, alphabet, raw_indices=None):
        if raw_indices is None:
            indices = [alphabet.symbol_to_index(v) for v in values]
        else:
            indices = raw_indices
        self.values = np.array(indices, dtype=np.int32)
        self.alphabet = alphabet

    def __len__(self):
        return self.values.shape[0]


class ProfileTrack(Track):
    """Per-position symbol counts [L, A] with a lazily derived fp32 profile (sequence.py:163-239)."""
    tid = "praline.container.ProfileTrack"

    def __init__(self, counts, alphabet):
        self.counts = np.array(counts, dtype=int)  # truncating cast, as the reference (sequence.py:184)
        self.alphabet = alphabet
        self._profile = None

    def __len__(self):
        return self.counts.shape[0]

    @property
    def profile(self):
        if self._profile is None:
            totals = np.array(self.counts.sum(axis=1), dtype=np.float32)
            self._profile = np.array(self.counts / totals[:, np.newaxis], dtype=np.float32)
        return self._profile

    def merge(self, track, path):
        """New ProfileTrack whose column i sums the counts of the positions that advance in alignment
        column i (sequence.py:205-239): the sum is formed in float32, then truncated to int."""
        if track.tid != self.tid:
            raise DataError("can not merge with non-profile track {0}".format(track.tid))
        if self.alphabet.aid != track.alphabet.aid:
            raise DataError("our alphabet {0} does not match track alphabet {1}".format(
                self.alphabet.aid, track.alphabet.aid))
        path = np.asarray(path)
        adv = (path[1:] - path[:-1]) > 0
        merged = np.zeros((path.shape[0] - 1, self.counts.shape[1]), dtype=np.float32)
        rows0 = np.nonzero(adv[:, 0])[0]
        merged[rows0] += self.counts[path[rows0 + 1, 0] - 1].astype(np.float32)
        rows1 = np.nonzero(adv[:, 1])[0]
        merged[rows1] += track.counts[path[rows1 + 1, 1] - 1].astype(np.float32)
        return ProfileTrack(merged, self.alphabet)


class Sequence(Container):
    """A named bundle of equally long tracks (sequence.py:19-117)."""
    tid = "praline.container.Sequence"

    def __init__(self, name, tracks):
        self.name = name
        self._length = None
        self._tracks = {}
        for trid, track in tracks:
            self.add_track(trid, track)

    def __len__(self):
        return self._length

    def add_track(self, trid, track):
        if trid in self._tracks:
            raise DataError("track with id {0} already present in this sequence".format(trid))
        if self._length is None:
            self._length = len(track)
        if len(track) != len(self):
            raise DataError("track length {0} does not match sequence length {1}".format(len(track), len(self)))
        self._tracks[trid] = track

    def del_track(self, trid):
        if trid not in self._tracks:
            raise DataError("track with id {0} not found".format(trid))
        del self._tracks[trid]
        if not self._tracks:
            self._length = None

    def replace_track(self, trid, track):
        self.del_track(trid)
        self.add_track(trid, track)

    def get_track(self, trid):
        if trid not in self._tracks:
            raise DataError("track with id {0} not found".format(trid))
        return self._tracks[trid]

    @property
    def tracks(self):
        return list(self._tracks.items())

    def __repr__(self):
        return "<Sequence name='{0}' length={1}>".format(self.name, len(self))


class ScoreMatrix(Container):
    """n-dimensional float32 score matrix, one alphabet per dimension (score.py:71-134)."""
    tid = "praline.container.ScoreMatrix"

    def __init__(self, scores, alphabets, matrix=None):
        if len(alphabets) < 2:
            raise DataError("need at least 2 alphabets for a score matrix, got {0}".format(len(alphabets)))
        self.alphabets = alphabets
        if scores is not None:
            self.matrix = np.zeros(tuple(a.size for a in alphabets), dtype=np.float32)
            for symbols, score in scores.items():
                idx = tuple(a.symbol_to_index(s) for a, s in zip(alphabets, symbols))
                self.matrix[idx] = score
        else:
            self.matrix = matrix

    def score(self, symbols):
        return self.matrix[tuple(a.symbol_to_index(s) for a, s in zip(self.alphabets, symbols))]


def blosum62():
    """BLOSUM62 over ALPHABET_AA as the reference's loader produces it (praline/__init__.py:67-102)."""
    from .matrices import blosum62_matrix
    return ScoreMatrix(None, [ALPHABET_AA, ALPHABET_AA], matrix=blosum62_matrix())


def nucleotide_matrix():
    from .matrices import nucleotide_matrix as nm
    return ScoreMatrix(None, [ALPHABET_DNA, ALPHABET_DNA], matrix=nm())


class MatchScoreModel(Container):
    """Match scores float32 [L1, L2] of a pair (score.py:16-42)."""
    tid = "praline.container.MatchScoreModel"

    def __init__(self, sequence_one, sequence_two, scores):
        if len(sequence_one) != scores.shape[0]:
            raise DataError("sequence length {0} does not correspond to array shape {1}".format(
                len(sequence_one), scores.shape[0]))
        if len(sequence_two) != scores.shape[1]:
            raise DataError("sequence length {0} does not correspond to array shape {1}".format(
                len(sequence_two), scores.shape[1]))
        self.sequence_one = sequence_one
        self.sequence_two = sequence_two
        self.scores = scores


class GapScoreModel(Container):
    """Gap scores float32 [L, 2] = per-position (open, extend) (score.py:45-68)."""
    tid = "praline.container.GapScoreModel"

    def __init__(self, sequence, scores):
        if len(sequence) != scores.shape[0]:
            raise DataError("sequence length {0} does not correspond to array shape {1}".format(
                len(sequence), scores.shape[0]))
        self.sequence = sequence
        self.scores = scores


class Alignment(Container):
    """Aligned sequences + the path through the DP matrix (align.py:14-61)."""
    tid = "praline.container.Alignment"

    def __init__(self, items, path):
        self.items = items
        self.path = path

    def merge(self, alignment, path):
        """Merge along a pairwise path: row i of the result takes row path[i, 0] of this alignment's
        path and row path[i, 1] of the other's; -1 rows stay -1 (align.py:30-61)."""
        path = np.asarray(path)
        mine, theirs = np.asarray(self.path), np.asarray(alignment.path)
        one = np.where((path[:, 0] >= 0)[:, None], mine[np.maximum(path[:, 0], 0)], -1)
        two = np.where((path[:, 1] >= 0)[:, None], theirs[np.maximum(path[:, 1], 0)], -1)
        return Alignment(self.items + alignment.items, np.hstack([one, two]).astype(int))


class SequenceTree(Container):
    tid = "praline.container.SequenceTree"

    def __init__(self, sequences, merge_orders):
        self.merge_orders = merge_orders
        self.sequences = sequences
