"""FASTA in, aligned FASTA out - the two I/O functions the MSA pipeline needs at its ends
(praline/__init__.py:67-136 load_sequence_fasta, 252-304 write_alignment_fasta).  Host-side text
handling; nothing here touches the device."""
import io as _io

import numpy as np

from .container import PlainTrack, Sequence, TRACK_ID_INPUT
from .core import DataError


def load_sequence_fasta(source, alphabet, track_id=TRACK_ID_INPUT):
    """Sequences of a FASTA file (path or text file object), residues upper-cased and mapped through
    `alphabet`; the record name is the header up to the first whitespace."""
    handle = open(source, "r") if isinstance(source, str) else source
    try:
        records = []
        name, chunks = None, []
        for line in handle:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if name is not None:
                    records.append((name, "".join(chunks)))
                fields = line[1:].split()
                name, chunks = (fields[0] if fields else ""), []
            elif name is not None:
                chunks.append(line)
        if name is not None:
            records.append((name, "".join(chunks)))
    finally:
        if isinstance(source, str):
            handle.close()
    out = []
    for name, residues in records:
        track = PlainTrack([c for c in residues.upper()], alphabet)
        out.append(Sequence(name, [(track_id, track)]))
    return out


def alignment_rows(alignment, track_id=TRACK_ID_INPUT):
    """One gapped string per aligned sequence: column i shows the residue a sequence consumes between
    path rows i and i+1, '-' where it does not advance."""
    path = np.asarray(alignment.path)
    rows = []
    for j, sequence in enumerate(alignment.items):
        track = sequence.get_track(track_id)
        if track.tid != PlainTrack.tid:
            raise DataError("can only write FASTA alignments for plain tracks")
        if np.any(path[:, j] == -1):
            raise DataError("the FASTA format does not currently support local alignments")
        advance = (path[1:, j] - path[:-1, j]) > 0
        symbols = np.array([track.alphabet.index_to_symbol(int(v)) for v in track.values] + ["-"])
        idx = np.where(advance, path[1:, j] - 1, len(track.values))
        rows.append("".join(symbols[idx]))
    return rows


def write_alignment_fasta(target, alignment, track_id=TRACK_ID_INPUT, line_length=72):
    """Aligned FASTA: '>name' (cut to line_length - 1 characters), then the gapped sequence wrapped
    at line_length columns, every line newline-terminated."""
    text = _io.StringIO()
    for sequence, row in zip(alignment.items, alignment_rows(alignment, track_id)):
        text.write(">{0}\n".format(sequence.name[:line_length - 1]))
        for k in range(0, len(row), line_length):
            text.write(row[k:k + line_length] + "\n")
    data = text.getvalue()
    if isinstance(target, str):
        with open(target, "w", encoding="utf-8") as f:
            f.write(data)
    else:
        target.write(data)
    return data
