#!/bin/sh
# Builds the REAL reference C extension (praline/util/cext.c) from the sources where
# they lie under /root/reference, straight with gcc (the reference's own setup.py is
# not run).  Flags are the reference's own (setup.py:26-28).  Output only into
# oracle/_ref/ (git-ignored, but it travels to the GPU box with the snapshot).
# TEST INFRASTRUCTURE ONLY: nothing in the product path may load oracle/_ref.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${PRALINE_REFERENCE:-/root/reference}"
SRC="$REF/praline/util/cext.c"
OUT="$HERE/_ref"
if [ ! -f "$SRC" ]; then
    echo "build_ref.sh: $SRC not present (GPU box?) - keeping prebuilt files" >&2
    exit 0
fi
mkdir -p "$OUT"
PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
NPINC="$(python3 -c 'import numpy; print(numpy.get_include())')"
EXT="$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
gcc -shared -fPIC -std=c99 -ffast-math -O3 -w -I"$PYINC" -I"$NPINC" "$SRC" -o "$OUT/cext$EXT"
echo "built $OUT/cext$EXT"
