#!/bin/bash
# oracle/build_ref_cython_merge.sh -- TEST INFRASTRUCTURE (build container only).
# Compiles the reference's own Cython greedy-merge (search/cython_merge/sources/cython_merge.pyx)
# from where it lies under /root/reference into oracle/_ref/ (git-ignored build output; nothing of
# the reference is copied into the repository).  Used by oracle/gen_golden_greedy.py to produce
# golden vectors for the greedy edge-merge decoder.  The reference's prebuilt sources/build/ is
# never loaded.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC=/root/reference/search/cython_merge/sources/cython_merge.pyx
OUT="$HERE/_ref"
mkdir -p "$OUT"
python3 -m cython -3 "$SRC" -o "$OUT/cython_merge.c"
INC=$(python3 -c "import sysconfig, numpy; print('-I' + sysconfig.get_paths()['include'] + ' -I' + numpy.get_include())")
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
gcc -O2 -fPIC -shared -w $INC "$OUT/cython_merge.c" -o "$OUT/cython_merge$EXT"
echo "built $OUT/cython_merge$EXT"
