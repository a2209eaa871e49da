#!/bin/sh
# Apply one patch of this directory to a scratch copy of the tree and run the CPU-harness parity tests of the
# touched stage against it (bit-exact vs the oracle).  Usage: experiments/check.sh experiments/<name>.patch [pytest args]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PATCH="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"
shift
TMP="$(mktemp -d /tmp/cc_exp.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
cd "$ROOT"
git ls-files -z | tar --null -T - -cf - | tar -xf - -C "$TMP"
[ -d "$ROOT/oracle/_ref" ] && cp -r "$ROOT/oracle/_ref" "$TMP/oracle/" || true
cd "$TMP"
patch -p1 -s < "$PATCH"
if [ $# -eq 0 ]; then set -- tests/test_emu_ingest.py tests/test_emu_query.py; fi
python -m pytest -x -q "$@"
