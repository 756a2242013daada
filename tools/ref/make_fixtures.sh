#!/bin/bash
# Reference-produced golden fixtures: builds the REAL elPrep from its Go sources and runs it on every case of
# tools/ref/cases.py - `elprep filter` (--mark-duplicates --mark-optical-duplicates --sorting-order coordinate --bqsr) and `elprep sfm`
# (split / per-split filter / merge: the split rule, the sr-tagged copies, the merge order) on the synthetic reads of the parity tests,
# the hand-derived edge cases of tests/kat_cases.py, CleanSam - and stores what it writes (record order, FLAGs, MAPQ / CIGAR, recalibrated
# QUALs, the duplication metrics text, the recalibration table) under tests/golden/ref/, where tests/test_oracle_golden.py compares the CPU
# oracle with it: the day this has run, the oracle is pinned on the reference itself.
#
# It needs a Go toolchain and the module github.com/exascience/pargo v1.1.0 (go.mod of the reference; from the module cache, GOFLAGS=-mod=vendor
# or a proxy).  Neither exists in the build container nor on the GPU box of this project (profiles/r2a_reference_toolchain_probe.txt):
# the script has never run here and says so in DESIGN.md section 6.  Nothing of the reference's SOURCE enters the repository; the binary goes
# to oracle/_ref/ (git-ignored).  Somebody with Go but without this repository's Python side can run the same commands from the bundle
# tools/ref/bundle.sh writes.
# The recipe's plumbing is exercised without Go by tests/test_oracle_golden.py::test_fixture_recipe_dry_run (tools/ref/oracle_as_elprep.py stands
# in for the binary, in a temporary directory; it pins nothing).
# usage: tools/ref/make_fixtures.sh [reference-dir (default /root/reference)] [pairs (default 20000)]
#        ELPREP=/path/to/elprep tools/ref/make_fixtures.sh - [pairs]      (a binary somebody built elsewhere)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
REF="${1:-/root/reference}"
PAIRS="${2:-20000}"
mkdir -p "$ROOT/oracle/_ref" "$ROOT/tests/golden/ref"
if [ -z "${ELPREP:-}" ]; then
  command -v go >/dev/null 2>&1 || { echo "make_fixtures: no Go toolchain on PATH - nothing done (the fixtures stay absent, the oracle stays unpinned)"; exit 2; }
  [ -f "$REF/go.mod" ] || { echo "make_fixtures: $REF is not the elPrep source tree"; exit 2; }
  ( cd "$REF" && go build -o "$ROOT/oracle/_ref/elprep" . )
  ELPREP="$ROOT/oracle/_ref/elprep"
fi
W="$(mktemp -d)"
python3 "$ROOT/tools/ref/write_inputs.py" --all "$W" "$PAIRS"
for D in "$W"/*/; do
  NAME="$(basename "$D")"
  python3 "$ROOT/tools/ref/run_case.py" "$ELPREP" "$D"
  python3 "$ROOT/tools/ref/collect.py" "$D" "$ROOT/tests/golden/ref/$NAME.json"
done
rm -rf "$W"
( [ -d "$REF" ] && cd "$REF" && git rev-parse HEAD 2>/dev/null || true; command -v go >/dev/null 2>&1 && go version || true; "$ELPREP" 2>&1 | head -2 || true ) > "$ROOT/tests/golden/ref/PROVENANCE.txt"
echo "fixtures written to tests/golden/ref/ - run: python -m pytest tests/test_oracle_golden.py -q"
