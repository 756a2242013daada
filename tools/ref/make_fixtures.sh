#!/bin/bash
# Reference-produced golden fixtures: builds the REAL elPrep from its Go sources and runs `elprep filter` with
# --mark-duplicates --mark-optical-duplicates --sorting-order coordinate --bqsr on the synthetic reads of the parity tests; what it writes
# (record order, FLAGs, recalibrated QUALs, the duplication metrics, the recalibration table) is stored under tests/golden/ref/, where
# tests/test_oracle_golden.py compares the CPU oracle with it - the day this has run, the oracle is pinned on the reference itself.
#
# It needs a Go toolchain and the module github.com/exascience/pargo v1.1.0 (go.mod of the reference; from the module cache, GOFLAGS=-mod=vendor
# or a proxy).  Neither exists in the build container nor on the GPU box of this project (profiles/r2a_reference_toolchain_probe.txt):
# the script has never run here and says so in DESIGN.md section 6.  Nothing of the reference's SOURCE enters the repository; the binary goes
# to oracle/_ref/ (git-ignored).
# The recipe's plumbing is exercised without Go by tests/test_oracle_golden.py::test_fixture_recipe_dry_run (tools/ref/oracle_as_elprep.py stands
# in for the binary, in a temporary directory; it pins nothing).
# usage: tools/ref/make_fixtures.sh [reference-dir (default /root/reference)] [pairs (default 20000)]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
REF="${1:-/root/reference}"
PAIRS="${2:-20000}"
command -v go >/dev/null 2>&1 || { echo "make_fixtures: no Go toolchain on PATH - nothing done (the fixtures stay absent, the oracle stays unpinned)"; exit 2; }
[ -f "$REF/go.mod" ] || { echo "make_fixtures: $REF is not the elPrep source tree"; exit 2; }
mkdir -p "$ROOT/oracle/_ref" "$ROOT/tests/golden/ref"
( cd "$REF" && go build -o "$ROOT/oracle/_ref/elprep" . )
ELPREP="$ROOT/oracle/_ref/elprep"
for SEED in 0 1; do
  W="$(mktemp -d)"
  python3 "$ROOT/tools/ref/write_inputs.py" "$W" "$PAIRS" "$SEED"
  "$ELPREP" fasta-to-elfasta "$W/ref.fasta" "$W/ref.elfasta"
  "$ELPREP" bed-to-elsites "$W/sites.bed" "$W/sites.elsites"
  # one thread: the reference's duplicate tournaments are racy on exact (score, QNAME) ties only, and its stable sort is deterministic
  # either way; --nr-of-threads 1 makes the run the sequential execution the oracle restates
  "$ELPREP" filter "$W/in.sam" "$W/out.sam" --mark-duplicates --mark-optical-duplicates "$W/metrics.txt" --optical-duplicates-pixel-distance 100 \
      --sorting-order coordinate --bqsr "$W/recal.txt" --reference "$W/ref.elfasta" --known-sites "$W/sites.elsites" --max-cycle 500 --nr-of-threads 1
  python3 "$ROOT/tools/ref/collect.py" "$W" "$ROOT/tests/golden/ref/filter_tiny_seed$SEED.json"
  rm -rf "$W"
done
( cd "$REF" && git rev-parse HEAD 2>/dev/null || true; go version ) > "$ROOT/tests/golden/ref/PROVENANCE.txt"
echo "fixtures written to tests/golden/ref/ - run: python -m pytest tests/test_oracle_golden.py -q"
