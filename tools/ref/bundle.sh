#!/bin/bash
# A self-contained bundle for somebody who has Go (or an elprep 5.1.3 binary) but not this repository's environment: the inputs of every
# case of tools/ref/cases.py, a run.sh with the elprep command lines, and collect.py (plain Python 3, no dependencies).  They run
#     ./run.sh /path/to/elprep
# and send back the fixtures/ directory; its *.json files go to tests/golden/ref/, where tests/test_oracle_golden.py compares the CPU oracle
# with them (round 6, VERDICT r5 next #3b).  Nothing of the reference's source is in the bundle.
# usage: tools/ref/bundle.sh [out.tar.gz (default /tmp/elprep_ref_bundle.tar.gz)] [pairs (default 20000)]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="${1:-/tmp/elprep_ref_bundle.tar.gz}"
PAIRS="${2:-20000}"
W="$(mktemp -d)"
B="$W/elprep_ref_bundle"
mkdir -p "$B/cases" "$B/fixtures"
python3 "$ROOT/tools/ref/write_inputs.py" --all "$B/cases" "$PAIRS"
cp "$ROOT/tools/ref/collect.py" "$B/collect.py"
ELP_ROOT="$ROOT" python3 - "$B" "$PAIRS" <<'PY'
import os, sys
sys.path.insert(0, os.environ["ELP_ROOT"])
from tools.ref import cases
b, pairs = sys.argv[1], int(sys.argv[2])
with open(os.path.join(b, "run.sh"), "w") as f:
    f.write("#!/bin/bash\n# usage: ./run.sh /path/to/elprep   (elPrep 5.1.3; every command runs with --nr-of-threads 1)\nset -euo pipefail\n"
            "ELPREP=\"${1:?path to the elprep binary}\"\ncd \"$(dirname \"$0\")\"\n")
    for c in cases.all_cases(pairs):
        w = os.path.join("cases", c["name"])
        f.write(f"\necho '== {c['name']}'\nmkdir -p {w}/tmp\n")
        for argv in cases.commands(c, "$ELPREP", w):
            f.write(" ".join(('"$ELPREP"' if a == "$ELPREP" else a) for a in argv) + "\n")
        f.write(f"python3 collect.py {w} fixtures/{c['name']}.json\n")
    f.write("\n\"$ELPREP\" 2>&1 | head -2 > fixtures/PROVENANCE.txt || true\necho 'done: send back the fixtures/ directory (its *.json files go to tests/golden/ref/)'\n")
os.chmod(os.path.join(b, "run.sh"), 0o755)
open(os.path.join(b, "README.txt"), "w").write(
    "Inputs for pinning the elprep_amd CPU oracle on the real elPrep (v5.1.3).\n"
    "  cases/<name>/in.sam [ref.fasta sites.bed]   inputs; case.json says how they were generated\n"
    "  run.sh <elprep>                             runs every case and collects fixtures/<name>.json\n"
    "  collect.py                                  turns a case's out.sam / metrics.txt / recal.txt into the fixture (Python 3, no dependencies)\n"
    "Expected layout afterwards: fixtures/*.json + fixtures/PROVENANCE.txt -> copy into tests/golden/ref/ of the repository and run\n"
    "  python -m pytest tests/test_oracle_golden.py -q\n")
PY
tar -C "$W" -czf "$OUT" elprep_ref_bundle
rm -rf "$W"
echo "bundle written to $OUT ($(du -h "$OUT" | cut -f1))"
