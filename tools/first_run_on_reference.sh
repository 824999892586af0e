#!/bin/bash
# First-run kit for the one thing neither of our boxes can execute (missing #2): the REFERENCE's own engines
# (COTR/inference/sparse_engine.py SparseEngine / FasterSparseEngine, what demo_single_pair.py:25-45 drives) running on the
# cotr_amd binding through INTEGRATION.md section 1's one-line switch.  Needs a machine that has BOTH a checkout of
# ubc-vision/COTR (with its Python dependencies: torchvision, opencv, imageio, ...) and an MI355X with this repository built.
#
#   tools/first_run_on_reference.sh /path/to/COTR            # apply the switch (backup kept), run the checker, print PASS / FAIL
#   tools/first_run_on_reference.sh /path/to/COTR --revert   # put COTR/models/__init__.py back
#
# No trained checkpoint is needed: the checker loads the same seeded random weights into the binding and into the reference's
# own torch model and compares (a) the two models on identical inputs (1e-3 px), (b) the reference's engines driving the binding
# against cotr_amd.inference's engines on the same seeds (same correspondences).
set -euo pipefail
COTR_DIR=$(cd "${1:?usage: $0 /path/to/COTR [--revert]}" && pwd)
HERE=$(cd "$(dirname "$0")/.." && pwd)
INIT="$COTR_DIR/COTR/models/__init__.py"
[ -f "$INIT" ] || { echo "not a COTR checkout: $INIT missing"; exit 2; }
if [ "${2:-}" = "--revert" ]; then
  [ -f "$INIT.orig" ] && mv "$INIT.orig" "$INIT" && echo "reverted $INIT" || echo "nothing to revert"
  exit 0
fi
python "$HERE/tools/first_run_check.py" --patch "$INIT"
python -m cotr_amd.build >/dev/null
cd "$COTR_DIR"
PYTHONPATH="$HERE:$COTR_DIR:${PYTHONPATH:-}" python "$HERE/tools/first_run_check.py" --run "$COTR_DIR"
