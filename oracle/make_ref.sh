#!/bin/sh
# TEST / MEASUREMENT INFRASTRUCTURE -- not part of the product.
#
# Stages the reference itself (maciejkula/spotlight, pure Python: nothing to compile) where the GPU box
# can import it: /root/reference/spotlight -> oracle/_ref/spotlight.  oracle/_ref/ is git-ignored (the
# reference's sources never enter this repository's history) but NOT gpurun-ignored, so the staged copy
# travels to the GPU box with the snapshot, like the built .so files.  Only bench.py's `cpu_baseline`
# leg (oracle/ref_cpu_baseline.py) and the fixture generators import it; nothing under spotlight_amd/ may.
#
#   sh oracle/make_ref.sh            (also run by __graft_entry__.build() when /root/reference exists)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="${SPOTLIGHT_REFERENCE:-/root/reference}"
if [ ! -d "$SRC/spotlight" ]; then
    echo "make_ref: $SRC/spotlight not found (GPU box: the staged copy under oracle/_ref is used as is)" >&2
    exit 0
fi
rm -rf "$HERE/_ref"
mkdir -p "$HERE/_ref"
cp -r "$SRC/spotlight" "$HERE/_ref/spotlight"
# the reference's own test modules that need no dataset download (synthetic data only): run UNMODIFIED against this
# package through its `spotlight` import aliases by tests/test_gpu_reference_tests.py
mkdir -p "$HERE/_ref/tests/sequence"
cp "$SRC/tests/test_layers.py" "$HERE/_ref/tests/test_layers.py"
cp "$SRC/tests/sequence/test_sequence_implicit.py" "$HERE/_ref/tests/sequence/test_sequence_implicit.py"
find "$HERE/_ref" -name __pycache__ -type d -prune -exec rm -rf {} +
# provenance of the staged copy (checked by tests/test_oracle.py when /root/reference is present)
( cd "$SRC/spotlight" && find . -name '*.py' | LC_ALL=C sort | xargs sha256sum ) > "$HERE/_ref/MANIFEST.sha256"
echo "make_ref: staged $(find "$HERE/_ref/spotlight" -name '*.py' | wc -l) reference files under oracle/_ref/spotlight"
