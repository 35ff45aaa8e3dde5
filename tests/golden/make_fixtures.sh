#!/bin/bash
# Regenerates the golden fixture files from the reference checkout (run in the build container,
# where /root/reference exists).  These are the reference's own tiny test DATA files (no source):
#   model.bin          <- resources/model.bin                       (lib.rs:25-41, predictor.rs:383-431)
#   kytea-model.bin    <- resources/kytea-model.bin                 (kytea_model.rs:401-422)
#   tantivy_model.bin  <- zstd -d vaporetto_tantivy/test_model/model.zst (vaporetto_tantivy/src/lib.rs:231-364)
#   docs.tok           <- resources/docs.tok  (expected tokenisation + tags for model.bin)
set -e
REF=${1:-/root/reference}
OUT=$(dirname "$0")
cp "$REF/resources/model.bin" "$OUT/model.bin"
cp "$REF/resources/kytea-model.bin" "$OUT/kytea-model.bin"
cp "$REF/resources/docs.tok" "$OUT/docs.tok"
LD_LIBRARY_PATH=/opt/conda/lib /opt/conda/bin/zstd -d -f -q "$REF/vaporetto_tantivy/test_model/model.zst" -o "$OUT/tantivy_model.bin"
chmod 644 "$OUT"/*.bin "$OUT/docs.tok"
