#!/bin/bash
# tools/build_variant.sh <git-ref> <name> [extra hipcc flags]: libodd_hip.so of <git-ref>'s sources as variants_<name>.so in the repo
# root (git-ignored; travels with gpurun; ODDIO_HIP_LIB selects it) -- the "A" of same-box A/B runs (tools/ab_env.sh).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=$1; NAME=$2; shift 2
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REF" oddio_amd/csrc include | tar -x -C "$TMP"
make -s -C "$TMP/oddio_amd/csrc" OUT="$ROOT/variants_$NAME.so" FLAGS_EXTRA="$*" >/dev/null
rm -rf "$TMP"
ls -la "$ROOT/variants_$NAME.so"
