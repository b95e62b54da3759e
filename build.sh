#!/usr/bin/env bash
# Build libuniver_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")" && pwd)"
SRC="$ROOT/univer-ocr_amd/csrc"
OUT="$ROOT/univer-ocr_amd/libuniver_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++20 -fPIC -I"$ROOT/include" -I"$SRC" -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-local-typedef)
# UOCR_BUILD_EXPERIMENTS=1: also the kernels that were measured and left off (conv_t32w.hip, most of conv_t32.hip)
OBJ="$SRC/.obj"
if [[ -n "${UOCR_BUILD_EXPERIMENTS:-}" ]]; then FLAGS+=(-DUOCR_EXPERIMENTS); OBJ="$SRC/.obj_exp"; fi
mkdir -p "$OBJ"
pids=()
for f in "$SRC"/*.hip; do
  o="$OBJ/$(basename "${f%.hip}").o"
  if [[ ! -f "$o" || "$f" -nt "$o" || "$SRC/uocr_common.h" -nt "$o" || "$ROOT/include/univer_hip.h" -nt "$o" \
        || -n "$(find "$SRC" -name '*.h' -newer "$o" -print -quit)" ]]; then
    # per-file extra flags: a line "// hipcc-flags: ..." in the source
    extra=$(sed -n 's|^// hipcc-flags: ||p' "$f" | head -1)
    # shellcheck disable=SC2086
    "$HIPCC" "${FLAGS[@]}" $extra -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
