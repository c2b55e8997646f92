#!/usr/bin/env bash
# Builds the reference's ONLY native file (bblean/csrc/similarity.cpp, pybind11 module
# `_cpp_similarity`) straight from /root/reference into oracle/_ref/ (git-ignored).
# Flags are the reference's defaults (setup.py:34-44). Nothing is copied into the repo.
# This is test infrastructure: only tests/, smoke() and bench.py's cpu_baseline may load it.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="${BB_REFERENCE_ROOT:-/root/reference}/bblean/csrc/similarity.cpp"
OUT="$HERE/_ref"
mkdir -p "$OUT"
if [ ! -f "$SRC" ]; then
  echo "reference source not present ($SRC); keeping prebuilt oracle/_ref if any" >&2
  exit 0
fi
EXT="$(python3-config --extension-suffix 2>/dev/null || python3 -c 'import sysconfig;print(sysconfig.get_config_var("EXT_SUFFIX"))')"
c++ -O3 -march=nocona -mtune=haswell -mpopcnt -shared -std=c++17 -fPIC -fvisibility=hidden \
    $(python3 -m pybind11 --includes) "$SRC" -o "$OUT/_cpp_similarity$EXT"
echo "built $OUT/_cpp_similarity$EXT"
