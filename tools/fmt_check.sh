#!/bin/bash
# Proof that tools/fmt_lines.py changed white space only: every translation unit of blockmaze_amd/csrc is compiled to assembly from the tree as committed (git HEAD) and from the
# working tree — device code and host code of the .hip files, host code of the .cpp files — and the two sets are compared.  bash tools/fmt_check.sh [tree to compare with HEAD, default: the working tree]   (about ten minutes on 8 cores)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); W=${TMPDIR:-/tmp}/fmt_check
if [ -z "$FMT_CHECK_COMPARE_ONLY" ]; then rm -rf "$W"; mkdir -p "$W/before" "$W/after" "$W/head"; git -C "$ROOT" archive HEAD blockmaze_amd/csrc include | tar -x -C "$W/head"; fi
asm() {   # asm <source tree> <output dir>
  local src=$1 out=$2
  ( cd "$src/blockmaze_amd/csrc"
    for f in *.hip; do echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -o $out/${f%.hip}.device.s $f"; echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-host-only -o $out/${f%.hip}.host.s $f"; done
    for f in *.cpp; do echo "g++ -O3 -std=c++17 -fPIC -S -o $out/${f%.cpp}.s $f"; done ) | ( cd "$src/blockmaze_amd/csrc" && xargs -P 8 -I{} sh -c '{} 2>/dev/null' )
}
AFTER=${1:-$ROOT}; if [ -z "$FMT_CHECK_COMPARE_ONLY" ]; then asm "$W/head" "$W/before"; asm "$AFTER" "$W/after"; fi
# .file / .ident lines name paths and compilers, and hipcc derives the ids of a compilation unit (__hip_cuid_*, __hip_fatbin_*, __hip_gpubin_handle_*) from its source text
norm() { grep -v '^\s*\.file\|^\s*\.ident' "$1" | sed 's/__hip_\(cuid\|gpubin_handle\|fatbin\)_[0-9a-f]*/__hip_\1_ID/g'; }
bad=0
# every expected assembly file must exist on BOTH sides: a unit that fails to compile (its errors go to /dev/null above) is a failure, not a silent skip
for src in "$ROOT"/blockmaze_amd/csrc/*.hip; do n=$(basename "${src%.hip}"); for s in "$n.device.s" "$n.host.s"; do for side in before after; do [ -s "$W/$side/$s" ] || { echo "MISSING: $side/$s (did not compile)"; bad=1; }; done; done; done
for src in "$ROOT"/blockmaze_amd/csrc/*.cpp; do s=$(basename "${src%.cpp}").s; for side in before after; do [ -s "$W/$side/$s" ] || { echo "MISSING: $side/$s (did not compile)"; bad=1; }; done; done
for f in "$W"/before/*.s; do
  b=$(basename "$f"); [ -s "$W/after/$b" ] || continue
  # (an assert() carries its line number as an immediate: such a difference is listed, not hidden)
  if ! diff <(norm "$f") <(norm "$W/after/$b") > "$W/$b.diff"; then echo "DIFFERENT: $b ($(wc -l < "$W/$b.diff") diff lines, $W/$b.diff)"; bad=1; else echo "same: $b"; fi
done
exit $bad
