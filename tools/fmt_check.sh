#!/bin/bash
# Proof that tools/fmt_lines.py changed white space only: every translation unit of blockmaze_amd/csrc is compiled to assembly from the tree as committed (git HEAD) and from the
# working tree — device code and host code of the .hip files, host code of the .cpp files — and the two sets are compared.  bash tools/fmt_check.sh   (about ten minutes on 8 cores)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); W=${TMPDIR:-/tmp}/fmt_check; rm -rf "$W"; mkdir -p "$W/before" "$W/after" "$W/head"
git -C "$ROOT" archive HEAD blockmaze_amd/csrc include | tar -x -C "$W/head"
asm() {   # asm <source tree> <output dir>
  local src=$1 out=$2
  ( cd "$src/blockmaze_amd/csrc"
    for f in *.hip; do echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -o $out/${f%.hip}.device.s $f"; echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-host-only -o $out/${f%.hip}.host.s $f"; done
    for f in *.cpp; do echo "g++ -O3 -std=c++17 -fPIC -S -o $out/${f%.cpp}.s $f"; done ) | ( cd "$src/blockmaze_amd/csrc" && xargs -P 8 -I{} sh -c '{} 2>/dev/null' )
}
asm "$W/head" "$W/before"; asm "$ROOT" "$W/after"
bad=0
for f in "$W"/before/*.s; do
  b=$(basename "$f")
  # (.file / .ident lines name paths and compilers; an assert() carries its line number as an immediate: those differences are listed, not hidden)
  if ! diff <(grep -v '^\s*\.file\|^\s*\.ident' "$f") <(grep -v '^\s*\.file\|^\s*\.ident' "$W/after/$b") > "$W/$b.diff"; then echo "DIFFERENT: $b ($(wc -l < "$W/$b.diff") diff lines, $W/$b.diff)"; bad=1; else echo "same: $b"; fi
done
exit $bad
