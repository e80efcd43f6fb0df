#!/bin/bash
# Build librdyhip.so from several git revisions (HERE, no GPU needed) into ab_libs/<rev>.so,
# to be compared on the GPU box with tools/ab_libs_run.sh.   usage: tools/ab_revisions.sh <rev> [<rev> ...]
set -e
mkdir -p ab_libs
for rev in "$@"; do
  tmp=$(mktemp -d)
  git archive "$rev" rdycore_amd/csrc include | tar -x -C "$tmp"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$tmp/include" -I"$tmp/rdycore_amd/csrc" \
      -o "ab_libs/$(echo $rev | tr '/~^' '___').so" "$tmp/rdycore_amd/csrc/rdyhip_api.hip"
  rm -rf "$tmp"
done
ls -la ab_libs
