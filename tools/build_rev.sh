#!/bin/bash
# build_rev.sh REV NAME -- build the HIP library of git revision REV into build_ab/libpinn_hip_NAME.so (git-ignored,
# travels with gpurun), for A/B timing against the working tree's library on one box:
#   PINN_HIP_LIB=build_ab/libpinn_hip_NAME.so python tools/kbench.py --arch wide4x128x3 --points 100000
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REV="$1"; NAME="$2"
WT="$(mktemp -d /tmp/pinn_wt.XXXXXX)"
git -C "$ROOT" worktree add -f "$WT" "$REV" -q
(cd "$WT" && python -c "from pinn_for_quantum_wavefunction_surfaces_amd import build as b; print(b.build_hip(force=True))")
mkdir -p "$ROOT/build_ab"
cp "$WT/pinn_for_quantum_wavefunction_surfaces_amd/libpinn_hip.so" "$ROOT/build_ab/libpinn_hip_$NAME.so"
git -C "$ROOT" worktree remove --force "$WT"
echo "$ROOT/build_ab/libpinn_hip_$NAME.so"
