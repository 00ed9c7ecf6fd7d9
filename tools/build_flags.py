#!/usr/bin/env python3
"""build_flags.py NAME FLAG [FLAG ...] -- the working tree's library with extra hipcc flags (e.g. -DPINN_STASH_PAD=4096) into
build_ab/libpinn_hip_NAME.so, for A/B runs with tools/ab_libs.sh (PINN_HIP_LIB)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stamps  # noqa: E402

if __name__ == "__main__":
    name, flags = sys.argv[1], sys.argv[2:]
    os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
    stamps.build(flags, os.path.join(ROOT, "build_ab", "libpinn_hip_%s.so" % name), "ab_" + name)
