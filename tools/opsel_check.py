#!/usr/bin/env python3
"""List packed-fp32 VALU instructions whose LOW result lane reads the HIGH dword of an operand pair
(v_pk_{mul,fma,add}_f32 ... op_sel:[..1..]) per kernel of the library's translation units (the matrix-core trunk files csrc/pinn_trunk_*.hip,
csrc/pinn_rnet_mfma.hip, and csrc/pinn_kernels.hip).

    python tools/opsel_check.py [substring ...]

Why: on MI355X these forms gave run-to-run different products in k_trunk2 when two workgroups shared a CU
(profiles/r02_packed_opsel.md); the ring-fed arithmetic of the merged-pass kernels is written so that none is
emitted there (lin2 / fmac_bcast in the source; bcast_elem in pinn_math.h for the fp32 vector kernels, round 4).  The
tool compiles the translation units to assembly (~60 s, in parallel)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pinn_for_quantum_wavefunction_surfaces_amd", "csrc")


SCRATCH = {}     # kernel symbol -> bytes of scratch per lane (filled by scan)


def scan(asm_path):
    cur, hits, tot = None, collections.defaultdict(list), collections.Counter()
    for line in open(asm_path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
        if cur and re.search(r"\tv_pk_(mul|fma|add)_f32", line):
            tot[cur] += 1
            if re.search(r"op_sel:\[", line):
                hits[cur].append(line.strip())
        m = re.match(r"^; ScratchSize: (\d+)", line)
        if m and cur:
            SCRATCH[cur] = int(m.group(1))
    return hits, tot


def compile_and_scan():
    sys.path.insert(0, ROOT)
    from pinn_for_quantum_wavefunction_surfaces_amd.build import UNITS
    hits, tot = collections.defaultdict(list), collections.Counter()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for objname, src, extra in UNITS:
            if src in ("pinn_xchg.hip", "pinn_trunk_mfma.hip"):      # no arithmetic kernels (exchange; dispatch only)
                continue
            out = os.path.join(d, objname + ".s")
            procs.append((out, subprocess.Popen(
                ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value"] + extra +
                ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)],
                stderr=subprocess.DEVNULL)))
        for out, pr in procs:
            if pr.wait() != 0:
                raise RuntimeError("hipcc failed")
            h, t = scan(out)
            hits.update(h)
            tot.update(t)
    return hits, tot


def main():
    filt = sys.argv[1:]
    hits, tot = compile_and_scan()
    names = subprocess.run(["c++filt"], input="\n".join(tot), capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(tot, names):
        if filt and not all(f in k or f in n for f in filt):
            continue
        print("%3d of %4d packed ops  %s" % (len(hits[k]), tot[k], k if n.startswith("_Z") else n[:100]))
        for h in hits[k]:
            print("        " + h)


if __name__ == "__main__":
    main()
