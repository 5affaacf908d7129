#!/usr/bin/env python3
"""Lists the COMPILER-generated s_waitcnt vmcnt(..) of every kernel (those outside inline-asm blocks) with the loop depth
they sit at, the number of global_load_lds / global_load / scratch ops of the kernel, and its scratch size.  A vmcnt(0)
that the compiler puts in front of the first use of a plain or scratch load also drains every LDS-DMA piece in flight:
inside a tile / step loop that is a serialised round trip (this is how the stem_block prefetch drain and the
whole-stage kernels' patch-address reloads were found).  usage: isa_waits.py [-Dflag ...] [--kernel substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fastintercu-vvc_amd", "csrc", "mlt_kernels.hip")


def collect(defs):
    """{demangled kernel name: {glds, gload, scratch, waits: [(asm line, loop depth, text)]}} of mlt_kernels.hip built with defs"""
    out = "/tmp/isa_waits.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *defs, SRC, "-o", out],
                          stderr=subprocess.DEVNULL)
    name, in_asm, depth, stats = None, False, 0, {}
    for ln, line in enumerate(open(out), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void ", "", name)[:100]
            depth, stats[name] = 0, {"glds": 0, "gload": 0, "scratch": 0, "waits": []}
            continue
        if name is None:
            continue
        if "s_endpgm" in line:
            name = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        m = re.search(r"Depth[= ](\d+)", line)
        if m and line.lstrip().startswith((".LBB", ";")):
            depth = int(m.group(1))
        elif line.startswith(".LBB") and "Loop" not in line:
            depth = 0
        st = stats[name]
        t = line.strip()
        if t.startswith("global_load_lds"):
            st["glds"] += 1
        elif (t.startswith("global_load") or t.startswith("buffer_load")) and not in_asm:
            st["gload"] += 1
        elif t.startswith("scratch_"):
            st["scratch"] += 1
        elif t.startswith("s_waitcnt") and "vmcnt" in t and not in_asm:
            st["waits"].append((ln, depth, t))
    return stats


def scratch_in_loops(prefix, path="/tmp/isa_waits.s"):
    """[(asm line, depth, text)] of the scratch_* instructions of the kernels whose demangled name starts with `prefix` that sit inside a loop (collect() wrote `path`)"""
    hits, name, depth = [], None, 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = re.sub(r"^void ", "", subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip())
            depth = 0
            continue
        if name is None or not name.startswith(prefix):
            continue
        if "s_endpgm" in line:
            name = None
            continue
        m = re.search(r"Depth[= ](\d+)", line)
        if m and line.lstrip().startswith((".LBB", ";")):
            depth = int(m.group(1))
        elif line.startswith(".LBB") and "Loop" not in line:
            depth = 0
        if line.strip().startswith("scratch_") and depth >= 1:
            hits.append((ln, depth, line.strip()))
    return hits


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    want = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else ""
    for k, st in collect(defs).items():
        if want not in k:
            continue
        inner = [w for w in st["waits"] if w[1] >= 1]
        print(f"{k}\n    glds {st['glds']}  loads {st['gload']}  scratch ops {st['scratch']}  compiler vmcnt waits {len(st['waits'])} ({len(inner)} inside loops)")
        for ln, d, t in inner[:40]:
            print(f"      line {ln:6d} depth {d}: {t}")


if __name__ == "__main__":
    main()
