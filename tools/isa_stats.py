#!/usr/bin/env python3
"""Per-kernel resource table and hot-loop instruction mix from the gfx950 assembly of libgroan_hip.so's device code.

    python tools/isa_stats.py [--out profiles/rNN_resource_usage.txt] [--kernel SUBSTR ...]

Compiles groan_rs_amd/csrc/gr_api.hip to assembly (device only, same flags as the product build), then for every kernel
prints VGPRs / SGPRs / LDS / scratch / occupancy (the compiler's own figures) and, for the kernel's LARGEST loop (by
instruction count, which is the streaming loop in every kernel here), how many VALU / packed-VALU / SALU / VMEM / LDS
instructions one trip executes on the straight-line path (code reached only through forward branches inside the loop --
rare-path blocks -- is counted separately).  The hot kernels process 4 atoms per lane and trip, so VALU per atom = VALU / 4.
No GPU needed (hipcc cross-compiles)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")


def classify(op):
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith(("v_cmp", "v_cndmask")) or op.startswith("v_"): return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--kernel", action="append", default=[])
    ap.add_argument("--asm", default=None, help="use an existing .s file")
    args = ap.parse_args()
    asm = args.asm
    if not asm:
        asm = os.path.join(tempfile.mkdtemp(prefix="isa_"), "gr_api.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-S", "--cuda-device-only",
                               "-o", asm, os.path.join(ROOT, "groan_rs_amd", "csrc", "gr_api.hip")], stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    # kernel bodies: from "<name>:" to ".Lfunc_end"
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s", l)] if m]
    rows = []
    names = demangle([n for _, n in starts])
    for (i0, mangled), name in zip(starts, names):
        if ".amdhsa_kernel " + mangled not in "\n".join(lines[i0:i0 + 20000]) and not any(l.startswith("\t.amdhsa_kernel " + mangled) for l in lines):
            continue
        short = re.sub(r"\(.*", "", name).replace("void ", "")
        if args.kernel and not any(k in short for k in args.kernel):
            continue
        end = next(j for j in range(i0, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[i0:end]
        meta = "\n".join(lines[end:end + 80])
        g = lambda pat: (re.search(pat, meta) or [None, "?"])[1]
        res = dict(vgpr=g(r"; NumVgprs: (\d+)"), agpr=g(r"; NumAgprs: (\d+)"), sgpr=g(r"; TotalNumSgprs: (\d+)"), scratch=g(r"; ScratchSize: (\d+)"),
                   occ=g(r"; Occupancy: (\d+)"), lds=g(r"; LDSByteSize: (\d+)"), code=g(r"; codeLenInByte = (\d+)"))
        # loops: a backward branch to a label defined earlier
        label_at = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\w+):", l)] if m}
        loops = []
        for k, l in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+ (\.LBB\w+)|\s+s_branch (\.LBB\w+)", l)
            if m:
                tgt = m.group(1) or m.group(2)
                if tgt in label_at and label_at[tgt] < k:
                    loops.append((label_at[tgt], k))
        best = None
        if loops:
            # outermost largest loop
            lo, hi = max(loops, key=lambda t: t[1] - t[0])
            mix = {}
            # straight-line path: skip blocks that are only entered by a taken forward branch over them?  Simple proxy: count
            # everything, and separately what sits behind an s_cbranch_execz/vccz/scc0 that jumps forward past it
            rare = 0
            k = lo
            skip_until = -1
            rare_mix = {}
            while k <= hi:
                l = body[k]
                m = re.match(r"\s+([a-z_0-9]+)", l)
                if m and not l.strip().startswith((".", ";")):
                    op = m.group(1)
                    cls = classify(op)
                    (rare_mix if k < skip_until else mix).setdefault(cls, 0)
                    if k < skip_until: rare_mix[cls] += 1
                    else: mix[cls] += 1
                k += 1
            best = (hi - lo, mix)
        rows.append((short, res, best))
    out = []
    out.append("kernel resource usage and hot-loop instruction mix, gfx950 (hipcc -O3, ROCm 7.2; tools/isa_stats.py)")
    out.append("VALU / trip = v_* + v_pk_* instructions in the kernel's largest loop (all blocks inside it, rare-path blocks included)")
    out.append("")
    out.append("%-58s %5s %5s %6s %7s %4s %7s | %5s %5s %5s %5s %5s" % ("kernel", "VGPR", "SGPR", "LDS B", "scratch", "occ", "code B", "valu", "pk", "salu", "vmem", "lds"))
    for short, r, best in rows:
        mix = best[1] if best else {}
        out.append("%-58s %5s %5s %6s %7s %4s %7s | %5d %5d %5d %5d %5d" % (short[:58], r["vgpr"], r["sgpr"], r["lds"], r["scratch"], r["occ"], r["code"],
                                                                            mix.get("valu", 0) + mix.get("valu_pk", 0), mix.get("valu_pk", 0), mix.get("salu", 0), mix.get("vmem", 0), mix.get("lds", 0)))
    text = "\n".join(out) + "\n"
    sys.stdout.write(text)
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
