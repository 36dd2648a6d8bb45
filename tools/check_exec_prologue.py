#!/usr/bin/env python
"""Static check of gfx950 assembly for the block-prologue miscompile of ROCm 7.2 clang (tools/repro/README.md, row
"the lanes that skipped a branch lose a register").

A divergent branch is `s_and_saveexec_b64 s[a:b], cond ; s_cbranch_execz JOIN ; <body> ; JOIN: s_or_b64 exec, exec, s[a:b]`:
the lanes that skip the body are switched off, and the `s_or_b64` at the top of JOIN switches them on again.  Everything
that moves per-lane data in JOIN must come AFTER that `s_or_b64`.  This compiler's register allocator sometimes puts the
copies of a live-range split (`v_mov_b32 vA, vB`) and spill stores (`scratch_store ... Folded Spill`) at the top of JOIN
IN FRONT of the restore: they run with the skipping lanes still off (with NO lane on when JOIN is entered through the
`s_cbranch_execz`), so those lanes' values are not copied / not stored, and what the code after the branch reads is stale.
Found in round 6 — it is the round-3/4 "load_lds_grid inlining" failure: in the 1024 x 3 correspondence-pass
instantiation of ieskf_lds_kernel the JOIN in front of walk_lds's task loop copies, among others, the register that
holds the partner lane of merge_query_lanes<3>; the line queries skip the branch before it (the plane queries' class-2
seed), read another lane's candidate, and get grid position 0 as their second point.  Whether the allocator splits there
depends on register pressure, i.e. on unrelated changes (inlining load_lds_grid, a reordered statement).

The check: a block that is the target of an `s_cbranch_execz` must not have anything but scalar instructions,
`v_readlane_b32` / `v_readfirstlane_b32` / `v_writelane_b32` (they ignore exec) and waits in front of its first
`s_or_b64 exec, exec, ...` (looked at up to the first instruction that narrows exec again: from there on the block is
inside a region of its own).  (A block entered with `s_cbranch_execnz` / a scalar branch is a branch BODY; when the compiler
merges a body with its closing restore, work in front of the restore is what it should be — not reported.)

usage: tools/check_exec_prologue.py file.s [...]      device assembly (hipcc -S --offload-device-only, same flags)
       tools/check_exec_prologue.py file.so|file.o ...  the gfx950 code objects INSIDE a built library / object (what ships): disassembled
                                                        with llvm-objdump --symbolize-operands; __graft_entry__.build() runs this on the library
       tools/check_exec_prologue.py --build            compiles every csrc/*.hip to assembly with the Makefile's flags and checks it
exit code 1 when a site is found; each is printed with its kernel, block label and the instructions in front of the restore."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lins---lidar-inertial-slam_amd", "csrc")

LABEL = re.compile(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):|^[0-9a-f]+ <(L\d+)>:")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s+; @|^[0-9a-f]+ <([A-Za-z_][\w$.]*)>:")
EXEC_OR = re.compile(r"^\s+s_or_b64 exec, exec,")
EXECZ = re.compile(r"^\s+s_cbranch_execz (\.LBB\d+_\d+|L\d+)\b")
HARMLESS = re.compile(r"^\s+(s_\w+|v_readlane_b32|v_readfirstlane_b32|v_writelane_b32)\b")
NARROW = re.compile(r"^\s+s_\w+_saveexec_b64\b|^\s+s_(and|andn2|xor|mov|cmov|wqm)_b64 exec,")  # the block opens a region of its own: what follows runs under THAT mask
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def check_lines(lines):
    # functions first: objdump's labels (L0, L1, ...) start again in every function
    funcs, cur = [], ("?", [])
    for line in lines:
        m = FUNC.match(line)
        if m and not LABEL.match(line):
            funcs.append(cur)
            cur = (m.group(1) or m.group(2), [])
        else:
            cur[1].append(line)
    funcs.append(cur)
    sites = []
    for func, body in funcs:
        joins = {m.group(1) for m in (EXECZ.match(l) for l in body) if m}
        block, pre, live = None, [], False
        for line in body:
            m = LABEL.match(line)
            if m:
                block = m.group(1) or m.group(3)
                pre, live = [], block in joins
                continue
            if not live or not line.startswith("\t") or line.lstrip().startswith((";", ".")):
                continue
            if EXEC_OR.match(line):
                bad = [p for p in pre if not HARMLESS.match(p)]
                if bad:
                    sites.append((func, block, [re.split(r";|//", b)[0].strip() for b in bad]))
                live = False  # only the first restore of a block has a prologue in front of it
                continue
            if NARROW.match(line):
                live = False
                continue
            pre.append(line)
    return sites


def check(path):
    if not path.endswith((".so", ".o")):
        return check_lines(open(path, errors="replace").read().splitlines())
    sites = []
    with tempfile.TemporaryDirectory() as tmp:  # the device code objects of a fat binary
        subprocess.run(["cp", path, os.path.join(tmp, "in.bin")], check=True)
        subprocess.run([OBJDUMP, "--offloading", "in.bin"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in sorted(os.listdir(tmp)) if "gfx950" in f]
        if not cos:
            sys.exit(f"{path}: no gfx950 code object found")
        for co in cos:
            out = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", os.path.join(tmp, co)], stdout=subprocess.PIPE, check=True).stdout.decode(errors="replace")
            sites += check_lines(out.splitlines())
    return sites


def build_asm(outdir):
    """device assembly of every .hip of the library, compiled as the Makefile compiles it (make -n gives the command lines)"""
    out = subprocess.run(["make", "-n", "-B", "all"], cwd=CSRC, stdout=subprocess.PIPE, check=True).stdout.decode()
    files = []
    for cmd in out.splitlines():
        m = re.search(r"-c -o (\S+)\.o (?:-x hip )?(\S+)$", cmd)
        if not m or "hipcc" not in cmd:
            continue
        src = m.group(2)
        s = os.path.join(outdir, os.path.basename(m.group(1)) + ".s")
        cmd2 = re.sub(r"-c -o \S+\.o", f"--offload-device-only -S -o {s}", cmd)
        files.append((src, s, cmd2))
    procs = [(src, s, subprocess.Popen(c, shell=True, cwd=CSRC, stderr=subprocess.PIPE)) for src, s, c in files]
    done = []
    for src, s, p in procs:
        err = p.communicate()[1].decode()
        if p.returncode:
            sys.exit(f"{src}: {err[-2000:]}")
        done.append(s)
    return done


def main():
    args = sys.argv[1:]
    if args == ["--build"]:
        tmp = tempfile.mkdtemp(prefix="exec_prologue_")
        args = build_asm(tmp)
    n = 0
    for path in args:
        sites = check(path)
        n += len(sites)
        print(f"{os.path.basename(path)}: {len(sites)} site(s)")
        for func, block, bad in sites:
            print(f"  {func[:110]}  {block}: {len(bad)} in front of the exec restore: " + " | ".join(bad[:8]) + (" ..." if len(bad) > 8 else ""))
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())
