"""Assembles an instrumented kernel, putting trampolines where the inserted counters pushed a branch out of reach.
A gfx950 branch holds a 16-bit offset in instructions' dwords (+-128 KB); tools/blockprof/instrument.py with `lanes` more than triples the
kernel.  For every branch the assembler refuses ("branch size exceeds simm16") a label with one `s_branch <target>` is put about half way,
behind an instruction control never falls through (s_branch, s_endpgm, s_setpc_b64), and the refused branch goes there instead; repeated until
the file assembles.  Trampolines are not counted blocks: the report's branch count misses the hops over them.
usage: relax.py in.s out.o    (rewrites in.s in place)"""
import re, subprocess, sys

src, obj = sys.argv[1], sys.argv[2]
CLANG = "/opt/rocm/lib/llvm/bin/clang"
NO_FALL = re.compile(r"^\s+(s_branch|s_endpgm|s_setpc_b64)\b")
n_tramp = 0
for it in range(200):
    r = subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-ferror-limit=0", "-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode == 0:
        print(f"relax: assembled after {it} rounds, {n_tramp} trampolines")
        sys.exit(0)
    bad = [int(m.group(1)) for m in re.finditer(r":(\d+):\d+: error: branch size exceeds simm16", r.stderr)]
    if not bad:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(1)
    lines = open(src).read().split("\n")
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^([.\w$]+):", l)
        if m:
            labels[m.group(1)] = i
    inserts = {}  # line index -> list of lines to put after it
    for ln in bad:
        i = ln - 1
        m = re.match(r"^(\s+s_c?branch\w*\s+)([.\w$]+)\s*$", lines[i].split(";")[0].rstrip())
        assert m, lines[i]
        tgt = labels[m.group(2)]
        mid = (i + tgt) // 2
        spot = None
        for d in range(abs(tgt - i) // 2 - 1):
            for c in (mid + d, mid - d):
                if min(i, tgt) < c < max(i, tgt) and NO_FALL.match(lines[c]):
                    spot = c
                    break
            if spot is not None:
                break
        assert spot is not None, f"no place for a trampoline between lines {i} and {tgt}"
        name = f".Lbp_tramp_{n_tramp}"
        n_tramp += 1
        inserts.setdefault(spot, []).extend([f"{name}:", f"\ts_branch {m.group(2)}"])
        lines[i] = f"{m.group(1)}{name}"
    out = []
    for i, l in enumerate(lines):
        out.append(l)
        out.extend(inserts.get(i, []))
    open(src, "w").write("\n".join(out))
sys.exit("relax: did not converge")
