"""Static instruction histogram per source line of one kernel, from assembly made with -gline-tables-only
(.loc directives).  usage: asm_by_line.py file.s <kernel-substring> [lo hi]   -- lines lo..hi of the source only"""
import re, sys, collections
f, key = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1 << 30)
text = open(f).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l and ":" in l[:400] and l.split(":")[0].startswith("_Z") and not l.startswith("\t"))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
cur = 0
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for l in text[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)", l)
    if m:
        cur = int(m.group(1)); continue
    m = re.match(r"\s+([sv]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|scratch_[a-z0-9_]+|flat_[a-z0-9_]+|buffer_[a-z0-9_]+)", l)
    if m and lo <= cur <= hi:
        cnt[cur] += 1
        op = m.group(1)
        k = "readlane" if "readlane" in op or "writelane" in op else ("salu" if op.startswith("s_") else ("lds" if op.startswith("ds_") else ("valu" if op.startswith("v_") else "mem")))
        kinds[cur][k] += 1
tot = sum(cnt.values())
print(f"{tot} instructions in lines {lo}..{hi}")
for line, n in sorted(cnt.items()):
    print(f"{line:5d} {n:5d}  {dict(kinds[line])}")
