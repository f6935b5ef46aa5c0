"""Static VALU instructions of one kernel per SOURCE LINE (hipcc -S -gline-tables-only listing): where the code of a block
lives.  usage: isa_lines.py file.s kernel-substring [source-file-substring]"""
import re, sys, collections
path, want = sys.argv[1], sys.argv[2]
src = sys.argv[3] if len(sys.argv) > 3 else "rt_kernels.h"
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
start = [i for i, l in enumerate(lines) if l.startswith("_ZN") and want in l and l.split(";")[0].rstrip().endswith(":")][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
cur = None
valu = collections.Counter()
allc = collections.Counter()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    if cur and src in cur[0]:
        allc[cur[1]] += 1
        if t.startswith("v_"):
            valu[cur[1]] += 1
srcl = open(src if "/" in src else "/root/repo/raytracing_amd/csrc/" + src).read().split("\n")
tot = sum(valu.values())
print("total VALU attributed to", src, tot)
for ln in sorted(valu):
    print(f"{ln:5d} {valu[ln]:4d} {allc[ln]:4d}  {srcl[ln - 1].strip()[:110]}")
