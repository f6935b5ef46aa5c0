"""Static instruction statistics of one kernel in a hipcc -S listing: per basic block the VALU count, SGPR-spill lane
moves (v_readlane / v_writelane), scratch accesses and register copies.  usage: isa_stats.py file.s [kernel-substring] [top]"""
import re, sys
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "render_kernelIfLb1ELi1024E"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN") and want in l and l.rstrip().split(";")[0].rstrip().endswith(":")][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
blocks, cur = [], ("entry", start, [])
for n in range(start, end + 1):
    l = lines[n]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or l.startswith("; %bb."):
        blocks.append(cur)
        cur = (m.group(1) if m else l.split()[1].rstrip(":"), n + 1, [])
    else:
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur[2].append(t)
blocks.append(cur)
rows = []
for name, line, ins in blocks:
    valu = sum(t.startswith("v_") for t in ins)
    lane = sum(t.startswith(("v_readlane", "v_writelane")) for t in ins)
    scr = sum(t.startswith("scratch_") for t in ins)
    mov = sum(t.startswith(("v_mov", "v_pk_mov", "v_accvgpr")) for t in ins)
    lds = sum(t.startswith("ds_") for t in ins)
    rows.append((name, line, len(ins), valu, lane, scr, mov, lds))
print(f"{len(rows)} blocks, {sum(r[2] for r in rows)} instructions, {sum(r[3] for r in rows)} VALU of which "
      f"{sum(r[4] for r in rows)} lane moves and {sum(r[6] for r in rows)} copies; {sum(r[5] for r in rows)} scratch, {sum(r[7] for r in rows)} LDS")
print("name line instr valu lane scratch mov lds")
for r in sorted(rows, key=lambda r: -(r[4] + r[5]))[:top]:
    print(*r)
