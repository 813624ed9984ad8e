"""Audit of csrc/attention_v3.hip's code object (make -C csrc audit_v3): the kernel names the accumulator registers a[0:191] literally, so the
compiler must never have put a value of its own there. Requires: no VGPR spills, no scratch, and no compiler-generated
v_accvgpr_* (outside an ;;#ASMSTART / ;;#ASMEND block) that names a0..a191 — hipcc may park long-lived values of the cold
partial-record path in accumulator registers of its own choice above a191 (they are clobber-free), never in ours."""
import re, sys

path = sys.argv[1]
txt = open(path).read()
meta = {k: int(v) for k, v in re.findall(r"\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|agpr_count|vgpr_count):\s+(\d+)", txt)}
inasm, stray = False, []
for n, line in enumerate(txt.split("\n"), 1):
    if "#ASMSTART" in line:
        inasm = True
    elif "#ASMEND" in line:
        inasm = False
    elif "v_accvgpr" in line and not inasm:
        regs = [int(x) for x in re.findall(r"\ba(\d+)\b", line)] + [int(x) for x in re.findall(r"a\[(\d+):", line)]
        if not regs or min(regs) < 192:
            stray.append((n, line.strip()))
# the steady-state tile loop = the first innermost loop of the listing: no scratch access may sit inside it (spills in the per-segment
# prologue / drain / combine code are tolerated and reported)
lines = txt.split("\n")
hot = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
hot_scratch = None
if hot:
    end = next((i for i in range(hot[0], len(lines)) if "s_cbranch_scc" in lines[i] or "s_cbranch_vcc" in lines[i] and False), len(lines))
    # walk to the loop's back edge: the first s_cbranch_scc* after 64 MFMAs
    n = 0
    for i in range(hot[0], len(lines)):
        if "v_mfma" in lines[i]:
            n += 1
        if n >= 64 and "s_cbranch_scc" in lines[i]:
            end = i
            break
    hot_scratch = sum("scratch_" in l for l in lines[hot[0]:end])
ok = hot_scratch == 0 and meta.get("agpr_count", 0) >= 192 and not stray
meta["hot_loop_scratch_ops"] = hot_scratch
print(("OK  " if ok else "FAIL") + f" {meta} stray_accvgpr={len(stray)}")
for s in stray[:10]:
    print("   ", s)
sys.exit(0 if ok else 1)
