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
ok = meta.get("vgpr_spill_count") == 0 and meta.get("private_segment_fixed_size") == 0 and meta.get("agpr_count", 0) >= 192 and not stray
print(("OK  " if ok else "FAIL") + f" {meta} stray_accvgpr={len(stray)}")
for s in stray[:10]:
    print("   ", s)
sys.exit(0 if ok else 1)
