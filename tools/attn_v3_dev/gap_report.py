"""Per-MFMA-gap instruction counts of attention_v3's steady-state tile loop (from `make -C csrc audit_v3`'s .s): for every gap
vector-ALU / transcendental / LDS / (D)MA / s_nop / s_waitcnt / other scalar instructions, and the issue-cycle estimate of
MI355X_MICROARCH.md's constants (MFMA 8, exp 8, other VALU 4, LDS 4, scalar 4, DMA piece 60)."""
import sys
p = sys.argv[1] if len(sys.argv) > 1 else "arabic-text-image-generation-reptext_amd/csrc/build/attention_v3.s"
lines = [l.strip() for l in open(p)]
hot = next(i for i, l in enumerate(lines) if "Inner Loop Header" in l)
gaps, cur, n = [], None, 0
for l in lines[hot:]:
    if not l or l.startswith(";") or l.startswith("."):
        continue
    if l.startswith("v_mfma"):
        n += 1
        if cur is not None:
            gaps.append(cur)
        if n > 72:
            cur = None
            break
        cur = dict(v=0, x=0, ds=0, dma=0, nop=0, w=0, s=0)
    elif cur is not None:
        if l.startswith("v_exp"): cur["x"] += 1
        elif l.startswith("v_"): cur["v"] += 1
        elif l.startswith("ds_"): cur["ds"] += 1
        elif l.startswith("buffer_load"): cur["dma"] += 1
        elif l.startswith("s_nop"): cur["nop"] += 1
        elif l.startswith("s_waitcnt"): cur["w"] += 1
        elif l.startswith("s_"): cur["s"] += 1
est = lambda g: 8 + 8 * g["x"] + 4 * (g["v"] + g["ds"] + g["nop"] + g["w"] + g["s"]) + 60 * g["dma"]
tail, gaps = gaps[-1], gaps[:-1]      # the 72nd gap runs from the tile's last MFMA to the next MFMA in PROGRAM order (back-edge, other tile variants): not a steady-state gap
tot = {k: sum(g[k] for g in gaps) for k in gaps[0]}
print(f"{len(gaps)} gaps; totals {tot}; issue estimate {sum(est(g) for g in gaps)} cycles, pipe floor {32 * len(gaps)}; over-32 excess {sum(max(0, est(g) - 32) for g in gaps)}")
for i in range(0, len(gaps), 18):
    print(" ".join(f"{est(g):3d}" for g in gaps[i:i + 18]))
