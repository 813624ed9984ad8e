"""CPU checks of the boundary: the shared library loads without a GPU, exports every symbol include/reptext_hip.h
declares, the ctypes binding covers exactly that set, struct layouts agree, and the product never touches oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "arabic-text-image-generation-reptext_amd")
HEADER = os.path.join(ROOT, "include", "reptext_hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    from reptext_amd import native

    lib = native.load()
    assert lib.rt_version().startswith(b"reptext_hip")
    syms = header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in reptext_hip.h but not exported"
    bound = set(native.SIGNATURES) | {"rt_version"}
    assert bound == set(syms), f"binding/header mismatch: {bound ^ set(syms)}"


def test_gemm_group_struct_layout_matches_c():
    """Compile a tiny host program against the header and compare sizeof/offsetof with the ctypes mirror."""
    from reptext_amd import native

    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "reptext_hip.h"
int main(){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(rt_gemm_group), offsetof(rt_gemm_group, lda), offsetof(rt_gemm_group, strideA),
  offsetof(rt_gemm_group, M), offsetof(rt_gemm_group, out_f32), offsetof(rt_gemm_group, alpha)); return 0; }
'''
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    src, exe = os.path.join(d, "_layout.c"), os.path.join(d, "_layout")
    open(src, "w").write(prog)
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    G = native.GemmGroup
    assert [int(x) for x in out] == [ctypes.sizeof(G), G.lda.offset, G.strideA.offset, G.M.offset, G.out_f32.offset, G.alpha.offset]


def test_calls_are_rejected_not_crashed_without_gpu_memory():
    """Argument validation happens on the host before any launch: null pointers / bad shapes return RT_E_* codes."""
    from reptext_amd import native

    lib = native.load()
    g = native.GemmGroup()
    assert lib.rt_gemm_bf16(ctypes.pointer(g), 1, None) == -1            # RT_E_BADARG (null A/W/C)
    assert lib.rt_gemm_bf16(ctypes.pointer(g), 9, None) == -1            # too many groups
    assert lib.rt_euler_step(None, None, 0.0, 10, None) == -1
    assert lib.rt_layernorm_modulate(8, 8, 0, 0, 8, 8, 0, None, None, 0, 1, 1, 12, 1e-6, None) in (-2, -3)   # D % 8 != 0


def test_product_never_imports_the_oracle_and_has_no_cpu_fallback():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle." in txt and f.endswith(".py") and "import oracle" in txt:
                    bad.append(f)
    for f in ("controlnet_flux.py", "pipeline_flux_controlnet.py", "pipeline_flux_controlnet_inpaint.py", "reptext_amd.py"):
        p = os.path.join(ROOT, f)
        if os.path.exists(p) and re.search(r"^\s*(from|import)\s+oracle\b", open(p).read(), flags=re.M):
            bad.append(f)
    assert not bad, f"product files import the oracle: {bad}"


def test_ops_refuse_cpu_tensors():
    import torch

    import reptext_amd.ops as ops

    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(a, a, a.clone())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.euler_step_(a.clone(), a, 0.1)


def test_missing_library_is_loud(monkeypatch):
    from reptext_amd import native

    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/librt_reptext_hip.so")
    with pytest.raises(native.NativeLibraryMissing):
        native.load()


def test_attention_v3_code_object_keeps_out_of_the_asm_owned_registers():
    """csrc/attention_v3.hip names the accumulator registers a0..a191 literally (O and Q live there for a whole work item). The
    compiler must not have put anything of its own in them: no VGPR spill, no scratch, no compiler-generated v_accvgpr_* that
    names a0..a191 (tools/audit_v3_asm.py over the -S output of the same flags the Makefile builds with)."""
    import shutil
    import subprocess

    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    r = subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "audit_v3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK " in r.stdout
