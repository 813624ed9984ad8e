"""Timing-only variants of the block-scaled GEMM (generated copies of csrc/, never the product library): which of the additions costs
what. variants listed below."""
import os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "arabic-text-image-generation-reptext_amd", "csrc")
for name in (sys.argv[1:] or ("OPSEL0", "NOREAD", "NODMA", "FIXADDR")):
    dst = os.path.join(root, "tools", "mx_dev", "lib_" + name)
    shutil.rmtree(dst, ignore_errors=True)
    shutil.copytree(src, os.path.join(dst, "csrc"), ignore=shutil.ignore_patterns("build"))
    p = os.path.join(dst, "csrc", "gemm_bf16.hip")
    s = open(p).read()
    n0 = s
    if name == "OPSEL0":                   # every fragment takes byte 0 of the scale dword (wrong values; timing only)
        s = s.replace("MX ? ((i) & 3) : 0, MX ? sc : 0x7F7F7F7F", "0, MX ? sc : 0x7F7F7F7F")
    if name == "NOREAD":                   # no ds_read_b32 of the scales: constant 2^0 operand
        s = s.replace("if constexpr (MX) sc = *reinterpret_cast<const int*>(smem + sc_rd", "if constexpr (false) sc = *reinterpret_cast<const int*>(smem + sc_rd")
    if name == "NODMA":                    # the scale piece is never issued
        s = s.replace("if constexpr (MX) { if ((kt & 7) == 7) issue_scales((kt + 1) >> 3); }", "")
    if name == "FIXADDR":                  # the scale dword is read from a K-tile-independent address (wrong values; timing only)
        s = s.replace("(((kt_) & 15) << 10) + (ah)*256);", "(ah)*256);")
    assert s != n0, name
    open(p, "w").write(s)
    subprocess.check_call(["make", "-s", "-C", os.path.join(dst, "csrc"), "ROOT=" + root], stderr=subprocess.DEVNULL)
    print(name, "built")
