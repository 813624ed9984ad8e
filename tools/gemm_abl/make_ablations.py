"""Timing-only ablations of gemm_pp_kernel<bf16> (results are wrong by construction): which part of a tile costs what?
    python tools/gemm_abl/make_ablations.py ; then on the GPU box:  for b in tools/gemm_abl/abl_*; do $b 4608 21504 3072; done"""
import os, subprocess, shutil
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
csrc = os.path.join(root, "arabic-text-image-generation-reptext_amd", "csrc")
s = open(os.path.join(csrc, "gemm_bf16.hip")).read()
shutil.copy(os.path.join(csrc, "rt_common.h"), here)
i = s.index("template <bool FP8>\n__global__ __launch_bounds__(THREADS, 2) void gemm_pp_kernel")
head, tail = s[:i], s[i:]


def rep(old, new, where="tail"):
    global head, tail
    src = tail if where == "tail" else head
    assert old in src, old[:60]
    src = src.replace(old, new, 1)
    if where == "tail":
        tail = src
    else:
        head = src


# the epilogue ablation lives in the generated copy only (the product kernel carries no timing-only switches)
rep("  if (g.out_f32) epilogue_tile<true, FP8>(g, bidx, mrow, ncol, acc);\n",
    "#ifdef ABL_NOEPI\n  { float sacc = 0.f;\n    _Pragma(\"unroll\") for (int i = 0; i < 8; ++i) _Pragma(\"unroll\") for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];\n"
    "    if (sacc == 12345.f) reinterpret_cast<float*>(g.C)[0] = sacc; }\n  if (false)\n#endif\n  if (g.out_f32) epilogue_tile<true, FP8>(g, bidx, mrow, ncol, acc);\n")
rep("  auto issue = [&](int part, int buf, int koff) {                        // koff in BYTES along the row",
    "  auto issue = [&](int part, int buf, int koff) {\n#ifdef ABL_NODMA\n    if (koff != 0) return;\n#endif")
rep("#define RT_READ_A(ah)                                                                                             \\\n  _Pragma(\"unroll\") for (int i = 0; i < 4; ++i) {",
    "#ifdef ABL_NOLDS\n#define RT_LDS_ON (kt == 0)\n#else\n#define RT_LDS_ON true\n#endif\n"
    "#define RT_READ_A(ah)                                                                                             \\\n  if (RT_LDS_ON) _Pragma(\"unroll\") for (int i = 0; i < 4; ++i) {")
rep("#define RT_READ_B(bh)                                                                                             \\\n  _Pragma(\"unroll\") for (int j = 0; j < 2; ++j) {",
    "#define RT_READ_B(bh)                                                                                             \\\n  if (RT_LDS_ON) _Pragma(\"unroll\") for (int j = 0; j < 2; ++j) {")
rep("  {                                            // last K-tile: nothing left to issue, drain\n", "  {                                            // last K-tile\n    const int kt = nk - 1; (void)kt;\n")
rep("#define RT_BAR()                              \\\n  do {                                        \\\n    __builtin_amdgcn_sched_barrier(0);        \\\n    __builtin_amdgcn_s_barrier();             \\",
    "#ifdef ABL_NOBAR\n#define RT_HWBAR()\n#else\n#define RT_HWBAR() __builtin_amdgcn_s_barrier()\n#endif\n"
    "#define RT_BAR()                              \\\n  do {                                        \\\n    __builtin_amdgcn_sched_barrier(0);        \\\n    RT_HWBAR();                               \\", "head")
rep("    __builtin_amdgcn_s_setprio(1);                                                                                \\\n    if constexpr (FP8) {",
    "    __builtin_amdgcn_s_setprio(1);                                                                                \\\n    if (ABL_MFMA_ON) { if constexpr (FP8) {")
rep("                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[bh][j][kk], af[i][kk], acc[(ah)*4 + i][(bh)*2 + j], 0, 0, 0); \\\n    }                                                                                                             \\",
    "                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[bh][j][kk], af[i][kk], acc[(ah)*4 + i][(bh)*2 + j], 0, 0, 0); \\\n    } }                                                                                                           \\")
head = head.replace("namespace {\n", "namespace {\n#ifdef ABL_NOMFMA\n#define ABL_MFMA_ON (nk < 0)\n#else\n#define ABL_MFMA_ON true\n#endif\n", 1)
open(os.path.join(here, "gemm_abl.hip"), "w").write(head + tail)
variants = ["BASE", "NOEPI", "NODMA", "NOLDS", "NOBAR", "NOMFMA", "NOEPI -DABL_NODMA -DABL_NOLDS -DABL_NOBAR", "NODMA -DABL_NOLDS", "NOLDS -DABL_NOBAR -DABL_NODMA"]
for v in variants:
    name = "abl_" + v.replace(" ", "").replace("-DABL_", "_")
    cmd = (f"hipcc --offload-arch=gfx950 -O3 -std=c++17 -I{root}/include -I{here} -DABL_{v} {here}/gemm_abl.hip "
           f"{csrc}/version.hip {here}/main.cpp -o {here}/{name}")
    print(name, flush=True)
    subprocess.run(cmd, shell=True, check=True, stderr=subprocess.DEVNULL)
