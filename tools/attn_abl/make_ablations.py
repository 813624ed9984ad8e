"""Build the attention ablation executables used for DESIGN.md's 'attention anatomy'.

Copies csrc/attention.hip, wraps parts of attention_fwd_pipe_kernel in ABL_* macros (timing only — results are wrong by
construction), adds s_memtime / s_memrealtime stamps around the steady loop, and compiles one executable per variant with
main.cpp. Run from the repo root:  python tools/attn_abl/make_ablations.py ; then on the GPU box:  tools/attn_abl/abl_BASE 4608
"""
import os, subprocess, shutil
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
csrc = os.path.join(root, "arabic-text-image-generation-reptext_amd", "csrc")
s = open(os.path.join(csrc, "attention.hip")).read()
shutil.copy(os.path.join(csrc, "rt_common.h"), here)
i = s.index("__global__ __launch_bounds__(ATT_THREADS, 2) void attention_fwd_pipe_kernel")
head, tail = s[:i], s[i:]


def rep(old, new, where="tail", count=1):
    global head, tail
    src = tail if where == "tail" else head
    assert old in src, old[:60]
    src = src.replace(old, new, count)
    if where == "tail":
        tail = src
    else:
        head = src


rep("    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[e >> 4][e & 15], scale_log2, -m_run));",
    "#ifdef ABL_NOEXP\n    const float p = __builtin_fmaf(cur[e >> 4][e & 15], scale_log2, -m_run);\n#else\n"
    "    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[e >> 4][e & 15], scale_log2, -m_run));\n#endif")
rep("  auto soft = [&](const f32x16 (&cur)[2], int e) {", "  auto soft = [&](const f32x16 (&cur)[2], int e) {\n#ifdef ABL_NOSOFT\n    if (e != 0) return;\n#endif")
rep("    if constexpr (!TAIL) {\n      stage_one(rsrcK, PAR * K_SLOT_B, t + 2, false);",
    "#ifdef ABL_NODMA\n    if (false) {\n#else\n    if constexpr (!TAIL) {\n#endif\n      stage_one(rsrcK, PAR * K_SLOT_B, t + 2, false);")
rep("    __syncthreads();     // K(t+1), V(t) landed", "#ifndef ABL_NOBAR\n    __syncthreads();\n#endif\n    //")
rep("        } else if (has_next) {\n          const int g = n - 8;",
    "        } else if (has_next) {\n#ifdef ABL_NOMAX\n          if (n != 8) { __builtin_amdgcn_sched_barrier(0); continue; }\n#endif\n          const int g = n - 8;")
rep("  auto k_read = [&](int i, int kb) -> bf16x8 {", "  auto k_read = [&](int i, int kb) -> bf16x8 {\n#ifdef ABL_NOLDS\n    return qf[i & 7];\n#endif")
rep("  auto v_read = [&](int n, int vb) -> bf16x8 {", "  auto v_read = [&](int n, int vb) -> bf16x8 {\n#ifdef ABL_NOLDS\n    return qf[n & 7];\n#endif")
rep("constexpr int K_SLOT_B = TILE_B;", "__device__ unsigned long long g_stamps[4 * 8192];\nconstexpr int K_SLOT_B = TILE_B;", "head")
rep("  int t = 0;\n  for (; t + 3 < nfull; t += 2) {",
    "  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();\n  int t = 0;\n  for (; t + 3 < nfull; t += 2) {")
rep("  // ---- epilogue: O[q][d] = Oᵀ / l ; lane holds q = l31, d = 32dt + (r&3) + 8(r>>2) + 4hh\n  const float l_tot",
    "  {\n    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();\n"
    "    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;\n"
    "    if (tid == 0 && wg < 8192) { g_stamps[4 * wg] = c1 - c0; g_stamps[4 * wg + 1] = r1 - r0; g_stamps[4 * wg + 2] = r0; g_stamps[4 * wg + 3] = r1; }\n  }\n"
    "  // ---- epilogue\n  const float l_tot")
rep('extern "C" int rt_attention_fwd',
    'extern "C" int rt_abl_read_stamps(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), (size_t)n * 8); }\n\nextern "C" int rt_attention_fwd')
# the harness always runs the pipelined kernel
rep("variant = (e && e[0] == 'p') ? 1 : 0;", "variant = 1; (void)e;")
open(os.path.join(here, "attn_abl.hip"), "w").write(head + tail)
variants = ["BASE", "NOEXP", "NOSOFT", "NOMAX", "NODMA", "NOBAR", "NOLDS", "NOSOFT -DABL_NOMAX -DABL_NODMA -DABL_NOBAR",
            "NOLDS -DABL_NOSOFT -DABL_NOMAX -DABL_NODMA -DABL_NOBAR"]
for v in variants:
    name = "abl_" + v.replace(" ", "").replace("-DABL_", "_")
    cmd = f"hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -I{root}/include -I{here} -DABL_{v} {here}/attn_abl.hip {here}/main.cpp -o {here}/{name}"
    print(cmd, flush=True)
    subprocess.run(cmd, shell=True, check=True, stderr=subprocess.DEVNULL)
