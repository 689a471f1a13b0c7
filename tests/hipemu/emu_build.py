"""tests/hipemu/emu_build.py -- build the host stand-in of selected kernel files (test infrastructure only).

Copies knowhere_amd/csrc/{pq_filter,mfma_scan,topk,range}.hip into tests/hipemu/_build/ with the few hardware-specific lines
rewritten (LDS-offset addressing, inline ISA, dynamic LDS declarations), and compiles them with the host clang++ against
tests/hipemu/hip/hip_runtime.h into _build/libpqf_emu.so.  Every rewrite asserts how often its pattern occurs, so a
change of the kernel source that the emulation does not cover fails here instead of passing silently."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "knowhere_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

DYN_SMEM = "extern __shared__ __align__(16) unsigned char smem[];"
DYN_SMEM_EMU = "unsigned char* smem = hipemu::tl.g->smem;"


def _sub(s, pattern, repl, count, flags=0, what=""):
    out, n = re.subn(pattern, repl, s, flags=flags)
    assert n == count, f"emu patch '{what or pattern[:40]}': expected {count} match(es), found {n}"
    return out


# occupancy hints of the device compiler
WAVES_PER_EU = re.compile(r"__attribute__\(\(amdgpu_waves_per_eu\(\d+, \d+\)\)\) ")


def patch_pq_filter(s):
    s = _sub(s, re.escape(DYN_SMEM), DYN_SMEM_EMU, 2, what="dynamic LDS")
    # the LDS-offset-0 assertion of the token addressing
    s = _sub(s, r"    if \(\(uint32_t\)\(size_t\)\(\(__attribute__\(\(address_space\(3\)\)\) unsigned char\*\)smem\) != 0u\) \{\n"
                r"        __builtin_trap\(\);[^\n]*\n    \}\n", "", 2, what="LDS offset check")
    # token -> LDS byte address: the SDWA shift, restated
    s = _sub(s, r'    asm\("v_lshlrev_b32_sdwa %0, %2, %1 [^"]*src1_sel:WORD_0"\n\s*: "=v"\(a\)\n\s*: "v"\(w\), "s"\(one\)\);\n',
             "    a = (w & 0xffffu) << one;\n", 1, what="sdwa lo")
    s = _sub(s, r'    asm\("v_lshlrev_b32_sdwa %0, %2, %1 [^"]*src1_sel:WORD_1"\n\s*: "=v"\(a\)\n\s*: "v"\(w\), "s"\(one\)\);\n',
             "    a = (w >> 16) << one;\n", 1, what="sdwa hi")
    # LDS reads by byte offset
    s = _sub(s, r"        typedef __attribute__\(\(address_space\(3\)\)\) const pf_h8 lds_h8;\n"
                r"        auto lut_read = \[&\]\(uint32_t addr\) -> pf_h8 \{ return \*reinterpret_cast<lds_h8\*>\(addr\); \};\n",
             "        auto lut_read = [&](uint32_t addr) -> pf_h8 { return *reinterpret_cast<const pf_h8*>(smem + addr); };\n",
             1, what="lut_read")
    s = _sub(s, r"        typedef __attribute__\(\(address_space\(3\)\)\) const pf_i4 lds_i4;\n"
                r"        auto lut_read = \[&\]\(uint32_t addr\) -> pf_i4 \{ return \*reinterpret_cast<lds_i4\*>\(addr\); \};\n",
             "        auto lut_read = [&](uint32_t addr) -> pf_i4 { return *reinterpret_cast<const pf_i4*>(smem + addr); };\n",
             1, what="lut_read i8")
    s = _sub(s, r'    asm volatile\("s_getreg_b32 %0, hwreg\(HW_REG_XCC_ID\)" : "=s"\(xcc\)\);\n',
             "    xcc = (uint32_t)blockIdx.x;\n", 2, what="xcc id")
    s = _sub(s, r'        asm volatile\("" : "\+v"\(lane_i\)\);[^\n]*\n', "", 2, what="lane launder")
    s = _sub(s, r'        asm volatile\("" : "\+v"\(tau_g\)[^\n]*\n', "", 1, what="load-order fence")
    s = _sub(s, r'        asm volatile\("" : "\+v"\(pq_qv\)\);\n', "", 1, what="query index launder")
    return WAVES_PER_EU.sub("", s)


def patch_dyn_smem_only(s):
    n = s.count(DYN_SMEM)
    assert n >= 1
    s = s.replace(DYN_SMEM, DYN_SMEM_EMU)
    return s


patch_mfma_scan = patch_dyn_smem_only
patch_topk = patch_dyn_smem_only
patch_range = patch_dyn_smem_only


def strip_inline_isa(s):
    """every `asm(...)` / `asm volatile(...)` statement -> a call that aborts: files that hold inline ISA compile, their
    asm-free kernels (layout builders, table kernels) run, the ISA kernels themselves cannot be emulated"""
    out, i = [], 0
    pat = re.compile(r"\basm\s*(volatile\s*)?\(")
    while True:
        m = pat.search(s, i)
        if not m:
            out.append(s[i:])
            break
        out.append(s[i:m.start()])
        j, depth = m.end(), 1
        while depth:
            c = s[j]
            if c == '"':
                j += 1
                while s[j] != '"':
                    j += 2 if s[j] == "\\" else 1
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            j += 1
        out.append("hipemu_unreachable()")
        i = j
    return "".join(out)


def patch_generic(s):
    """dynamic LDS declarations of any element type, LDS address-space casts, inline ISA"""
    s = re.sub(r"extern __shared__ (?:__align__\(\d+\) )?([A-Za-z_][\w ]*?) (\w+)\[\];",
               lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(hipemu::tl.g->smem);", s)
    # the XCD id of a persistent workgroup (placement only): the workgroup index stands in
    s = re.sub(r'asm volatile\("s_getreg_b32 %0, hwreg\(HW_REG_XCC_ID\)" : "=s"\((\w+)\)\);', r"\1 = (uint32_t)blockIdx.x;", s)
    s = re.sub(r'asm volatile\("" : "\+v"\(([\w\[\]\.]+)\)\);', "", s)  # (compiler fences on a lane value)
    s = strip_inline_isa(s)
    # "the LUT sits at LDS offset 0" assertions of the token-addressed kernels (host pointers are never 0)
    s = re.sub(r"    if \(\(uint32_t\)\(size_t\)\(\(__attribute__\(\(address_space\(3\)\)\) unsigned char\*\)smem\) != 0u\) \{\n"
               r"        __builtin_trap\(\);[^\n]*\n    \}\n", "", s)
    return WAVES_PER_EU.sub("", s).replace("__attribute__((address_space(3)))", "")


# the whole library (C ABI + orchestration + kernels) for API-level emulation; prims.hip / build.hip stay out (their
# entry points resolve to aborting stubs generated from the link's undefined symbols)
API_FILES = ["knhip_api.hip", "knhip_api_build.hip", "knhip_api_prims.hip", "knhip_api_rows.hip", "knhip_api_range.hip", "flat_scan.hip", "sq_scan.hip", "topk.hip", "worktable.hip", "coarse_gemm.hip", "refine.hip",
             "range.hip", "mfma_scan.hip", "mfma_scan_bf16.hip", "pq_filter.hip", "pq_decode.hip", "pq_scan.hip", "pq_scan_v2.hip", "pq_scan_q4.hip", "pq_scan_any.hip"]


def build_api(force=False):
    """_build/libknhip_emu.so: load it through KNHIP_LIB to drive the emulated kernels with the product's own ctypes
    harness (knowhere_amd/index.py) and orchestration (knhip_api.hip)"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libknhip_emu.so")
    srcs = [os.path.join(CSRC, f) for f in API_FILES + ["common.h", "kernels.h", "ms_common.h", "knhip_internal.h", "knhip_env.h"]]
    srcs += [os.path.join(HERE, f) for f in ("emu_runtime.cpp", "emu_build.py", "hip/hip_runtime.h")]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(p) for p in srcs):
        return so
    flags = ["-std=c++20", "-O1", "-g", "-fPIC", "-pthread", "-ffp-contract=off", "-Xclang", "-ffloat16-excess-precision=none",
             "-Wno-unused-value", "-Wno-unknown-pragmas", "-Wno-pass-failed", "-I", HERE, "-I", CSRC,
             "-I", os.path.join(ROOT, "include")]
    objs = []
    jobs = []
    for name in API_FILES + ["emu_runtime.cpp"]:
        if name.endswith(".hip"):
            with open(os.path.join(CSRC, name)) as f:
                src = f.read()
            src = patch_pq_filter(src) if name == "pq_filter.hip" else patch_generic(src)
            cpp = os.path.join(BUILD, "api_" + name.replace(".hip", "_emu.cpp"))
            with open(cpp, "w") as f:
                f.write(src)
        else:
            cpp = os.path.join(HERE, name)
        obj = os.path.join(BUILD, "api_" + os.path.basename(cpp) + ".o")
        objs.append(obj)
        extra = []  # (pq_scan.hip without KN_PQ_M: the common part -- table kernels, skew layout, dispatch)
        jobs.append((name, subprocess.Popen([CLANG] + flags + extra + ["-c", cpp, "-o", obj], stdout=subprocess.PIPE,
                                            stderr=subprocess.PIPE, text=True)))
    for name, j in jobs:
        out, err = j.communicate()
        if j.returncode != 0:
            raise RuntimeError(f"emulation build of {name} failed:\n" + err[-6000:])
    # entry points of the files left out (and of the other pq_scan.hip widths): aborting stubs
    first = subprocess.run([CLANG, "-shared", "-pthread", "-o", so] + objs, capture_output=True, text=True)
    if first.returncode != 0:
        raise RuntimeError("emulation link failed:\n" + first.stderr[-6000:])
    und = subprocess.run(["nm", "-u", "--defined-only", so], capture_output=True, text=True)
    und = subprocess.run(["nm", "-u", so], capture_output=True, text=True).stdout.split("\n")
    names = sorted({l.split()[-1] for l in und if l.strip() and ("knhip" in l.split()[-1])})
    stub = os.path.join(BUILD, "api_stubs.S")
    with open(stub, "w") as f:
        f.write(".text\n")
        for n in names:
            f.write(f".globl {n}\n.type {n}, @function\n{n}:\n    call abort@PLT\n")
        f.write('.section .note.GNU-stack,"",@progbits\n')
    r = subprocess.run([CLANG, "-shared", "-pthread", "-o", so] + objs + [stub], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulation link failed:\n" + r.stderr[-6000:])
    return so


def build(force=False):
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libpqf_emu.so")
    srcs = [os.path.join(CSRC, f) for f in ("pq_filter.hip", "mfma_scan.hip", "topk.hip", "range.hip", "common.h", "kernels.h",
                                            "ms_common.h")]
    srcs += [os.path.join(HERE, f) for f in ("emu_runtime.cpp", "harness.cpp", "emu_build.py", "hip/hip_runtime.h")]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(p) for p in srcs):
        return so
    for name, fn in (("pq_filter.hip", patch_pq_filter), ("mfma_scan.hip", patch_mfma_scan), ("topk.hip", patch_topk),
                     ("range.hip", patch_range)):
        with open(os.path.join(CSRC, name)) as f:
            src = fn(f.read())
        with open(os.path.join(BUILD, name.replace(".hip", "_emu.cpp")), "w") as f:
            f.write(src)
    cmd = [CLANG, "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
           "-Xclang", "-ffloat16-excess-precision=none", "-Wno-unused-value", "-Wno-unknown-pragmas", "-Wno-pass-failed",
           "-I", HERE, "-I", CSRC, "-I", os.path.join(ROOT, "include"),
           os.path.join(BUILD, "pq_filter_emu.cpp"), os.path.join(BUILD, "mfma_scan_emu.cpp"),
           os.path.join(BUILD, "topk_emu.cpp"), os.path.join(BUILD, "range_emu.cpp"),
           os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "harness.cpp"), "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulation build failed:\n" + r.stdout[-3000:] + r.stderr[-6000:])
    return so


if __name__ == "__main__":
    print(build(force=True))
    print(build_api(force=True))
