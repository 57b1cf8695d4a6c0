#!/usr/bin/env python3
"""Developer tool: register / scratch evidence of every shipped self-play kernel variant (profiles/rNN_resource_usage.txt).

Compiles the three translation units of the library to ISA with the Makefile's flags plus -Rpass-analysis=kernel-resource-usage
(hipcc cross-compiles gfx950 without a GPU) and prints, for every `selfplay_kernel*` instantiation: VGPRs, AGPRs, SGPRs,
occupancy, scratch bytes per lane, VGPR / SGPR spills, and the static count of scratch instructions in the whole kernel, inside
its MFMA range and inside loops (a scratch op between a backward branch's target and the branch).
    usage: tools/isa/resource_usage.py > profiles/r03_resource_usage.txt"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "synthesis_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-S", "--cuda-device-only",
         "-Rpass-analysis=kernel-resource-usage"]


# the device translation units of the DEFAULT library (synthesis_amd/csrc/Makefile without DEBUG_SHAPES); `--debug-shapes` adds the
# forced-only kernels (two trees per lane, trees unbound from the lanes)
TUS = ["engine", "engine_conv", "engine_lanes_fast", "engine_lanes_gen", "engine_lanes_ref", "engine_lanes_f16", "engine_lanes_f16_gen",
       "engine_free"]
if "--debug-shapes" in sys.argv:
    TUS += ["engine_lanes2", "engine_pool", "engine_pool_f16"]
    FLAGS.append("-DSYN_DEBUG_SHAPES")


def compile_tu(name, tmp):
    out = os.path.join(tmp, name + ".s")
    p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, os.path.join(CSRC, name + ".hip")], capture_output=True, text=True)
    return out, p.stderr


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.split("\n")))


def loop_scratch(body):
    """body: list of ISA lines of one kernel. Returns (scratch ops, of them inside the MFMA range, of them inside a loop)."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = i
    in_loop = [False] * len(body)
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\w*\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", l)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t < i:
                for k in range(t, i + 1):
                    in_loop[k] = True
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    sc = [i for i, l in enumerate(body) if re.match(r"\s+scratch_", l)]
    return len(sc), sum(1 for i in sc if mf and mf[0] <= i <= mf[-1]), sum(1 for i in sc if in_loop[i])


def main():
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(3) as ex:
            res = list(ex.map(lambda n: compile_tu(n, tmp), TUS))
        rows = []
        for path, err in res:
            if not os.path.exists(path):
                sys.stderr.write(err[-2000:])
                raise SystemExit("compile failed: " + path)
            lines = open(path).read().split("\n")
            cur = None
            info = {}
            for l in err.split("\n"):
                m = re.search(r"Function Name: (\S+)", l)
                if m:
                    cur = m.group(1)
                    info[cur] = {}
                    continue
                m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", l)
                if m and cur:
                    info[cur][m.group(1).strip()] = m.group(2)
            for fn, d in info.items():
                if "selfplay_kernel" not in fn:
                    continue
                start = [i for i, l in enumerate(lines) if l.startswith(fn + ":")]
                if not start:
                    continue
                end = [i for i in range(start[0], len(lines)) if lines[i].startswith(".Lfunc_end")][0]
                d["_scratch"] = loop_scratch(lines[start[0]:end])
                rows.append((fn, d))
        names = demangle([r[0] for r in rows])
        print("# kernel-resource-usage of every shipped self-play kernel instantiation (gfx950; hipcc -Rpass-analysis=kernel-resource-usage,")
        print("# the Makefile's flags). scratch ops = static scratch_load/store instructions: whole kernel / inside the MFMA range / inside loops.")
        print("# template arguments: selfplay_kernel_lanes<MODE (0 self-play, 1 search), COUNT, FAST, waves, PROF, POLICY (0 Connect4Net, 1 rollout, 2 conv)>,")
        print("#                     (POLICY 3 = Connect4Net in the f16x2 arithmetic; FAST 0 runtime-switched, 1 parity family, 2 the reference's self-play configuration)")
        print("#                     selfplay_kernel<MODE, COUNT, WPS, FAST, PROF>, selfplay_kernel_quads<MODE, COUNT, FAST, NQ, PROF>, selfplay_kernel_free<MODE, COUNT, FAST, PROF>,")
        print("#                     selfplay_kernel_lanes2<MODE, COUNT, FAST, waves, POLICY, TILE>, selfplay_kernel_pool<MODE, COUNT, FAST, waves, POLICY>")
        print("%-96s %5s %5s %5s %4s %8s %7s %7s  %s" % ("kernel", "VGPR", "AGPR", "SGPR", "occ", "scratchB", "vspill", "sspill", "scratch ops all/mfma/loops"))
        for fn, d in sorted(rows, key=lambda r: names[r[0]]):
            n = names[fn].replace("syn::", "").replace("(syn::EngineParams)", "").replace("void ", "")
            print("%-96s %5s %5s %5s %4s %8s %7s %7s  %d/%d/%d" % (n, d.get("VGPRs"), d.get("AGPRs"), d.get("TotalSGPRs"), d.get("Occupancy [waves/SIMD]"),
                                                            d.get("ScratchSize [bytes/lane]"), d.get("VGPRs Spill"), d.get("SGPRs Spill"), *d["_scratch"]))


if __name__ == "__main__":
    main()
