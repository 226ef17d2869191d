#!/usr/bin/env python3
"""Search the strict build's phase-pad masks (csrc/sepaihrd_lane_split.inc: SEPAIHRD_STRICT_STAGE_PADS for Dopri5,
SEPAIHRD_STRICT_HEAD_PADS_CK / SEPAIHRD_STRICT_STAGE_PADS_CK for Cash-Karp) on the code the build ships: device assembly ->
csrc/phase_pass.py -> assembler -> addresses.  Cost model of DESIGN.md ("instruction fetch"): 0.9 cycles per 8-byte encoding that
starts 4 bytes off inside a run of >= 4, 4.3 cycles per s_nop in the body.  No GPU needed; the best masks are then confirmed on
the GPU with tools/ab.sh.      usage: phase_pad_search.py dopri5|cashkarp [max pads per mask, default 2]"""
import itertools, multiprocessing, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mathematical-modeling-of-infectious-diseases-v1_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_code_phase as cp  # noqa: E402


def evaluate(job):
    key, flags = job
    tmp = tempfile.mkdtemp(prefix="pads_")
    dev, phased, obj = (os.path.join(tmp, n) for n in ("dev.s", "phased.s", "phased.o"))
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
            "-ffp-contract=off", "-DSEPAIHRD_ARITH_FMA=0", *flags, "--cuda-device-only", "-S", os.path.join(CSRC, "sepaihrd_kernels.hip"), "-o", dev]
    subprocess.run(base, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run(["python3", os.path.join(CSRC, "phase_pass.py"), dev, phased], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([cp.LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", phased, "-o", obj], check=True)
    dis = subprocess.run([cp.LLVM + "/llvm-objdump", "-d", obj], check=True, capture_output=True, text=True).stdout
    best = None
    for name, ins in cp.kernels(dis).items():
        if key not in name:
            continue
        for blk in cp.body_blocks(ins, 250):
            rep = cp.phase_report(blk)
            rep["nops"] = sum(1 for _, _, m in blk if m == "s_nop")
            if best is None or rep["instructions"] > best["instructions"]:
                best = rep
    best["cost"] = 0.9 * best["wide_off_in_runs"] + 4.3 * best["nops"]
    return flags, best


def main():
    solver = sys.argv[1] if len(sys.argv) > 1 else "dopri5"
    most = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    masks = [sum(1 << b for b in bits) for k in range(most + 1) for bits in itertools.combinations(range(6), k)]
    if solver == "dopri5":
        key = "sepaihrd_eval_quad_kernelILi0ELi0ELb1ELb0EE"
        jobs = [(key, [f"-DSEPAIHRD_STRICT_STAGE_PADS={m}", f"-DSEPAIHRD_PHASE_NOPS_HEAD={h}"] if h else [f"-DSEPAIHRD_STRICT_STAGE_PADS={m}"])
                for m in masks for h in (0, 1)]
    else:
        key = "sepaihrd_eval_quad_kernelILi1ELi0ELb1ELb0EE"
        jobs = [(key, [f"-DSEPAIHRD_STRICT_HEAD_PADS_CK={h}", f"-DSEPAIHRD_STRICT_STAGE_PADS_CK={m}"]) for m in masks for h in (0, 1, 2)]
    with multiprocessing.Pool(8) as pool:
        res = pool.map(evaluate, jobs)
    res.sort(key=lambda r: r[1]["cost"])
    for flags, rep in res[:12]:
        print("%-80s cost %6.1f  off in runs %3d of %3d  nops %2d  instructions %d" % (" ".join(flags), rep["cost"], rep["wide_off_in_runs"], rep["wide_in_runs"], rep["nops"], rep["instructions"]))


if __name__ == "__main__":
    main()
