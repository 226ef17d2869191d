"""Build-time checks that need no GPU: the DPP read hazard on the kernels' listings (inline-asm DPP instructions are
opaque to the compiler's hazard pass)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_hazard_checker_sees_a_violation(tmp_path):
    chk = _checker()
    listing = tmp_path / "k.s"
    listing.write_text(
        "kernel_a:\n"
        "\tv_mul_f64 v[10:11], v[2:3], v[4:5]\n"
        "\tv_add_f64 v[20:21], v[6:7], v[8:9]\n"
        "\tv_fmac_f64_dpp v[30:31], v[10:11], v[12:13] row_newbcast:2 row_mask:0xf bank_mask:15\n"      # 1 wait state: hazard
        "\tv_mul_f64 v[40:41], v[2:3], v[4:5]\n"
        "\ts_nop 1\n"
        "\tv_fmac_f64_dpp v[30:31], v[40:41], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:15\n"      # 2 wait states: fine
        "\tv_mov_b32_e32 v50, v1\n"
        "\tv_mov_b32_dpp v51, v50 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"        # 0 wait states: hazard
        "\tv_mul_f64 v[60:61], v[2:3], v[4:5]\n"
        "\tv_fmac_f64_dpp v[30:31], v[12:13], v[60:61] row_newbcast:3 row_mask:0xf bank_mask:15\n"      # v60 is not the DPP operand
        "\ts_endpgm\n")
    violations, n = chk.check(str(listing))
    assert n == 4 and len(violations) == 2
    assert "v[10, 11]" in violations[0] and "1 wait state" in violations[0]
    assert "v[50]" in violations[1] and "0 wait state" in violations[1]


def test_no_dpp_read_hazard_in_the_shipped_kernels():
    """tools/check_dpp_hazards.py on fresh `hipcc -S` listings of the translation units with inline-asm DPP instructions
    (csrc/sepaihrd_kernels.hip in the tolerance build, csrc/sepaihrd_kernels_f32.hip): every DPP read is at least two
    wait states behind the VALU write of its source."""
    chk = _checker()
    for listing in chk.build_listings():
        violations, n = chk.check(listing)
        assert n > 100 and violations == [], listing


def test_strict_headline_kernel_sits_in_its_fast_code_placement():
    """A lone wave is fed ~1.9 bytes of instructions per cycle: a run of 8-byte encodings that starts 4 bytes off an 8-byte
    boundary issues every 5.2 cycles instead of every 4.2-4.3 (tools/ubench/phase.hip).  The strict build's RK body is almost
    purely 8-byte VOP3 encodings, and every 4-byte encoding in front of it flips its phase: the build has a fast and a slow
    placement 6-8 % apart (0.878 against 0.935 ms per 4096-chain evaluation), one s_nop at the loop head switches between them
    (-DSEPAIHRD_PHASE_NOPS_HEAD=1).  tools/check_code_phase.py reads the placement off the disassembly: in the fast one 27 %
    of the body's 8-byte encodings inside runs start 4 bytes off, in the slow one 73 %.  If this fails after a change to the
    loop, add (or remove) one pad at the loop head of the strict build and look again."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_code_phase", os.path.join(ROOT, "tools", "check_code_phase.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    res = chk.analyse("strict")
    # With the pads chosen on this figure (SEPAIHRD_STRICT_STAGE_PADS / SEPAIHRD_STRICT_HEAD_PADS_CK in sepaihrd_lane_split.inc:
    # a search over the pad masks with this script, the best ones confirmed on the GPU, 0.879 -> 0.855 ms) 2-3 % are off phase.
    for key in ("ILi0ELi0ELb1ELb0EE", "ILi1ELi0ELb1ELb0EE"):   # Dopri5 / Cash-Karp, strict, fused, no trajectory
        body = [v for k, v in res.items() if key in k]
        assert len(body) == 1 and body[0]["wide"] > 400, res.keys()
        assert body[0]["share_off_in_runs"] < 0.15, (key, body[0])


def test_assembly_phase_pass_aligns_the_runs_of_the_tolerance_bodies(tmp_path):
    """csrc/phase_pass.py (the build's pass over the device assembly, csrc/Makefile): on a fresh `hipcc -S` listing of the
    tolerance build its output assembles, holds the same instructions but for `_e32` -> `_e64` re-encodings and `s_nop 0` pads,
    and leaves at most 30 of the 16-lane common body's ~190 eight-byte encodings off phase (80 without it; none inside a run
    of three or more)."""
    import importlib.util, re, subprocess
    spec = importlib.util.spec_from_file_location("check_code_phase", os.path.join(ROOT, "tools", "check_code_phase.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    csrc = os.path.join(ROOT, "mathematical-modeling-of-infectious-diseases-v1_amd", "csrc")
    src, dst, obj = str(tmp_path / "dev.s"), str(tmp_path / "phased.s"), str(tmp_path / "phased.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    "-ffp-contract=fast", "-DSEPAIHRD_ARITH_FMA=1", "--cuda-device-only", "-S", os.path.join(csrc, "sepaihrd_kernels.hip"), "-o", src],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run(["python3", os.path.join(csrc, "phase_pass.py"), src, dst], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([chk.LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", dst, "-o", obj], check=True)
    # the same instruction stream: drop the pads, undo the re-encodings, compare line by line
    a = [l for l in open(src).read().split("\n")]
    b = [l.replace("_e64", "_e32") for l in open(dst).read().split("\n") if l != "\ts_nop 0"]
    a_n = [l.replace("_e64", "_e32") for l in a if l != "\ts_nop 0"]
    assert a_n == b
    dis = subprocess.run([chk.LLVM + "/llvm-objdump", "-d", obj], check=True, capture_output=True, text=True).stdout
    bodies = [chk.phase_report(blk) for name, ins in chk.kernels(dis).items() if "sepaihrd_eval_quad_kernelILi0ELi1ELb1ELb0EE" in name
              for blk in chk.body_blocks(ins, 250)]
    common = max(bodies, key=lambda r: r["instructions"])
    assert common["wide"] > 150 and common["wide_off"] <= 30 and common["wide_off_in_runs"] == 0, common
