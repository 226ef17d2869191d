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
