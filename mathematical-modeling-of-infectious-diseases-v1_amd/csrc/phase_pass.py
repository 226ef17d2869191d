#!/usr/bin/env python3
"""Part of the build of the tolerance-arithmetic kernels (csrc/Makefile, kernels_fma.o): a pass over the device assembly of
csrc/sepaihrd_kernels.hip that sets the 4-byte phase of runs of 8-byte encodings INSIDE the RK bodies ("The strict
build's two speeds").

A lone wave is fed ~1.9 bytes of instructions per cycle; a run of 8-byte encodings that starts 4 bytes off an 8-byte boundary
issues every 5.2 cycles instead of 4.2-4.3.  Pads between the stages (what the shipped build uses) cannot reach the flips
inside a stage.  This pass walks the big straight-line blocks of the chosen kernels and, in front of every run of >= MIN_RUN
8-byte encodings that would start off phase, re-encodes the nearest preceding 4-byte VOP1 / VOP2 / VOPC instruction of the same
block as its 8-byte VOP3 form (`_e32` -> `_e64`: the same instruction, no issue slot, four more bytes), or -- if there is none
close enough -- inserts one `s_nop 0` when the run is long enough to pay for it.

usage: phase_pass.py <in.s> <out.s> [--kernels substr,substr] [--min-run N] [--llvm DIR] [--mcpu ARCH]
Needs the assembler of the image: addresses come from assembling the input once, and the OUTPUT is assembled again at the end --
the phases the pass reports are read off that second object, not off its own bookkeeping (alignment padding inside a function
moves code in ways the running byte count does not see); an output that does not assemble is an error (exit status 1).

Never touched: the instructions from an `s_getpc_b64` to the last `@rel32` operand behind it.  Their literals encode distances
from the `s_getpc_b64` (`sym@rel32@lo+4`, `@hi+12`): a pad or a wider encoding BETWEEN them would silently corrupt the address."""
import argparse, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"   # --llvm overrides (csrc/Makefile passes its $(LLVM))
MCPU = "gfx950"                   # --mcpu overrides (csrc/Makefile passes its $(ARCH))
E64_OK = re.compile(r"^\s+(v_fmac_f64_e32|v_mov_b32_e32|v_mov_b64_e32|v_add_u32_e32|v_sub_u32_e32|v_subrev_u32_e32|v_mul_f32_e32|v_add_f32_e32|"
                    r"v_cvt_f32_f64_e32|v_cvt_f64_f32_e32|v_cvt_f64_i32_e32|v_rcp_f64_e32|v_log_f32_e32|v_exp_f32_e32|v_and_b32_e32|v_or_b32_e32|"
                    r"v_lshlrev_b32_e32|v_max_i32_e32|v_min_i32_e32)\b")


def assemble_sizes(path):
    """[(function, [sizes of its instructions in order])] from assembling `path`"""
    tmp = tempfile.mkdtemp(prefix="pp_")
    obj = os.path.join(tmp, "a.o")
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + MCPU, "-c", path, "-o", obj], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", obj], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-Fa-f]+):\s+((?:[0-9A-Fa-f]{8}\s*)+)", line)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), 4 * len(m.group(3).split()), m.group(1)))
    return out


def protected_spans(lines, idxs):
    """Positions (indices into idxs) that lie strictly behind an s_getpc_b64 and up to the last instruction within reach that
    names an @rel32 operand: nothing may be inserted in front of them or re-encoded among them."""
    out = set()
    for k, li in enumerate(idxs):
        if not lines[li].strip().startswith("s_getpc_b64"):
            continue
        last = k
        for m in range(k + 1, min(k + 8, len(idxs))):
            if "@rel32" in lines[idxs[m]]:
                last = m
        out.update(range(k + 1, last + 1))
    return out


def off_phase_runs(sizes, wanted, min_run, min_block):
    """(runs of >= min_run wide encodings inside the big straight-line blocks of the wanted kernels, those that start 4 bytes
    off an 8-byte boundary) -- read off an assembled object's real addresses"""
    n_runs = n_off = 0
    for name, ins in sizes.items():
        if not any(w in name for w in wanted):
            continue
        start = 0
        for k in range(len(ins) + 1):
            if not (k == len(ins) or ins[k][2].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc"))):
                continue
            blk = ins[start:min(k + 1, len(ins))]
            start = k + 1
            if len(blk) < min_block:
                continue
            j = 0
            while j < len(blk):
                if blk[j][1] != 8:
                    j += 1
                    continue
                r = j
                while r < len(blk) and blk[r][1] == 8:
                    r += 1
                if r - j >= min_run:
                    n_runs += 1
                    n_off += blk[j][0] % 8 == 4
                j = r
    return n_runs, n_off


def is_instruction(line):
    t = line.strip()
    return bool(t) and not t.startswith((";", ".", "//", "#")) and not re.match(r"^[\w.$]+:", t)


def main():
    global LLVM, MCPU
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("dst")
    ap.add_argument("--kernels", default="sepaihrd_eval_quad_kernel,sepaihrd_eval_kernelILi16E")
    ap.add_argument("--min-run", type=int, default=3)
    ap.add_argument("--min-block", type=int, default=200)
    ap.add_argument("--nop-run", type=int, default=7, help="insert s_nop 0 in front of an off-phase run at least this long when nothing can be re-encoded")
    ap.add_argument("--look-back", type=int, default=12)
    ap.add_argument("--body-residue", type=int, default=-1,
                    help="0, 8, .., 56: pad the ENTRY of every sepaihrd_eval_quad_kernel (s_nop pairs, executed once per wave) so that its "
                         "largest straight-line block -- the common RK body -- starts at this offset within a 64-byte line.  The 8-byte phase "
                         "is untouched (pads come in pairs).  Measured (round 4, tools/ab.sh): five instructions fewer in the consumer's "
                         "logarithm moved the body from offset 60 to 20 and cost the headline kernel 1 % at an unchanged phase report")
    ap.add_argument("--llvm", default=LLVM, help="directory of clang / llvm-objdump")
    ap.add_argument("--mcpu", default=MCPU)
    args = ap.parse_args()
    LLVM, MCPU = args.llvm, args.mcpu
    wanted = args.kernels.split(",")
    sizes = assemble_sizes(args.src)
    lines = open(args.src).read().split("\n")
    # functions of the listing: name -> list of line indices of its instructions, in order
    func, funcs = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\w+):\s*(;.*)?$", l)
        if m and m.group(1) in sizes:
            func = m.group(1); funcs[func] = []
            continue
        if l.startswith("\t.end_amdhsa_kernel") or l.strip().startswith(".Lfunc_end"):
            func = None
        if func is not None and is_instruction(l):
            funcs[func].append(i)
    n_re = n_nop = n_runs = n_fixed = 0
    edits = {}      # line index -> new text (re-encode) ; inserts: line index -> text inserted before
    inserts = {}
    for name, idxs in funcs.items():
        if not any(w in name for w in wanted):
            continue
        ins = sizes[name]
        if len(ins) != len(idxs):
            print(f"skip {name[:70]}: {len(ins)} disassembled vs {len(idxs)} listed instructions", file=sys.stderr)
            continue
        guarded = protected_spans(lines, idxs)
        # straight-line blocks
        start = 0
        shift = 0  # bytes added in this function so far: every edit moves everything behind it
        for k in range(len(ins) + 1):
            end_block = k == len(ins) or ins[k][2].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc"))
            if not end_block:
                continue
            blk = list(range(start, min(k + 1, len(ins))))
            start = k + 1
            if len(blk) < args.min_block:
                continue
            j = 0
            last_touch = -1
            while j < len(blk):
                a = blk[j]
                if ins[a][1] != 8:
                    j += 1
                    continue
                r = j
                while r < len(blk) and ins[blk[r]][1] == 8:
                    r += 1
                run_len = r - j
                if run_len >= args.min_run:
                    n_runs += 1
                    if (ins[a][0] + shift) % 8 == 4:
                        fixed = False
                        for b in range(j - 1, max(j - 1 - args.look_back, last_touch), -1):
                            li = idxs[blk[b]]
                            # a wider encoding INSIDE an s_getpc_b64 .. @rel32 span would move the literal's reference point;
                            # in front of the s_getpc_b64 it moves both alike
                            if blk[b] in guarded:
                                continue
                            if ins[blk[b]][1] == 4 and li not in edits and E64_OK.match(lines[li]) and "0x" not in lines[li] and "dpp" not in lines[li] and "sdwa" not in lines[li]:
                                # what lies between it and the run is 4-byte stuff and short runs: their phase flips too, accepted
                                edits[li] = lines[li].replace("_e32", "_e64", 1)
                                shift += 4; n_re += 1; fixed = True; last_touch = b
                                break
                        if not fixed and run_len >= args.nop_run and a not in guarded:
                            inserts[idxs[a]] = "\ts_nop 0"
                            shift += 4; n_nop += 1; fixed = True; last_touch = j
                        n_fixed += fixed
                j = r
    out = []
    for i, l in enumerate(lines):
        if i in inserts:
            out.append(inserts[i])
        out.append(edits.get(i, l))
    open(args.dst, "w").write("\n".join(out))
    # the output's phases from ITS OWN addresses (the byte count kept above does not see alignment padding inside a function)
    try:
        after = assemble_sizes(args.dst)
    except subprocess.CalledProcessError:
        print("phase_pass: the rewritten assembly does not assemble", file=sys.stderr)
        sys.exit(1)
    if args.body_residue >= 0:
        # where the largest block of each 16-lane kernel starts now; pad the function's entry by the even number of s_nop that
        # moves it to the wanted residue modulo 64
        out_lines = open(args.dst).read().split("\n")
        inserts_at = {}
        for name, ins in after.items():
            if "sepaihrd_eval_quad_kernel" not in name or not ins:
                continue
            blocks, start = [], 0
            for k in range(len(ins) + 1):
                if k == len(ins) or ins[k][2].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    blocks.append(ins[start:min(k + 1, len(ins))]); start = k + 1
            # RK bodies: big blocks of mostly FP64 / DPP instructions; the tolerance build has two copies (DESIGN_HISTORY.md:
            # one fed a single beta*kappa value -- the common path, the SHORTER one -- and one fed a value per stage)
            bodies = [b for b in blocks if len(b) >= args.min_block and sum(1 for _, _, m in b if "f64" in m or "dpp" in m) * 2 > len(b)]
            if not bodies:
                continue
            body = min(bodies, key=len)
            delta = (args.body_residue - body[0][0]) % 64
            delta -= delta % 8                      # pairs of 4-byte pads only: the 8-byte phase of everything stays
            if delta:
                inserts_at[name] = delta // 4
        if inserts_at:
            padded = []
            for l in out_lines:
                padded.append(l)
                m = re.match(r"^(\w+):\s*(;.*)?$", l)
                if m and m.group(1) in inserts_at:
                    padded.extend(["\ts_nop 0"] * inserts_at[m.group(1)])
            open(args.dst, "w").write("\n".join(padded))
            try:
                after = assemble_sizes(args.dst)
            except subprocess.CalledProcessError:
                print("phase_pass: the padded assembly does not assemble", file=sys.stderr)
                sys.exit(1)
            print(f"body residue {args.body_residue}: entry pads {sorted(set(inserts_at.values()))} s_nop in {len(inserts_at)} kernels")
    runs_before, off_before = off_phase_runs(sizes, wanted, args.min_run, args.min_block)
    runs_after, off_after = off_phase_runs(after, wanted, args.min_run, args.min_block)
    print(f"runs of >= {args.min_run} wide encodings: {n_runs}; off phase and fixed: {n_fixed} ({n_re} re-encoded, {n_nop} s_nop); "
          f"off-phase runs by the assembled addresses: {off_before} of {runs_before} before, {off_after} of {runs_after} after")
    if off_after > off_before:
        print("phase_pass: the rewritten assembly has MORE off-phase runs than its input", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
