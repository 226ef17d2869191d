#!/usr/bin/env python3
"""Exhaustive search over the compartment -> (lane of the quad, slot) assignments of the 16-lanes-per-chain form
(csrc/sepaihrd_lane_split.inc), scored by the instructions ONE wave issues per right-hand-side call and per RK attempt.

Why the per-wave count and not the "padded FP64 work" of the PMC summary: configs[1] puts exactly one integrator wave on every
SIMD, and a lone wave's attempt costs (instructions it issues) x (4.1 - 5 cycles) whatever the lanes do with them.  What the
other lanes of a quad execute with zero coefficients is padding in the counter (14.5 G FP64 flops issued against 8.46 G
algorithmic = 1.71 x) but costs no time: a layout is better only if it SHORTENS the stream every lane executes.

The stream of one RHS call, identical in all 16 lanes of a chain (per-lane coefficient registers, zero where a term does not
apply; tolerance build, rhs_quad):
    2 instructions per quad rotation (r, slot) that any term needs (v_mov_b32_dpp x 2: the FP64 ALU takes quad_perm on no operand)
    2               infectious pressure  P + A + theta I  in the lane that has I   (one add, one fma)
    7               lambda = sum_j m_ij pressure_j (4 v_fmac_f64_dpp row_newbcast), * beta kappa, max(0, .), W = lambda S
    T_s             one multiply-add per DISTINCT source (rotation, slot) the four compartments of output slot s draw on
The stage sums cost 26 multiply-adds per slot that holds a compartment some derivative reads, 11 per slot of pure quadratures
(R, D, CumH, CumICU are never read back: their intermediate stage values are dead, only new value and error estimate remain),
and the error norm 6 instructions per slot.  Six RHS calls per Dopri5 attempt.

Constraints of the form: 4 lanes x 3 slots = 12 places for 11 compartments; E shares a lane with S (both take W = lambda S; a
rotated W would cost one more rotation -- such layouts are scored with that cost, not excluded).  The strict build embeds the
reference's expression tree of every compartment in one template per slot; that template is written per layout (for the shipped
one: 10 + 7 + 11 instructions against 11 + 11 + 9 for the rounds 1-3 layout, plus the same three rotations instead of four), so
the search scores the tolerance build's stream and the strict build follows the layout it finds.  Among the winners the shipped
layout is the one in which every lane observes at most ONE of D, CumH, CumICU (the observer reads one value per lane) and all
three sit in the same slot.

    python3 tools/layout_search.py            # prints the table DESIGN.md quotes; ~1 minute
"""
import itertools
import json
import sys
from collections import Counter

COMPS = ["S", "E", "P", "A", "I", "H", "ICU", "R", "D", "CumH", "CumICU"]
DYNAMIC = ["S", "E", "P", "A", "I", "H", "ICU"]          # read by some derivative
QUADRATURES = ["R", "D", "CumH", "CumICU"]                # read by none
# linear terms of every derivative (tolerance build; W = lambda S is not a linear term): compartment -> sources
TERMS = {"S": [], "E": ["E"], "P": ["E", "P"], "A": ["P", "A"], "I": ["P", "I"], "H": ["I", "H"], "ICU": ["H", "ICU"],
         "R": ["I", "H", "A", "ICU"], "D": ["H", "I", "ICU"], "CumH": ["I"], "CumICU": ["H"]}
PRESSURE = ["P", "A", "I"]                                 # formed in the lane that holds I
# shape of the reference's expression tree (strict build: one template per slot must embed all four of a slot's trees)
SHAPE = {"S": "in-out", "P": "in-out", "I": "in-out", "ICU": "in-out", "E": "in+W-out", "A": "in+W-out", "H": "in+W-out",
         "R": "sum", "D": "sum", "CumH": "sum", "CumICU": "sum"}
ROUND3 = {"S": (0, 0), "E": (0, 1), "D": (0, 2), "P": (1, 0), "A": (1, 1), "I": (2, 0), "H": (2, 1), "CumH": (2, 2),
          "ICU": (3, 0), "CumICU": (3, 1), "R": (3, 2)}   # the layout of rounds 1-3
SHIPPED = {"S": (0, 0), "E": (0, 1), "CumICU": (0, 2), "P": (1, 1), "CumH": (1, 2), "A": (2, 0), "I": (2, 1), "R": (2, 2),
           "ICU": (3, 0), "H": (3, 1), "D": (3, 2)}       # csrc/sepaihrd_lane_split.inc since round 4


def score(place):
    """place: compartment -> (lane, slot).  Returns (rhs instructions, attempt instructions, detail)."""
    rotations = set()
    per_slot = [set(), set(), set()]
    for comp, (lane, slot) in place.items():
        for src in TERMS[comp]:
            sl, ss = place[src]
            r = (sl - lane) % 4
            per_slot[slot].add((r, ss))
            if r:
                rotations.add((r, ss))
    il = place["I"][0]
    for src in PRESSURE:
        sl, ss = place[src]
        r = (sl - il) % 4
        if r:
            rotations.add((r, ss))
    w_rot = 0 if place["E"][0] == place["S"][0] else 1   # W = lambda S has to reach E's lane
    rhs = 2 * (len(rotations) + w_rot) + 2 + 7 + sum(len(s) for s in per_slot)
    slots_dynamic = {place[c][1] for c in DYNAMIC}
    slots_used = {s for _, s in place.values()}
    stage = sum(26 if s in slots_dynamic else 11 for s in slots_used)
    norm = 6 * len(slots_used)
    observers = Counter(place[c][0] for c in ("D", "CumH", "CumICU"))
    return rhs, 6 * rhs + stage + norm, {"rotations": len(rotations) + w_rot, "terms": [len(s) for s in per_slot], "stage_sums": stage,
                                         "one_observable_per_lane": max(observers.values()) == 1,
                                         "observables_in_one_slot": len({place[c][1] for c in ("D", "CumH", "CumICU")}) == 1}


def bits(place_of_sources, comp, lane, slot):
    """(rotation mask [bit 3 ss + r - 1], term mask of `slot` [bit 4 ss + r]) that `comp` at (lane, slot) contributes"""
    rot = term = 0
    for src in TERMS[comp]:
        sl, ss = place_of_sources[src]
        r = (sl - lane) % 4
        term |= 1 << (4 * ss + r)
        if r:
            rot |= 1 << (3 * ss + r - 1)
    return rot, term


POP = [bin(i).count("1") for i in range(1 << 12)]


def main():
    cur, shipped = score(ROUND3), score(SHIPPED)
    hist = Counter()
    best = []
    n = 0
    # S pinned to lane 0, slot 0 (the rotation of the quad and the naming of the slots do not matter).  The 6 other dynamic
    # compartments take 6 of the 11 other places (332 640 ways); for each, the four quadratures try every placement into the
    # places left (120 ways): 39.9 M assignments, each scored by the masks of rotations and per-slot sources it needs.
    places = [(l, s) for s in range(3) for l in range(4)]
    free = [p for p in places if p != (0, 0)]
    dyn_rest = [c for c in DYNAMIC if c != "S"]
    for chosen in itertools.permutations(free, len(dyn_rest)):
        base = dict(zip(dyn_rest, chosen), S=(0, 0))
        rot0, term0 = 0, [0, 0, 0]
        for c in DYNAMIC:
            l, sl = base[c]
            r, t = bits(base, c, l, sl)
            rot0 |= r
            term0[sl] |= t
        il = base["I"][0]
        for src in PRESSURE:
            sl, ss = base[src]
            r = (sl - il) % 4
            if r:
                rot0 |= 1 << (3 * ss + r - 1)
        w_rot = 0 if base["E"][0] == 0 else 1
        dyn_slots = {sl for _, sl in base.values()}
        left = [p for p in free if p not in chosen]
        contrib = {(q, p): bits(base, q, p[0], p[1]) for q in QUADRATURES for p in left}
        local_best = None
        for qp in itertools.permutations(left, len(QUADRATURES)):
            rot, t = rot0, term0[:]
            used = set(dyn_slots)
            for q, p in zip(QUADRATURES, qp):
                r, tm = contrib[(q, p)]
                rot |= r
                t[p[1]] |= tm
                used.add(p[1])
            rhs = 2 * (POP[rot] + w_rot) + 9 + POP[t[0]] + POP[t[1]] + POP[t[2]]
            attempt = 6 * rhs + sum(26 if sl in dyn_slots else 11 for sl in used) + 6 * len(used)
            n += 1
            hist[attempt] += 1
            if local_best is None or attempt < local_best[0]:
                local_best = (attempt, qp)
        place = dict(base, **dict(zip(QUADRATURES, local_best[1])))
        best.append((score(place), place))
    best.sort(key=lambda b: (b[0][1], b[0][0]))
    top = best[0][0][1]
    winners = [b for b in best if b[0][1] == top]
    print(json.dumps({"assignments_scored": n, "rounds_1_to_3_layout": {"rhs": cur[0], "attempt": cur[1], **cur[2]},
                      "shipped_layout": {"rhs": shipped[0], "attempt": shipped[1], **shipped[2]},
                      "best_attempt": top, "best_rhs": best[0][0][0], "shipped_is_a_winner": shipped[1] == top,
                      "dynamic_placements_reaching_the_best": len(winners),
                      "of_which_one_observable_per_lane_and_in_one_slot": sum(1 for b in winners if b[0][2]["one_observable_per_lane"] and b[0][2]["observables_in_one_slot"]),
                      "assignments_by_attempt_instructions": {str(k): hist[k] for k in sorted(hist)[:10]}}, indent=1))
    print("\nbest layouts (instructions per attempt of one wave, RK body + error norm; rounds 1-3: %d, shipped: %d)" % (cur[1], shipped[1]))
    for sc, place in winners:
        grid = [["-"] * 3 for _ in range(4)]
        for c, (l, sl) in place.items():
            grid[l][sl] = c
        print("  attempt %d  rhs %d  rotations %d  terms %s  one observable per lane %s, in one slot %s   lanes: %s" % (
            sc[1], sc[0], sc[2]["rotations"], sc[2]["terms"], "yes" if sc[2]["one_observable_per_lane"] else "no",
            "yes" if sc[2]["observables_in_one_slot"] else "no", " | ".join(",".join(r) for r in grid)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
