#!/usr/bin/env python3
"""Generates octcubem_amd/csrc/attn_bwd1w_body.inc: the hand-placed tile body of the one-wave-per-SIMD attention backward
(csrc/attn_bwd1w.hip).  The body of one 64-query tile = two 32-query sub-steps; each sub-step is a list of BUNDLES -- one MFMA
followed by the vector / LDS instructions placed in its shadow -- pinned with __builtin_amdgcn_sched_barrier(0) between bundles,
so the order below IS the order of the emitted instruction stream (the compiler still allocates registers, inserts the
s_waitcnt for its own LDS reads and pads MFMA hazards).  The placement follows MI355X_MICROARCH.md, 'vector-instruction ISSUE
cost': an MFMA holds the issue port 8 of its 32 cycles, v_exp_f32 8, v_mul / v_cvt_pk 4-5, LDS 4-13; fillers are dealt to the
MFMA slots so that every bundle costs about the same, in dependency order.

Software pipeline of sub-step i (32 queries x this wave's 128 keys), group-step g = 0..3 (32 keys each):
    C1(g-1)  dV^T, dK^T += (second 16 queries of the previous group)          2 MFMA   (accumulators in AGPRs, inline asm)
    A(g+1)   S, dP of the NEXT group (for g = 3: group 0 of the next sub-step)  4 MFMA
    B(g)     exp2, dS = P * dP, bf16 conversions, dS -> private LDS image       48 VALU + 4 ds_write
    C0(g)    dV^T, dK^T += (first 16 queries)                                   2 MFMA
    D(i-1)   dQ^T partial of the PREVIOUS sub-step over this wave's keys        8 MFMA over g = 0..2, operands read from the image
                                                                                 before this sub-step's writes reach their rows
    g = 0: reduce(i-2): the four waves' partial tiles of two sub-steps ago + workspace value -> workspace (1 store)
    g = 1 (first sub-step of a tile): LDS-DMA requests of the tile three ahead and of the next tile's workspace values
    g = 2: row constants and Q / dO row fragments of the next sub-step (after A(3) has consumed the current ones)
    g = 3: dQ^T partial of sub-step i-1 -> partial-tile buffer; transposed Q / dO fragments of the next sub-step
    end:  s_waitcnt lgkmcnt(0); s_barrier
Run: python tools/gen_attn_bwd1w.py [--report]   (writes the .inc next to the kernel; --report prints the bundle table)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "octcubem_amd", "csrc", "attn_bwd1w_body.inc")

# issue cost in cycles of one wave's stream (MI355X_MICROARCH.md, cycle constants)
COST = {"mfma": 8, "exp": 8, "mul": 4, "cvt": 5, "add": 16, "ldsr": 4, "ldsw8": 6, "ldsw16": 13, "store": 8, "dma": 30, "salu": 2,
        "vaddr": 4}


class Op:
    def __init__(self, kind, code, tag=""):
        self.kind, self.code, self.tag = kind, code, tag

    @property
    def cost(self):
        return COST[self.kind]


def sub_step(s):
    """Returns the list of bundles [(mfma Op or None, [filler Ops])] of sub-step s (0 / 1) of a tile."""
    cur_par, nxt_par = ("a", "b") if s == 0 else ("b", "a")       # parity of the transposed fragments (qT / oT)
    prev_par = nxt_par
    bundles = []
    issued_m = {"MD(7)"} if False else set()     # MFMA tags issued so far in this sub-step

    def mfma_acc(acc, a, b, tag):
        return Op("mfma", f"MFMA_ACC({acc}, {a}, {b});", tag)

    def mfma_b(dst, a, b, c, tag):
        return Op("mfma", f"{dst} = mfma32({a}, {b}, {c});", tag)

    for g in range(4):
        X = "A" if g % 2 == 0 else "B"           # buffer of the CURRENT group's sa / dp
        Y = "B" if g % 2 == 0 else "A"           # buffer of the NEXT group's
        gp, ppar = (g - 1, cur_par) if g > 0 else (3, prev_par)
        gn = (g + 1) % 4
        sa, dp = f"sa{X}", f"dp{X}"
        mf = [mfma_acc(f"dv{gp}", f"oT{ppar}1", "pf1", f"CV1({gp})"),
              mfma_acc(f"dk{gp}", f"qT{ppar}1", "dsf1", f"CK1({gp})"),
              mfma_b(f"sa{Y}", "qrow0", f"kS[{gn}][0]", "lse_t", f"AS0({gn})"),
              mfma_b(f"dp{Y}", "orow0", f"vS[{gn}][0]", "dlt_t", f"AP0({gn})"),
              mfma_b(f"sa{Y}", "qrow1", f"kS[{gn}][1]", f"sa{Y}", f"AS1({gn})"),
              mfma_b(f"dp{Y}", "orow1", f"vS[{gn}][1]", f"dp{Y}", f"AP1({gn})"),
              mfma_acc(f"dv{g}", f"oT{cur_par}0", "pf0", f"CV0({g})"),
              mfma_acc(f"dk{g}", f"qT{cur_par}0", "dsf0", f"CK0({g})")]
        # D stage of the previous sub-step: k-steps j (16 keys each)
        dj = {0: [0, 1, 2], 1: [3, 4, 5], 2: [6, 7], 3: []}[g]
        for j in dj:
            c = "zero16" if j == 0 else "dq"
            mf.append(mfma_b("dq", f"kT[{j}]", f"cat4(bj{j}l, bj{j}h)", c, f"MD({j})"))
        # ---- fillers in dependency order.  Entries: (Op, not_before_mfma_tag or None)
        F = []

        def add(kind, code, after=None, tag=""):
            F.append((Op(kind, code, tag), after))

        if g == 0:
            for j in range(4):
                add("ldsr", f"const bf16x4 bj{j}l = lds_tr_ld(a_imglo + {j * 1024});", tag=f"RD{j}l")
                add("ldsr", f"const bf16x4 bj{j}h = lds_tr_ld(a_imghi + {j * 1024});", tag=f"RD{j}h")
            # reduce(i-2): partial tiles of the four waves (this wave's register quad) + workspace value
            for w in range(4):
                add("ldsr", f"const f32x4 rp{w} = lds_ld<f32x4>(a_red + {(s * 4 + w) * 4096});", tag=f"RP{w}")
            add("vaddr", f"const unsigned ao = a_old + s_oldr;", tag="AO")
            add("ldsr", f"const f32x4 rold = lds_ld<f32x4>(ao + {s * 4096});", tag="ROLD")
        if g == 1:
            for j in range(4, 8):
                add("ldsr", f"const bf16x4 bj{j}l = lds_tr_ld(a_imglo + {j * 1024});", tag=f"RD{j}l")
                add("ldsr", f"const bf16x4 bj{j}h = lds_tr_ld(a_imghi + {j * 1024});", tag=f"RD{j}h")
        wr_base = f"a_imgw + {g * 32 * 64}"
        for half in range(2):
            e0 = 8 * half
            pf, dsf = f"pf{half}", f"dsf{half}"
            for e in range(e0, e0 + 8):
                add("exp", f"{sa}[{e}] = fast_exp2({sa}[{e}]);", tag=f"E{e}")
            for e in range(e0, e0 + 8):
                add("mul", f"{dp}[{e}] = {sa}[{e}] * {dp}[{e}];", tag=f"MU{e}")
            for i in range(4):
                add("cvt", f"{pf}[{i}] = pack2bf({sa}[{e0 + 2 * i}], {sa}[{e0 + 2 * i + 1}]);", tag=f"CP{4 * half + i}")
            for i in range(4):
                add("cvt", f"{dsf}[{i}] = pack2bf({dp}[{e0 + 2 * i}], {dp}[{e0 + 2 * i + 1}]);", tag=f"CD{4 * half + i}")
            add("ldsw8", f"lds_st<u32x2>((a_imgw ^ {32 * half}u) + {g * 32 * 64}, u32x2{{{dsf}[0], {dsf}[1]}});", tag=f"WR{half}0")
            add("ldsw8", f"lds_st<u32x2>((a_imgw ^ {32 * half + 16}u) + {g * 32 * 64}, u32x2{{{dsf}[2], {dsf}[3]}});", tag=f"WR{half}1")
            if half == 0 and g == 0:
                # reduce arithmetic + store (fixed summation order: waves 0, 1, 2, 3, then the earlier key blocks)
                # element by element: a <4 x float> add becomes v_pk_add_f32, which costs more beside MFMAs than two v_add_f32
                add("add", "f32x4 rv; rv[0] = rp0[0] + rp1[0]; rv[1] = rp0[1] + rp1[1]; rv[2] = rp0[2] + rp1[2]; rv[3] = rp0[3] + rp1[3];", tag="RA0")
                for k_, src in enumerate(("rp2", "rp3", "rold")):
                    add("add", f"rv[0] += {src}[0]; rv[1] += {src}[1]; rv[2] += {src}[2]; rv[3] += {src}[3];", tag=f"RA{k_ + 1}")
                add("vaddr", f"const unsigned rvo = wsoff + s_redoff{s};", tag="RVO")
                add("store", "__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rv), rsWs, rvo, 0, 0);", tag="RST")
            if half == 0 and g == 1 and s == 0:
                add("dma", "issue_tile();", tag="DMA")
            if half == 1 and g == 2:
                # next sub-step's row constants and row fragments: A(3) (issued in this group-step) was the last reader of the old ones
                nx = "1" if s == 0 else "n"
                add("vaddr", f"const unsigned pc_ = a_const + s_pc{nx};", after="AP1(3)", tag="PC")
                add("vaddr", f"const unsigned pr_ = a_row + s_x{nx};", after="AP1(3)", tag="PR")
                for G in range(4):
                    add("ldsr", f"{{ const f32x4 c_ = lds_ld<f32x4>(pc_ + {32 * G}); lse_t[{4 * G}] = c_[0]; lse_t[{4 * G + 1}] = c_[1]; "
                                f"lse_t[{4 * G + 2}] = c_[2]; lse_t[{4 * G + 3}] = c_[3]; }}", after="AP1(3)", tag=f"LCa{G}")
                    add("ldsr", f"{{ const f32x4 c_ = lds_ld<f32x4>(pc_ + {256 + 32 * G}); dlt_t[{4 * G}] = c_[0]; dlt_t[{4 * G + 1}] = c_[1]; "
                                f"dlt_t[{4 * G + 2}] = c_[2]; dlt_t[{4 * G + 3}] = c_[3]; }}", after="AP1(3)", tag=f"LCd{G}")
                add("ldsr", "qrow0 = lds_ld<bf16x8>(pr_);", after="AP1(3)", tag="LQ0")
                add("ldsr", "orow0 = lds_ld<bf16x8>(pr_ + (OR_ - QR));", after="AP1(3)", tag="LO0")
                add("ldsr", "qrow1 = lds_ld<bf16x8>(pr_ ^ 32u);", after="AP1(3)", tag="LQ1")
                add("ldsr", "orow1 = lds_ld<bf16x8>((pr_ ^ 32u) + (OR_ - QR));", after="AP1(3)", tag="LO1")
            if half == 0 and g == 3:
                pb = (s + 1) % 2                   # partial-tile buffer of sub-step i-1
                for G in range(4):
                    add("ldsw16", f"lds_st<f32x4>(a_part + {pb * 4 * 4096 + G * 1024}, f32x4{{dq[{4 * G}], dq[{4 * G + 1}], dq[{4 * G + 2}], dq[{4 * G + 3}]}});",
                        after="MD(7)", tag=f"WP{G}")
            if half == 1 and g == 3:
                nx = "1" if s == 0 else "n"
                add("vaddr", f"const unsigned pl_ = a_trlo + s_x{nx};", tag="PL")
                add("vaddr", f"const unsigned ph_ = a_trhi + s_x{nx};", tag="PH")
                for ss in range(2):
                    add("ldsr", f"qT{nxt_par}{ss} = cat4(lds_tr_ld(pl_ + {ss * 16 * 64}), lds_tr_ld(ph_ + {ss * 16 * 64}));", tag=f"LTq{ss}")
                    add("ldsr", f"oT{nxt_par}{ss} = cat4(lds_tr_ld(pl_ + {ss * 16 * 64} + (OR_ - QR)), lds_tr_ld(ph_ + {ss * 16 * 64} + (OR_ - QR)));",
                        tag=f"LTo{ss}")
        # the two-read fillers above count double
        # ---- merge: deal the fillers to the MFMA slots by cost
        need = {  # MFMA tag -> filler tags that must precede it
            f"CV0({g})": ["CP3"], f"CK0({g})": ["CD3"],
        }
        for j in dj:
            need[f"MD({j})"] = [f"RD{j}h"] if (g, j) in ((0, 0), (0, 1), (0, 2), (1, 4), (1, 5)) else []
        total = sum(op.cost * (2 if op.code.count("lds_tr_ld") == 2 else 1) for op, _ in F)
        target = total / len(mf)
        fi = 0
        done_f = set()
        for mi, m in enumerate(mf):
            pre = []
            # fillers this MFMA needs first go into the PREVIOUS bundle (or a filler-only bundle at the start)
            for nt in need.get(m.tag, []):
                while nt not in done_f and fi < len(F):
                    op, after = F[fi]
                    if after is not None and after not in issued_m:
                        raise RuntimeError(f"deadlock: {m.tag} needs {nt} but {op.tag} waits for {after}")
                    pre.append(op); done_f.add(op.tag); fi += 1
            if pre:
                if bundles:
                    bundles[-1][1].extend(pre)
                else:
                    bundles.append((None, pre))
            issued_m.add(m.tag)
            fill = []
            c = 0.0
            remaining_m = len(mf) - mi - 1
            remaining_cost = sum(op.cost for op, _ in F[fi:])
            tgt = remaining_cost / (remaining_m + 1)
            while fi < len(F):
                op, after = F[fi]
                if after is not None and after not in issued_m:
                    break
                oc = op.cost * (2 if op.code.count("lds_tr_ld") == 2 else 1)
                if c + oc / 2 > tgt and fill:
                    break
                fill.append(op); done_f.add(op.tag); fi += 1
                c += oc
            bundles.append((m, fill))
        if fi < len(F):
            bundles[-1][1].extend(op for op, _ in F[fi:])
    return bundles


def guard(o):
    """Timing-only ablation switches (-DABL_...: results are wrong, the instruction stream is otherwise unchanged)."""
    t = o.tag
    g = None
    if t.startswith("WR"): g = "ABL_NO_WR"
    elif t.startswith("WP") or t.startswith("RP") or t in ("AO", "ROLD", "RVO", "RST") or t.startswith("RA"): g = "ABL_NO_WP"
    elif t.startswith("E") and t[1:].isdigit(): g = "ABL_NO_EXP"
    elif t.startswith("CV") or t.startswith("CK"): g = "ABL_NO_ACC"
    elif t.startswith("MD") or t.startswith("RD"): g = "ABL_NO_MD"
    if g is None:
        return o.code
    return f"\n#ifndef {g}\n  {o.code}\n#endif\n "


def emit():
    out = ["// GENERATED by tools/gen_attn_bwd1w.py -- do not edit; the bundle order is the instruction order (see the generator's header)."]
    report = []
    for s in range(2):
        out.append(f"// ======================== sub-step {s} of the tile ========================")
        out.append("{")
        bl = sub_step(s)
        for bi, (m, fill) in enumerate(bl):
            cost = (m.cost if m else 0) + sum(o.cost for o in fill)
            out.append(f"  // bundle {bi}: {m.tag if m else '-'} + {len(fill)} fillers, issue ~{cost} cycles")
            if m:
                out.append("  " + guard(m))
                out.append("  __builtin_amdgcn_sched_barrier(0);")     # the MFMA opens its bundle
            for o in fill:
                out.append("  " + guard(o))
                if o.tag in ("CP3", "CD3", "CP7", "CD7"):     # last producer of an asm MFMA's operand: nothing may sink below it
                    out.append("  __builtin_amdgcn_sched_barrier(0);")
            out.append("  __builtin_amdgcn_sched_barrier(0);")
            report.append((s, bi, m.tag if m else "-", [o.tag for o in fill], cost))
        out.append("  SUBSTEP_END();")
        out.append("}")
    return "\n".join(out) + "\n", report


if __name__ == "__main__":
    text, report = emit()
    with open(OUT, "w") as f:
        f.write(text)
    tot = sum(r[4] for r in report)
    nm = sum(1 for r in report if r[2] != "-")
    print(f"wrote {OUT}: {len(report)} bundles, {nm} MFMAs per tile, issue estimate {tot} cycles per tile, MFMA pipe {nm * 32}")
    if "--report" in sys.argv:
        for s, bi, m, fl, cost in report:
            print(f"s{s} b{bi:02d} {m:10s} {cost:5.0f}  " + " ".join(fl))
