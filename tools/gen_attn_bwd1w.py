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
    D(i-1)   this wave's 16 x 16 tile of dQ^T of the PREVIOUS sub-step over all      16 MFMA 16x16x32, 4 per group-step; operands by
             512 keys of the block (the four waves' dS images are one [512][32] image)  transposed reads of the other image buffer
    g = 0: the tile finished in sub-step i-1 + workspace value of the earlier key blocks -> workspace (1 store)
    g = 1 (first sub-step of a tile): LDS-DMA requests of the tile three ahead and of the next tile's workspace values
    g = 2: row constants and Q / dO row fragments of the next sub-step (after A(3) has consumed the current ones)
    g = 3: transposed Q / dO fragments of the next sub-step
    end:  s_waitcnt lgkmcnt(0); s_barrier
Run: python tools/gen_attn_bwd1w.py [--report]   (writes the .inc next to the kernel; --report prints the bundle table)
"""
import os
import sys

IMG_BUF = 512 * 64        # bytes of one dS image: [512 keys][32 queries] bf16
IMG_BUF64 = 256 * 64      # head_dim 64: [256 keys][32 queries]

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


def merge(mf, F, need, bundles, issued_m):
    """Deal the fillers F = [(Op, not-before-MFMA-tag or None)] (list order = dependency order; blocked ones are skipped, not
    waited for) to the MFMA slots mf, balancing the bundles' issue cost; `need`: MFMA tag -> filler tags that must precede it."""
    pending = list(F)
    done_f = set()

    def take(idx):
        op, _ = pending.pop(idx)
        done_f.add(op.tag)
        return op

    def cost_of(op):
        return op.cost * (2 if op.code.count("lds_tr_ld") == 2 else 1)

    for mi, m in enumerate(mf):
        pre = []
        for nt in need.get(m.tag, []):      # fillers this MFMA needs go into the previous bundle, with everything ahead of them
            while nt not in done_f:
                idx = next((k for k, (op, after) in enumerate(pending) if after is None or after in issued_m), None)
                if idx is None:
                    raise RuntimeError(f"deadlock: {m.tag} needs {nt}")
                pre.append(take(idx))
        if pre:
            if bundles:
                bundles[-1][1].extend(pre)
            else:
                bundles.append((None, pre))
        issued_m.add(m.tag)
        fill = []
        c = 0.0
        remaining_m = len(mf) - mi - 1
        tgt = sum(cost_of(op) for op, _ in pending) / (remaining_m + 1)
        while True:
            idx = next((k for k, (op, after) in enumerate(pending) if after is None or after in issued_m), None)
            if idx is None:
                break
            oc = cost_of(pending[idx][0])
            if c + oc / 2 > tgt and fill:
                break
            fill.append(take(idx))
            c += oc
        bundles.append((m, fill))
    if pending:
        raise RuntimeError("fillers left over: " + " ".join(op.tag for op, _ in pending))


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
        # row constants and Q / dO row fragments exist once per sub-step parity (c = the set the NEXT group's S / dP read: this
        # sub-step's for groups 1..3, the next sub-step's -- loaded in group-step 2 -- for its group 0), so that each set has a
        # single definition inside the loop and needs no register copies at the loop's back edge
        c = cur_par if g < 3 else nxt_par
        mf = [mfma_acc(f"dv{gp}", f"oT{ppar}1", "pf1", f"CV1({gp})"),
              mfma_acc(f"dk{gp}", f"qT{ppar}1", "dsf1", f"CK1({gp})"),
              mfma_b(f"sa{Y}", f"qrow{c}0", f"kS[{gn}][0]", f"lse_{c}", f"AS0({gn})"),
              mfma_b(f"dp{Y}", f"orow{c}0", f"vS[{gn}][0]", f"dlt_{c}", f"AP0({gn})"),
              mfma_b(f"sa{Y}", f"qrow{c}1", f"kS[{gn}][1]", f"sa{Y}", f"AS1({gn})"),
              mfma_b(f"dp{Y}", f"orow{c}1", f"vS[{gn}][1]", f"dp{Y}", f"AP1({gn})"),
              mfma_acc(f"dv{g}", f"oT{cur_par}0", "pf0", f"CV0({g})"),
              mfma_acc(f"dk{g}", f"qT{cur_par}0", "dsf0", f"CK0({g})")]
        # D stage of the previous sub-step: this wave's 16 x 16 tile of dQ^T over ALL 512 keys of the block, k-steps of 32 keys,
        # operands from the other (complete) dS image; accumulator alternates with the sub-step so that the finished tile can be
        # written out during the next one
        dqn = "dqA" if s == 0 else "dqB"
        dj = list(range(4 * g, 4 * g + 4))
        for j in dj:
            c = "zero4" if j == 0 else dqn
            mf.append(Op("mfma", f"{dqn} = mfma16(kT[{j}], cat4(bj{j}l, bj{j}h), {c});", f"MD({j})"))
        # ---- fillers in dependency order.  Entries: (Op, not_before_mfma_tag or None)
        F = []

        def add(kind, code, after=None, tag=""):
            F.append((Op(kind, code, tag), after))

        rb = ((s + 1) % 2) * IMG_BUF               # image of the previous sub-step
        wb = s * IMG_BUF                           # image this sub-step writes

        def rd(j):
            add("ldsr", f"const bf16x4 bj{j}l = lds_tr_ld(a_imglo + {rb + j * 2048});", tag=f"RD{j}l")
            add("ldsr", f"const bf16x4 bj{j}h = lds_tr_ld(a_imghi + {rb + j * 2048});", tag=f"RD{j}h")

        if g == 0:
            for j in range(4):
                rd(j)
            # write-out of the tile finished in the previous sub-step: + workspace value of the earlier key blocks -> workspace
            add("vaddr", f"const unsigned ao = a_old + s_oldr;", tag="AO")
            add("ldsr", f"const f32x4 rold = lds_ld<f32x4>(ao + {s * 4096});", tag="ROLD")
        for half in range(2):
            e0 = 8 * half
            pf, dsf = f"pf{half}", f"dsf{half}"
            if half == 1 and g == 2:   # listed first, placed as soon as A(3) has been issued
                # next sub-step's row constants and row fragments: A(3) (issued in this group-step) was the last reader of the old ones
                nx = "1" if s == 0 else "n"
                add("vaddr", f"const unsigned pc_ = a_const + s_pc{nx};", after="AP1(3)", tag="PC")
                add("vaddr", f"const unsigned pr_ = a_row + s_x{nx};", after="AP1(3)", tag="PR")
                for G in range(4):
                    add("ldsr", f"{{ const f32x4 c_ = lds_ld<f32x4>(pc_ + {32 * G}); lse_{nxt_par}[{4 * G}] = c_[0]; lse_{nxt_par}[{4 * G + 1}] = c_[1]; "
                                f"lse_{nxt_par}[{4 * G + 2}] = c_[2]; lse_{nxt_par}[{4 * G + 3}] = c_[3]; }}", after="AP1(3)", tag=f"LCa{G}")
                    add("ldsr", f"{{ const f32x4 c_ = lds_ld<f32x4>(pc_ + {256 + 32 * G}); dlt_{nxt_par}[{4 * G}] = c_[0]; dlt_{nxt_par}[{4 * G + 1}] = c_[1]; "
                                f"dlt_{nxt_par}[{4 * G + 2}] = c_[2]; dlt_{nxt_par}[{4 * G + 3}] = c_[3]; }}", after="AP1(3)", tag=f"LCd{G}")
                add("ldsr", f"qrow{nxt_par}0 = lds_ld<bf16x8>(pr_);", after="AP1(3)", tag="LQ0")
                add("ldsr", f"orow{nxt_par}0 = lds_ld<bf16x8>(pr_ + (OR_ - QR));", after="AP1(3)", tag="LO0")
                add("ldsr", f"qrow{nxt_par}1 = lds_ld<bf16x8>(pr_ ^ 32u);", after="AP1(3)", tag="LQ1")
                add("ldsr", f"orow{nxt_par}1 = lds_ld<bf16x8>((pr_ ^ 32u) + (OR_ - QR));", after="AP1(3)", tag="LO1")
            for e in range(e0, e0 + 8):
                add("exp", f"{sa}[{e}] = fast_exp2({sa}[{e}]);", tag=f"E{e}")
            for e in range(e0, e0 + 8):
                if e % 2 == 0:
                    add("mul", f"\n#ifdef BWD1W_PKMUL\n  {{ const f32x2 t_ = f32x2{{{sa}[{e}], {sa}[{e + 1}]}} * f32x2{{{dp}[{e}], {dp}[{e + 1}]}}; "
                               f"{dp}[{e}] = t_[0]; {dp}[{e + 1}] = t_[1]; }}\n#else\n  {dp}[{e}] = {sa}[{e}] * {dp}[{e}];\n#endif\n ", tag=f"MU{e}")
                else:
                    add("mul", f"\n#ifndef BWD1W_PKMUL\n  {dp}[{e}] = {sa}[{e}] * {dp}[{e}];\n#endif\n ", tag=f"MU{e}")
            for i in range(4):
                add("cvt", f"{pf}[{i}] = pack2bf({sa}[{e0 + 2 * i}], {sa}[{e0 + 2 * i + 1}]);", tag=f"CP{4 * half + i}")
            for i in range(4):
                add("cvt", f"{dsf}[{i}] = pack2bf({dp}[{e0 + 2 * i}], {dp}[{e0 + 2 * i + 1}]);", tag=f"CD{4 * half + i}")
            add("ldsw8", f"lds_st<u32x2>((a_imgw ^ {32 * half}u) + {wb + g * 32 * 64}, u32x2{{{dsf}[0], {dsf}[1]}});", tag=f"WR{half}0")
            add("ldsw8", f"lds_st<u32x2>((a_imgw ^ {32 * half + 16}u) + {wb + g * 32 * 64}, u32x2{{{dsf}[2], {dsf}[3]}});", tag=f"WR{half}1")
            if half == 0 and g == 0:
                dqo = "dqB" if s == 0 else "dqA"       # the tile finished in the previous sub-step
                # element by element: a <4 x float> add becomes v_pk_add_f32, which costs more beside MFMAs than two v_add_f32
                add("add", f"f32x4 rv; rv[0] = {dqo}[0] + rold[0]; rv[1] = {dqo}[1] + rold[1]; rv[2] = {dqo}[2] + rold[2]; rv[3] = {dqo}[3] + rold[3];", tag="RA0")
                add("vaddr", f"const unsigned rvo = BWD1W_RVO(wsoff + s_redoff{s});", tag="RVO")
                add("store", "__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rv), rsWs, rvo, 0, BWD1W_WS_ST_AUX);", tag="RST")
            if half == 1 and g < 3:
                for j in range(4 * g + 4, 4 * g + 8):
                    rd(j)
            if half == 0 and g == 1 and s == 0:
                add("dma", "issue_tile();", tag="DMA")
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
            need[f"MD({j})"] = [f"RD{j}h"] if g == 0 else []
        merge(mf, F, need, bundles, issued_m)
    return bundles



def sub_step64(s):
    """head_dim 64: a wave owns 64 keys (two 32-key groups) of a 256-key block; S / dP take 4 k-steps of 16 head dims, dV^T / dK^T
    two 32-wide head-dim blocks each; the wave's part of dQ^T[64][32] is head-dim tile `wid` (16 dims) of BOTH 16-query tiles over
    all 256 keys: 8 k-steps x 2 tiles of 16x16x32 MFMAs sharing their K^T fragments.  Same pipeline as head_dim 32 with two
    group-steps per sub-step.  Register economy (one set of row constants / row fragments / first-half transposed fragments,
    reloaded as soon as their last reader has been issued; only the second-half transposed fragments, which the C1 MFMAs of the
    next group-step still read, alternate with the sub-step)."""
    cur, nxt = ("a", "b") if s == 0 else ("b", "a")
    bundles = []
    issued_m = set()
    dqn = "dqA" if s == 0 else "dqB"
    dqo = "dqB" if s == 0 else "dqA"
    rb = ((s + 1) % 2) * IMG_BUF64
    wb = s * IMG_BUF64
    for g in range(2):
        X, Y = ("A", "B") if g == 0 else ("B", "A")
        gp = 1 - g                                          # previous group (of the previous sub-step when g == 0)
        gn = 1 - g
        sa, dp = f"sa{X}", f"dp{X}"
        mf = [Op("mfma", f"MFMA_ACC(dv{gp}0, oT1d0, pf1);", f"CV1a({gp})"), Op("mfma", f"MFMA_ACC(dv{gp}1, oT1d1, pf1);", f"CV1b({gp})"),
              Op("mfma", f"MFMA_ACC(dk{gp}0, qT1d0, dsf1);", f"CK1a({gp})"), Op("mfma", f"MFMA_ACC(dk{gp}1, qT1d1, dsf1);", f"CK1b({gp})")]
        for ks in range(4):
            mf.append(Op("mfma", f"sa{Y} = mfma32(qrow{ks}, kS[{gn}][{ks}], {'lse_t' if ks == 0 else 'sa' + Y});", f"AS{ks}({gn})"))
            mf.append(Op("mfma", f"dp{Y} = mfma32(orow{ks}, vS[{gn}][{ks}], {'dlt_t' if ks == 0 else 'dp' + Y});", f"AP{ks}({gn})"))
        mf += [Op("mfma", f"MFMA_ACC(dv{g}0, oT0d0, pf0);", f"CV0a({g})"), Op("mfma", f"MFMA_ACC(dv{g}1, oT0d1, pf0);", f"CV0b({g})"),
               Op("mfma", f"MFMA_ACC(dk{g}0, qT0d0, dsf0);", f"CK0a({g})"), Op("mfma", f"MFMA_ACC(dk{g}1, qT0d1, dsf0);", f"CK0b({g})")]
        for j in range(4 * g, 4 * g + 4):
            for qt in range(2):
                c = "zero4" if j == 0 else f"{dqn}{qt}"
                mf.append(Op("mfma", f"{dqn}{qt} = mfma16(kT[{j}], cat4(bj{j}q{qt}l, bj{j}q{qt}h), {c});", f"MD({j},{qt})"))
        F = []

        def add(kind, code, after=None, tag=""):
            F.append((Op(kind, code, tag), after))

        def rd(j, after):
            for qt in range(2):
                x = f" ^ 32u" if qt else ""
                add("ldsr", f"const bf16x4 bj{j}q{qt}l = lds_tr_ld((a_imglo{x}) + {rb + j * 2048});", after=after, tag=f"RD{j}q{qt}l")
                add("ldsr", f"const bf16x4 bj{j}q{qt}h = lds_tr_ld((a_imghi{x}) + {rb + j * 2048});", after=after, tag=f"RD{j}q{qt}h")

        # image reads of the D stage: two k-steps at a time, ~6 MFMA slots ahead of their consumers (register economy)
        for j in range(4 * g, 4 * g + 4):
            rd(j, f"AS1({gn})" if j % 4 < 2 else f"AS3({gn})")
        if g == 0:
            add("vaddr", "const unsigned ao = a_old + s_oldr;", tag="AO")
            for qt in range(2):
                add("ldsr", f"const f32x4 rold{qt} = lds_ld<f32x4>(ao + {(s * 2 + qt) * 4096});", tag=f"ROLD{qt}")
            # second-half transposed fragments of THIS sub-step (one set: the C1 MFMAs just issued were the last readers of the
            # previous sub-step's; the first reader of these is C1 of the next group-step)
            add("vaddr", f"const unsigned pl1_ = a_trlo + s_x{s};", after="CK1b(1)", tag="PL1")
            add("vaddr", f"const unsigned ph1_ = a_trhi + s_x{s};", after="CK1b(1)", tag="PH1")
            for d in range(2):
                add("ldsr", f"qT1d{d} = cat4(lds_tr_ld((pl1_ ^ {64 * d}u) + {16 * 128}), lds_tr_ld((ph1_ ^ {64 * d}u) + {16 * 128}));",
                    after="CK1b(1)", tag=f"LT1q{d}")
                add("ldsr", f"oT1d{d} = cat4(lds_tr_ld((pl1_ ^ {64 * d}u) + {16 * 128} + (OR_ - QR)), lds_tr_ld((ph1_ ^ {64 * d}u) + {16 * 128} + (OR_ - QR)));",
                    after="CK1b(1)", tag=f"LT1o{d}")
        for half in range(2):
            e0 = 8 * half
            pf, dsf = f"pf{half}", f"dsf{half}"
            if half == 1 and g == 0:
                # next sub-step's row constants and row fragments: A(1), issued in this group-step, was the last reader of the old ones
                nx = "1" if s == 0 else "n"
                add("vaddr", f"const unsigned pc_ = a_const + s_pc{nx};", after="AP3(1)", tag="PC")
                add("vaddr", f"const unsigned pr_ = a_row + s_x{nx};", after="AP3(1)", tag="PR")
                for G in range(4):
                    add("ldsr", f"{{ const f32x4 c_ = lds_ld<f32x4>(pc_ + {32 * G}); lse_t[{4 * G}] = c_[0]; lse_t[{4 * G + 1}] = c_[1]; "
                                f"lse_t[{4 * G + 2}] = c_[2]; lse_t[{4 * G + 3}] = c_[3]; }}", after="AP3(1)", tag=f"LCa{G}")
                    add("ldsr", f"{{ const f32x4 c_ = lds_ld<f32x4>(pc_ + {256 + 32 * G}); dlt_t[{4 * G}] = c_[0]; dlt_t[{4 * G + 1}] = c_[1]; "
                                f"dlt_t[{4 * G + 2}] = c_[2]; dlt_t[{4 * G + 3}] = c_[3]; }}", after="AP3(1)", tag=f"LCd{G}")
                for ks in range(4):
                    add("ldsr", f"qrow{ks} = lds_ld<bf16x8>(pr_ ^ {32 * ks}u);", after="AP3(1)", tag=f"LQ{ks}")
                    add("ldsr", f"orow{ks} = lds_ld<bf16x8>((pr_ ^ {32 * ks}u) + (OR_ - QR));", after="AP3(1)", tag=f"LO{ks}")
            for e in range(e0, e0 + 8):
                add("exp", f"{sa}[{e}] = fast_exp2({sa}[{e}]);", tag=f"E{e}")
            for e in range(e0, e0 + 8):
                add("mul", f"{dp}[{e}] = {sa}[{e}] * {dp}[{e}];", tag=f"MU{e}")
            for i in range(4):
                add("cvt", f"{pf}[{i}] = pack2bf({sa}[{e0 + 2 * i}], {sa}[{e0 + 2 * i + 1}]);", tag=f"CP{4 * half + i}")
            for i in range(4):
                add("cvt", f"{dsf}[{i}] = pack2bf({dp}[{e0 + 2 * i}], {dp}[{e0 + 2 * i + 1}]);", tag=f"CD{4 * half + i}")
            add("ldsw8", f"lds_st<u32x2>((a_imgw ^ {32 * half}u) + {wb + g * 32 * 64}, u32x2{{{dsf}[0], {dsf}[1]}});", tag=f"WR{half}0")
            add("ldsw8", f"lds_st<u32x2>((a_imgw ^ {32 * half + 16}u) + {wb + g * 32 * 64}, u32x2{{{dsf}[2], {dsf}[3]}});", tag=f"WR{half}1")
            if half == 0 and g == 0:
                for qt in range(2):
                    add("add", f"f32x4 rv{qt}; rv{qt}[0] = {dqo}{qt}[0] + rold{qt}[0]; rv{qt}[1] = {dqo}{qt}[1] + rold{qt}[1]; "
                               f"rv{qt}[2] = {dqo}{qt}[2] + rold{qt}[2]; rv{qt}[3] = {dqo}{qt}[3] + rold{qt}[3];", tag=f"RA{qt}")
                add("vaddr", f"const unsigned rvo = BWD1W_RVO(wsoff + s_redoff{s});", tag="RVO")
                add("store", "__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rv0), rsWs, rvo, 0, BWD1W_WS_ST_AUX);", tag="RST0")
                add("store", "__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rv1), rsWs, rvo + 4096u, 0, BWD1W_WS_ST_AUX);", tag="RST1")
            if half == 0 and g == 1 and s == 0:
                add("dma", "issue_tile();", tag="DMA")
        if g == 1:
            # first-half transposed fragments of the next sub-step: C0(1) of this one was their last reader
            nx = "1" if s == 0 else "n"
            add("vaddr", f"const unsigned pl0_ = a_trlo + s_x{nx};", after="CK0b(1)", tag="PL0")
            add("vaddr", f"const unsigned ph0_ = a_trhi + s_x{nx};", after="CK0b(1)", tag="PH0")
            for d in range(2):
                add("ldsr", f"qT0d{d} = cat4(lds_tr_ld(pl0_ ^ {64 * d}u), lds_tr_ld(ph0_ ^ {64 * d}u));", after="CK0b(1)", tag=f"LT0q{d}")
                add("ldsr", f"oT0d{d} = cat4(lds_tr_ld((pl0_ ^ {64 * d}u) + (OR_ - QR)), lds_tr_ld((ph0_ ^ {64 * d}u) + (OR_ - QR)));",
                    after="CK0b(1)", tag=f"LT0o{d}")
        need = {f"CV0a({g})": ["CP3"], f"CK0a({g})": ["CD3"]}
        for j in range(4 * g, 4 * g + 4):
            need[f"MD({j},0)"] = [f"RD{j}q1h"]
        merge(mf, F, need, bundles, issued_m)
    return bundles

def guard(o):
    """Timing-only ablation switches (-DABL_...: results are wrong, the instruction stream is otherwise unchanged)."""
    t = o.tag
    g = None
    if t.startswith("WR"): g = "ABL_NO_WR"
    elif t in ("AO", "RVO") or t.startswith(("RA", "ROLD", "RST")): g = "ABL_NO_WP"
    elif t.startswith("E") and t[1:].isdigit(): g = "ABL_NO_EXP"
    elif t.startswith("CV") or t.startswith("CK"): g = "ABL_NO_ACC"
    elif t.startswith("MD") or t.startswith("RD"): g = "ABL_NO_MD"
    if g is None:
        return o.code
    if t.startswith("RST"):
        return f"\n#if !defined({g}) && !defined(ABL_NO_RST)\n  {o.code}\n#endif\n "
    return f"\n#ifndef {g}\n  {o.code}\n#endif\n "


def emit(hd=32):
    out = ["// GENERATED by tools/gen_attn_bwd1w.py -- do not edit; the bundle order is the instruction order (see the generator's header)."]
    report = []
    for s in range(2):
        out.append(f"// ======================== sub-step {s} of the tile ========================")
        if s == 1:
            out.append("#ifndef BWD1W_ONLY_SUBSTEP0     // (the half iteration behind the loop: see the kernels)")
        out.append("{")
        bl = sub_step(s) if hd == 32 else sub_step64(s)
        nst = 0
        for bi, (m, fill) in enumerate(bl):
            if m and (m.tag.startswith(("CV1(", "CV0(", "CV1a", "CV0a"))):
                out.append(f"  STAMP({nst});")
                nst += 1
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
        flat = [o for m_, fl in bl for o in fl]
        last_wr = max(k for k, o in enumerate(flat) if o.tag.startswith("WR"))
        younger = sum(o.code.count("lds_ld") + o.code.count("lds_tr_ld") + o.code.count("lds_st") for o in flat[last_wr + 1:])
        assert younger <= 15
        out.append(f"  STAMP({nst});")
        out.append(f"  SUBSTEP_END({younger});     // the image writes are done when at most the {younger} LDS reads issued after them are pending")
        out.append(f"  STAMP({nst + 1});")
        out.append(f"  STAMP_ACCUM({s});")
        out.append("}")
        if s == 1:
            out.append("#endif")
    return "\n".join(out) + "\n", report


def emit_drain(hd=32):
    """The pipeline drain behind the last real sub-step X of a key block, instead of padding sub-steps run in full: SLIM(s) is
    sub-step s reduced to what still has an effect -- dV^T / dK^T of X's last group (second half), the dQ^T product of X from its dS
    image, the write-out of the tile before -- and TINY(s) is the last write-out alone.  Same instruction order as the full body."""
    out = ["// GENERATED by tools/gen_attn_bwd1w.py -- do not edit.  Selected by the kernels with BWD1W_DRAIN_SLIM0 / _SLIM1 / _TINY0 / _TINY1."]
    last_c1 = ("CV1(3)", "CK1(3)") if hd == 32 else ("CV1a(1)", "CV1b(1)", "CK1a(1)", "CK1b(1)")
    wo = lambda t: t in ("AO", "RVO") or t.startswith(("ROLD", "RA", "RST"))
    for kind in ("SLIM", "TINY"):
        for s in range(2):
            out.append(f"#ifdef BWD1W_DRAIN_{kind}{s}")
            out.append("{")
            bl = sub_step(s) if hd == 32 else sub_step64(s)
            for bi, (m, fill) in enumerate(bl):
                keep_m = m is not None and kind == "SLIM" and (m.tag in last_c1 or m.tag.startswith("MD("))
                keep_f = [o for o in fill if wo(o.tag) or (kind == "SLIM" and o.tag.startswith("RD"))]
                if not keep_m and not keep_f:
                    continue
                out.append(f"  // from bundle {bi}")
                if keep_m:
                    out.append("  " + guard(m))
                    out.append("  __builtin_amdgcn_sched_barrier(0);")
                for o in keep_f:
                    out.append("  " + guard(o))
                out.append("  __builtin_amdgcn_sched_barrier(0);")
            if kind == "SLIM":
                out.append("  SUBSTEP_END(0);     // every wave is done with the dS image: the epilogue may reuse the region")
            out.append("}")
            out.append("#endif")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text, report = emit(32)
    text64, report64 = emit(64)
    OUT64 = OUT.replace("_body.inc", "_body_hd64.inc")
    drains = ((OUT.replace("_body.inc", "_drain.inc"), emit_drain(32)), (OUT.replace("_body.inc", "_drain_hd64.inc"), emit_drain(64)))
    if "--check" in sys.argv:      # the committed bodies are what this generator produces (tests/test_cpu_host.py)
        stale = [p for p, t in ((OUT, text), (OUT64, text64)) + drains if not os.path.exists(p) or open(p).read() != t]
        print("stale: " + ", ".join(stale) if stale else "generated bodies are current")
        sys.exit(1 if stale else 0)
    with open(OUT, "w") as f:
        f.write(text)
    tot = sum(r[4] for r in report)
    nm = sum(1 for r in report if r[2] != "-")
    print(f"wrote {OUT}: {len(report)} bundles, {nm} MFMAs per tile, issue estimate {tot} cycles per tile")
    with open(OUT64, "w") as f:
        f.write(text64)
    for p_, t_ in drains:
        with open(p_, "w") as f:
            f.write(t_)
    print(f"wrote the head_dim-64 body: {len(report64)} bundles, issue estimate {sum(r[4] for r in report64)} cycles per tile")
    if "--report64" in sys.argv:
        report = report64
        sys.argv.append("--report")
    if "--report" in sys.argv:
        for s, bi, m, fl, cost in report:
            print(f"s{s} b{bi:02d} {m:10s} {cost:5.0f}  " + " ".join(fl))
