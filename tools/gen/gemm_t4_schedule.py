"""Generates the 128-slot K-tile schedule macro of gemm_bf16_t4_kernel (csrc/gemm.hip, variant 26): variant 18's pipeline
(two 64 KB buffers, register-resident K-tile, DMA of K-tile t+2 into the buffer just freed) issued as v_mfma_f32_16x16x32_bf16.
`python tools/gen/gemm_t4_schedule.py` prints it; paste between the GENERATED markers.

One K-tile (BK = 64) of a 128x128 wave tile = 2 k-steps x 64 MFMAs (16x16x32, 16 cycles each).  Slot j = MFMA j + at most one filler.
  T0: reads of k-step 1 on even slots 0..30 (k-step 0 was read by the previous iteration), lgkmcnt(0) + barrier #1 after slot 38 (the
      buffer is free), the 16 DMA pieces of K-tile t+2 on slots 40, 44, .. 100, vmcnt(pieces so far) + barrier #2 after slot 94
      (K-tile t+1 readable), its k-step-0 reads on odd slots 97..127 -- the same cadence in cycles as KTILE_S0 of variant 18
      (one LDS read per 32 cycles and wave = the LDS pipe's rate with four waves reading, one DMA piece per 64).
MFMA order inside a k-step: shells of the 8x8 block grid (block (mi, ni) belongs to shell max(mi, ni)), so that fragment s of
either operand is first needed at slot s*s and the read order m0 n0 m1 n1 .. delivers them in the order of first use."""


def shell_order():
    o = []
    for s in range(8):
        o += [(s, j) for j in range(s)]          # row s against the columns already open
        o += [(i, s) for i in range(s + 1)]      # column s
    assert len(o) == 64 and len(set(o)) == 64
    return o


ORD = shell_order()
FR = [(t, i) for i in range(8) for t in ("m", "n")]   # request order of a k-step's 16 fragments


def rd(ks, r, buf):
    t, i = FR[r]
    return "LDF(f%s[%d][%d], %s, %s, %s, %d, %d);" % (t, ks, i, "rb" if t == "n" else "ra", "rb_hi" if t == "n" else "ra_hi", buf, ks, i)


def schedule(name, reads, b1, dma, b2, xreads, merged=False, hooks=None, zero_first=False):
    """merged: ONE barrier per K-tile -- at b1 the wave also waits for its DMA pieces of K-tile t+1 (all issued during the previous
    iteration, nothing newer in flight: vmcnt(0)); the same barrier frees the current buffer and publishes K-tile t+1; b2 is None."""
    lines, issued = [], 0
    for j in range(128):
        ks, q = j >> 6, j & 63
        mi, ni = ORD[q]
        parts = ["%s(%d, %d, %d); SB();" % ("MMAZ" if (zero_first and ks == 0) else "MMA", ks, mi, ni)]
        for r in reads.get(j, []):
            parts.append(rd(1, r, "BUF") + " SB();")
        if j == b1 and merged:
            parts.append('asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();')
        elif j == b1:
            parts.append('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();')
        if hooks and j in hooks:
            parts.append("%s; SB();" % hooks[j])
        for p in dma.get(j + 1, []):
            if hooks and p % 4 == 0:   # (round 6d) M0 of a group of four pieces is written ONE SLOT before its first piece: the s_mov's latency hides under an MFMA
                parts.append("if (DMA) dma_m0(BUF, %d); SB();" % p)
        for p in dma.get(j, []):
            parts.append("if (DMA) dma(BUF, (TV) + 2, %d%s); SB();" % (p, ", true" if hooks else ""))
            issued += 1
        if b2 is not None and j == b2:
            parts.append("if (NEXT) { T4_WAIT_NEXT(DMA, %d); __builtin_amdgcn_s_barrier(); } SB();" % issued)
        for r in xreads.get(j, []):
            parts.append("if (NEXT) { " + rd(0, r, "(BUF) ^ 1") + " } SB();")
        lines.append("        " + " ".join(parts))
    assert issued == 16
    w = max(len(x) for x in lines) + 2
    head = "#define %s(BUF, TV, DMA, NEXT)" % name
    out = [head + " " * (w - len(head)) + "\\", "    do {" + " " * (w - 8) + "\\"]
    out += [x + " " * (w - len(x)) + "\\" for x in lines]
    out.append("    } while (0)")
    return "\n".join(out)


T0 = schedule("KTILE_T0", {2 * r: [r] for r in range(16)}, 38, {40 + 4 * p: [p] for p in range(16)}, 94, {97 + 2 * r: [r] for r in range(16)})
# T1: k-step-1 reads on EVERY slot 0..15, everything 8 slots earlier
T1 = schedule("KTILE_T1", {r: [r] for r in range(16)}, 30, {32 + 4 * p: [p] for p in range(16)}, 90, {91 + 2 * r: [r] for r in range(16)})
# T2: barrier #1 8 slots later (reads get 16 slots to land), X' reads on every slot 108..123
T2 = schedule("KTILE_T2", {2 * r: [r] for r in range(16)}, 46, {48 + 4 * p: [p] for p in range(16)}, 106, {108 + r: [r] for r in range(16)})
# T3: one barrier per K-tile (as KTILE_S3 of variant 22), X' reads on odd slots right after it
T3 = schedule("KTILE_T3", {2 * r: [r] for r in range(16)}, 38, {40 + 4 * p: [p] for p in range(16)}, None, {41 + 2 * r: [r] for r in range(16)}, merged=True)


def rd2(half, ks, r, buf):
    t, i = FR[r]
    return "LDF%s(f%s[%d][%d], %s, %s, %s, %d, %d);" % (half, t, ks, i, "rb" if t == "n" else "ra", "rb_hi" if t == "n" else "ra_hi", buf, ks, i)


def schedule_split(name, b1, dma, b2):
    """K-major instantiations (round 3): a transpose-read fragment is TWO LDS instructions (+ one v_xor); in one filler slot they take
    longer to issue than the 16 cycles of the MFMA beside them (measured: +6.5 % on the launch).  Here every fragment read is split over
    two adjacent slots (LDFA = address + first half, LDFB = second half; for a K-contiguous operand LDFA is the whole 128-bit read and
    LDFB nothing): k-step-1 reads fill slots 0..31, the 16 DMA pieces move to slots 40..92 (all issued before barrier #2), the next
    K-tile's k-step-0 reads fill slots 96..127."""
    lines, issued = [], 0
    for j in range(128):
        ks, q = j >> 6, j & 63
        mi, ni = ORD[q]
        parts = ["MMA(%d, %d, %d); SB();" % (ks, mi, ni)]
        if j < 32:
            parts.append(rd2("AB"[j & 1], 1, j >> 1, "BUF") + " SB();")
        if j == b1:
            parts.append('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();')
        if j in (33, 35):   # (round 6d) the scalar source offsets of K-tile t + 2 on two free slots instead of a burst in front of its first piece
            parts.append("T4_SETKX(%d, (TV) + 2); SB();" % (0 if j == 33 else 1))
        import os
        early = os.environ.get("T4_X_M0EARLY", "1") == "1"     # round 6d (product): every piece's M0 written one slot before the piece (no s_nop, the s_mov's latency under an MFMA): -0.5 ... -1 %
        if early:
            for p in dma.get(j + 1, []):
                parts.append("if (DMA) dma_m0x(BUF, %d); SB();" % p)
        for p in dma.get(j, []):
            parts.append("if (DMA) dma(BUF, (TV) + 2, %d%s); SB();" % (p, ", true" if early else ""))
            issued += 1
        if j == b2:
            parts.append("if (NEXT) { T4_WAIT_NEXT(DMA, %d); __builtin_amdgcn_s_barrier(); } SB();" % issued)
        if j >= 96:
            parts.append("if (NEXT) { " + rd2("AB"[j & 1], 0, (j - 96) >> 1, "(BUF) ^ 1") + " } SB();")
        lines.append("        " + " ".join(parts))
    assert issued == 16
    w = max(len(x) for x in lines) + 2
    head = "#define %s(BUF, TV, DMA, NEXT)" % name
    out = [head + " " * (w - len(head)) + "\\", "    do {" + " " * (w - 8) + "\\"]
    out += [x + " " * (w - len(x)) + "\\" for x in lines]
    out.append("    } while (0)")
    return "\n".join(out)


def dma_x(shift):
    import os
    if os.environ.get("T4_X_SPREAD") == "1":     # A/B (round 6d): the K-major schedule's pieces as far apart as the K-contiguous one's: 12 on slots 40 .. 89 (steps 4 / 5), 4 behind barrier #2
        sl = [40, 44, 49, 53, 58, 62, 67, 71, 76, 80, 85, 89, 98, 104, 110, 116]
        return {x + shift: [i] for i, x in enumerate(sl)}
    if os.environ.get("T4_X_SPREAD") == "2":     # ... or all 16 in front of barrier #2, evenly: 40 .. 92 is what there is (steps 3 / 4): the product schedule
        pass
    d, slot = {}, 40 + shift
    for p_ in range(16):
        d[slot] = [p_]
        slot += 3 if p_ % 2 == 0 else 4
    return d


X0 = schedule_split("KTILE_X0", 38, dma_x(0), 94)
import sys
if len(sys.argv) > 1 and sys.argv[1] == "x0":
    print(X0)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "inc":
    # round 6: csrc/gemm_t4_ktile.inc -- the default placement (T0 / X0).  `inc N` adds copies 1..N-1 whose 16 DMA pieces are issued that many
    # MFMA slots later (KTILE_T0_<s>): the per-wave stagger experiment of round 6 (wave w running copy w so that one piece per MFMA slot
    # reaches the CU's vector-memory path instead of four every fourth slot) measured NOTHING (370.7 vs 371.1 us, profiles/r06_xt_ab3.txt)
    # and cost the largest instantiation +15 % (four copies of the K loop: instruction cache); only copy 0 is built.
    ncopies = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("// GENERATED by tools/gen/gemm_t4_schedule.py inc -- do not edit by hand.  KTILE_T0_<s> / KTILE_X0_<s>: s = DMA issue stagger in MFMA slots.")
    for sh in range(ncopies):
        # round 6d: T4_SETK(n), n = 0..2 on the free slots 31 / 33 / 35: the cross-tile form computes the scalar source offsets of K-tile t + 2 in three
        # pieces of <= 4 SALU instructions under MFMAs, instead of 13-19 of them in one burst between two K-tiles (a 40-70 cycle hole in the matrix pipe)
        import os
        dstep = int(os.environ.get("T4_DMA_STEP", "4").rstrip("e"))    # A/B: MFMA slots between two DMA pieces (4 = the product schedule: slots 40, 44 .. 100)
        if os.environ.get("T4_EARLY_B1") == "1":           # A/B: k-step-1 reads on every slot 0..15, barrier #1 at slot 20, pieces every 6th slot 22 .. 112
            print(schedule("KTILE_T0_%d" % sh, {r: [r] for r in range(16)}, 20, {22 + 6 * p + sh: [p] for p in range(16)}, 94, {97 + 2 * r: [r] for r in range(16)},
                           hooks={16: "T4_SETK(0)", 17: "T4_SETK(1)", 18: "T4_SETK(2)"}))
        elif os.environ.get("T4_DMA_STEP", "5e") == "5e":    # THE PRODUCT SCHEDULE since round 6d: pieces 0-10 every 5th slot 40 .. 90, pieces 11-15 on the even slots
                                                             # 96 .. 112 (the odd ones carry the reads).  Until then: every 4th slot 40 .. 100 (T4_DMA_STEP=4).  Measured on cold
                                                             # operands (profiles/r06d_dma_step_ab.txt): every 2nd slot +7 %, every 3rd +2.5 %, every 5th -0.5 ... -2 % against every 4th
            print(schedule("KTILE_T0_%d" % sh, {2 * r: [r] for r in range(16)}, 38, {(40 + 5 * p if p < 11 else 96 + 4 * (p - 11)) + sh: [p] for p in range(16)}, 94,
                           {97 + 2 * r: [r] for r in range(16)}, hooks={31: "T4_SETK(0)", 33: "T4_SETK(1)", 35: "T4_SETK(2)"}))
            # the same K-tile as the FIRST of a tile in the cross-tile form (its peeled iteration 0): the 64 MFMAs of k-step 0 take a zero C operand (MMAZ),
            # so the 256 v_accvgpr_write that zeroed the accumulators between two tiles (~1.2 k cycles with the matrix pipe idle) are gone
            print(schedule("KTILE_T0F_%d" % sh, {2 * r: [r] for r in range(16)}, 38, {(40 + 5 * p if p < 11 else 96 + 4 * (p - 11)) + sh: [p] for p in range(16)}, 94,
                           {97 + 2 * r: [r] for r in range(16)}, hooks={31: "T4_SETK(0)", 33: "T4_SETK(1)", 35: "T4_SETK(2)"}, zero_first=True))
        else:
            print(schedule("KTILE_T0_%d" % sh, {2 * r: [r] for r in range(16)}, 38, {40 + dstep * p + sh: [p] for p in range(16)}, 94, {97 + 2 * r: [r] for r in range(16)},
                           hooks={31: "T4_SETK(0)", 33: "T4_SETK(1)", 35: "T4_SETK(2)"}))
        print(schedule_split("KTILE_X0_%d" % sh, 38, dma_x(sh), 94))
    sys.exit(0)
print(T0)
print(T1)
print(T2)
print(T3)
