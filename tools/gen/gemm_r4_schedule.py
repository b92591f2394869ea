"""Generates the 64-slot K-tile schedule macros of gemm_bf16_r4_kernel (csrc/gemm.hip, variants 18-20).  The output is
pasted into gemm.hip between the GENERATED markers: `python tools/gen/gemm_r4_schedule.py` prints it.

One K-tile (BK = 64) of a 128x128 wave tile = 4 k-steps x 16 MFMAs (32x32x16).  Slot j = MFMA j followed by its fillers.
Every schedule: reads of k-steps 1..3 early (k-step 0 was read by the previous iteration), then lgkmcnt(0) + barrier #1
(every wave holds the whole K-tile in registers -> the buffer is free), the 16 DMA pieces of K-tile t+2 into that buffer,
vmcnt(pieces issued so far) + barrier #2 (K-tile t+1, issued one iteration ago, is readable), reads of k-step 0 of K-tile
t+1 from the other buffer.
  S0: 1 read/slot in 0..23, B1 after 27, DMA on even slots 28..58, B2 after 48, X' reads on odd slots 49..63
  S1: 2 reads/slot in 0..11, B1 after 19, DMA on even slots 20..50, B2 after 42, X' reads on odd slots 43..57
      (buffer freed 8 slots earlier, every piece gets 8 more slots to land, the k-step-0 reads are 6 slots further ahead)
  S2: 1 read/slot in 0..23, B1 after 31, DMA on every slot 32..47, B2 after 50, X' reads 2/slot in 51..54
      (DMA issue compressed into one k-step, reads and DMA issue never overlap)
  S3 / S4: ONE barrier per K-tile (see `merged` below)
"""
ORD = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1), (0, 2), (1, 2), (2, 2), (3, 2), (0, 3), (1, 3), (2, 3), (3, 3)]
FR = [("n", 0), ("m", 0), ("m", 1), ("n", 1), ("m", 2), ("m", 3), ("n", 2), ("n", 3)]   # request order of a k-step's fragments


def rd(ks, r, buf):
    t, i = FR[r]
    return "LDF(f%s[%d][%d], %s, %s, %s, %d, %d);" % (t, ks, i, "rb" if t == "n" else "ra", "rb_hi" if t == "n" else "ra_hi", buf, ks, i)


def schedule(name, reads, b1, dma, b2, xreads, merged=False):
    """reads: {slot: [(ks, r), ...]}; dma: {slot: [piece, ...]}; xreads: {slot: [r, ...]}.  merged: ONE barrier per K-tile -- at b1 the
    wave also waits for its DMA pieces of K-tile t+1 (all issued during the previous iteration; nothing newer is in flight yet, hence
    vmcnt(0)), so the same barrier frees the current buffer AND publishes K-tile t+1; b2 is None."""
    lines = []
    issued = 0
    for j in range(64):
        ks, q = j >> 4, j & 15
        mi, ni = ORD[q]
        parts = ["MMA(%d, %d, %d); SB();" % (ks, mi, ni)]
        for (rks, r) in reads.get(j, []):
            parts.append(rd(rks, r, "BUF") + " SB();")
        if j == b1 and merged:
            parts.append('asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();')
        elif j == b1:
            parts.append('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();')
        for p in dma.get(j, []):
            parts.append("if (DMA) dma(BUF, (TV) + 2, %d); SB();" % p)
            issued += 1
        if b2 is not None and j == b2:
            parts.append('if (NEXT) { if (DMA) asm volatile("s_waitcnt vmcnt(%d)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); '
                         "__builtin_amdgcn_s_barrier(); } SB();" % issued)
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


def all_reads_1():
    d = {}
    for j in range(16):
        d[j] = [(1 if j % 2 == 0 else 2, j // 2)]
    for j in range(16, 24):
        d[j] = [(3, j - 16)]
    return d


def all_reads_2():
    seq = []
    for r in range(8):
        seq.append((1, r))
    for r in range(8):
        seq.append((2, r))
    for r in range(8):
        seq.append((3, r))
    return {j: [seq[2 * j], seq[2 * j + 1]] for j in range(12)}


S0 = schedule("KTILE_S0", all_reads_1(), 27, {j: [(j - 28) // 2] for j in range(28, 60, 2)}, 48, {j: [(j - 49) // 2] for j in range(49, 64, 2)})
S1 = schedule("KTILE_S1", all_reads_2(), 19, {j: [(j - 20) // 2] for j in range(20, 52, 2)}, 42, {j: [(j - 43) // 2] for j in range(43, 58, 2)})
S2 = schedule("KTILE_S2", all_reads_1(), 31, {j: [j - 32] for j in range(32, 48)}, 50, {51 + k: [2 * k, 2 * k + 1] for k in range(4)})
# merged-barrier schedules (variants 22, 23)
#  S3: reads as S0, ONE barrier after 27, DMA on even slots 28..58, X' reads on odd slots 29..43 (K-tile t+1 is visible right away)
#  S4: same, DMA on every slot 28..43 (the last piece gets 48 slots = 1500+ cycles to land instead of 33)
S3 = schedule("KTILE_S3", all_reads_1(), 27, {j: [(j - 28) // 2] for j in range(28, 60, 2)}, None, {j: [(j - 29) // 2] for j in range(29, 44, 2)}, merged=True)
S4 = schedule("KTILE_S4", all_reads_1(), 27, {j: [j - 28] for j in range(28, 44)}, None, {j: [j - 44] for j in range(44, 52)}, merged=True)
print(S0)
print(S1)
print(S2)
print(S3)
print(S4)
