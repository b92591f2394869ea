"""Generates the 64-slot K-tile schedule macro of gemm_bf16_r4_kernel (csrc/gemm.hip, variant 18).  The output is pasted
into gemm.hip between the GENERATED markers: `python tools/gen/gemm_r4_schedule.py` prints it.

One K-tile (BK = 64) of a 128x128 wave tile = 4 k-steps x 16 MFMAs (32x32x16).  Slot j = MFMA j, then at most ONE filler:
  j  0..15  reads of k-step 1 (even j) and k-step 2 (odd j) fragments        (k-step 0 was read by the previous iteration)
  j 16..23  reads of k-step 3 fragments
  j 27      lgkmcnt(0) + barrier #1: every wave holds the whole K-tile in registers -> the buffer is free
  j 28..58  (even) the 16 DMA pieces of K-tile t+2 into the buffer just freed
  j 48      vmcnt(pieces issued so far) + barrier #2: K-tile t+1 (issued one iteration ago) is readable
  j 49..63  (odd) reads of k-step 0 of K-tile t+1 from the other buffer
"""
ORD = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1), (0, 2), (1, 2), (2, 2), (3, 2), (0, 3), (1, 3), (2, 3), (3, 3)]
FR = [("n", 0), ("m", 0), ("m", 1), ("n", 1), ("m", 2), ("m", 3), ("n", 2), ("n", 3)]   # request order of a k-step's fragments


def rd(ks, r, buf):
    t, i = FR[r]
    return "LDF(f%s[%d][%d], %s, %s, %s, %d, %d);" % (t, ks, i, "rb" if t == "n" else "ra", "rb_hi" if t == "n" else "ra_hi", buf, ks, i)


lines = []
for j in range(64):
    ks, q = j >> 4, j & 15
    mi, ni = ORD[q]
    parts = ["MMA(%d, %d, %d); SB();" % (ks, mi, ni)]
    if j < 16:
        parts.append(rd(1 if j % 2 == 0 else 2, j // 2, "BUF") + " SB();")
    elif j < 24:
        parts.append(rd(3, j - 16, "BUF") + " SB();")
    if j == 27:
        parts.append('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();')
    if 28 <= j <= 58 and j % 2 == 0:
        parts.append("if (DMA) dma(BUF, (TV) + 2, %d); SB();" % ((j - 28) // 2))
    if j == 48:
        parts.append('if (NEXT) { if (DMA) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); '
                     "__builtin_amdgcn_s_barrier(); } SB();")
    if j >= 49 and j % 2 == 1:
        parts.append("if (NEXT) { " + rd(0, (j - 49) // 2, "(BUF) ^ 1") + " } SB();")
    lines.append("        " + " ".join(parts))
w = max(len(x) for x in lines) + 2
print("#define KTILE(BUF, TV, DMA, NEXT)" + " " * (w - 33) + "\\")
print("    do {" + " " * (w - 8) + "\\")
for x in lines:
    print(x + " " * (w - len(x)) + "\\")
print("    } while (0)")
