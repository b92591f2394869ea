"""Generates the 16-slot K-tile schedule macro of gemm_bf16_s4_kernel (csrc/gemm.hip, variant 25): 128x128 tile, 4 waves of 64x64
(2x2 blocks of 32x32), BK = 64 stages in a 4-deep LDS ring.  One K-tile = 4 k-steps x 4 MFMAs.  Slot j = MFMA j, then its fillers:
every slot requests one fragment of the NEXT k-step (k-step 0 of the next stage during the last k-step), odd slots also issue one of
the 8 DMA pieces of stage s+3.  `python tools/gen/gemm_s4_schedule.py` prints the macro pasted between the GENERATED markers."""
ORD = [(0, 0), (1, 0), (0, 1), (1, 1)]            # (mi, ni) order inside a k-step
FR = [("n", 0), ("m", 0), ("m", 1), ("n", 1)]     # request order of a k-step's four fragments

lines = []
piece = 0
for j in range(16):
    ks, q = j >> 2, j & 3
    mi, ni = ORD[q]
    parts = ["MMA(%d, %d, %d); SB();" % (ks, mi, ni)]
    t, i = FR[q]
    if ks < 3:
        parts.append("LDF(f%s[%d][%d], %s, S, %d, %d); SB();" % (t, ks + 1, i, "rb" if t == "n" else "ra", ks + 1, i))
    else:
        parts.append("if (NEXT) { LDF(f%s[0][%d], %s, SN, 0, %d); } SB();" % (t, i, "rb" if t == "n" else "ra", i))
    if j % 2 == 1:
        parts.append("if (DMA) dma(SD, (STEPV) + 3, %d); SB();" % piece)
        piece += 1
    lines.append("        " + " ".join(parts))
assert piece == 8
w = max(len(x) for x in lines) + 2
head = "#define KSTEP(S, STEPV, DMA, NEXT, VMW)"
print(head + " " * (w - len(head)) + "\\")
print("    do {" + " " * (w - 8) + "\\")
print("        constexpr int SN = ((S) + 1) & 3, SD = ((S) + 3) & 3;" + " " * (w - 63) + "\\")
for x in lines:
    print(x + " " * (w - len(x)) + "\\")
tail = '        asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();'
print(tail + " " * (w - len(tail)) + "\\")
print("    } while (0)")
