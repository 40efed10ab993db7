"""Brute-force the LDS image of the halo convolution kernels (conv_halo.hip): pixel pitch (in 16-byte slots) and an
optional XOR of the chunk index, scored with the ds_read_b128 banking rule of gfx950 (four 16-lane groups, 64 banks
x 4 B: a group is conflict-free when its 16 lanes hit 16 distinct 16-byte slots mod 256 B)."""
import itertools
import sys

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def cost(addr_of_lane):
    tot = 0
    for g in GROUPS:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            banks.setdefault(a % 16, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot          # 4 = conflict-free


def score(IH, IW, CI, KS, S, F, pitch, fpitch, swz):
    OH = (IH - KS) // S + 1
    OW = (IW - KS) // S + 1
    CH = CI // 8
    M = F * OH * OW
    tot = n = 0
    for t in range((M + 15) // 16):
        for tap in range(KS * KS):
            ky, kx = divmod(tap, KS)
            for half in range(CI // 32):
                def addr(l):
                    r, g = l & 15, l >> 4
                    m = min(t * 16 + r, M - 1)
                    f, rem = divmod(m, OH * OW)
                    oy, ox = divmod(rem, OW)
                    p = (S * oy + ky) * IW + S * ox + kx
                    c = half * 4 + g
                    return f * fpitch + p * pitch + (c ^ swz(p, CH))
                tot += cost(addr)
                n += 1
    return tot / n / 4.0


SWZ = {
    "none": lambda p, ch: 0,
    "p": lambda p, ch: p & (ch - 1),
    "p>>1": lambda p, ch: (p >> 1) & (ch - 1),
    "p>>2": lambda p, ch: (p >> 2) & (ch - 1),
}

if __name__ == "__main__":
    for name, (IH, IW, CI, KS, S, F) in {"conv2": (20, 20, 32, 4, 2, 1), "conv3": (9, 9, 64, 3, 1, 4),
                                         "dgrad3": (11, 11, 64, 3, 1, 4), "dgrad2": (11, 11, 64, 2, 1, 1)}.items():
        CH = CI // 8
        res = []
        for pitch in range(CH, CH + 6):
            for sname, sw in SWZ.items():
                if pitch == CH or sname == "none" or True:
                    fp = IH * IW * pitch
                    for fpad in (0, 1, 2, 3, 4, 5, 6, 7, 8):
                        res.append((score(IH, IW, CI, KS, S, F, pitch, fp + fpad, sw), pitch, sname, fpad))
        res.sort()
        print(name, res[:6])
