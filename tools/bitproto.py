# prototype of the bit-parallel row update for scores match +1 / mismatch -1 / gap -1, tie-break D >= L >= U
import random, sys
def plain(a, b):
    m, n = len(a), len(b)
    H = [[0]*(n+1) for _ in range(m+1)]
    D = [[0]*(n+1) for _ in range(m+1)]
    for j in range(n+1): H[0][j] = -j
    for i in range(m+1): H[i][0] = -i
    for i in range(1, m+1):
        for j in range(1, n+1):
            d = H[i-1][j-1] + (1 if a[i-1] == b[j-1] else -1)
            l = H[i][j-1] - 1
            u = H[i-1][j] - 1
            h = max(d, l, u)
            H[i][j] = h
            D[i][j] = 2 if h == d else (1 if h == l else 0)
    return H, D

def bits(a, b):
    m, n = len(a), len(b)
    mask = (1 << n) - 1
    eq = {c: sum(1 << j for j in range(n) if b[j] == c) for c in "ACGT"}
    H0 = H1 = H2 = 0          # thermometer planes of dH from above: w>=0, w>=1, w>=2 ; row 0: w = -1
    out = []
    for i in range(m):
        E = eq[a[i]]; nE = ~E & mask
        Wm1 = ~H0 & mask
        W0 = H0 & ~H1
        W1 = H1 & ~H2
        # chain for u>=2 : generate E&Wm1, propagate nE&Wm1 ; carry-in 0 (u_0 = -1)
        g2 = E & Wm1
        s2 = Wm1 + g2
        G2in = (s2 ^ Wm1 ^ g2) & mask                    # bit j: u_j >= 2 (incoming from the left)
        g1 = (E & ~H1) | (nE & W0 & G2in)
        p1 = nE & Wm1
        A1 = g1 | p1
        s1 = A1 + g1
        G1in = (s1 ^ A1 ^ g1) & mask
        G0out = (E & ~H2) | (nE & (Wm1 | (W0 & G1in) | (W1 & G2in)))
        G0out &= mask
        G0in = (G0out << 1) & mask                       # carry-in 0
        C1 = E | G2in | H2
        C0 = E | G1in | H1
        T2 = C1 & ~G0in
        T1 = (C1 & ~G1in) | (C0 & ~G0in)
        T0 = (C1 & ~G2in) | (C0 & ~G1in) | (~G0in & mask)
        Dm = (E | ~C0) & mask
        Lm = ~Dm & ~T0 & mask
        out.append((Dm, Lm))
        H2, H1, H0 = T2 & mask, T1 & mask, T0 & mask
    return out

random.seed(1)
for it in range(300):
    m = random.randint(1, 70); n = random.randint(1, 90)
    if it % 3 == 0:
        a = "".join(random.choice("ACGT") for _ in range(m)); b = "".join(random.choice("ACGT") for _ in range(n))
    elif it % 3 == 1:
        b = "".join(random.choice("ACGT") for _ in range(n)); a = "".join(c if random.random() > 0.1 else random.choice("ACGT") for c in b)[:m] or "A"
        m = len(a)
    else:
        a = "".join(random.choice("AC") for _ in range(m)); b = "".join(random.choice("AC") for _ in range(n))
    H, D = plain(a, b)
    out = bits(a, b)
    for i in range(1, m+1):
        Dm, Lm = out[i-1]
        for j in range(1, n+1):
            d = 2 if (Dm >> (j-1)) & 1 else (1 if (Lm >> (j-1)) & 1 else 0)
            if d != D[i][j]:
                print("MISMATCH", it, i, j, d, D[i][j]); sys.exit(1)
print("all ok")


# ---- word-level emulation of the kernel's step (32-bit words, packed hand-off word) ----------
M32 = 0xffffffff
A_, B_, C_ = 0xF0, 0xCC, 0xAA


def lut(f):
    return f(A_, B_, C_) & 0xff


def bitop3(a, b, c, t):
    r = 0
    for idx in range(8):
        if (t >> idx) & 1:
            ma = a if idx & 4 else ~a
            mb = b if idx & 2 else ~b
            mc = c if idx & 1 else ~c
            r |= ma & mb & mc
    return r & M32


def perm(s0, s1, sel):
    src = [(s1 >> (8 * i)) & 0xff for i in range(4)] + [(s0 >> (8 * i)) & 0xff for i in range(4)]
    out = 0
    for i in range(4):
        sb = (sel >> (8 * i)) & 0xff
        out |= (src[sb] if sb < 8 else 0) << (8 * i)
    return out


def sbfe(x, off):
    return M32 if (x >> off) & 1 else 0


def kernel_words(a, b):
    """rows of a against columns of b, 32 columns per word, returns per row (notdiag, left) bit masks"""
    m, n = len(a), len(b)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    nw = (n + 31) // 32
    B0 = [0] * nw
    B1 = [0] * nw
    for j, ch in enumerate(b):
        B0[j >> 5] |= (code[ch] & 1) << (j & 31)
        B1[j >> 5] |= (code[ch] >> 1) << (j & 31)
    nH0 = [M32] * nw
    H1 = [0] * nw
    H2 = [0] * nw
    out = []
    for i in range(m):
        PP = code[a[i]]                                       # word entering lane 0: no carries, the row letter
        nd_row = lf_row = 0
        for w in range(nw):
            PPin = PP
            R0, R1 = sbfe(PPin, 0), sbfe(PPin, 1)
            c2, c1 = (PPin >> 15) & 1, (PPin >> 23) & 1
            x0 = B0[w] ^ R0
            nE = bitop3(x0, B1[w], R1, lut(lambda a, b, c: a | (b ^ c)))
            g2 = bitop3(nE, nH0[w], 0, lut(lambda a, b, c: ~a & b))
            s2 = (nH0[w] + g2 + c2) & M32
            G2 = bitop3(s2, nH0[w], g2, lut(lambda a, b, c: a ^ b ^ c))
            O2 = bitop3(g2, nH0[w], G2, lut(lambda a, b, c: a | (b & c)))
            t1 = bitop3(nE, nH0[w], G2, lut(lambda a, b, c: ~a | (~b & c)))
            g1 = bitop3(t1, H1[w], 0, lut(lambda a, b, c: a & ~b))
            A1 = bitop3(g1, nE, nH0[w], lut(lambda a, b, c: a | (b & c)))
            s1 = (A1 + g1 + c1) & M32
            G1 = bitop3(s1, A1, g1, lut(lambda a, b, c: a ^ b ^ c))
            O1 = bitop3(g1, A1, G1, lut(lambda a, b, c: a | (b & c)))
            v = bitop3(H1[w], G2, G1, lut(lambda a, b, c: (a & b) | (~a & c)))
            ww = bitop3(nE, v, H2[w], lut(lambda a, b, c: ~c & (~a | b)))
            O0 = bitop3(ww, nE, nH0[w], lut(lambda a, b, c: a | (b & c)))
            G0 = ((O0 << 1) | (PPin >> 31)) & M32
            C1 = bitop3(nE, G2, H2[w], lut(lambda a, b, c: ~a | b | c))
            C0 = bitop3(nE, G1, H1[w], lut(lambda a, b, c: ~a | b | c))
            T2 = bitop3(C1, G0, 0, lut(lambda a, b, c: a & ~b))
            a1 = bitop3(C1, G1, 0, lut(lambda a, b, c: a & ~b))
            T1 = bitop3(G0, a1, C0, lut(lambda a, b, c: (a & b) | (~a & c)))
            b0 = bitop3(C0, G1, G0, lut(lambda a, b, c: c & (~a | b)))
            nT0 = bitop3(b0, C1, G2, lut(lambda a, b, c: a & (~b | c)))
            nd = C0 & nE
            lf = nd & nT0
            nd_row |= nd << (32 * w)
            lf_row |= lf << (32 * w)
            Tq = perm(O1, O2, 0x0c07030c)
            Pq = perm(O0, Tq, 0x0702010c)
            PP = bitop3(Pq, PPin, 0xff, lut(lambda a, b, c: a | (b & c)))
            nH0[w], H1[w], H2[w] = nT0, T1, T2
        out.append((nd_row, lf_row))
    return out


if __name__ == "__main__":
    random.seed(2)
    for it in range(200):
        m = random.randint(1, 60)
        n = random.randint(1, 200)
        if it % 2:
            b = "".join(random.choice("ACGT") for _ in range(n))
            a = ("".join(c if random.random() > 0.15 else random.choice("ACGT") for c in b)[:m]) or "A"
            m = len(a)
        else:
            a = "".join(random.choice("AC") for _ in range(m))
            b = "".join(random.choice("AC") for _ in range(n))
        H, D = plain(a, b)
        out = kernel_words(a, b)
        for i in range(1, m + 1):
            nd, lf = out[i - 1]
            for j in range(1, n + 1):
                d = (1 if (lf >> (j - 1)) & 1 else 0) if (nd >> (j - 1)) & 1 else 2
                assert d == D[i][j], ("word-level", it, i, j, d, D[i][j])
    print("word-level emulation ok")
