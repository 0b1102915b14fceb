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
