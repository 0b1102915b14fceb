"""oracle/anchor_oracle.py -- TEST INFRASTRUCTURE ONLY.

Restatement of the reference's anchor stage for SMALL inputs: the border nodes it prepares
(alignment.c:69-86, morenodeslinkedlists.c:259-330 and :547-620) written from their definition
(no suffix tree), and the anchor loop that turns them into the alignment map
(alignment.c:163-214, alignmentmap.c:9-31/:47-146/:259-316, morenodeslinkedlists.c:89-146,
:375-534), restated step by step because its results depend on the exact order of its list
operations.  It is pinned against oracle/_ref (ref_shim.c: csa_ref_alignment_map) in
tests/test_anchors.py before the product implements the same stage with suffix automata.

Definition used for the border nodes (after MarkUsedNodes/DeleteUnusedNodes the tree holds
exactly the suffixes of the ROTATED, LINEAR sequences): every suffix start p of sequence s is
credited to the string w = the longest prefix of T_s[p:] that occurs in every sequence; one
border node per distinct non-empty w, holding per sequence the ascending list of such p; nodes
lacking a sequence are dropped (morenodeslinkedlists.c:325-328).
"""

INT_MAX = 2 ** 31 - 1


def rotate(seq, rot):
    return seq[rot:] + seq[:rot]


def normalise(seq):
    """The tree compares every non-ACGT byte as one and the same symbol (gencycsuffixtrees.c:320)."""
    return bytes(c if c in b"ACGT" else ord("-") for c in seq)


def leaf_collision(rseqs):
    """True when a proper suffix of one rotated sequence is a whole rotation of another one.  The
    reference's suffix walk (morenodeslinkedlists.c:590-617) then leaves the sequence and wanders
    through the other sequence's rotation leaves: positions past the end of the text appear, the
    result depends on the processing order and the walk need not end.  Out of contract."""
    for j, T in enumerate(rseqs):
        for i, U in enumerate(rseqs):
            if i != j and len(U) < len(T) and T[len(T) - len(U):] in (U + U):
                return True
    return False


def border_nodes(rseqs):
    nodes = {}
    n = len(rseqs)
    for s, T in enumerate(rseqs):
        for p in range(len(T)):
            L = len(T) - p
            for t, U in enumerate(rseqs):
                if t == s or L == 0:
                    continue
                lo, hi = 0, L
                while lo < hi:
                    mid = (lo + hi + 1) // 2
                    if T[p:p + mid] in U:
                        lo = mid
                    else:
                        hi = mid - 1
                L = lo
            if L == 0:
                continue                      # credited to the root = the list's sentinel (alignment.c:47)
            nodes.setdefault(T[p:p + L], [[] for _ in range(n)])[s].append(p)
    out = [(len(w), pos) for w, pos in nodes.items() if all(pos)]
    out.sort(key=lambda b: b[1][0][0])
    return out


class BNode:
    __slots__ = ("size", "pos", "act", "hidden", "hiddennode", "next", "prev")

    def __init__(self, size, pos):
        self.size = size
        self.pos = pos
        self.act = [0] * len(pos)
        self.hidden = False
        self.hiddennode = None
        self.next = None
        self.prev = None

    @property
    def k0(self):
        return self.pos[0][0]


class Item:
    __slots__ = ("positions", "size", "weight", "backtrack", "next", "prev")


class Seg:
    __slots__ = ("positions", "size", "mingap", "maxgap", "dp", "next")

    def __init__(self, positions, size):
        self.positions = positions
        self.size = size
        self.mingap = INT_MAX
        self.maxgap = INT_MAX
        self.dp = 0
        self.next = None


class AnchorLoop:
    def __init__(self, sizes, border):
        self.n = len(sizes)
        self.sizes = list(sizes)
        self.first = BNode(0, [[-1] for _ in sizes])          # alignment.c:47-54
        tail = self.first
        for size, pos in border:
            b = BNode(size, [list(p) for p in pos])
            tail.next = b
            b.prev = tail
            tail = b
        self.firstseg = Seg([-1] * self.n, 1)                 # alignment.c:57-64
        self.lastseg = Seg(list(sizes), 0)
        self.firstseg.next = self.lastseg
        self.gaps(self.firstseg)
        self.start = [0] * self.n
        self.end = [0] * self.n
        self.chain = None

    # -- alignmentmap.c:239-256
    def gaps(self, seg):
        lo, hi = INT_MAX, -INT_MAX - 1
        for i in range(self.n):
            g = seg.next.positions[i] - (seg.positions[i] + seg.size)
            if g < 0:
                g += self.sizes[i]
            lo = min(lo, g)
            hi = max(hi, g)
        seg.mingap, seg.maxgap = lo, hi

    # -- morenodeslinkedlists.c:31-71 (list part)
    def delete(self, b):
        if b.prev is not None:
            b.prev.next = b.next
        else:
            self.first = b.next
        if b.next is not None:
            b.next.prev = b.prev

    # -- morenodeslinkedlists.c:106-128
    def hide(self, b):
        if b.hidden:
            return
        st = b.prev
        st.next = b.next
        if b.next is not None:
            b.next.prev = st
        b.next = None
        b.prev = st.hiddennode
        if st.hiddennode is not None:
            st.hiddennode.next = b
        st.hiddennode = b
        b.hidden = True

    # -- morenodeslinkedlists.c:131-146
    def unhide(self, node):
        aux = node.hiddennode
        if aux is None:
            return
        aux.hidden = False
        aux.next = node.next
        if node.next is not None:
            node.next.prev = aux
        while aux.prev is not None:
            aux = aux.prev
            aux.hidden = False
        aux.prev = node
        node.next = aux
        node.hiddennode = None

    # -- morenodeslinkedlists.c:398-443
    def sort(self):
        e0 = self.end[0]
        c = self.first.next
        while c is not None and c.k0 < e0:
            prev = c.prev
            if c.k0 < prev.k0:
                back = c.prev
                while back is not None and back.k0 > c.k0:
                    back = back.prev
                following = back.next
                back.next = c
                c.prev = back
                fwd = c
                while fwd.next is not None and fwd.next.k0 > fwd.k0 and fwd.next.k0 < following.k0:
                    fwd = fwd.next
                nxt = fwd.next
                fwd.next = following
                following.prev = fwd
                prev.next = nxt
                if nxt is not None:
                    nxt.prev = prev
            else:
                nxt = c.next
            c = nxt

    # -- morenodeslinkedlists.c:446-462
    def resort(self, node):
        if node.next is None or node.next.k0 > node.k0:
            return
        cur = node.next
        while cur.next is not None and cur.next.k0 < node.k0:
            cur = cur.next
        prevn, nextn = node.prev, node.next
        if prevn is not None:
            prevn.next = nextn
        if nextn is not None:
            nextn.prev = prevn
        nextn = cur.next
        cur.next = node
        node.prev = cur
        if nextn is not None:
            nextn.prev = node
        node.next = nextn

    # -- morenodeslinkedlists.c:465-534.  The call that should restore the positions hidden by
    # the chaining step returns immediately (its guard tests hiddennode, already cleared two
    # lines above, :175), so hidden positions never come back: they are simply dropped here.
    def update_active(self):
        e0 = self.end[0]
        b = self.first.next
        while b is not None and b.k0 < e0:
            if b.hiddennode is not None:
                self.unhide(b)
            nxt = b.next
            for i in range(self.n):
                p = b.pos[i]
                while p and p[0] < self.start[i]:
                    p.pop(0)
                    b.act[i] -= 1
                if not p:
                    self.delete(b)
                    break
            b = nxt
        self.sort()
        active = 0
        b = self.first.next
        while b is not None and b.k0 < e0:
            active += 1
            ok = True
            for i in range(self.n):
                c = 0
                for k in b.pos[i]:
                    if k < self.end[i]:
                        c += 1
                    else:
                        break
                if c == 0:
                    ok = False
                    break
                b.act[i] = c
            nxt = b.next
            if not ok:
                self.hide(b)
                active -= 1
                b = nxt
                continue
            for i in range(1, self.n):
                if b.act[i] != b.act[0]:
                    self.hide(b)
                    active -= 1
                    break
            b = nxt
        return active

    # -- alignmentmap.c:9-31
    def new_item(self, b):
        it = Item()
        it.positions = [0] * self.n
        size = b.size
        for i in range(self.n):
            pos = b.pos[i][0]
            it.positions[i] = pos
            if pos + b.size >= self.end[i]:
                aux = self.end[i] - pos
                if aux < size:
                    size = aux
        it.size = size
        it.weight = size
        it.backtrack = None
        it.next = None
        it.prev = None
        return it

    @staticmethod
    def greater(a, b):
        return all(pa >= pb + b.size for pa, pb in zip(a.positions, b.positions))

    # -- alignmentmap.c:70-105
    def heaviest_chain(self):
        self.chain = None
        b = self.first.next
        while b is not None and b.k0 < self.end[0]:
            new = self.new_item(b)
            cur = None
            nxt = self.chain
            while nxt is not None and not self.greater(new, nxt):
                cur = nxt
                nxt = cur.next
            if nxt is not None:
                new.weight += nxt.weight
                new.backtrack = nxt
            prev = cur
            cur = nxt
            while prev is not None and new.weight >= prev.weight:
                cur = prev
                prev = cur.prev
            if prev is None:
                self.chain = new
            else:
                prev.next = new
            new.prev = prev
            if cur is not None:
                cur.prev = new
            new.next = cur
            nextnode = b.next
            if b.act[0] > 1:
                for i in range(self.n):                       # HideFirstPositions (:151-172)
                    b.pos[i].pop(0)
                    b.act[i] -= 1
                self.resort(b)
                if b.next is nextnode:
                    nextnode = b
            b = nextnode

    # -- alignmentmap.c:259-316
    def set_segments(self, startseg, endseg):
        cur = endseg
        item = self.chain
        count = 0
        while item is not None:
            new = Seg(item.positions, item.size)
            new.next = cur
            self.gaps(new)
            total = 0
            for i in range(self.n):
                g = cur.positions[i] - (new.positions[i] + new.size)
                if g < 0:
                    g += self.sizes[i]
                total += g
            lo, hi = new.mingap, new.maxgap
            avgmin = c_div(total - lo, self.n - 1)
            avgmax = c_div(total - hi, self.n - 1)
            if lo < c_div(avgmin, 2) or hi > c_div(avgmax * 3, 2):
                pass
            else:
                cur = new
                count += 1
            item = item.backtrack
        startseg.next = cur
        self.gaps(startseg)
        self.chain = None
        return count

    # -- alignment.c:163-214
    def run(self):
        startseg = self.firstseg
        while startseg is not self.lastseg:
            endseg = startseg.next
            if startseg.mingap == 0:
                startseg = startseg.next
                continue
            for i in range(self.n):
                self.start[i] = startseg.positions[i] + startseg.size
                self.end[i] = endseg.positions[i]
            count = self.update_active()
            if count > 0:
                self.heaviest_chain()
                count = self.set_segments(startseg, endseg)
            if count == 0:
                startseg.dp = 1
                startseg = startseg.next
        out = []
        seg = self.firstseg
        while seg is not None:
            out.append((seg.size, seg.dp, list(seg.positions)))
            seg = seg.next
        return out


def c_div(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def alignment_map(seqs, rotations):
    """seqs: un-rotated circular sequences; returns (border nodes, segments) like
    tests/helpers.py:ref_alignment_map."""
    rseqs = [normalise(rotate(s, r)) for s, r in zip(seqs, rotations)]
    if leaf_collision(rseqs):
        raise ValueError("a suffix of one sequence is a whole rotation of another (reference walk undefined)")
    border = border_nodes(rseqs)
    segs = AnchorLoop([len(s) for s in seqs], border).run()
    return border, segs
