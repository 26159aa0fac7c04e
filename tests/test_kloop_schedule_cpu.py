"""The synchronisation argument of gemm_nt's 4-phase K loop (csrc/gemm_nt.hip, PIPE 4), checked by simulation on the CPU.

The loop keeps LDS-DMA half-tiles in flight across barriers behind COUNTED ``s_waitcnt vmcnt(n)`` waits; nothing but the issuing wave's wait
plus a barrier orders a ``ds_read`` behind a DMA, and nothing but retired reads plus a barrier orders a DMA behind a ``ds_read``.  The two
wave-rows run one barrier apart, so "before" has to hold for the slower row as well.  This file replays the per-wave instruction order of the
kernel (same issue points, same counts) for both wave-rows and asserts, for every contraction length:

* RAW -- a half-tile is read in barrier interval g only if EVERY wave-row retired its own share of that half-tile's DMA by a wait it executed
  in an interval < g (i.e. before a barrier that precedes the read);
* WAR -- a half-tile buffer is re-staged in interval g only if every wave-row's reads of the previous occupant retired (``lgkmcnt(0)`` right
  behind the barrier that ends the reading interval) at least one full interval earlier;
* every read sees the K-tile it expects.

The 8-phase loop (kept for tile-blocked weights and operands beyond 2^31 elements) is replayed too: with the wait where rounds 2-4 had it -- at
the end of P4's MFMA cluster -- the checker flags a one-barrier-short wait for the lagging wave-row (the finding that put the new loop's waits into
its read segments, docs/LAB_r01-r05.md section 4.1e); with the wait in front of P4's first barrier (round 5) it is ordered as well."""
from collections import deque

import pytest

HALVES = ("A0", "A1", "B0", "B1")


class Row:
    """One wave-row (its four waves execute the same stream): outstanding DMA queue with in-order retirement."""

    def __init__(self, lag):
        self.lag = lag                      # barrier intervals this row runs behind the leading one
        self.q = deque()                    # outstanding (half, ktile), oldest first; one entry = the two instructions of a half-tile
        self.retired_at = {}                # (half, ktile) -> interval in which the covering wait was executed
        self.issued_at = {}                 # (half, ktile) -> interval of the issue
        self.reads = []                     # (half, ktile, interval)

    def issue(self, half, k, g):
        self.q.append((half, k))
        self.issued_at[(half, k)] = g

    def wait(self, n_instr, g):             # s_waitcnt vmcnt(n): everything but the newest n INSTRUCTIONS (2 per half-tile) has landed
        keep = n_instr // 2
        while len(self.q) > keep:
            self.retired_at[self.q.popleft()] = g

    def read(self, half, k, g, safe_from=None):
        """``safe_from``: first interval in which another wave may overwrite the bytes -- g + 2 when the reader's lgkmcnt(0) sits BEHIND the barrier
        that ends interval g (default), g' + 1 when it waits in interval g' in front of that interval's barrier."""
        self.reads.append((half, k, g, g + 2 if safe_from is None else safe_from))


def four_phase(nk):
    """gemm_nt.hip PIPE 4: per K-tile  R-S1 | M-S1 | R-S2 | M-S2, one barrier between segments; row r runs r intervals behind."""
    rows = [Row(0), Row(1)]
    for r in rows:
        g = 0                                                # prologue interval
        for h in ("A0", "B0", "B1", "A1"):
            r.issue(h, 0, g)
        if nk > 1:
            for h in ("A0", "B0", "B1"):
                r.issue(h, 1, g)
            r.wait(6, g)
        else:
            r.wait(0, g)
        for t in range(nk):
            more1, more2 = t + 1 < nk, t + 2 < nk
            g = 1 + 4 * t + r.lag                            # R-S1
            for h in ("A0", "B0", "B1"):
                r.read(h, t, g)
            r.wait(6 if more1 else 0, g)
            g += 1                                           # M-S1
            if more1:
                r.issue("A1", t + 1, g)
            g += 1                                           # R-S2
            r.read("A1", t, g)
            r.wait(2 if more1 else 0, g)
            g += 1                                           # M-S2
            if more2:
                for h in ("A0", "B0", "B1"):
                    r.issue(h, t + 2, g)
    return rows


def eight_phase(nk, fixed):
    """The 8-phase loop (PIPE 2): P1 reads A0 + B0/B1(j0) ... ; DMA from the MFMA clusters: P1 A1(t+1), P2 A0(t+2), P3 B0(t+2), P4 B1(t+2) and
    ONE wait per K-tile -- rounds 2-4: vmcnt(6) at the end of P4's cluster; ``fixed`` (round 5): vmcnt(4) in front of P4's first barrier.
    Intervals: 8 per K-tile (read and cluster of each phase)."""
    rows = [Row(0), Row(1)]
    for r in rows:
        g = 0
        for h in ("A0", "B0", "B1", "A1"):
            r.issue(h, 0, g)
        if nk > 1:
            for h in ("A0", "B0", "B1"):
                r.issue(h, 1, g)
            r.wait(6, g)
        else:
            r.wait(0, g)
        for t in range(nk):
            more1, more2 = t + 1 < nk, t + 2 < nk
            g = 1 + 8 * t + r.lag                            # P1 read
            r.read("A0", t, g); r.read("B0", t, g); r.read("B1", t, g)     # (each wave reads one of B0 / B1: both checked)
            g += 1                                           # P1 cluster
            if more1:
                r.issue("A1", t + 1, g)
            g += 1                                           # P2 read: the second 32 columns of the wave's B half-tile
            r.read("B0", t, g); r.read("B1", t, g)
            g += 1                                           # P2 cluster
            if more2:
                r.issue("A0", t + 2, g)
            g += 1                                           # P3 read
            r.read("A1", t, g)
            g += 1                                           # P3 cluster
            if more2:
                r.issue("B0", t + 2, g)
            g += 1                                           # P4 (no reads)
            if fixed:
                r.wait(4 if more2 else 0, g)
            g += 1                                           # P4 cluster
            if more2:
                r.issue("B1", t + 2, g)
            if not fixed:
                r.wait(6 if more2 else 0, g)
    return rows


def tn_eight_phase(nk):
    """gemm_tn8_kernel (csrc/gemm_tn.hip): the same four quadrant phases on transposed fragment reads, with the half-tiles named as in gemm_nt
    (A0 / A1 = X left / right, B0 / B1 = Y left / right).  Here the DMA is issued from the READ segments -- P1: A1(t+1), P2: A0(t+2), P3: B0(t+2),
    P4: B1(t+2) + vmcnt(6) -- and the reads of a half-tile that is re-staged one phase later are retired by an lgkmcnt wait IN FRONT of the barrier
    of their own segment (P1: lgkmcnt(8) = the X reads; P2: lgkmcnt(0) = all Y reads)."""
    rows = [Row(0), Row(1)]
    for r in rows:
        g = 0
        for h in ("A0", "B0", "B1", "A1"):
            r.issue(h, 0, g)
        if nk > 1:
            for h in ("A0", "B0", "B1"):
                r.issue(h, 1, g)
            r.wait(6, g)
        else:
            r.wait(0, g)
        for t in range(nk):
            more1, more2 = t + 1 < nk, t + 2 < nk
            g = 1 + 8 * t + r.lag                            # P1 read: X left (retired here), Y (its first 32 columns)
            r.read("A0", t, g, safe_from=g + 1)
            r.read("B0", t, g, safe_from=g + 3); r.read("B1", t, g, safe_from=g + 3)      # retired by P2's lgkmcnt(0), two intervals on
            if more1:
                r.issue("A1", t + 1, g)
            g += 2                                           # P2 read: Y second 32 columns, lgkmcnt(0) in front of the barrier
            r.read("B0", t, g, safe_from=g + 1); r.read("B1", t, g, safe_from=g + 1)
            if more2:
                r.issue("A0", t + 2, g)
            g += 2                                           # P3 read: X right (lgkmcnt(0) behind the barrier)
            r.read("A1", t, g)
            if more2:
                r.issue("B0", t + 2, g)
            g += 2                                           # P4 read segment: last issue + the counted wait
            if more2:
                r.issue("B1", t + 2, g)
            r.wait(6 if more2 else 0, g)
    return rows


def violations(rows):
    bad = []
    for r in rows:
        for half, k, g, _ in r.reads:
            for o in rows:                                   # RAW: every row's share retired in an interval before the read's
                ra = o.retired_at.get((half, k))
                if ra is None or ra >= g:
                    bad.append(("RAW", half, k, f"read by row {r.lag} in interval {g}", f"row {o.lag} retired its share in {ra}"))
    for r in rows:
        for (half, k), gi in r.issued_at.items():
            if k < 2:
                continue
            for o in rows:                                   # WAR: every read of the previous occupant (k - 2) is retired AND a barrier lies in between
                for h2, k2, gr, safe in o.reads:
                    if h2 == half and k2 == k - 2 and gi < safe:
                        bad.append(("WAR", half, k, f"issued by row {r.lag} in interval {gi}", f"row {o.lag} read {half}({k2}) in {gr}"))
    return bad


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5, 12, 48])
def test_four_phase_loop_is_ordered(nk):
    rows = four_phase(nk)
    assert violations(rows) == []
    for r in rows:                                           # every half-tile of every K-tile was read exactly once per row, nothing left in flight
        assert sorted((h, k) for h, k, _, _ in r.reads) == sorted((h, k) for k in range(nk) for h in HALVES)
        assert not r.q
        # in flight across a barrier: at most three half-tiles (48 KB per CU) behind the counted waits
    # landing time: every half-tile has at least two full intervals between its issue and the wait that retires it
    for r in rows:
        for key, gi in r.issued_at.items():
            if gi > 0:
                assert r.retired_at[key] - gi >= 2, (key, gi, r.retired_at[key])


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 12, 36])
def test_eight_phase_loop_with_the_wait_in_front_of_the_barrier_is_ordered(nk):
    rows = eight_phase(nk, fixed=True)
    assert violations(rows) == []
    for r in rows:
        assert {(h, k) for h, k, _, _ in r.reads} == {(h, k) for k in range(nk) for h in HALVES} and not r.q


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 12, 225])
def test_weight_gradient_loop_is_ordered(nk):
    rows = tn_eight_phase(nk)
    assert violations(rows) == []
    for r in rows:
        assert {(h, k) for h, k, _, _ in r.reads} == {(h, k) for k in range(nk) for h in HALVES} and not r.q


@pytest.mark.parametrize("nk", [3, 12])
def test_the_checker_flags_the_round_2_to_4_wait_placement(nk):
    bad = violations(eight_phase(nk, fixed=False))
    raw = [b for b in bad if b[0] == "RAW"]
    assert raw and all(b[1] in ("A0", "B0", "B1") for b in raw)          # the lagging row waits in the interval in which the leading row already reads
    assert all("row 0" in b[3] and "row 1" in b[4] for b in raw)
    assert not [b for b in bad if b[0] == "WAR"]


# ---- the attention kernels' workgroup -> (part, head, batch) map (csrc/attention.hip: wg_id), restated: the division-free form of round 5 against the
# quotient / remainder form of rounds 1-4.  Workgroups are dealt to the 8 XCDs round-robin by linear id L = x + G (y + nh z); in groups of 8 G ids the
# parts of ONE (batch, head) pair are the ids j, 8 + j, 16 + j, ... (same XCD): part = r >> 3, pair = 8 group + (r & 7), r = L mod 8 G.
def _wg_id_div(x, y, z, G, nh, B):
    L, pairs = x + G * (y + nh * z), nh * B
    if G == 1 or pairs & 7:
        return (x, y, z)
    group = L // (8 * G)
    r = L - group * 8 * G
    pair = group * 8 + (r & 7)
    return (r >> 3, pair % nh, pair // nh)


def _wg_id_nodiv(x, y, z, G, nh, B):
    if G == 1 or (nh * B) & 7:
        return (x, y, z)
    P = y + nh * z
    c = P & 7
    r = G * c + x                      # = L mod 8 G, because x < G; L div 8 G = P >> 3
    hh, bb = y + ((r & 7) - c), z      # pair = P + ((r & 7) - c): fewer than 8 positions away from (y, z)
    while hh < 0:
        hh, bb = hh + nh, bb - 1
    while hh >= nh:
        hh, bb = hh - nh, bb + 1
    return (r >> 3, hh, bb)


@pytest.mark.parametrize("G,nh,B", [(4, 12, 256), (4, 12, 8), (2, 12, 8), (3, 4, 6), (4, 2, 4), (5, 1, 8), (7, 3, 8), (4, 12, 2), (2, 16, 3), (1, 12, 8), (4, 12, 3)])
def test_attention_workgroup_map_without_division_is_the_old_map_and_a_bijection(G, nh, B):
    seen = set()
    for z in range(B):
        for y in range(nh):
            for x in range(G):
                a, b = _wg_id_div(x, y, z, G, nh, B), _wg_id_nodiv(x, y, z, G, nh, B)
                assert a == b, (x, y, z, a, b)
                assert 0 <= b[0] < G and 0 <= b[1] < nh and 0 <= b[2] < B
                seen.add(b)
    assert len(seen) == G * nh * B
    if G > 1 and (nh * B) % 8 == 0:      # the point of the map: the G parts of a pair have linear ids 8 apart, i.e. land on one XCD
        by_pair = {}
        for z in range(B):
            for y in range(nh):
                for x in range(G):
                    part, h, b = _wg_id_nodiv(x, y, z, G, nh, B)
                    by_pair.setdefault((h, b), set()).add((x + G * (y + nh * z)) % 8)
        assert all(len(v) == 1 for v in by_pair.values())
