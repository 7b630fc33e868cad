"""CPU model check of the peer-memory exchange protocol (csrc/exchange.hip): every interleaving of the ranks' memory operations, for small
configurations.  The kernel cannot be run across GPUs here, so its hand-off logic is restated as a transition system and explored
exhaustively:

  kernel k of rank r at point e = k mod NP, generation g = k div NP + 1 (the device-side launch counter), P workgroups in parallel:
      workgroup for peer q != r:  rows[q][e][r] := k ;  flag[q][e][r] := g
      workgroup for r itself:     rows[r][e][r] := k ;  flag[r][e][r] := g ;  wait flag[r][e][j] >= g for every j (lane j polls its own word)
  (flags are never lowered: round 4 — a flag that arrives late is one generation behind the next use of its point)
  then (stream order) the consumer reads rows[r][e][*] and the rank's next kernel starts.

Checked: no deadlock, and the consumer of kernel k sees k in every slice — for the re-use rule the kernel states (consecutive exchanges
alternate between >= 2 points).  And the converse: with ONE point the same exploration finds a lost flag or a clobbered row, i.e. the rule
is what makes the protocol safe, not luck of timing."""
import itertools


def _explore(P, NP, T, limit=3_000_000):
    """returns (states, failure or None).  A rank's state: (kernel k, phase, per-workgroup program counters)."""
    # workgroup programs: list of ops.  ops: ("row", q), ("raise", q), ("see", j)
    def programs(r):
        progs = []
        for q in range(P):
            if q != r:
                progs.append((("row", q), ("raise", q)))
        # the waiting workgroup: lanes poll in parallel: modelled as one thread taking the P observations in a fixed order — enough,
        # because an observation only blocks and flags only grow
        progs.append(tuple([("row", r), ("raise", r)] + [("see", j) for j in range(P)]))
        return progs

    progs = [programs(r) for r in range(P)]
    # memory: flags[q][e][r], rows[q][e][r]
    flags0 = tuple(0 for _ in range(P * NP * P))
    rows0 = tuple(-1 for _ in range(P * NP * P))
    idx = lambda q, e, r: (q * NP + e) * P + r
    start = (tuple((0, tuple(0 for _ in range(P))) for _ in range(P)), flags0, rows0)
    seen = {start}
    stack = [start]
    while stack:
        ranks, flags, rows = stack.pop()
        succ = []
        done_all = True
        for r in range(P):
            k, pcs = ranks[r]
            if k >= T:
                continue
            done_all = False
            e = k % NP
            g = k // NP + 1
            kernel_done = all(pc == len(progs[r][w]) for w, pc in enumerate(pcs))
            if kernel_done:
                # consumer (next kernel in stream order) reads the rank's rows of this point
                for j in range(P):
                    if rows[idx(r, e, j)] != k:
                        return len(seen), f"rank {r}, kernel {k}: slice of rank {j} holds {rows[idx(r, e, j)]}"
                nr = list(ranks)
                nr[r] = (k + 1, tuple(0 for _ in range(P)))
                succ.append((tuple(nr), flags, rows))
                continue
            for w, pc in enumerate(pcs):
                if pc == len(progs[r][w]):
                    continue
                op, arg = progs[r][w][pc]
                nf, nw = flags, rows
                if op == "row":
                    nw = list(rows); nw[idx(arg, e, r)] = k; nw = tuple(nw)
                elif op == "raise":
                    nf = list(flags); nf[idx(arg, e, r)] = g; nf = tuple(nf)
                elif op == "see":
                    if flags[idx(r, e, arg)] < g:
                        continue           # blocked
                npcs = list(pcs); npcs[w] = pc + 1
                nr = list(ranks); nr[r] = (k, tuple(npcs))
                succ.append((tuple(nr), nf, nw))
        if not succ and not done_all:
            return len(seen), "deadlock: " + repr(ranks)
        for s in succ:
            if s not in seen:
                seen.add(s)
                if len(seen) > limit:
                    return len(seen), "state limit"
                stack.append(s)
    return len(seen), None


def test_two_ranks_two_points_every_interleaving():
    n, fail = _explore(P=2, NP=2, T=5)
    assert fail is None, fail
    assert n > 1000


def test_three_ranks_two_points_every_interleaving():
    n, fail = _explore(P=3, NP=2, T=3)
    assert fail is None, fail


def test_two_ranks_four_points_as_in_a_decoder_block():
    n, fail = _explore(P=2, NP=4, T=6)
    assert fail is None, fail


def test_one_point_is_unsafe_and_the_model_finds_it():
    """without the re-use rule a fast rank overwrites a slice before its consumer has read it (the generations keep the flags safe, not the rows)"""
    n, fail = _explore(P=2, NP=1, T=3)
    assert fail is not None and fail != "state limit"
