"""The hand-over protocol of the chained launch (boltzmann_machines_amd/csrc/bm_chain.h), as a discrete-event model.

The kernel's argument for "no deadlock, every tile exactly once, no tile before its inputs" does not depend on the GPU:
a team's tiles are ordered (round, pass, tile column) and handed out by ONE counter; a workgroup that claimed tile n
waits only for the tiles of the previous pass of the same round, all of which have smaller numbers and are therefore
held by workgroups that are already running.  This test replays that argument with adversarial schedules - any number
of resident workgroups per team (down to one), workgroups that arrive late (the CU was busy with another kernel),
random tile times, the next claim issued BEFORE the current tile is published (as the kernel does, under its epilogue) -
and checks the three properties on every schedule.  (The kernel itself is tested against the oracle on the GPU:
tests/test_chain_gpu.py.)"""
import heapq
import random

import pytest


def simulate(n_passes, tiles_per_pass, n_rounds, n_workers, rng, late_frac=0.3):
    """one team.  Returns (order in which tiles were published, start/publish times); raises on deadlock."""
    prefix = [0]
    for t in tiles_per_pass:
        prefix.append(prefix[-1] + t)
    per_round = prefix[-1]
    total = per_round * n_rounds
    counter = 0                                   # the team's claim counter
    published = {}                                # tile number -> publish time
    started = {}
    # event queue: (time, seq, kind, worker, tile)
    q, seq = [], 0
    for w in range(n_workers):
        arrive = rng.uniform(0.0, 50.0) if rng.random() < late_frac else 0.0
        heapq.heappush(q, (arrive, seq, 'claim', w, None)); seq += 1
    waiting = []                                  # (worker, tile, time it asked to start)

    def decode(n):
        r, rem = divmod(n, per_round)
        p = max(i for i in range(n_passes) if prefix[i] <= rem)
        return r, p, rem - prefix[p]

    def deps_ready(n):
        r, p, _ = decode(n)
        if p == 0:
            return True
        lo = r * per_round + prefix[p - 1]
        return all(m in published for m in range(lo, lo + tiles_per_pass[p - 1]))

    def start(t, w, n):
        nonlocal seq
        assert n not in started, 'tile %d claimed twice' % n
        assert deps_ready(n), 'tile %d started before its inputs' % n
        started[n] = t
        dur = rng.uniform(5.0, 15.0)
        # the next claim goes out under the epilogue, i.e. BEFORE this tile is published
        heapq.heappush(q, (t + dur - 1.0, seq, 'claim_next', w, n)); seq += 1
        heapq.heappush(q, (t + dur, seq, 'publish', w, n)); seq += 1

    pending_next = {}                             # worker -> tile it holds for after the publish
    while q:
        t, _, kind, w, n = heapq.heappop(q)
        if kind in ('claim', 'claim_next'):
            mine = counter
            counter += 1
            if kind == 'claim_next':
                pending_next[w] = mine
                continue
            if mine >= total:
                continue                          # nothing left: the workgroup exits
            if deps_ready(mine):
                start(t, w, mine)
            else:
                waiting.append((w, mine))
        else:                                     # publish
            published[n] = t
            nxt = pending_next.pop(w)
            if nxt < total:
                if deps_ready(nxt):
                    start(t, w, nxt)
                else:
                    waiting.append((w, nxt))
            still = []
            for (ww, nn) in waiting:              # pollers see the new flag
                if deps_ready(nn):
                    start(t, ww, nn)
                else:
                    still.append((ww, nn))
            waiting = still
    if waiting or len(published) != total:
        raise AssertionError('deadlock: %d of %d tiles published, %d workgroups waiting' % (len(published), total, len(waiting)))
    return started, published


@pytest.mark.parametrize('seed', range(40))
def test_claimed_tiles_never_deadlock_and_run_once_after_their_inputs(seed):
    rng = random.Random(seed)
    n_passes = rng.randint(2, 24)
    tiles = [rng.choice([25, 32, 7, 1, 64]) for _ in range(n_passes)]
    n_rounds = rng.randint(1, 3)
    n_workers = rng.choice([1, 2, 5, 16, 31, 32, 40])
    started, published = simulate(n_passes, tiles, n_rounds, n_workers, rng)
    total = sum(tiles) * n_rounds
    assert sorted(published) == list(range(total))
    # a tile starts only after every tile of the previous pass of its round was published
    per_round = sum(tiles)
    prefix = [0]
    for t in tiles:
        prefix.append(prefix[-1] + t)
    for n, t0 in started.items():
        r, rem = divmod(n, per_round)
        p = max(i for i in range(n_passes) if prefix[i] <= rem)
        if p:
            lo = r * per_round + prefix[p - 1]
            assert all(published[m] <= t0 for m in range(lo, lo + tiles[p - 1])), n


def test_one_workgroup_per_team_is_enough():
    """the degenerate schedule: a single resident workgroup walks the whole list in order"""
    rng = random.Random(1)
    started, published = simulate(6, [32, 25, 32, 25, 32, 25], 2, 1, rng, late_frac=0.0)
    order = sorted(published, key=lambda n: published[n])
    assert order == list(range(2 * 171))
