"""The protocol of the chained DBM update (boltzmann_machines_amd/csrc/bm_dbmchain.h) as a discrete-event model: TWO pass
families handed out to the workgroups of a team - the mean-field tiles of the main sequence with slots for particle tiles,
the particle tiles from a counter of their own - and the data-dependent end of the mean-field loop decided inside the launch
from arrival words that cross the teams late.

What the kernel's argument claims, and what is checked here on adversarial schedules (any number >= 1 of resident workgroups
per team, late workgroups, random tile times, verdicts that take a random time to become visible on another team, several
teams with different speeds):
  * no deadlock, although workgroups spin for flags, for verdicts of OTHER teams' tiles and - after the end of the loop -
    for producers that will never run (the stop word);
  * every tile of a sweep <= n runs exactly once, after its inputs, n being the first sweep in which no tile saw a residual
    above the tolerance (or the cap);
  * at most ONE sweep past n starts (the speculative one), none of sweep n + 2 writes anything: with three rotating
    buffers the result mu_n is never overwritten;
  * every particle tile runs exactly once in dependency order, whatever the mean-field does;
  * the trip count derived from the arrival words equals n.
(The kernel itself is tested against the oracle on the GPU: tests/test_dbm_chain_gpu.py.)"""
import heapq
import random

import pytest


def simulate(rng, n_teams, n1, n2, slots, max_sweeps, conv_at, pc_passes, pc_sweeps, workers_per_team, verdict_delay):
    """one launch.  conv_at: the first sweep whose tiles all report "no residual above tol" (None: never); every arrival
    word / stop word becomes visible to the other workgroups after a random delay <= verdict_delay.  A particle tile remembers
    whether its workgroup came from a slot of the main sequence or drains the particle counter (tail)."""
    T = n1 + slots + n2
    total_per_sweep = n_teams * (n1 + n2)
    pc_per_sweep = sum(pc_passes)
    n_pc = pc_per_sweep * pc_sweeps
    pc_prefix = [0]
    for t in pc_passes:
        pc_prefix.append(pc_prefix[-1] + t)
    arrived, stop = {}, []
    flags_mf = [dict() for _ in range(n_teams)]
    flags_pc = [dict() for _ in range(n_teams)]
    main_ctr, pc_ctr = [0] * n_teams, [0] * n_teams
    started_mf, started_pc, aborted = {}, {}, []
    q, seq = [], [0]

    def push(t, kind, team, w, payload=None):
        heapq.heappush(q, (t, seq[0], kind, team, w, payload)); seq[0] += 1

    for team in range(n_teams):
        for w in range(workers_per_team[team]):
            push(rng.uniform(0.0, 40.0) if rng.random() < 0.3 else 0.0, 'next', team, w, 'main')

    def verdict(s, t):
        vis = [v for (tv, v) in arrived.get(s, []) if tv <= t]
        if len(vis) < total_per_sweep:
            return False, False
        return True, not any(vis)

    def stopped(lim, t):
        return any(tv <= t and s <= lim for (tv, s) in stop)

    def mf_ready(team, sweep, p):
        if p == 0:
            return sweep == 1 or all((sweep - 1, 1, i) in flags_mf[team] for i in range(n2))
        return all((sweep, 0, i) in flags_mf[team] for i in range(n1))

    def pc_ready(team, n):
        t_, rem = divmod(n, pc_per_sweep)
        p = max(i for i in range(3) if pc_prefix[i] <= rem)
        if p == 0:
            if t_ == 0:
                return True
            prev = (t_ - 1) * pc_per_sweep
            need = range(prev + pc_prefix[1], prev + pc_prefix[3])
        else:
            need = range(t_ * pc_per_sweep, t_ * pc_per_sweep + pc_passes[0])
        return all(m in flags_pc[team] for m in need)

    waiting = []

    def try_start(t, team, w, kind, payload, mode):
        if kind == 'mf':
            sweep, p, ti = payload
            if sweep >= 3:
                comp, conv = verdict(sweep - 2, t)
                if comp and conv:
                    stop.append((t + rng.uniform(0.0, verdict_delay), sweep - 2))
                    aborted.append(payload)
                    push(t, 'next', team, w, 'tail')
                    return True
                if stopped(sweep - 2, t):
                    aborted.append(payload)
                    push(t, 'next', team, w, 'tail')
                    return True
                if not comp:
                    return False
            if not mf_ready(team, sweep, p):
                return False
            key = (team,) + payload
            assert key not in started_mf, 'mean-field tile %r claimed twice' % (key,)
            assert payload[0] < 3 or verdict(payload[0] - 2, t) == (True, False)
            started_mf[key] = t
            push(t + rng.uniform(4.0, 12.0), 'publish_mf', team, w, payload)
            return True
        if not pc_ready(team, payload):
            return False
        assert (team, payload) not in started_pc, 'particle tile claimed twice'
        started_pc[(team, payload)] = t
        push(t + rng.uniform(4.0, 20.0), 'publish_pc', team, w, (payload, mode))
        return True

    def poll(t):
        still = []
        for item in list(waiting):
            if not try_start(t, *item):
                still.append(item)
        waiting[:] = still

    steps = 0
    while q:
        steps += 1
        assert steps < 3000000, 'runaway'
        t, _, kind, team, w, payload = heapq.heappop(q)
        if kind == 'next':
            if payload == 'main':
                n = main_ctr[team]; main_ctr[team] += 1
                s, r = divmod(n, T)
                sweep = s + 1
                if sweep > max_sweeps or (sweep >= 3 and stopped(sweep - 2, t)):
                    push(t, 'next', team, w, 'tail')
                elif n1 <= r < n1 + slots:
                    p = pc_ctr[team]; pc_ctr[team] += 1
                    if p >= n_pc:
                        push(t, 'next', team, w, 'main')
                    elif not try_start(t, team, w, 'pc', p, 'main'):
                        waiting.append((team, w, 'pc', p, 'main'))
                else:
                    pl = (sweep, 0 if r < n1 else 1, r if r < n1 else r - n1 - slots)
                    if not try_start(t, team, w, 'mf', pl, 'main'):
                        waiting.append((team, w, 'mf', pl, 'main'))
            else:
                p = pc_ctr[team]; pc_ctr[team] += 1
                if p < n_pc and not try_start(t, team, w, 'pc', p, 'tail'):
                    waiting.append((team, w, 'pc', p, 'tail'))
        elif kind == 'publish_mf':
            sweep = payload[0]
            flags_mf[team][payload] = t
            violated = not (conv_at is not None and sweep >= conv_at)
            arrived.setdefault(sweep, []).append((t + rng.uniform(0.0, verdict_delay), violated))
            push(t, 'next', team, w, 'main')
            poll(t)
        elif kind == 'publish_pc':
            n, mode = payload
            flags_pc[team][n] = t
            push(t, 'next', team, w, mode)
            poll(t)
        else:
            poll(t)
        if not q and waiting:
            future = [tv for lst in arrived.values() for (tv, _) in lst if tv > t] + [tv for (tv, _) in stop if tv > t]
            if future:
                push(min(future) + 1e-6, 'tick', 0, 0)
    return dict(waiting=waiting, started_mf=started_mf, started_pc=started_pc, aborted=aborted, arrived=arrived,
                total_per_sweep=total_per_sweep, n_pc=n_pc)


def _run(seed):
    rng = random.Random(seed)
    n_teams = rng.choice([1, 2, 8])
    n1, n2 = rng.choice([(16, 32), (6, 10), (8, 6), (1, 1)])
    slots = rng.choice([0, 32 - n1 if n1 < 32 else 0, 3])
    max_sweeps = rng.choice([1, 2, 3, 5, 12, 30])
    conv_at = rng.choice([None, 1, 2, 3, 7, max_sweeps, max_sweeps + 3])
    pc_passes = rng.choice([(16, 32, 25), (6, 10, 8), (1, 1, 1)])
    pc_sweeps = rng.choice([1, 2, 5])
    workers = [rng.choice([1, 2, 7, 32, 40]) for _ in range(n_teams)]
    delay = rng.choice([0.0, 3.0, 30.0, 200.0])
    return rng, dict(n_teams=n_teams, n1=n1, n2=n2, slots=slots, max_sweeps=max_sweeps, conv_at=conv_at, pc_passes=pc_passes,
                     pc_sweeps=pc_sweeps, workers_per_team=workers, verdict_delay=delay)


@pytest.mark.parametrize('seed', range(24))
def test_two_families_and_the_late_verdict(seed):
    rng, cfg = _run(seed)
    out = simulate(rng, **cfg)
    n_expected = cfg['max_sweeps'] if (cfg['conv_at'] is None or cfg['conv_at'] > cfg['max_sweeps']) else cfg['conv_at']
    per_team = cfg['n1'] + cfg['n2']
    # every tile of the sweeps 1 .. n ran, on every team, exactly once (claimed-twice is asserted inside the model)
    for team in range(cfg['n_teams']):
        for s in range(1, n_expected + 1):
            assert sum(1 for k in out['started_mf'] if k[0] == team and k[1] == s) == per_team, (team, s)
    # at most one speculative sweep; nothing of sweep n + 2 ever started
    assert all(k[1] <= n_expected + 1 for k in out['started_mf']), max(k[1] for k in out['started_mf'])
    # the trip count from the arrival words (dch_finish_kernel): first complete sweep without a violation, else the cap
    n_words = cfg['max_sweeps']
    for s in range(1, cfg['max_sweeps'] + 1):
        lst = out['arrived'].get(s, [])
        assert len(lst) == out['total_per_sweep'], 'sweep %d incomplete before the end of the loop' % s
        if not any(v for (_, v) in lst):
            n_words = s
            break
    assert n_words == n_expected
    # the particle family: every tile once
    for team in range(cfg['n_teams']):
        assert sorted(n for (tm, n) in out['started_pc'] if tm == team) == list(range(out['n_pc']))
    assert not out['waiting'], 'deadlock: %d workgroups still waiting' % len(out['waiting'])


