"""Metropolis acceptance, replica exchange and the replica sharding across GPUs (host side).

Counterpart of the reference's ``utils/replica_exchange_monte_carlo.py``: ``mc_delta`` (:26-57),
``metropolis_score`` (:60-78), ``replica_exchange_attempt`` (:81-110), ``replica_exchange`` (:113-173),
and of the ``mp.Pool(R)`` fan-out of ``mutate_sequence_re`` (:233-271).  The reference
forks one worker per replica and gathers whole pickled ``ScoreSeq`` objects every exchange step; here
the replicas of a design job are folded in batches on the GPU(s) and the only thing that crosses
GPUs is one all-gather of R scoring-function values per exchange step: every rank then
replays the same swap decisions (they touch only the temperature labels), so no sequence ever moves.

RNG semantics kept from the reference: ``mc_delta`` draws one ``random()`` only when the mutant is
worse; ``replica_exchange`` draws one only when the upper neighbour is worse; swaps use the MAIN
process stream (``DesiRNA.py:659-664``), which every rank seeds identically.
"""
import math

import numpy as np

L_CONST = 504.12  # reference DesiRNA.py:568 (DesignOptions.L)


def get_rep_temps(replicas, T_min, T_max):
    """Linear temperature ladder, rounded to 3 dp (reference utils/sequence_utils.py:826-843)."""
    if replicas == 1:
        return [T_max]
    delta = (T_max - T_min) / (replicas - 1)
    temps, cur = [], T_min
    for _ in range(replicas):
        temps.append(round(cur, 3))
        cur += delta
    return temps


def metropolis_score(temp, dE, L=L_CONST):
    return math.exp((-L / temp) * dE)


def mc_delta(deltaF_o, deltaF_m, T_replica, rng, L=L_CONST):
    """(accept, better): accept iff the mutant is not worse, else with probability exp(-L/T * dE)."""
    if deltaF_m <= deltaF_o:
        return True, True
    p = metropolis_score(T_replica, deltaF_m - deltaF_o, L)
    return p > rng.random(), False


def replica_exchange_attempt(T0, T1, dE0, dE1, rng, L=L_CONST):
    if dE1 <= dE0:
        return True, True
    u = rng.random()
    p = math.exp(L * (1 / T0 - 1 / T1) * (dE0 - dE1))
    return p > u, False


def replica_exchange(temps, scores, global_step, rng, L=L_CONST):
    """One exchange step over all replicas.

    temps[r], scores[r]: temperature shelf and scoring function of replica r (replica_num order).
    Returns (new_temps, n_accepted, n_accepted_better, n_rejected).  Even global steps pair the
    temperature-sorted neighbours (1,2),(3,4),...; odd steps pair (0,1),(2,3),... (reference :140-146);
    an accepted pair swaps its temperature labels only (:163-164).
    """
    R = len(temps)
    temps = list(temps)
    order = sorted(range(R), key=lambda r: temps[r])       # stable, like sorted(..., key=temp_shelf)
    n_shelfs = R - 1
    if global_step % 2 == 0:
        pairs = [(i + 1, i + 2) for i in range(0, n_shelfs - 1, 2)]
    else:
        pairs = [(i, i + 1) for i in range(0, n_shelfs, 2)]
    acc = better = rej = 0
    for a, b in pairs:
        ra, rb = order[a], order[b]
        ok, bet = replica_exchange_attempt(temps[ra], temps[rb], scores[ra], scores[rb], rng, L)
        if ok:
            temps[ra], temps[rb] = temps[rb], temps[ra]
            acc += 1
            better += int(bet)
        else:
            rej += 1
    return temps, acc, better, rej


class ReplicaShards:
    """Replica r lives on rank r mod world (interleaving spreads the temperature ladder)."""

    def __init__(self, n_replicas, rank=0, world=1):
        self.R, self.rank, self.world = int(n_replicas), int(rank), int(world)
        self.local = list(range(self.rank, self.R, self.world))
        self.max_local = -(-self.R // self.world)

    def owner(self, r):
        return r % self.world

    def allgather_scores(self, local_scores, device=None, extras=None):
        """local_scores: this rank's scores in self.local order -> all R scores in replica order.

        THE collective of an exchange step: one all-gather of ceil(R/world) (+ len(extras)) fp64 per rank (RCCL over xGMI
        with backend 'nccl' -- the tensors then live on this rank's GPU --, gloo on CPU).  `extras` are a few per-rank
        control values that ride in the same buffer (solved / time-is-up flags of the design loop); with extras the
        return value is (scores, extras_of_every_rank[world, len(extras)])."""
        ne = 0 if extras is None else len(extras)
        if self.world == 1:
            full = np.asarray(local_scores, dtype=np.float64).copy()
            return full if extras is None else (full, np.asarray(extras, dtype=np.float64).reshape(1, ne))
        import torch
        import torch.distributed as dist
        if device is None and dist.get_backend() == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        width = self.max_local + ne
        host = np.full(width, np.nan, dtype=np.float64)
        host[:len(self.local)] = np.asarray(local_scores, dtype=np.float64)
        if ne:
            host[self.max_local:] = np.asarray(extras, dtype=np.float64)
        buf = torch.as_tensor(host, device=device)
        out = torch.empty(self.world * width, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, buf)
        out = out.cpu().numpy().reshape(self.world, width)
        full = np.empty(self.R, dtype=np.float64)
        for rk in range(self.world):
            idx = list(range(rk, self.R, self.world))
            full[idx] = out[rk, :len(idx)]
        return full if extras is None else (full, out[:, self.max_local:].copy())

    def broadcast_seed(self, seed):
        """Every rank must replay the same swap decisions and start from the same sequence, so the main stream's seed
        is rank 0's: an unseeded run (seed 0 = random.seed(random.random()) in the reference, DesiRNA.py:659-664) draws
        it on rank 0 and sends it to the others."""
        if self.world == 1:
            return seed
        import torch.distributed as dist
        box = [seed]
        dist.broadcast_object_list(box, src=0)
        return box[0]


def shard_puzzles(lengths, world):
    """BASELINE config 4 (a set of independent design problems, e.g. the 100 Eterna100-V1 puzzles): puzzle -> rank by greedy
    longest-processing-time balancing of sum n^3.  Every rank computes the same assignment; no data-path
    collective is needed, only the final gather of the results."""
    order = sorted(range(len(lengths)), key=lambda k: (-int(lengths[k]) ** 3, k))
    load = [0] * world
    owner = [0] * len(lengths)
    for k in order:
        r = min(range(world), key=lambda x: (load[x], x))
        owner[k] = r
        load[r] += int(lengths[k]) ** 3
    return owner


def gather_results(local_results, world):
    """local_results: {puzzle index: result} of this rank -> the merged dict on every rank (one all_gather_object)."""
    if world == 1:
        return dict(local_results)
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local_results)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged
