"""Item-column sharding of the sampler over torch.distributed (one process per GPU, RCCL over xGMI).

Given L and theta the m item columns of draw_f / draw_fstar / draw_beta are conditionally
independent (src/draw-f.cpp:69-71, src/draw-fstar.cpp:23-29, src/draw-beta.cpp:16-38), so rank r
owns columns [m*r/G, m*(r+1)/G).  What is not item-separable:
  * draw_theta sums the log-likelihood over ALL items per respondent (src/draw-theta.cpp:15-19).
    `theta="gather"` (default): the ranks all-gather their f* columns (N* x m, 8 MB at the metric size), every
    rank forms the log-posterior of ITS BLOCK OF RESPONDENTS over all items in one MFMA GEMM (the same sums in
    the same order as on one GPU) and draws theta for that block; the n draws are then combined (64 KB).
    `theta="allreduce"`: every rank computes the partial sum over its items for all respondents and the
    N* x n partial log-posteriors (66 MB) are all-reduced; every rank then draws the same theta.
    Either way the RNG is keyed by the global respondent index.
  * K + chol (src/gpirtMCMC.cpp:76-78): `chol="replicated"` factors on every rank (no traffic);
    `chol="bcast"` factors on rank 0 and broadcasts L (one RCCL broadcast per iteration);
    `chol="distributed"`: 1-D block-cyclic ownership of the 1024-column outer panels -- the owner
    factors a panel, the finished panel is broadcast while every rank applies the PREVIOUS panel to the block columns
    it owns (the next panel's columns first, so its owner can start factoring), and every rank ends with the full L:
    the panel broadcasts ARE the broadcast of L.  Same launches as the single-GPU factorisation => bit-identical L.
The per-item RNG sub-streams are keyed by the GLOBAL item index, so the draws do not depend on G.

`engine_factory(y_local, theta_init, pm, ps, step, item0, m_total)` must return an object with the
stage methods of gpirt_amd.Sampler; tests substitute a CPU engine to exercise this host logic
under gloo.
"""
from __future__ import annotations

import numpy as np


def item_range(m: int, rank: int, world: int):
    """Contiguous block partition of the item columns."""
    lo = (m * rank) // world
    hi = (m * (rank + 1)) // world
    return lo, hi


class ShardedSampler:
    def __init__(self, engine_factory, y, theta_init, pm=None, ps=None, step=None, *, dist=None,
                 chol="replicated", theta="gather"):
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        y = np.asfortranarray(np.asarray(y, dtype=np.float64))
        self.n, self.m_total = y.shape
        self.lo, self.hi = item_range(self.m_total, self.rank, self.world)
        m = self.m_total
        pm = np.zeros((2, m)) if pm is None else np.asarray(pm, dtype=np.float64)
        ps = np.full((2, m), 3.0) if ps is None else np.asarray(ps, dtype=np.float64)
        step = np.full((2, m), 0.1) if step is None else np.asarray(step, dtype=np.float64)
        sl = slice(self.lo, self.hi)
        self.engine = engine_factory(y[:, sl], np.asarray(theta_init, dtype=np.float64), pm[:, sl], ps[:, sl],
                                     step[:, sl], self.lo, m)
        self.chol = chol
        if chol not in ("replicated", "bcast", "distributed"):
            raise ValueError("chol must be 'replicated', 'bcast' or 'distributed'")
        self._panel_bufs = None
        if theta not in ("gather", "allreduce"):
            raise ValueError("theta must be 'gather' or 'allreduce'")
        self.theta_mode = theta if self.world > 1 else "local"
        self._views = {}
        if self.theta_mode == "gather":
            self.i0, self.i1 = item_range(self.n, self.rank, self.world)     # this rank's block of respondents
            self.engine.set_theta_block(y[self.i0:self.i1, :], self.i0, m)
            # The gather path is chosen ONCE, from facts every rank agrees on (backend, m % world): a rank that
            # switched paths on its own (e.g. after a failed collective) would leave its peers in a different
            # collective.  Equal shards on a backend with the flat all-gather (RCCL, gloo) use it; anything else
            # all-reduces disjoint supports.
            self._flat_gather = (m % self.world == 0) and hasattr(dist, "all_gather_into_tensor")

    def _view(self, name):
        """torch view of an engine buffer, created once (the buffers live as long as the engine)"""
        v = self._views.get(name)
        if v is None:
            v = self._views[name] = self.engine.device_tensor(name)
        return v

    # -- collectives on the engine's buffers ------------------------------------------------
    def _allreduce_logpost(self):
        if self.world > 1:
            self.dist.all_reduce(self._view("logpost"))

    def _gather_fstar(self):
        """every rank's f* columns -> the full N* x m array on every rank (column-major: shards are contiguous)"""
        full, loc = self._view("fstar_full"), self._view("fstar")
        if self._flat_gather:
            self.dist.all_gather_into_tensor(full, loc)
            return
        N = full.numel() // self.m_total
        full.zero_()
        full[N * self.lo: N * self.hi].copy_(loc[: N * (self.hi - self.lo)])
        self.dist.all_reduce(full)                  # disjoint supports: the sum is the concatenation, exactly

    def _combine_theta(self):
        self.dist.all_reduce(self._view("theta_stage"))   # zero outside each rank's block

    def _factor_distributed(self):
        """Right-looking over outer panels with one panel of look-ahead.  Panel p is owned by rank p % world.
        Every block column receives the panels' updates in ascending order on its owner's (in-order) stream."""
        import torch
        e, W, n = self.engine, self.engine.panel_width, self.n
        NP = (n + W - 1) // W
        rows = e.panel_rows                         # n, plus the rows of a bordered factorisation
        owner = lambda p: p % self.world
        if self._panel_bufs is None:
            # two broadcast buffers (panel p + 1 travels while panel p is still being applied)
            self._panel_bufs = [torch.empty(rows * min(W, n), dtype=torch.float64, device=e.torch_device) for _ in range(2)]
        buf = lambda p: self._panel_bufs[p % 2][: (rows - p * W) * min(W, n - p * W)]
        e.build_cov()
        work = None

        def send(p):            # the finished panel p leaves its owner; collective: every rank calls it
            if owner(p) == self.rank:
                e.panel_copy(p, buf(p), True)
            return self.dist.broadcast(buf(p), src=owner(p), async_op=True)

        def receive(p, w):
            w.wait()
            if owner(p) != self.rank:
                e.panel_copy(p, buf(p), False)

        if owner(0) == self.rank:
            e.panel_factor(0)
        receive(0, send(0))
        for p in range(NP - 1):
            nxt = p + 1
            if owner(nxt) == self.rank:
                e.panel_update(p, nxt)              # the next panel's columns first ...
                e.panel_factor(nxt)                 # ... so its owner can factor it
            if nxt < NP - 1:
                work = send(nxt)                    # travels while panel p is applied to the remaining columns
            for c in range(nxt + 1, NP):
                if owner(c) == self.rank:
                    e.panel_update(p, c)
            if nxt < NP - 1:
                receive(nxt, work)
        # the last panel has nothing to update: it only has to reach everybody
        if NP > 1:
            receive(NP - 1, send(NP - 1))
        e.adopt_factor(True)                        # the panel copies carry the rows below the factor too (panel_rows)

    def _factor(self):
        if self.chol == "distributed" and self.world > 1:
            self._factor_distributed()
        elif self.chol in ("replicated", "distributed") or self.world == 1:
            self.engine.factor()
        else:
            if self.rank == 0:
                self.engine.factor()
            else:
                self.engine.adopt_factor(True)      # the "L" view is the WHOLE ldl x n buffer: the bordered rows travel too
            self.dist.broadcast(self._view("L"), src=0)

    def init(self):
        self.engine.init()

    STAGES = ("draw_f", "draw_fstar", "theta_gemm", "theta_allreduce", "theta_sample", "draw_beta", "factor")

    def step(self, timer=None):
        """One MCMC iteration.  `timer(name)` (optional) is called after each stage is enqueued --
        bench.py passes a function that records a device event, to attribute time to stages."""
        e = self.engine
        t = timer if timer is not None else (lambda name: None)
        e.draw_f(); t("draw_f")
        e.draw_fstar(); t("draw_fstar")
        if self.theta_mode == "gather":
            self._gather_fstar(); t("theta_allreduce")
            e.theta_block(); t("theta_gemm")
            self._combine_theta(); e.theta_commit(); t("theta_sample")
        else:
            e.theta_partial(); t("theta_gemm")
            self._allreduce_logpost(); t("theta_allreduce")
            e.theta_finish(); t("theta_sample")
        e.draw_beta(); t("draw_beta")
        self._factor(); t("factor")

    def gather(self, name: str):
        """All ranks receive the full (column-concatenated) array `name` of item-sharded state."""
        local = np.ascontiguousarray(self.engine.get(name).T)     # (m_local, rows)
        if self.world == 1:
            return np.asfortranarray(local.T)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, local)
        return np.asfortranarray(np.concatenate(parts, axis=0).T)
