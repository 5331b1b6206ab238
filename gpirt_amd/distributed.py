"""Item-column sharding of the sampler over torch.distributed (one process per GPU, RCCL over xGMI).

Given L and theta the m item columns of draw_f / draw_fstar / draw_beta are conditionally
independent (src/draw-f.cpp:69-71, src/draw-fstar.cpp:23-29, src/draw-beta.cpp:16-38), so rank r
owns columns [m*r/G, m*(r+1)/G).  What is not item-separable:
  * draw_theta sums the log-likelihood over ALL items per respondent (src/draw-theta.cpp:15-19):
    every rank computes the partial sum over its items (one MFMA GEMM) and the N* x n partial
    log-posteriors are all-reduced; every rank then draws the same theta (RNG keyed by respondent).
  * K + chol (src/gpirtMCMC.cpp:76-78): `chol="replicated"` factors on every rank (no traffic);
    `chol="bcast"` factors on rank 0 and broadcasts L (one RCCL broadcast per iteration).
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
                 chol="replicated"):
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
        if chol not in ("replicated", "bcast"):
            raise ValueError("chol must be 'replicated' or 'bcast'")

    # -- collectives on the engine's buffers ------------------------------------------------
    def _allreduce_logpost(self):
        if self.world > 1:
            self.dist.all_reduce(self.engine.device_tensor("logpost"))

    def _factor(self):
        if self.chol == "replicated" or self.world == 1:
            self.engine.factor()
        else:
            if self.rank == 0:
                self.engine.factor()
            else:
                self.engine.skip_factor()
            self.dist.broadcast(self.engine.device_tensor("L"), src=0)

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
