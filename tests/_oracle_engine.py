"""A CPU stand-in for gpirt_amd.Sampler built on the oracle -- tests only.

It exists so the multi-process host logic of gpirt_amd/distributed.py (item partition, all-reduce of
the partial log-posterior, broadcast of L, gathers) can be exercised under gloo on a machine without
a GPU.  The HIP path itself is covered by the -m gpu tests.
"""
import numpy as np
import torch

from oracle import oracle as O


class OracleEngine:
    def __init__(self, y, theta0, pm, ps, step, item0, m_total, seed=11):
        self.y = np.asfortranarray(y)
        self.n, self.m = self.y.shape
        self.pm, self.ps, self.step = (np.asfortranarray(a) for a in (pm, ps, step))
        self.theta = np.array(theta0, dtype=np.float64)
        self.rng = O.ItemStream(seed, item_base=item0)
        self.it = 0
        self.ts = O.theta_star()
        self.N = len(self.ts)
        self._logpost = torch.zeros(self.N * self.n, dtype=torch.float64)
        self._L = torch.zeros(self.n * self.n, dtype=torch.float64)
        self._fstar_t = torch.zeros(self.N * self.m, dtype=torch.float64)   # persistent, like the device array

    # state views shared with torch (what the collectives operate on)
    def device_tensor(self, name):
        return {"logpost": self._logpost, "L": self._L, "fstar": self._fstar_t,
                "fstar_full": getattr(self, "_fstar_full", None),
                "theta_stage": getattr(self, "_theta_stage", None)}[name]

    @property
    def fstar(self):
        return self._fstar_t.numpy().reshape(self.N, self.m, order="F")

    @fstar.setter
    def fstar(self, value):
        self._fstar_t.numpy()[:] = np.asarray(value).reshape(-1, order="F")

    # respondent-block form of draw_theta (gpirt_sampler_set_theta_block / theta_block / theta_commit)
    def set_theta_block(self, y_block, i0, m_total):
        self.y_blk = np.asfortranarray(y_block)
        self.blk_i0 = i0
        self._fstar_full = torch.zeros(self.N * m_total, dtype=torch.float64)
        self._theta_stage = torch.zeros(self.n, dtype=torch.float64)

    def theta_block(self):
        fs = self._fstar_full.numpy().reshape(self.N, -1, order="F")
        prior = np.array([O.lib().orc_dnorm_log(t, 0.0, 1.0) for t in self.ts])
        st = self._theta_stage.numpy()
        st[:] = 0.0
        for ib in range(self.y_blk.shape[0]):
            i = self.blk_i0 + ib
            ok = ~np.isnan(self.y_blk[ib])
            a = fs[:, ok] * self.y_blk[ib, ok][None, :]
            P = prior + (-np.sum(np.log(1 + np.exp(-a)), axis=1))
            P = np.cumsum(np.exp(P - P.max()))
            P = (P - P.min()) / (P.max() - P.min())
            u = O.item_uniform(self.rng.s.seed, self.it + 1, O.ST_THETA, i, 0)
            st[i] = self.ts[np.nonzero(P > u)[0][0]]

    def theta_commit(self):
        self.theta = self._theta_stage.numpy().copy()

    @property
    def L(self):
        return self._L.numpy().reshape(self.n, self.n, order="F")

    def _factor_into_L(self):
        L, info = O.factor(self.theta)
        assert info == 0
        self.L[:, :] = L

    def init(self):
        self._factor_into_L()
        f = np.empty((self.n, self.m), order="F")
        for j in range(self.m):
            self.rng.substream(0, O.ST_INIT_F, j)
            f[:, j], _ = O.rmvnorm(self.rng, self.L)
        self.f = f
        beta = np.empty((2, self.m), order="F")
        for j in range(self.m):
            self.rng.substream(0, O.ST_INIT_BETA, j)
            for p in range(2):
                self.rng.s.index = p
                beta[p, j] = O.lib().orc_rnorm(self.rng.ref, self.pm[p, j], self.ps[p, j])
        self.beta = beta
        self._means()
        self.fstar, _, _ = O.draw_fstar(self.rng, self.f, self.theta, self.L, self.mu_star, it=0)

    def _means(self):
        self.mu = self.beta[0][None, :] + self.theta[:, None] * self.beta[1][None, :]
        self.mu_star = self.beta[0][None, :] + self.ts[:, None] * self.beta[1][None, :]

    def draw_f(self):
        self.f, self.k = O.draw_f(self.rng, self.f, self.y, self.L, self.mu, it=self.it + 1)

    def draw_fstar(self):
        self.fstar, _, _ = O.draw_fstar(self.rng, self.f, self.theta, self.L, self.mu_star, it=self.it + 1)

    def theta_partial(self):
        # partial log-likelihood sums over this rank's items: N x n, column i = respondent i
        lp = np.zeros((self.N, self.n), order="F")
        for i in range(self.n):
            ok = ~np.isnan(self.y[i])
            a = self.fstar[:, ok] * self.y[i, ok][None, :]
            lp[:, i] = -np.sum(np.log(1 + np.exp(-a)), axis=1)
        self._logpost.numpy()[:] = lp.reshape(-1, order="F")

    def theta_finish(self):
        lp = self._logpost.numpy().reshape(self.N, self.n, order="F")
        prior = np.array([O.lib().orc_dnorm_log(t, 0.0, 1.0) for t in self.ts])
        out = np.empty(self.n)
        for i in range(self.n):
            P = prior + lp[:, i]
            P = np.cumsum(np.exp(P - P.max()))
            P = (P - P.min()) / (P.max() - P.min())
            u = O.item_uniform(self.rng.s.seed, self.it + 1, O.ST_THETA, i, 0)
            out[i] = self.ts[np.nonzero(P > u)[0][0]]
        self.theta = out

    def draw_beta(self):
        self.beta = O.draw_beta(self.rng, self.beta, self.theta, self.y, self.f, self.pm, self.ps, self.step,
                                it=self.it + 1)
        self._means()

    def factor(self):
        self._factor_into_L()
        self.it += 1

    def skip_factor(self):
        self.it += 1

    def adopt_factor(self, rows_with_L):
        self.it += 1

    # -- the factorisation in pieces (gpirt_potrf_panel_* / gpirt_sampler_build_cov), NumPy, small panels ----------
    panel_width = 16
    torch_device = torch.device("cpu")

    @property
    def panel_rows(self):
        return self.n

    def build_cov(self):
        S = O.se_kernel(self.theta, self.theta)
        S[np.diag_indices_from(S)] += 0.001
        self.L[:, :] = S

    def _cols(self, p):
        K0 = p * self.panel_width
        return K0, min(K0 + self.panel_width, self.n)

    def panel_factor(self, p):
        L = self.L
        K0, c1 = self._cols(p)
        for j in range(K0, c1):                       # left-looking inside the panel, unblocked
            L[j:, j] -= L[j:, K0:j] @ L[j, K0:j]
            d = np.sqrt(L[j, j])
            L[j, j] = d
            L[j + 1:, j] /= d
            L[:j, j] = 0.0                            # arma::chol leaves zeros above the diagonal

    def panel_update(self, p, c):
        L = self.L
        K0, c1 = self._cols(p)
        lo, hi = self._cols(c)
        L[lo:, lo:hi] -= L[lo:, K0:c1] @ L[lo:hi, K0:c1].T

    # ... by halves of an outer panel (gpirt_potrf_panel_*_part): first sub-panel of `subpanel_width` columns / the rest
    subpanel_width = 8

    def _half(self, p, half):
        K0, c1 = self._cols(p)
        mid = min(K0 + self.subpanel_width, c1)
        return (K0, mid) if half == 0 else ((mid, c1) if half == 1 else (K0, c1))

    def panel_factor_part(self, p, half):
        L = self.L
        K0, _ = self._cols(p)
        j0, j1 = self._half(p, half)
        for j in range(j0, j1):
            L[j:, j] -= L[j:, K0:j] @ L[j, K0:j]
            d = np.sqrt(L[j, j])
            L[j, j] = d
            L[j + 1:, j] /= d
            L[:j, j] = 0.0

    def panel_update_part(self, p, c, part):
        L = self.L
        K0, c1 = self._cols(p)
        mid = min(K0 + self.subpanel_width, c1)
        lo, hi = self._cols(c)
        a_hi = min(lo + self.subpanel_width, hi)
        if c != p + 1:
            if part != 0:
                L[lo:, lo:hi] -= L[lo:, K0:c1] @ L[lo:hi, K0:c1].T
            return
        if part in (0, 2) and mid < c1:
            L[lo:, lo:a_hi] -= L[lo:, K0:mid] @ L[lo:a_hi, K0:mid].T
        if part in (1, 2):
            k0 = mid if mid < c1 else K0
            L[lo:, lo:a_hi] -= L[lo:, k0:c1] @ L[lo:a_hi, k0:c1].T
            if a_hi < hi:
                L[lo:, a_hi:hi] -= L[lo:, K0:c1] @ L[a_hi:hi, K0:c1].T

    def panel_copy_part(self, p, half, buf, to_buf):
        k0, k1 = self._half(p, half)
        if k1 <= k0:
            return
        view = buf.numpy()[: (self.n - k0) * (k1 - k0)].reshape(self.n - k0, k1 - k0, order="F")
        if to_buf:
            view[:, :] = self.L[k0:, k0:k1]
        else:
            self.L[k0:, k0:k1] = view
            self.L[:k0, k0:k1] = 0.0

    def panel_copy(self, p, buf, to_buf):
        K0, c1 = self._cols(p)
        view = buf.numpy()[: (self.n - K0) * (c1 - K0)].reshape(self.n - K0, c1 - K0, order="F")
        if to_buf:
            view[:, :] = self.L[K0:, K0:c1]
        else:
            self.L[K0:, K0:c1] = view
            self.L[:K0, K0:c1] = 0.0

    def get(self, name):
        return np.asfortranarray(getattr(self, name) if name != "L" else self.L)
