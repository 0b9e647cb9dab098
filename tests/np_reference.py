"""A second, independent restatement of the reference's CD-k train op in plain NumPy (float64
arithmetic, matrix form), written to read like boltzmann_machines/rbm/base_rbm.py:415-479.
It shares only the pinned Philox stream with the C oracle; used by tests/test_oracle.py to
cross-check oracle/bm_oracle.c (which uses the canonical fp32 chain order)."""
import numpy as np

from boltzmann_machines_amd.utils import philox


def sigmoid(x):
    return 1. / (1. + np.exp(-x))


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def bernoulli(p, seed, site, call, row0=0):
    """Bernoulli(probs=p).sample(): uniform < p (layers.py:50-51)."""
    B, n = p.shape
    u = philox.uniform(seed, site, call, B * n, idx0=row0 * n).reshape(B, n)
    return (u < p.astype(np.float32)).astype(np.float64), u


def dropout_input(X, keep, seed, call):
    """tf.nn.dropout(X, keep_prob) (base_rbm.py:417-418): x / keep * floor(keep + u)."""
    u = philox.uniform(seed, 1, call, X.size).reshape(X.shape)
    return (X.astype(np.float32) / np.float32(keep) * np.floor(np.float32(keep) + u)).astype(np.float64)


def cd_step(P, X, lr, momentum, k, seed, call, l2=1e-4, sample_v=False, sample_h=True, dropout=None,
            sp_target=0.1, sp_cost=0., sp_damping=0.9, dbm_first=False, dbm_last=False):
    """One `session.run(train_op)`; P = dict(W, vb, hb, dW, dvb, dhb, q_means) of float64 arrays
    (updated in place).  Returns the intermediates."""
    W, vb, hb = P['W'], P['vb'], P['hb']
    up, down = 1. + dbm_first, 1. + dbm_last
    X = X.astype(np.float64)
    if dropout is not None:                                                 # :417-418
        X = dropout_input(X, dropout, seed, call)
    h0_means = sigmoid(up * X.dot(W) + up * hb)                             # :421, :339-345
    h0_samples, u_h0 = bernoulli(h0_means, seed, 2, call)                   # :422
    h_states = h0_samples if sample_h else h0_means                         # :423
    us = []
    for t in range(k):                                                      # :367-378
        v_means = sigmoid(down * h_states.dot(W.T) + down * vb)
        v_states = v_means
        if sample_v:
            v_states, u = bernoulli(v_means, seed, 3 + 16 * t, call)
            us.append((u, v_means))
        h_means = sigmoid(up * v_states.dot(W) + up * hb)
        h_states = h_means
        if sample_h:
            h_states, u = bernoulli(h_means, seed, 4 + 16 * t, call)
            us.append((u, h_means))
    N = float(len(X))
    dW = (X.T.dot(h0_means) - v_states.T.dot(h_means)) / N - l2 * W         # :447-449
    dvb = np.mean(X - v_states, axis=0)                                     # :451
    dhb = np.mean(h0_means - h_means, axis=0)                               # :453
    q = sp_damping * P['q_means'] + (1 - sp_damping) * np.sum(h_means, axis=0)   # :457-459
    P['q_means'][...] = q
    pen = sp_cost * (q - sp_target)                                         # :460
    dhb -= pen
    dW -= pen                                                               # :462 (broadcast over rows)
    P['dW'][...] = lr * (momentum * P['dW'] + dW); P['W'] += P['dW']        # :467-468
    P['dvb'][...] = lr * (momentum * P['dvb'] + dvb); P['vb'] += P['dvb']   # :470-471
    P['dhb'][...] = lr * (momentum * P['dhb'] + dhb); P['hb'] += P['dhb']   # :473-474
    return dict(X=X, h0_means=h0_means, u_h0=u_h0, v_means=v_means, v_states=v_states, h_means=h_means, us=us)


def free_energy(P, X):
    """BernoulliRBM._free_energy (rbm/rbm.py:17-22)."""
    return np.mean(-X.dot(P['vb']) - np.sum(softplus(X.dot(P['W']) + P['hb']), axis=1))


# ====================================================================== DBM (dbm.py:385-759)
# Float64, matrix-form restatement of the DBM graph, written from boltzmann_machines/dbm.py
# independently of oracle/bm_oracle.c (which walks the same graph element by element in the
# canonical fp32 chain order).  The only things shared with the C oracle are the pinned Philox
# stream (sites: 8 + i hidden layer i, 12 visible, 13 AIS x0; counter word = site + 16 * sweep;
# global row offsets) and the variable names of DbmEngine / OracleDBM.

def _sfx(i):
    return '' if i == 0 else '_%d' % i


class NumpyDBM(object):
    """P: dict with W*, hb*, vb, dW*, dhb*, dvb, q_means*, mu_means*, mu*, v, h* (float64 copies)."""

    def __init__(self, P, n_layers, N, M, sample_v=True, sample_h=None, max_mf=10, mf_tol=1e-7, l2=0.,
                 max_norm=np.inf, sp_target=None, sp_cost=None, sp_damping=0.9):
        self.P, self.L, self.N, self.M = P, n_layers, N, M
        self.smp_v, self.smp_h = sample_v, sample_h or [True] * n_layers
        self.max_mf, self.mf_tol, self.l2, self.max_norm = max_mf, mf_tol, l2, max_norm
        self.sp_target = sp_target or [0.1] * n_layers
        self.sp_cost = sp_cost or [0.] * n_layers
        self.sp_damping = sp_damping
        self.seed, self.call, self.prow0 = 0, 0, 0
        self.ties = 0            # draws whose uniform was within 1e-6 of the probability
        self.allreduce_max = None    # data-parallel: max over ranks of the mean-field residual (dbm.py:449-452 is over ALL rows)

    def W(self, i): return self.P['W' + _sfx(i)]
    def hb(self, i): return self.P['hb' + _sfx(i)]

    def _draw(self, means, site, t, seed=None, call=None, row0=None):
        seed = self.seed if seed is None else seed
        call = self.call if call is None else call
        row0 = self.prow0 if row0 is None else row0
        s, u = bernoulli(means, seed, site + 16 * t, call, row0)
        self.ties += int(np.sum(np.abs(u.astype(np.float64) - means) < 1e-6))
        return s

    def gibbs_step(self, v, H, update_v=True, sample=True, t=0):
        """_make_gibbs_step (dbm.py:385-427): returns (v_new, H_new)."""
        L = self.L
        H_new = [None] * L
        T = v.dot(self.W(0))                                                    # :390
        if L >= 2:
            T = T + H[1].dot(self.W(1).T)                                       # :391-392
        H_new[0] = sigmoid(T + self.hb(0))                                      # :393
        if sample and self.smp_h[0]:
            H_new[0] = self._draw(H_new[0], 8 + 0, t)                           # :394-396
        for i in range(1, L - 1):                                               # :399-407
            T1 = H_new[i - 1].dot(self.W(i))
            T2 = H[i + 1].dot(self.W(i + 1).T)
            H_new[i] = sigmoid(T1 + T2 + self.hb(i))
            if sample and self.smp_h[i]:
                H_new[i] = self._draw(H_new[i], 8 + i, t)
        if L >= 2:                                                              # :410-416
            H_new[-1] = sigmoid(H_new[-2].dot(self.W(L - 1)) + self.hb(L - 1))
            if sample and self.smp_h[-1]:
                H_new[-1] = self._draw(H_new[-1], 8 + L - 1, t)
        v_new = None
        if update_v:                                                            # :419-425
            v_new = sigmoid(H_new[0].dot(self.W(0).T) + self.P['vb'])
            if sample and self.smp_v:
                v_new = self._draw(v_new, 12, t)
        return v_new, H_new

    def mean_field(self, X):
        """_make_mf (dbm.py:429-478): returns n_mf_updates, leaves mu* in P."""
        L = self.L
        mu_new, T = [], None
        for i in range(L):                                                      # :434-446
            if i == 0:
                T = 2. * X.dot(self.W(0))
            else:
                T = T.dot(self.W(i))
                if i < L - 1:
                    T = T * 2.
            T = sigmoid(T + self.hb(i))
            mu_new.append(T)
        for i in range(L):
            self.P['mu_new' + _sfx(i)] = mu_new[i].copy()
        mu = [self.P['mu' + _sfx(i)] for i in range(L)]
        step = 0
        def resid():
            r = max(np.max(np.abs(u - w)) for u, w in zip(mu, mu_new))
            return self.allreduce_max(r) if self.allreduce_max is not None else r
        while step < self.max_mf and resid() > self.mf_tol:                    # :449-452
            _, out = self.gibbs_step(X, mu, update_v=False, sample=False)      # :455 (reads mu, overwrites mu_new)
            mu, mu_new = out, mu                                                # :457 swap
            step += 1
        for i in range(L):                                                      # :477
            self.P['mu' + _sfx(i)] = mu[i]
        return step

    def particles_update(self, k, sample=True, t0=0):
        """_make_particles_update (dbm.py:480-509): k sweeps with swap; returns (v, H) and assigns when asked."""
        v, H = self.P['v'], [self.P['h' + _sfx(i)] for i in range(self.L)]
        for t in range(k):
            v, H = self.gibbs_step(v, H, update_v=True, sample=sample, t=t0 + t)
        return v, H

    def train_step(self, X, lr, mom, k):
        """session.run(train_op) (dbm.py:515-621); returns (n_mf, msre)."""
        n_mf, msre, S = self.raw_sums(X, k)
        self.apply(S, float(self.N), float(self.M), lr, mom)
        self.call += 1
        return n_mf, msre

    def raw_sums(self, X, k):
        """mean-field + PCD, then the UN-normalised sums the update needs (what a data-parallel rank
        contributes to the all-reduce): pos/neg outer products per layer and the column sums."""
        P, L = self.P, self.L
        n_mf = self.mean_field(X)                                               # :517
        v, H = self.particles_update(k)                                         # :521
        P['v'] = v
        for i in range(L):
            P['h' + _sfx(i)] = H[i]
        mu = [P['mu' + _sfx(i)] for i in range(L)]
        msre = np.mean((X - sigmoid(mu[0].dot(self.W(0).T) + P['vb'])) ** 2)    # :625-630 (W before the update)
        S = dict(sX=np.sum(X, axis=0), sv=np.sum(v, axis=0))
        for i in range(L):
            below_p, below_n = (X, v) if i == 0 else (mu[i - 1], H[i - 1])
            S['pos%d' % i] = below_p.T.dot(mu[i])
            S['neg%d' % i] = below_n.T.dot(H[i])
            S['smu%d' % i] = np.sum(mu[i], axis=0)
            S['sH%d' % i] = np.sum(H[i], axis=0)
        return n_mf, msre, S

    def apply(self, S, N, M, lr, mom):
        """the parameter update of dbm.py:550-621 from (possibly all-reduced) raw sums and the GLOBAL N, M"""
        P, L = self.P, self.L
        dvb = S['sX'] / N - S['sv'] / M                                         # :553
        dW = [S['pos%d' % i] / N - S['neg%d' % i] / M - self.l2 * self.W(i) for i in range(L)]    # :558-569
        dhb = [S['smu%d' % i] / N - S['sH%d' % i] / M for i in range(L)]        # :573-576
        d = self.sp_damping
        for i in range(L):                                                      # :580-592
            q_means = S['sH%d' % i]
            q_update = d * P['q_means' + _sfx(i)] + (1 - d) * q_means[i]        # q_means[i]: scalar (layer index!)
            P['q_means' + _sfx(i)] = q_update
            mu_means = S['smu%d' % i]
            mu_update = d * P['mu_means' + _sfx(i)] + (1 - d) * mu_means[i]
            P['mu_means' + _sfx(i)] = mu_update
            pen = self.sp_cost[i] * (q_update - self.sp_target[i])
            pen = pen + self.sp_cost[i] * (mu_update - self.sp_target[i])
            dW[i] = dW[i] - pen
            dhb[i] = dhb[i] - pen
        P['dvb'] = lr * (mom * P['dvb'] + dvb)                                  # :597
        P['vb'] = P['vb'] + P['dvb']
        for i in range(L):                                                      # :601-609
            P['dW' + _sfx(i)] = lr * (mom * P['dW' + _sfx(i)] + dW[i])
            Wu = self.W(i) + P['dW' + _sfx(i)]
            nrm = np.sqrt(np.sum(Wu ** 2, axis=0))                              # :511-513
            P['W' + _sfx(i)] = Wu * np.minimum(nrm, self.max_norm) / np.maximum(nrm, 1e-8)
        for i in range(L):                                                      # :611-615
            P['dhb' + _sfx(i)] = lr * (mom * P['dhb' + _sfx(i)] + dhb[i])
            P['hb' + _sfx(i)] = self.hb(i) + P['dhb' + _sfx(i)]

    def sample_v(self, k):
        """_make_sample_v (dbm.py:641-648): k sampled sweeps (assigned), k mean sweeps, v <- v_means."""
        v, H = self.particles_update(k)
        self.P['v'] = v
        for i in range(self.L):
            self.P['h' + _sfx(i)] = H[i]
        v_means, _ = self.particles_update(k, sample=False, t0=k)
        self.P['v'] = v_means
        self.call += 1
        return v_means

    # ---- AIS (dbm.py:650-736), 2-layer Bernoulli DBM
    def log_p_H0(self, x, beta):
        """_unnormalized_log_prob_H0 (:650-660)"""
        lp = x.dot(self.hb(0)) * beta
        lp = lp + np.sum(softplus((x.dot(self.W(0).T) + self.P['vb']) * beta), axis=1)
        lp = lp + np.sum(softplus((x.dot(self.W(1)) + self.hb(1)) * beta), axis=1)
        return lp

    def ais_next(self, x, beta, k, seed, step, chain0):
        """_make_ais_next_sample (:662-694)"""
        for t in range(k):
            v = sigmoid(beta * x.dot(self.W(0).T) + beta * self.P['vb'])
            if self.smp_v:
                v = self._draw(v, 12, t, seed, step, chain0)
            h2 = sigmoid(beta * x.dot(self.W(1)) + beta * self.hb(1))
            if self.smp_h[1]:
                h2 = self._draw(h2, 8 + 1, t, seed, step, chain0)
            x = sigmoid(beta * (v.dot(self.W(0)) + h2.dot(self.W(1).T)) + beta * self.hb(0))
            if self.smp_h[0]:
                x = self._draw(x, 8 + 0, t, seed, step, chain0)
        return x

    def ais(self, n_betas, n_runs, k, seed, chain0=0):
        """_make_ais (:696-736) with delta_beta = 1/n_betas (:929); beta accumulates in the model dtype like the graph
        (`self.real`: float32 unless a test sets float64, together with a float64 `uniform0`)."""
        R = getattr(self, 'real', np.float32)
        H1 = self.W(0).shape[1]
        uniform0 = getattr(self, 'uniform0', philox.uniform)
        u = uniform0(seed, 13, 0, n_runs * H1, idx0=chain0 * H1).reshape(n_runs, H1)
        x = (u < R(0.5)).astype(np.float64)                                     # :699-702
        db = R(1.0) / R(n_betas)
        x = self.ais_next(x, float(db), k, seed, 0, chain0)                     # :704-705
        log_Z = -self.log_p_H0(x, 0.)                                           # :708
        beta, step = db, 1
        while beta < R(1.) - db + R(1e-5):                                      # :710-711
            log_Z = log_Z + self.log_p_H0(x, float(beta))                       # :714
            x = self.ais_next(x, float(R(beta + db)), k, seed, step, chain0)    # :716
            log_Z = log_Z - self.log_p_H0(x, float(beta))                       # :718
            beta = R(beta + db)
            step += 1
        log_Z = log_Z + self.log_p_H0(x, 1.)                                    # :728
        V, H2 = self.W(0).shape[0], self.W(1).shape[1]
        return log_Z + (V + H1 + H2) * float(np.log(np.float32(2.)))            # :731-734: tf.log(2.) is a float32 node in every dtype

    def log_proba(self, X):
        """_make_log_proba (:738-759): ELBO terms per row (log Z not subtracted)."""
        self.mean_field(X)
        mu0, mu1 = self.P['mu'], self.P['mu_1']
        mE = np.sum(X.dot(self.W(0)) * mu0, axis=1)
        mE = mE + np.sum(mu0.dot(self.W(1)) * mu1, axis=1)
        mE = mE + X.dot(self.P['vb']) + mu0.dot(self.hb(0)) + mu1.dot(self.hb(1))
        s1, s2 = np.clip(mu0, 1e-7, 1. - 1e-7), np.clip(mu1, 1e-7, 1. - 1e-7)
        S1 = -s1 * np.log(s1) - (1. - s1) * np.log(1. - s1)
        S2 = -s2 * np.log(s2) - (1. - s2) * np.log(1. - s2)
        self.call += 1
        return mE + np.sum(S1, axis=1) + np.sum(S2, axis=1)


def dbm_exact_log_Z(W0, W1, vb, hb0, hb1):
    """log sum_{v,h1,h2} exp(v.vb + h1.hb0 + h2.hb1 + v'W0 h1 + h1'W1 h2) by enumeration of h1
    (v and h2 summed analytically: prod (1 + exp(.))), float64.  Ground truth for AIS (tex/chapter_3)."""
    H1 = W0.shape[1]
    n = 1 << H1
    h1 = ((np.arange(n)[:, None] >> np.arange(H1)[None, :]) & 1).astype(np.float64)
    lp = h1.dot(hb0) + np.sum(softplus(h1.dot(W0.T) + vb), axis=1) + np.sum(softplus(h1.dot(W1) + hb1), axis=1)
    m = lp.max()
    return m + np.log(np.sum(np.exp(lp - m)))
