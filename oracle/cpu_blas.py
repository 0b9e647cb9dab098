"""NumPy float32 + BLAS restatement of the reference's CD-k train op (base_rbm.py:415-479), for SPEED.

TEST / BENCH INFRASTRUCTURE ONLY (like everything under oracle/): imported by bench.py's cpu_baseline
leg and by tests/test_oracle.py, never by the product.

The parity checker (bm_oracle.c) computes every dot product as one sequential fma chain, which no blocked
sgemm reproduces; it is therefore a slow CPU program.  BASELINE.md §2 asks for the CPU figure of "the same
maths built for speed": this file is that — the five GEMMs of a CD-1 update go to the BLAS numpy links
(OpenBLAS sgemm, all host threads), the elementwise work is vectorised NumPy, and the Bernoulli draws use
NumPy's own Philox bit generator (TF on CPU also draws from Philox; the draw ADDRESSING is not the pinned
one, which does not matter for a throughput baseline).  With `uniforms=` the pinned stream can be injected,
which is how tests/test_oracle.py checks this restatement against the C oracle.
"""
import numpy as np


def _sigmoid(x):
    # float32 throughout, in place where possible
    np.negative(x, out=x)
    np.exp(x, out=x)
    x += np.float32(1)
    np.reciprocal(x, out=x)
    return x


class BlasRBM(object):
    def __init__(self, W, l2=1e-4, sample_v=False, sample_h=True, sp_target=0.1, sp_cost=0., sp_damping=0.9, seed=0):
        V, H = W.shape
        f = np.float32
        self.W = np.ascontiguousarray(W, dtype=f)
        self.vb, self.hb = np.zeros(V, f), np.zeros(H, f)
        self.dW, self.dvb, self.dhb, self.q = np.zeros((V, H), f), np.zeros(V, f), np.zeros(H, f), np.zeros(H, f)
        self.l2, self.sample_v, self.sample_h = f(l2), sample_v, sample_h
        self.sp_target, self.sp_cost, self.sp_damping = f(sp_target), f(sp_cost), f(sp_damping)
        self.rng = np.random.Generator(np.random.Philox(seed))

    def _draw(self, p, uniforms):
        u = uniforms.pop(0) if uniforms is not None else self.rng.random(p.shape, dtype=np.float32)
        return (u < p).astype(np.float32)

    def train_step(self, X, lr, mom, k, uniforms=None):
        """one session.run(train_op); `uniforms`: optional list of [B, n] float32 arrays consumed in draw order
        (h0, then per step: v (if sample_v), h (if sample_h))"""
        f = np.float32
        X = np.ascontiguousarray(X, dtype=f)
        N = f(len(X))
        h0 = _sigmoid(X @ self.W + self.hb)                                  # :421
        h0s = self._draw(h0, uniforms)                                       # :422 (always drawn)
        hs = h0s if self.sample_h else h0                                    # :423
        for _ in range(k):                                                   # :367-378
            vm = _sigmoid(hs @ self.W.T + self.vb)
            vs = self._draw(vm, uniforms) if self.sample_v else vm
            hm = _sigmoid(vs @ self.W + self.hb)
            hs = self._draw(hm, uniforms) if self.sample_h else hm
        g = X.T @ h0                                                         # :447
        g -= vs.T @ hm                                                       # :448
        g /= N
        g -= self.l2 * self.W                                                # :449
        dvb = (X - vs).mean(axis=0, dtype=f)                                 # :451
        dhb = (h0 - hm).mean(axis=0, dtype=f)                                # :453
        self.q = self.sp_damping * self.q + (f(1) - self.sp_damping) * hm.sum(axis=0, dtype=f)   # :457-459 (SUM)
        pen = self.sp_cost * (self.q - self.sp_target)                       # :460
        dhb -= pen
        if self.sp_cost != 0:
            g -= pen                                                         # :462
        lr, mom = f(lr), f(mom)
        self.dW *= mom; self.dW += g; self.dW *= lr; self.W += self.dW       # :467-468
        self.dvb = lr * (mom * self.dvb + dvb); self.vb += self.dvb          # :470-471
        self.dhb = lr * (mom * self.dhb + dhb); self.hb += self.dhb          # :473-474
        return dict(h0=h0, vm=vm, vs=vs, hm=hm)
