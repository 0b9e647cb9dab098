"""Stream addressing of the reference's UNSEEDED random ops when it runs on the TF-1 shim (tests/tf1_shim).

Real TF 1.3 keys such an op with (graph seed, id of the op in the graph) - unknowable without TF.  The engine
 replaces that by a documented table: key = graph seed of the public call (`tf.set_random_seed(
model.make_random_seed())`, base/tf_model.py:20-21), counter = (block, site + 16 t, call).  This module maps the
reference's OWN graph structure - the name scopes its builders open (rbm/base_rbm.py:329-378, dbm.py:385-427), the
order in which they create ops, the `tf.while_loop` iteration a draw happens in, the number of `session.run`s that
drew random numbers before - onto that table.  Test infrastructure (fixture generation only).

site  reference op                                                        where
   1  tf.nn.dropout mask                                                  base_rbm.py:417-418
   2  h0 samples                                                          base_rbm.py:422 (`gibbs_chain/sample_h_given_v`)
   3  v samples of Gibbs step t                                           base_rbm.py:371-372 (`gibbs_step[_t]/sample_v_given_h`)
   4  h samples of Gibbs step t                                           base_rbm.py:375-376
   5  PLL flip index (int32 uniform)                                      base_rbm.py:501
   6  Multinomial h_hat of MultinomialRBM._free_energy (t = 0 free_energy_op, 1 F(x), 2 F(x corrupted))   rbm.py:56
 8+i  DBM hidden layer i samples, sweep t                                 dbm.py:394-416
  12  DBM visible samples, sweep t                                        dbm.py:423-425
  13  AIS x_0 ~ Ber(1/2)  (the op carries a seed drawn at graph build, dbm.py:701; addressed like the others)
  20  particle initialisers (`layer.init`): 20 v, 21 v_new, 22 + 2i h_i, 23 + 2i h_new_i      dbm.py:362-383
AIS transitions (dbm.py:662-694) use sites 12 (v), 9 (h2), 8 (x_hat) with t = Gibbs step inside the transition and
call = index of the beta step (0 for x_1 ~ T_1(. | x_0), then 1, 2, ... inside the loop of dbm.py:713-726).
"""
import re

DEFAULT_GRAPH_SEED = 87654321


def _suffix(component, base):
    m = re.match(r'^%s(?:_(\d+))?$' % re.escape(base), component)
    if not m:
        return None
    return int(m.group(1) or 0)


def _siblings(site, pred):
    """creation-ordered random ops of the graph that satisfy pred(node)"""
    return sorted((n for n in site.node.graph.nodes
                   if n.op in ('random_uniform', 'random_normal', 'multinomial') and pred(n)),
                  key=lambda n: n.creation_index)


def policy(site):
    key = site.graph_seed if site.graph_seed is not None else DEFAULT_GRAPH_SEED
    comps = site.scope.split('/') if site.scope else []
    inner_it = site.loop_iterations[-1] if site.loop_iterations else 0

    # ---- DBM particle initialisers (evaluated by global_variables_initializer) -------------------------------
    if site.initializer and 'negative_particles' in comps:
        if site.node.attrs.get('seed') is not None:
            return None
        top = [c for c in comps if c.startswith('h_particle')]
        if not top:                                   # v, v_new (v absent when v_particle_init is given)
            ops = _siblings(site, lambda n: n.scope.split('/')[:1] == ['negative_particles'] and
                            not any(c.startswith('h_particle') for c in n.scope.split('/')))
            idx = ops.index(site.node) + (2 - len(ops))
            return key, 20 + idx, 0
        layer = _suffix(top[0], 'h_particle')
        ops = _siblings(site, lambda n: top[0] in n.scope.split('/'))
        idx = ops.index(site.node) + (2 - len(ops))
        return key, 22 + 2 * layer + idx, 0
    if site.node.attrs.get('seed') is not None and 'annealed_importance_sampling' not in comps[:1]:
        return None                                   # TF-literal stream of a seeded op (W init)

    # ---- AIS (dbm.py:696-736) ---------------------------------------------------------------------------------
    if comps[:1] == ['annealed_importance_sampling']:
        ops = _siblings(site, lambda n: n.scope.split('/')[:1] == ['annealed_importance_sampling'])
        i = ops.index(site.node)
        if i == 0:
            return key, 13, 0
        which = (i - 1) % 3                           # creation order inside _make_ais_next_sample: v, h2, x_hat
        first_transition = (i - 1) < 3
        step = 0 if first_transition else site.loop_iterations[0] + 1
        return key, (12, 9, 8)[which] + 16 * inner_it, step

    call = site.call
    # ---- RBM graph ------------------------------------------------------------------------------------------------
    if site.role == 'dropout':
        return key, 1, call
    if 'pseudo_loglik' in comps and site.kind == 'random_uniform' and site.dtype.kind == 'i':
        return key, 5, call
    fe = [c for c in comps if _suffix(c, 'free_energy') is not None]
    if fe and site.kind == 'multinomial':
        if 'pseudo_loglik' in comps:
            t = 2 if _suffix(fe[0], 'free_energy') == 0 else 1
        else:
            t = 0
        return key, 6 + 16 * t, call
    step = [(_suffix(c, 'gibbs_step')) for c in comps if _suffix(c, 'gibbs_step') is not None]
    if 'sample_h_given_v' in comps:
        if step:
            return key, 4 + 16 * (step[0] + inner_it), call
        return key, 2, call
    if 'sample_v_given_h' in comps:
        return key, 3 + 16 * (step[0] + inner_it if step else inner_it), call

    # ---- DBM sweeps (dbm.py:385-427) ---------------------------------------------------------------------------
    for c in comps:
        m = re.match(r'^sample_h(\d+)_hat_given', c)
        if m:
            return key, 8 + int(m.group(1)) + 16 * inner_it, call
        if c.startswith('sample_v_hat_given_h_hat'):
            return key, 12 + 16 * inner_it, call
    raise RuntimeError('no stream rule for random op %s (scope %r, role %r)' % (site.name, site.scope, site.role))
