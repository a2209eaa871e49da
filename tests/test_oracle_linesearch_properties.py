"""Properties of the restated Ceres LineSearchMinimizer (oracle/orc_gmm.h: L-BFGS direction + Wolfe line search with cubic
interpolation; correlation.h:206-238 sets only max_num_iterations = 10), checked step by step on recorded correlation
problems with an INDEPENDENT numpy implementation of what the published algorithm guarantees -- Ceres is absent here, so the
individual steps cannot be compared with its own, but every accepted step must
  * satisfy the strong Wolfe conditions with Ceres' default constants (sufficient decrease 1e-4, curvature 0.9) along the
    direction it was taken in (Nocedal & Wright, Numerical Optimization, (3.7a, 3.7b)): what WolfeLineSearch promises
    whenever it returns success with a bracket;
  * point along the L-BFGS two-loop direction of the history so far (Nocedal & Wright, Algorithm 7.4, H0 = I: Ceres'
    use_approximate_eigenvalue_bfgs_scaling defaults to false), recomputed here from the iterates alone.
(The converged optimum is compared with scipy's BFGS in tests/test_oracle_crosschecks.py::test_gmm_gradient_and_optimum.)"""
import numpy as np


def _problems(cc, oracle):
    """(src scan, tgt scan, tf_init) of loop closures of a short synthetic sequence: tf_init = the accepted pose, nudged."""
    L = oracle.L
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    n = 64
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=32, azim=900)
    xs = x.numpy()
    res, _, _ = oracle.run_sequence(xs.reshape(-1, 4), np.arange(n + 1, dtype=np.int64) * x.shape[1], ts, np.arange(n, dtype=np.int32), dcfg=dcfg)
    hit = np.nonzero(res["n_res"] > 0)[0]
    assert len(hit) >= 6
    rng = np.random.default_rng(3)
    out = []
    for qi in hit[:8]:
        a = oracle.Scan(xs[int(res["cand_gidx"][qi])], int_id=int(res["cand_gidx"][qi]))
        b = oracle.Scan(xs[qi], int_id=int(qi))
        out.append((a, b, res["tf"][qi] + rng.normal(0, [0.5, 0.5, 0.02])))
    return out


def _two_loop(g, S, Y):
    q = g.copy()
    al = []
    for s, y in zip(reversed(S), reversed(Y)):
        a = (s @ q) / (y @ s)
        q -= a * y
        al.append(a)
    for (s, y), a in zip(zip(S, Y), reversed(al)):
        b = (y @ q) / (y @ s)
        q += s * (a - b)
    return -q


def test_accepted_steps_are_wolfe_steps_along_lbfgs_directions(cc, oracle):
    c1, c2 = 1e-4, 0.9
    n_steps = n_dir = 0
    for a, b, tf0 in _problems(cc, oracle):
        xs, term = oracle.gmm_trace(a, b, tf0)
        assert term in (0, 1, 2, 3) and len(xs) >= 2 and np.allclose(xs[0], tf0)
        f = lambda p: oracle.gmm_eval(a, b, tf0, p)[0]
        g = lambda p: oracle.gmm_eval(a, b, tf0, p)[1]
        S, Y = [], []
        for k in range(len(xs) - 1):
            x0, x1 = xs[k], xs[k + 1]
            p = x1 - x0
            f0, g0, f1, g1 = f(x0), g(x0), f(x1), g(x1)
            assert f1 <= f0 + c1 * (g0 @ p) + 1e-12 * abs(f0), ("sufficient decrease", k, f0, f1, g0 @ p)
            assert abs(g1 @ p) <= c2 * abs(g0 @ p) * (1 + 1e-9) + 1e-14, ("curvature", k, g1 @ p, g0 @ p)
            assert g0 @ p < 0                                                  # a descent direction
            d = _two_loop(g0, S, Y)                                            # what L-BFGS would walk along from x_k
            cosang = (d @ p) / (np.linalg.norm(d) * np.linalg.norm(p))
            assert cosang > 1 - 1e-8, ("direction", k, cosang)
            n_dir += int(k > 0)
            s, y = p, g1 - g0
            if s @ y > 1e-14:   # LowRankInverseHessian::Update keeps a pair only if it has positive curvature
                S.append(s)
                Y.append(y)
            n_steps += 1
        # the refinement improved the correlation
        assert f(xs[-1]) < f(xs[0])
    assert n_steps >= 20 and n_dir >= 10
