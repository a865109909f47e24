"""GPU: the C ABI is re-entrant per call (threading row; `Lasso.cpp:42-50`: every .Call owns its solver).
Four host threads call admm_hip_lasso (tall and wide), admm_hip_enet, admm_hip_lad and admm_hip_bp AT ONCE on different problems,
several rounds; every result must be bit-equal to the same call made alone.  What this exercises in the library: the per-thread
stream pools, the per-thread pinned read-back buffer and host-to-device staging ring, the process-wide device info cache and
error string, and the kernels of different solvers interleaving on one device."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problems():
    from admm_amd import admm_bp, admm_enet, admm_lad, admm_lasso
    rng = np.random.default_rng(77)

    def lasso(n, p, m):
        x = rng.standard_normal((n, p)) * 2.0
        b = np.concatenate([rng.uniform(size=m), np.zeros(p - m)])
        return x, x @ b + rng.standard_normal(n)

    xt, yt = lasso(900, 260, 30)          # tall
    xw, yw = lasso(240, 1500, 20)         # wide
    xe, ye = lasso(700, 300, 25)          # elastic net
    xl = rng.standard_normal((2600, 40)) * 2 + 0.3
    yl = xl @ rng.uniform(size=40) + rng.standard_t(3, size=2600)
    a = rng.standard_normal((150, 500))
    bt = np.zeros(500); bt[rng.choice(500, 14, replace=False)] = rng.uniform(size=14)
    jobs = {
        "tall": lambda: admm_lasso(xt, yt).penalty(nlambda=12).fit(),
        "wide": lambda: admm_lasso(xw, yw).penalty(nlambda=8).fit(),
        "enet": lambda: admm_enet(xe, ye).penalty(nlambda=10, alpha=0.5).fit(),
        "lad": lambda: admm_lad(xl, yl).fit(),
        "bp": lambda: admm_bp(a, a @ bt).fit(),
        "par": lambda: admm_lasso(xt, yt).penalty(nlambda=6).parallel(3).fit(),
    }
    return jobs


def _key(fit):
    beta = fit.beta_dense if hasattr(fit, "beta_dense") else (np.asarray(fit.beta.todense()).ravel() if hasattr(fit.beta, "todense") else np.asarray(fit.beta))
    return np.ascontiguousarray(beta).tobytes(), np.asarray(fit.niter).tobytes()


def test_concurrent_calls_are_bit_equal_to_sequential_ones():
    jobs = _problems()
    alone = {k: _key(f()) for k, f in jobs.items()}
    names = list(jobs)
    rounds, nthreads = 3, 4
    errors, results = [], {}
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            start.wait()
            for r in range(rounds):
                for j in range(len(names)):
                    name = names[(t + j + r) % len(names)]            # every thread starts on a different solver, all run all of them
                    results[(t, r, name)] = _key(jobs[name]())
        except Exception as e:                                        # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) == nthreads * rounds * len(names)
    bad = [k for k, v in results.items() if v != alone[k[2]]]
    assert not bad, ("results of concurrent calls differ from the same calls made alone", bad[:8])


def test_two_threads_run_two_different_variants_at_the_same_time():
    """Variant selectors belong to the calling thread (admm_hip_options_set / admm_hip_option_set, round 6; they were process-wide
    environment variables): thread A fits the consensus problem in the one-pass form, thread B the SAME problem in the reference's
    two-pass form, and a tall problem with / without the refinement, concurrently and repeatedly.  Every result must be bit-equal to
    the same variant run alone, the two variants must really have been different executions, and the options of one thread must
    never show in the other."""
    import admm_amd
    from admm_amd import admm_lasso
    rng = np.random.default_rng(12)
    x = rng.standard_normal((403, 900)) * 2
    y = x[:, :15] @ rng.uniform(size=15) + rng.standard_normal(403)
    job = lambda: admm_lasso(x, y).penalty(nlambda=4, lambda_min_ratio=0.2).parallel(4).opts(maxit=300).fit()      # noqa: E731  (Woodbury workers)
    variants = {"one-pass": {}, "two-pass": dict(consensus_two_pass=1)}
    alone = {}
    for name, fields in variants.items():
        admm_amd.options.struct(**fields)
        alone[name] = _key(job())
        admm_amd.options.reset()
    assert alone["one-pass"] != alone["two-pass"], "the two variants are different float executions of one algorithm"
    errors, results = [], {}
    start = threading.Barrier(2)

    def worker(name):
        try:
            admm_amd.options.struct(**variants[name])                  # this thread's options only
            assert (admm_amd.load().admm_hip_option_get(b"PAR_ONEPASS") is not None) == (name == "two-pass")
            start.wait()
            for r in range(6):
                results[(name, r)] = _key(job())
        except Exception as e:                                        # noqa: BLE001
            errors.append((name, repr(e)))

    th = [threading.Thread(target=worker, args=(n,)) for n in variants]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    bad = [k for k, v in results.items() if v != alone[k[0]]]
    assert len(results) == 12 and not bad, bad
    assert admm_amd.load().admm_hip_option_get(b"PAR_ONEPASS") is None      # nothing leaked into the main thread
