"""Dev tool (GPU box): the persistent active-set stretches of the wide solver against the two-launch path on the C3 shape, for
several column limits.  The two paths group the partial sums of A x differently (32 against 256 workgroup partials), so
they are separate float executions: deterministic each (the same limit twice gives identical bits), equal to summation
rounding, and late lambdas may stop an iteration apart (measured: 18 447 iterations against 18 444 / 18 445, first
difference at lambda 87 resp. 72 of 100).  wide_persist_check.py [n p nlambda]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from admm_amd import DevicePtr, admm_lasso
    n, p, nl = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(123)
    xt = torch.empty((p, n), dtype=torch.float64, device=dev)
    for c0 in range(0, p, 20000):
        c1 = min(p, c0 + 20000)
        xt[c0:c1] = torch.randn((c1 - c0, n), generator=g, device=dev, dtype=torch.float64)
    b = torch.zeros(p, dtype=torch.float64, device=dev); b[:100] = torch.rand(100, generator=g, device=dev, dtype=torch.float64)
    y = b @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
    torch.cuda.synchronize()
    fit = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=nl).fit()
    np.savez(sys.argv[5], niter=fit.niter, beta=fit.beta_dense, persist=np.array([fit.stats["persist_iter"]]))
else:
    import numpy as np
    n, p, nl = (sys.argv[1:4] + ["2000", "200000", "100"][len(sys.argv) - 1:])[:3] if len(sys.argv) > 1 else ("2000", "200000", "100")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    ref = None
    for tag, env in [("two-launch", {"ADMM_HIP_WIDE_PERSIST": "0"}), ("512", {}), ("512 again", {}), ("640", {"ADMM_HIP_WIDE_PERSIST_COLS": "640"}),
                     ("768", {"ADMM_HIP_WIDE_PERSIST_COLS": "768"}), ("768 again", {"ADMM_HIP_WIDE_PERSIST_COLS": "768"}),
                     ("1024", {"ADMM_HIP_WIDE_PERSIST_COLS": "1024"}), ("2048", {"ADMM_HIP_WIDE_PERSIST_COLS": "2048"})]:
        out = "/tmp/wpc.npz"                                  # 80 MB of coefficients: not into gpurun_out
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", n, p, nl, out], env=dict(os.environ, **env), check=True)
        r = dict(np.load(out))
        if ref is None:
            ref = r
        same_n = np.array_equal(r["niter"], ref["niter"]); same_b = np.array_equal(r["beta"], ref["beta"])
        first = int(np.nonzero(r["niter"] != ref["niter"])[0][0]) if not same_n else -1
        print(f"{tag:12s} iterations {int(r['niter'].sum())} in persistent stretches {int(r['persist'][0])}  niter identical {same_n} beta identical {same_b}"
              + (f"  first differing lambda {first}: {int(r['niter'][first])} vs {int(ref['niter'][first])}" if first >= 0 else ""), flush=True)
