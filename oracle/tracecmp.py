"""Lock-step comparison of two decision traces (libadmm_hip's and the NumPy oracle's) -- TEST INFRASTRUCTURE."""
import numpy as np


def compare_traces(trace_gpu, trace_ref):
    """Walk the GPU decision trace (include/admm_hip.h, ADMM_TRACE_*; its first record is the cold-start decision) and
    the oracle's (FADMM.trace) in lock step.  Returns a dict:
      first_div   index (in oracle records) of the first decision that differs, or None
      div_lambda  lambda index of that decision
      kind        'stop' (converged on one side only) or 'restart' (accelerate vs restart)
      margin_gpu / margin_ref   |q / threshold - 1| of the deciding quantity on each side: q = max(r_p/eps_p, r_d/eps_d)
                  for 'stop', c / (0.999 c_old) for 'restart'
      max_dev     largest deviation between the two traces before first_div, each quantity on the scale its test uses:
                  |d eps| / eps for the two thresholds, |d r| / max(r, eps) for the two residuals,
                  |d c| / (0.999 c_old) for the combined residual
    """
    tg = np.asarray(trace_gpu, dtype=np.float64)
    if len(tg) and tg[0, 8] == -1:
        tg = tg[1:]
    tr = np.asarray(trace_ref, dtype=np.float64)
    out = dict(first_div=None, div_lambda=None, kind=None, margin_gpu=0.0, margin_ref=0.0, max_dev=0.0)

    def margins(rec):
        stop = max(rec[4] / rec[2] if rec[2] > 0 else np.inf, rec[5] / rec[3] if rec[3] > 0 else np.inf)
        return stop, (rec[6] / (0.999 * rec[7]) if rec[7] > 0 else np.inf)

    n = min(len(tg), len(tr))
    for k in range(n):
        g, r = tg[k], tr[k]
        assert g[0] == r[0] and g[1] == r[1], ("trace positions out of step before any decision differed", k, g[:2], r[:2])
        if g[8] != r[8]:
            sg, cg = margins(g)
            sr, cr = margins(r)
            out["first_div"], out["div_lambda"] = k, int(r[0])
            if (g[8] == 0) != (r[8] == 0):
                out["kind"], out["margin_gpu"], out["margin_ref"] = "stop", abs(sg - 1), abs(sr - 1)
            else:
                out["kind"], out["margin_gpu"], out["margin_ref"] = "restart", abs(cg - 1), abs(cr - 1)
            break
        devs = []
        if r[2] > 0 and r[3] > 0:
            devs += [abs(g[2] - r[2]) / r[2], abs(g[3] - r[3]) / r[3],
                     abs(g[4] - r[4]) / max(r[2], r[4]), abs(g[5] - r[5]) / max(r[3], r[5])]
        if r[8] != 0 and r[7] > 0:
            devs.append(abs(g[6] - r[6]) / (0.999 * r[7]))
        if devs:
            out["max_dev"] = max(out["max_dev"], max(devs))
    if out["first_div"] is None:
        assert len(tg) == len(tr), (len(tg), len(tr))
        out["summary"] = f"all {n} decisions identical, scalars within {out['max_dev']:.1e}"
    else:
        out["summary"] = (f"first differing decision: record {out['first_div']} (lambda {out['div_lambda']}), {out['kind']} test, "
                          f"distance to threshold gpu {out['margin_gpu']:.1e} ref {out['margin_ref']:.1e}; scalars before within {out['max_dev']:.1e}")
    return out
