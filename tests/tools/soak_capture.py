"""Dev tool (GPU box): soak of the strict trace-based parity rule (tests/helpers.py) over many seeds of the random small
problems of tests/fuzz_cases.py, several worker processes sharing the GPU.  Unlike soak.sh it KEEPS its evidence:

  <out>/cases.jsonl      one line per judged case: seed, case, kind, sizes, decisions, near-ties taken from the GPU by kind,
                         the ulps the largest one needed, pass / fail and -- for a failure -- the rule's message and the
                         band that WOULD have been needed (the case judged again with an unbounded band)
  <out>/fail_s<seed>_c<case>.npz   the GPU's outputs for a failing case (beta, niter, decision trace): with the seed this
                         is everything needed to judge the case again on a CPU (tests/tools/soak_replay.py)
  <out>/summary.md       totals, the distribution of near-ties, the failures -- copied to profiles/r03_soak_summary.md

   python tests/tools/soak_capture.py <first_seed> <nseeds> [ncases=150] [procs=16] [out=gpurun_out/soak_r03]
   python tests/tools/soak_capture.py worker <seed> <ncases> <out>          (what the master spawns)
"""
import json
import os
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _report_stats(rep):
    forced = (rep or {}).get("forced", []) or []
    by = {}
    for f in forced:
        by[f["kind"]] = by.get(f["kind"], 0) + 1
    ulps = max([f["ulps"] for f in forced if f["kind"] != "rho"], default=0.0)
    out = dict(forced=len(forced), forced_by_kind=by, max_ulps=float(ulps))
    if rep:
        out["max_err"] = float(rep.get("max_err", rep.get("err", 0.0)))
        out["loose"] = len(rep.get("loose", []) or [])
    return out


def worker(seed, ncases, out):
    import io
    import contextlib
    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process)
    from fuzz_cases import cases
    import test_gpu_fuzz as T
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, f"seed_{seed}.jsonl")
    with open(path, "w") as fh:
        for cs in cases(ncases, seed):
            orig = cs["kind"]
            if os.environ.get("SOAK_SWAP") == "1":              # round 6: the SAME data with the prox labels swapped (is a failure the label's or the data's?)
                if orig == "tall":
                    cs["kind"], cs["alpha"] = "enet_tall", [0.1, 0.5, 0.9, 1.0][(7 * seed + cs["c"]) % 4]
                elif orig == "enet_tall":
                    cs["kind"], cs["alpha"] = "tall", None
            rec = dict(seed=seed, case=cs["c"], kind=cs["kind"], orig_kind=orig, alpha=cs.get("alpha"), n=cs["n"], p=cs["p"], K=cs.get("K", 0), icpt=int(cs["icpt"]),
                       stdz=int(cs["stdz"]), scale=cs["scale"])
            t0 = time.time()
            try:
                # round 4: the wide, LAD and BP kinds carry their iterate dump too and are ALWAYS held to the stepwise rule as well
                always_sw = cs["kind"] in ("lad", "bp") or (cs["n"] <= cs["p"] and not (cs["kind"] == "par" and cs.get("K", 0) > 1))
                cap = T.gpu_capture(cs, state=always_sw)
                rec["decisions"] = int(len(cap["trace"]))
                rec["longest_lambda"] = int(np.asarray(cap["trace"])[:, 1].max()) + 1 if len(cap["trace"]) else 0
                rec["t_gpu"] = round(time.time() - t0, 3)
                buf = io.StringIO()
                try:
                    if always_sw and "state" in cap:
                        from oracle import stepcheck
                        sw = T.stepwise_capture(cs, cap)
                        if cs["kind"] in ("lad", "bp"):
                            stepcheck.assert_stepwise_dense(sw, label=T.case_label(cs))
                            rec["stepwise_ok"] = dict(records=sw["records"], x_vs_ref_max=float(sw["x_vs_ref_max"]), ties=len(sw.get("rounding_ties", [])))
                        else:
                            stepcheck.assert_stepwise_wide(sw, label=T.case_label(cs))
                            rec["stepwise_ok"] = dict(records=sw["records"], xt=float(sw["xt_ratio_max"]), ax=float(sw["ax_ratio_max"]), ties=len(sw.get("rounding_ties", [])))
                    with contextlib.redirect_stdout(buf):
                        rep = T.judge_capture(cs, cap, budget=False)          # the near-ties are RECORDED here, not bounded
                    rec.update(ok=True, judged=rep is not None, **_report_stats(rep))
                except Exception as e:                          # noqa: BLE001  (the rule said no)
                    rec.update(ok=False, judged=True, error=type(e).__name__, message=str(e)[:600])
                    np.savez_compressed(os.path.join(out, f"fail_s{seed}_c{cs['c']}.npz"), beta=cap["beta"], niter=cap["niter"], trace=cap["trace"])
                    if "state" in cap and "elementwise steps differ" in str(e):        # a bit mismatch of the stepwise rule: keep the dump that showed it
                        np.savez_compressed(os.path.join(out, f"state_s{seed}_c{cs['c']}.npz"), **{k: v for k, v in cap.items()})
                    if cs["kind"] in ("tall", "enet_tall") or (cs["kind"] == "par" and cs.get("K", 0) > 1):
                        try:                                    # the stronger, drift-free statement: every iteration on its own (oracle/stepcheck.py)
                            cap2 = T.gpu_capture(cs, state=True)
                            sw = T.stepwise_capture(cs, cap2)
                            rec["stepwise"] = dict(records=sw["records"], x_ratio_max=float(sw["x_ratio_max"]), x_vs_ref_max=float(sw.get("x_vs_ref_max", 0.0)),
                                                   bit_mismatch=len(sw["bit_mismatch"]), first_mismatch=[list(map(str, m)) for m in sw["bit_mismatch"][:3]],
                                                   accum_ties=len(sw["accum_ties"]), norm_rel_max=float(sw["norm_rel_max"]))
                            if sw["bit_mismatch"]:
                                np.savez_compressed(os.path.join(out, f"state_s{seed}_c{cs['c']}.npz"), **cap2)
                        except Exception as e3:                 # noqa: BLE001
                            rec["stepwise"] = dict(error=type(e3).__name__, message=str(e3)[:400])
                    try:                                        # what band WOULD have been needed (and what else fails then)
                        with contextlib.redirect_stdout(buf):
                            rep = T.judge_capture(cs, cap, band=1e9, budget=False)
                        rec["unbounded"] = dict(ok=True, **_report_stats(rep))
                    except Exception as e2:                     # noqa: BLE001
                        rec["unbounded"] = dict(ok=False, error=type(e2).__name__, message=str(e2)[:400])
            except Exception as e:                              # noqa: BLE001  (the library or the tool itself failed)
                rec.update(ok=False, judged=False, error=type(e).__name__, message=str(e)[:600], tb=traceback.format_exc()[-800:])
            rec["t"] = round(time.time() - t0, 3)
            fh.write(json.dumps(rec) + "\n")
            fh.flush()


def summarise(out, seeds, ncases, wall):
    recs = []
    for s in seeds:
        p = os.path.join(out, f"seed_{s}.jsonl")
        if os.path.exists(p):
            recs += [json.loads(ln) for ln in open(p)]
    with open(os.path.join(out, "cases.jsonl"), "w") as fh:
        for r in recs:
            fh.write(json.dumps(r) + "\n")
    judged = [r for r in recs if r.get("judged")]
    fails = [r for r in recs if not r.get("ok")]
    kinds = sorted({r["kind"] for r in recs})
    lines = ["# Soak of the trace-based parity rule (tests/tools/soak_capture.py)", "",
             f"seeds {seeds[0]}..{seeds[-1]} x {ncases} cases of tests/fuzz_cases.py (`cases(ncases, seed)`), band 8 ulps; "
             f"{len(recs)} cases run, {len(judged)} judged (consensus cases with K = 1 are the serial solver and are skipped), "
             f"**{len(judged) - len([r for r in fails if r.get('judged')])} pass, {len(fails)} fail**; wall {wall:.0f} s.", "",
             "| kind | judged | pass | decisions | near-ties taken (stop / restart / rho) | per 1000 decisions | largest ulps | cases with any |",
             "|---|---|---|---|---|---|---|---|"]
    for k in kinds:
        rk = [r for r in judged if r["kind"] == k]
        ok = [r for r in rk if r["ok"]]
        dec = sum(r.get("decisions", 0) for r in ok)
        st = sum(r.get("forced_by_kind", {}).get("stop", 0) for r in ok)
        rs = sum(r.get("forced_by_kind", {}).get("restart", 0) for r in ok)
        rh = sum(r.get("forced_by_kind", {}).get("rho", 0) for r in ok)
        mu = max([r.get("max_ulps", 0.0) for r in ok], default=0.0)
        nany = sum(1 for r in ok if r.get("forced", 0))
        lines.append(f"| {k} | {len(rk)} | {len(ok)} | {dec} | {st} / {rs} / {rh} | {1000.0 * (st + rs + rh) / max(dec, 1):.2f} | {mu:.2f} | {nany} |")
    allu = sorted(r.get("max_ulps", 0.0) for r in judged if r.get("ok") and r.get("forced", 0))
    if allu:
        import numpy as np
        q = np.quantile(allu, [0.5, 0.9, 0.99, 1.0])
        lines += ["", f"Largest near-tie per case (cases with at least one, n = {len(allu)}): median {q[0]:.2f}, 90 % {q[1]:.2f}, 99 % {q[2]:.2f}, max {q[3]:.2f} ulps."]
    sw_ok = [r for r in recs if r.get("stepwise_ok")]
    if sw_ok:
        nrec = sum(r["stepwise_ok"]["records"] for r in sw_ok)
        wide = [r["stepwise_ok"] for r in sw_ok if "xt" in r["stepwise_ok"]]
        dense = [r["stepwise_ok"] for r in sw_ok if "x_vs_ref_max" in r["stepwise_ok"]]
        lines += ["", f"Stepwise rule (oracle/stepcheck.py check_wide / check_dense) on every wide / LAD / BP case: {len(sw_ok)} cases, {nrec} iterations replayed, "
                      f"all elementwise steps bit-exact; wide mat-vecs at most {max([w['xt'] for w in wide], default=0):.2f} (X't) / {max([w['ax'] for w in wide], default=0):.2f} (A x) "
                      f"float-dot yardsticks; LAD / BP projection at most {max([d['x_vs_ref_max'] for d in dense], default=0):.2f} x the reference route's error; "
                      f"{sum(w['ties'] for w in wide) + sum(d['ties'] for d in dense)} decisions that are exact ties up to the order of a sum."]
    lines += ["", "## Failures", ""]
    if not fails:
        lines.append("none")
    for r in fails:
        ub = r.get("unbounded")
        sw = r.get("stepwise")
        lines.append(f"* seed {r['seed']} case {r['case']} `{r['kind']}` n={r['n']} p={r['p']} K={r['K']} icpt={r['icpt']} std={r['stdz']} "
                     f"scale={r['scale']:g}: {r.get('error')}: {r.get('message', '')[:300]}"
                     + (f" — with an unbounded band: {json.dumps(ub)[:300]}" if ub else "")
                     + (f" — **stepwise** (every iteration replayed from the library's own previous iterates): {json.dumps(sw)[:400]}" if sw else ""))
    open(os.path.join(out, "summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def main():
    if sys.argv[1] == "worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        return
    first, nseeds = int(sys.argv[1]), int(sys.argv[2])
    ncases = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "gpurun_out", "soak_r04")
    os.makedirs(out, exist_ok=True)
    seeds = list(range(first, first + nseeds))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t0 = time.time()
    budget = float(os.environ.get("SOAK_BUDGET_S", "1e9"))       # stop cleanly (and still summarise) inside a gpurun time limit
    pending, running = list(seeds), []
    while pending or running:
        if time.time() - t0 > budget:
            for _, pr in running:
                pr.terminate()                                   # exact children only
            print(f"budget of {budget:.0f} s used up: {len(pending)} seeds not started, {len(running)} cut short")
            break
        while pending and len(running) < procs:
            s = pending.pop(0)
            log = open(os.path.join(out, f"seed_{s}.log"), "w")
            running.append((s, subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(s), str(ncases), out],
                                                env=env, stdout=log, stderr=subprocess.STDOUT)))
        time.sleep(1.0)
        running = [(s, pr) for s, pr in running if pr.poll() is None]
    summarise(out, seeds, ncases, time.time() - t0)


if __name__ == "__main__":
    main()
