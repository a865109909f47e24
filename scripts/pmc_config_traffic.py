#!/usr/bin/env python
"""HBM traffic PER ADMM ITERATION of one BASELINE config from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only,
separate runs as MI355X_MICROARCH.md prescribes) of `scripts/bench_configs.py <config>`:

    pmc_config_traffic.py <config> fetch.db write.db bench.jsonl out.md [--update profiles/pmc_traffic.json]

Sums FETCH_SIZE / WRITE_SIZE over every launch of the config's LOOP kernels (name patterns below; setup kernels are listed but not
counted), corrects them as the guide says for gfx950 (read = 2 x FETCH_SIZE x 1024 bytes, write = WRITE_SIZE x 1024 bytes) and divides
by the iteration count the profiled run reported.  With --update the result goes into profiles/pmc_traffic.json under
configs.<config>, together with the sha256 of the source files the loop kernels live in: bench.py quotes it as roofline.traffic
only while those files are unchanged."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOP = {
    "c3": (["wide_x_kernel", "wide_tail_kernel", "wide_rows_persist_kernel", "wide_ax_kernel", "wide_t_kernel", "wide_state_kernel"], ["lasso_wide.hip"]),
    "c4": (["gemv_t_batch_kernel", "par_A_batch_resid_kernel", "gather_batch_kernel", "par_head_kernel", "par_pack_kernel", "par_z_kernel", "par_wb_", "par_gather_fix_kernel"], ["padmm_lasso.hip", "gemv_kernels.h", "gather_kernels.h"]),
    "c5lad": (["lad_rows_kernel", "gemv_t_kernel<double", "reduce_partials_kernel<double", "dense_head_kernel", "dense_tail_kernel"], ["fadmm_dense.hip", "gemv_kernels.h"]),
    "c5bp": (["gemv_t_kernel<double", "bp_gather_kernel", "dense_head_kernel", "dense_tail_kernel"], ["fadmm_dense.hip", "gemv_kernels.h", "gather_kernels.h"]),
    "c5parbp": (["sbp_xreg_kernel", "sbp_xreg_screen_kernel", "sbp_list_kernel", "sbp_xact_kernel", "sbp_tail_kernel", "sbp_gs_", "gather_batch_kernel"], ["sharing_bp.hip", "gather_kernels.h"]),
    "dantzig": (["dz_head_kernel", "dz_mid_kernel", "dz_tail_kernel", "gemv_t_kernel<double", "reduce_partials_kernel<double"], ["dantzig.hip", "gemv_kernels.h"]),
}


def source_sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "admm_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def load(path, counter):
    db = sqlite3.connect(path)
    q = "select kernel_name, count(*), sum(value), max(value) from counters_collection where counter_name = ? group by kernel_name"
    return {r[0]: r[1:] for r in db.execute(q, (counter,)).fetchall()}


def main():
    cfg, fdb, wdb, bench, out = sys.argv[1:6]
    pats, files = LOOP[cfg]
    f, w = load(fdb, "FETCH_SIZE"), load(wdb, "WRITE_SIZE")
    line = [json.loads(l) for l in open(bench) if l.startswith("{")][-1]
    iters = int(line["iterations"])
    rows, rd, wr = [], 0.0, 0.0
    for k in sorted(f, key=lambda k: -f[k][1]):
        if "admm::" not in k:
            continue
        loop = any(p in k for p in pats)
        r_b, w_b = 2.0 * f[k][1] * 1024.0, w.get(k, (0, 0.0, 0.0))[1] * 1024.0
        if loop:
            rd += r_b
            wr += w_b
        rows.append(f"| `{k[:90]}` | {'loop' if loop else 'setup'} | {f[k][0]} | {r_b / f[k][0] / 1e6:.2f} | {w_b / max(f[k][0], 1) / 1e6:.2f} | {r_b / 1e9:.2f} | {w_b / 1e9:.2f} |")
    per_it = (rd + wr) / iters
    head = (f"{line['config']}\n\n{iters} iterations in the profiled run.  HBM bytes of the LOOP kernels: read {rd / 1e9:.2f} GB + written {wr / 1e9:.2f} GB "
            f"= **{per_it / 1e6:.1f} MB per iteration** (read {rd / iters / 1e6:.1f} + written {wr / iters / 1e6:.1f}); algorithmic bytes per iteration as bench.py counts them: "
            f"{line['alg_GB_per_iter'] * 1e3:.1f} MB -> traffic / algorithmic = {per_it / (line['alg_GB_per_iter'] * 1e9):.3f}.\n"
            "(rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate kernel-trace passes; read = 2 x FETCH_SIZE x 1024, write = WRITE_SIZE x 1024: gfx950 corrections of MI355X_MICROARCH.md)\n\n"
            "| kernel | part | launches | HBM read MB/launch | HBM write MB/launch | read GB total | written GB total |\n|---|---|---|---|---|---|---|\n")
    open(out, "w").write(head + "\n".join(rows) + "\n")
    print(head.split("\n\n")[1])
    if "--update" in sys.argv:
        path = sys.argv[sys.argv.index("--update") + 1]
        pmc = json.load(open(path))
        pmc.setdefault("configs", {})[cfg] = {"hbm_read_bytes_per_iteration": rd / iters, "hbm_write_bytes_per_iteration": wr / iters, "iterations": iters,
                                              "source": os.path.join("profiles", os.path.basename(out)), "kernel_source_sha16": source_sha(files), "source_files": files}
        json.dump(pmc, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
