"""Static audit of memory scopes on EXCHANGE memory (multi-GPU readiness that can be checked without a second GPU, VERDICT r4 task 5b).

The PEER back-end's payloads and flags live in other devices' memory (hipIpc-mapped): every store / load / wait on them must be at
SYSTEM scope -- an agent-scope atomic is only coherent inside one device and would pass every single-GPU test (producer and consumer
share the device there) and fail across xGMI.  Rules checked on the sources:
  1. comm.hip / peer_device.h: an atomic whose address is exchange memory (peer_flag / peer_dst_slot / peer_src_slot / a mapped remote
     or local buffer) names __HIP_MEMORY_SCOPE_SYSTEM; every __HIP_MEMORY_SCOPE_AGENT in these files is on a LOCAL object of the
     producing launch (the arrival counters `count[...]`, the error word `err`);
  2. every __global__ kernel that takes a PeerExchange / PeerArgs (the producers and consumers the `_dist` entry points launch over PEER)
     contains no agent- or workgroup-scope ATOMIC in its body and calls no device helper that contains one -- except the three
     exchange-layer helpers whose agent-scope objects rule 1 has vetted;
  3. the persistent stretch (wide_rows_persist_kernel), which DOES use agent scope by design for its hand-overs inside one device, touches
     exchange memory (round 6: the AUX region, in its column-sharded instantiation) through the system-scope helpers only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "admm_amd", "csrc")
VETTED_HELPERS = {"peer_publish", "peer_wait_relaxed", "peer_wait"}          # their AGENT-scope words are local counters / the error word (rule 1)


def _read(name):
    return open(os.path.join(CSRC, name)).read()


def _functions(src, qualifier):
    """(name, params, body) of every function whose declaration carries `qualifier` (brace matching; templates included)."""
    out = []
    for m in re.finditer(qualifier + r"[^;{]*?\b(\w+)\s*\(([^)]*)\)\s*\{", src, flags=re.S):
        i = m.end()
        depth = 1
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        out.append((m.group(1), m.group(2), src[m.end():i]))
    return out


def test_exchange_layer_uses_system_scope_on_exchange_memory():
    exch = re.compile(r"peer_flag\(|peer_dst_slot|peer_src_slot|\.remote\[|\blocal\b|\bdst\b|\bnd\b")
    local_ok = re.compile(r"count\[|\berr\b|\.err\b")
    seen_agent = 0
    for name in ("peer_device.h", "comm.hip"):
        for ln, line in enumerate(_read(name).splitlines(), 1):
            if "__hip_atomic" not in line:
                continue
            if "__HIP_MEMORY_SCOPE_AGENT" in line or "__HIP_MEMORY_SCOPE_WORKGROUP" in line:
                seen_agent += 1
                assert local_ok.search(line) and not re.search(r"peer_flag\(|peer_dst_slot|peer_src_slot|\.remote\[", line), (name, ln, line.strip())
            elif exch.search(line):
                assert "__HIP_MEMORY_SCOPE_SYSTEM" in line, (name, ln, line.strip())
    assert seen_agent >= 4                                    # the arrival counters and the error word are there (the rule is not vacuous)
    pd = _read("peer_device.h")
    for helper in ("peer_store_f32", "peer_store_u64", "peer_load_f32x2"):
        body = next(b for n, _, b in _functions(pd, "__device__") if n == helper)
        assert "__HIP_MEMORY_SCOPE_SYSTEM" in body and "SCOPE_AGENT" not in body, helper


def test_kernels_of_the_sharded_solvers_use_no_agent_scope_atomics():
    files = [f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    tainted = set()                                           # device helpers that contain an agent / workgroup-scope atomic
    for f in files:
        for name, _, body in _functions(_read(f), "__device__"):
            if re.search(r"__hip_atomic[^;]*__HIP_MEMORY_SCOPE_(AGENT|WORKGROUP)", body, flags=re.S):
                tainted.add(name)
    assert {"peer_publish", "peer_wait_relaxed"} <= tainted
    tainted -= VETTED_HELPERS
    peer_kernels = []
    for f in files:
        for name, params, body in _functions(_read(f), "__global__"):
            if "PeerExchange" in params or "PeerArgs" in params:
                peer_kernels.append((f, name))
                direct = re.findall(r"__hip_atomic[^;]*__HIP_MEMORY_SCOPE_(?:AGENT|WORKGROUP)[^;]*;", body, flags=re.S)
                if f == "comm.hip":                           # the generic push / sum kernels: only their local arrival counters (rule 1)
                    direct = [d for d in direct if "count[" not in d]
                assert not direct, (f, name, direct[:2])
                called = {t for t in tainted if re.search(r"\b" + t + r"\s*\(", body)}
                assert not called, (f, name, "calls a helper with an agent-scope atomic", called)
    names = {n for _, n in peer_kernels}
    assert {"par_pack_kernel", "par_z_kernel", "wide_tail_kernel", "wide_ax_push_kernel", "tall_tail_kernel", "peer_push_kernel", "peer_sum_kernel"} <= names, names


def test_the_persistent_stretch_touches_exchange_memory_at_system_scope_only():
    """wide_rows_persist_kernel hands its results from workgroup to workgroup of ONE device through agent-scope atomics by design (ps.*:
    flags, partial dots, norm shares -- all device-local allocations of the plan).  Round 6: its column-sharded instantiation also writes
    and reads the AUX region of the PEER exchange buffers (other devices' memory).  Checked on the source: every statement of the kernel
    that names AUX memory (aux_slot / aux_flag / ax_.remote / ax_.local) goes through the system-scope helpers of peer_device.h
    (peer_store_*, peer_load_*, aux_wait) and carries no agent- or workgroup-scope atomic; the agent-scope atomics that mention the
    exchange descriptor at all are on its two device-local words (the sequence number `ax_.seq`, the error word); those helpers are
    system scope themselves; and the host enqueues the sharded instantiation only over the PEER back-end."""
    wide = _read("lasso_wide.hip")
    _, params, body = next(f for f in _functions(wide, "__global__") if f[0] == "wide_rows_persist_kernel")
    assert "PeerAux" in params and "PeerExchange" not in params and "__HIP_MEMORY_SCOPE_AGENT" in body
    stmts = [st for st in body.split(";") if re.search(r"aux_slot|aux_flag|ax_\.remote|ax_\.local", st)]
    assert len(stmts) >= 5, len(stmts)
    for st in stmts:
        assert not re.search(r"__HIP_MEMORY_SCOPE_(AGENT|WORKGROUP)", st), st.strip()[:200]
        assert re.search(r"peer_store_f32|peer_store_u64|peer_load_f32|peer_load_u64", st), st.strip()[:200]
    for st in body.split(";"):
        if "ax_." in st and re.search(r"__hip_atomic[^;]*__HIP_MEMORY_SCOPE_(AGENT|WORKGROUP)", st, flags=re.S):
            assert "ax_.seq" in st, st.strip()[:200]
    pd = _read("peer_device.h")
    for helper in ("peer_load_f32", "peer_load_u64"):
        b = next(b for n, _, b in _functions(pd, "__device__") if n == helper)
        assert "__HIP_MEMORY_SCOPE_SYSTEM" in b and "SCOPE_AGENT" not in b, helper
    b = next(b for n, _, b in _functions(pd, "__device__") if n == "aux_wait")
    loads = re.findall(r"__hip_atomic_load\([^;]*\)", b)
    assert loads and all("__HIP_MEMORY_SCOPE_SYSTEM" in ld for ld in loads)
    assert re.search(r"persist_rows\s*=\s*\(!cshard\s*\|\|\s*\(peer_fused", wide)          # sharded: PEER only (RCCL / SHM cannot exchange inside a launch)
    assert "wide_rows_persist_kernel<true>" in wide and "comm_peer_aux()" in wide
