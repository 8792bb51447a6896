#!/usr/bin/env python3
"""Reads the gfx950 code objects out of the SHIPPED library and answers two questions from their disassembly:

1. Is the in-launch hand-over protocol what sdtw_kernels.hpp (drain_stores) says it is?  Every store a consumer wave of the
   same launch reads must be write-through (`sc1`), an `s_waitcnt vmcnt(0)` must stand between those stores and the counter
   / progress word that publishes them, and the consumer must run `buffer_inv sc1` between its poll and its first plain load.
   Round 2 shipped a workgroup-scope release fence in that place, which emits NO wait on this compiler: this check is what
   keeps a compiler bump or an edit from silently undoing the protocol again.
2. How many VALU instructions does the steady-state loop of the headline fill kernel spend per DP cell?  (bench.py's VALU
   roofline used to hard-code 49/16.)

Usage: tools/isa_check.py [path/to/libsigfish_amd.so] [--json out.json]
Used by tests/test_publish_isa.py (CPU) and by the Makefile (writes lib/isa_stats.json next to the library).
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP_CANDIDATES = ["/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump") or ""]


def objdump():
    for c in OBJDUMP_CANDIDATES:
        if c and os.path.exists(c):
            return c
    return None


def disassemble(so_path):
    """{mangled symbol: [instruction text, ...]} over every gfx950 code object bundled in so_path."""
    od = objdump()
    if od is None:
        raise RuntimeError("llvm-objdump not found")
    funcs = {}
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, "lib.so")
        shutil.copy(so_path, local)  # --offloading writes the bundles next to its input
        subprocess.run([od, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for name in sorted(os.listdir(td)):
            if "gfx950" not in name:
                continue
            txt = subprocess.run([od, "-d", os.path.join(td, name)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
                if m:
                    cur = funcs.setdefault(m.group(1), [])
                    continue
                if cur is None or not line.startswith("\t"):
                    continue
                ins = line.split("//")[0].strip()
                addr = line.split("//")[1].split(":")[0].strip() if "//" in line else ""
                if ins and re.fullmatch(r"[0-9A-Fa-f]+", addr):
                    cur.append((ins, addr))
    return funcs


def _is(ins, prefix):
    return ins.split()[0].startswith(prefix)


def _waits_vm0(ins):
    return ins.startswith("s_waitcnt") and "vmcnt(0)" in ins


def _cfg(ins_list):
    """successor / predecessor lists over instruction indices (s_branch / s_cbranch_* targets from their word offsets)"""
    idx = {int(a, 16): i for i, (_, a) in enumerate(ins_list)}
    succ = [[] for _ in ins_list]
    for i, (ins, addr) in enumerate(ins_list):
        op = ins.split()[0]
        nxt = [i + 1] if i + 1 < len(ins_list) else []
        if op in ("s_endpgm", "s_setpc_b64"):  # end of the kernel / return from a function
            continue
        m = re.match(r"s_(c?)branch\w*\s+(\d+)", ins)
        if m:
            off = int(m.group(2))
            if off >= 32768:
                off -= 65536
            tgt = idx.get(int(addr, 16) + 4 + off * 4)
            succ[i] = ([tgt] if tgt is not None else []) + (nxt if m.group(1) else [])
        else:
            succ[i] = nxt
    pred = [[] for _ in ins_list]
    for i, ss in enumerate(succ):
        for j in ss:
            pred[j].append(i)
    return succ, pred


def _opcode(ins):
    return ins.split()[0]


def check_producer(ins_list, what, is_publish_store, is_signal):
    """Walking BACKWARDS from every signal instruction (the counter add / the progress store) along every control-flow path, an
    `s_waitcnt vmcnt(0)` must come before any publish store or the function's entry.  Returns a list of violations."""
    bad = []
    succ, pred = _cfg(ins_list)
    n_signal = 0
    for i, (ins, addr) in enumerate(ins_list):
        if not is_signal(ins):
            continue
        n_signal += 1
        seen, todo = set(), list(pred[i])
        while todo:
            j = todo.pop()
            if j in seen:
                continue
            seen.add(j)
            p = ins_list[j][0]
            if _waits_vm0(p):
                continue  # this path is fine
            if is_publish_store(p):
                bad.append(f"{what}: a path reaches `{ins}` at {addr} from the store at {ins_list[j][1]} without an s_waitcnt vmcnt(0)")
                break
            if not pred[j] and j == 0:
                bad.append(f"{what}: a path reaches `{ins}` at {addr} from the entry without an s_waitcnt vmcnt(0)")
                break
            todo.extend(pred[j])
    if n_signal == 0:
        bad.append(f"{what}: found no publishing instruction at all")
    return bad


def _is_vector_load(ins):
    return _opcode(ins).startswith(("global_load", "flat_load", "buffer_load"))


def check_consumer(name, ins_list):
    """Walking FORWARDS from every polling loop (s_sleep) along every control-flow path, `buffer_inv sc1` must come before
    any vector load that is not itself sc1 (the poll is)."""
    bad = []
    succ, pred = _cfg(ins_list)
    n = 0
    for i, (ins, addr) in enumerate(ins_list):
        if not ins.startswith("s_sleep"):
            continue
        n += 1
        seen, todo = set(), list(succ[i])
        while todo:
            j = todo.pop()
            if j in seen:
                continue
            seen.add(j)
            p, pa = ins_list[j]
            if p.startswith("buffer_inv") and "sc1" in p:
                continue
            if _is_vector_load(p) and " sc1" not in p:
                bad.append(f"{name}: plain load `{p}` at {pa} is reachable from the poll at {addr} without a buffer_inv sc1")
                break
            todo.extend(succ[j])
    if n == 0:
        bad.append(f"{name}: found no polling loop (s_sleep)")
    return bad


def check_fused_fill(name, ins_list):
    bad = []
    stores = [(i, a) for i, a in ins_list if _opcode(i).startswith(("global_store", "flat_store"))]
    if not stores:
        bad.append(f"{name}: no global stores found")
    for ins, addr in stores:
        if " sc1" not in ins:
            bad.append(f"{name}: store without sc1 (not write-through): `{ins}` at {addr}")
    # the completion counter: the only atomic add whose value is not used (no sc0); the ticket add returns (sc0)
    # (a kernel that also holds private-memory accesses addresses global memory with flat_ instructions: same protocol)
    bad += check_producer(ins_list, name, lambda s: _opcode(s).startswith(("global_store", "flat_store")),
                          lambda s: _opcode(s) in ("global_atomic_add", "flat_atomic_add") and " sc0" not in s)
    return bad


def check_strip_pipe(name, ins_list):
    bad = []
    rows = [(i, a) for i, a in ins_list if _opcode(i) == "global_store_dwordx4"]
    if not rows:
        bad.append(f"{name}: no 16-byte boundary-row stores found")
    for ins, addr in rows:
        if " sc1" not in ins:
            bad.append(f"{name}: boundary-row store without sc1: `{ins}` at {addr}")
    # progress words: the 4-byte sc1 stores
    bad += check_producer(ins_list, name, lambda s: _opcode(s) == "global_store_dwordx4",
                          lambda s: _opcode(s) == "global_store_dword" and " sc1" in s)
    bad += check_consumer(name, ins_list)
    return bad


def protocol_violations(funcs):
    bad = []
    seen = {"fused": 0, "pass2": 0, "pipe": 0}
    for name, ins in funcs.items():
        if re.search(r"sdtw_fill_kernelILi\d+ELb[01]ELb0ELb[01]ELb1EE", name):  # <MAXR, STD, false, LCK, FUSED>: pass 2 by ticket in the launch
            seen["fused"] += 1
            bad += check_fused_fill(name, ins)
        elif "fused_trace_dispatch" in name:
            seen["pass2"] += 1
            bad += check_consumer(name, ins)
        elif "sdtw_strip_pipe_kernel" in name:
            seen["pipe"] += 1
            bad += check_strip_pipe(name, ins)
    for k, v in seen.items():
        if v == 0:
            bad.append(f"no `{k}` kernel found in the library")
    return bad, seen


VALU_PREFIXES = ("v_",)
NOT_VALU = ("v_readlane", "v_readfirstlane", "v_writelane")  # executed by the scalar side / SALU-like


def loops(ins_list):
    """innermost loops: (start, end) index pairs of backward branches that contain no other backward branch target"""
    addr_to_idx = {}
    for i, (ins, addr) in enumerate(ins_list):
        addr_to_idx[int(addr, 16)] = i
    out = []
    for i, (ins, addr) in enumerate(ins_list):
        m = re.match(r"s_c?branch\w*\s+(\d+)", ins)
        if not m:
            continue
        off = int(m.group(1))
        if off < 32768:
            continue
        tgt = int(addr, 16) + 4 + (off - 65536) * 4
        if tgt in addr_to_idx:
            out.append((addr_to_idx[tgt], i))
    inner = []
    for s, e in out:
        if not any((s2 >= s and e2 <= e) and (s2, e2) != (s, e) for s2, e2 in out):
            inner.append((s, e))
    return inner


def _classify(op):
    if op.startswith("v_") and not op.startswith(NOT_VALU):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def hot_path_mix(ins_list, s, e, succ):
    """instruction mix along the CHEAPEST path (fewest VALU instructions) from the loop head s to its back edge e: the
    steady-state trip -- the blocks a trip may skip (checkpoints, priority updates) are skipped; the cells cannot be"""
    import heapq
    dist = {s: (1 if _classify(_opcode(ins_list[s][0])) == "valu" else 0)}
    prev = {}
    heap = [(dist[s], s)]
    while heap:
        d, i = heapq.heappop(heap)
        if d > dist.get(i, 1 << 30):
            continue
        if i == e:
            break
        for j in succ[i]:
            if j < s or j > e:
                continue
            nd = d + (1 if _classify(_opcode(ins_list[j][0])) == "valu" else 0)
            if nd < dist.get(j, 1 << 30):
                dist[j] = nd
                prev[j] = i
                heapq.heappush(heap, (nd, j))
    if e not in dist:
        return None
    path = [e]
    while path[-1] != s:
        path.append(prev[path[-1]])
    mix = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0, "cells": 0, "min3": 0, "instructions": len(path)}
    for i in path:
        op = _opcode(ins_list[i][0])
        mix[_classify(op)] += 1
        if op.startswith("v_sub_f32"):
            mix["cells"] += 1
        if op.startswith("v_min3"):
            mix["min3"] += 1
    return mix


def fill_loop_stats(funcs, pattern, min_cells=64):
    """steady-state loops (>= min_cells cells per trip) of the kernels whose symbol matches `pattern`: VALU per cell"""
    best = None
    for name, ins in funcs.items():
        if not re.search(pattern, name):
            continue
        succ, _ = _cfg(ins)
        per = []
        for s, e in loops(ins):
            m = hot_path_mix(ins, s, e, succ)
            if m and m["cells"] >= min_cells:
                per.append(m)
        if not per:
            continue
        per.sort(key=lambda m: m["valu"] / m["cells"])
        rep = per[len(per) // 2]
        best = {"kernel": name, "loops": len(per), "valu_per_cell_min": per[0]["valu"] / per[0]["cells"],
                "valu_per_cell_max": per[-1]["valu"] / per[-1]["cells"], "valu_per_cell": rep["valu"] / rep["cells"], "median_loop": rep}
    return best


def scratch_in_hot_loops(funcs, pattern, min_valu=300):
    """(scratch loads, scratch stores) inside the big loops (>= min_valu VALU instructions per trip: the unrolled step loops) of
    the functions matching `pattern` -- a spill there is HBM traffic per step (r03: 19 GB per --dtw-std launch)"""
    out = {}
    for name, ins in funcs.items():
        if not re.search(pattern, name):
            continue
        worst = (0, 0)
        for s, e in loops(ins):
            ops = [_opcode(i) for i, _ in ins[s:e + 1]]
            if sum(1 for o in ops if _classify(o) == "valu") < min_valu:
                continue
            sl, ss = sum(o.startswith("scratch_load") for o in ops), sum(o.startswith("scratch_store") for o in ops)
            if sl + ss > sum(worst):
                worst = (sl, ss)
        out[name] = worst
    return out


def main(argv):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(here, "sigfish_amd", "lib", "libsigfish_amd.so")
    out = None
    args = argv[1:]
    while args:
        a = args.pop(0)
        if a == "--json":
            out = args.pop(0)
        else:
            so = a
    funcs = disassemble(so)
    bad, seen = protocol_violations(funcs)
    # <MAXR, STD, SEG, LCK, FUSED>: the kernels bench.py's workloads run at full batch size
    stats = {"kernels_checked": seen, "violations": bad,
             "headline_fill": fill_loop_stats(funcs, r"sdtw_fill_kernelILi16ELb0ELb0ELb1ELb1EE"),
             "std_fill": fill_loop_stats(funcs, r"sdtw_fill_kernelILi16ELb1ELb0ELb1ELb1EE"),
             "scratch_in_pass2_loops_of_the_fused_launch": scratch_in_hot_loops(funcs, r"fused_trace_dispatch"),
             "fill32": fill_loop_stats(funcs, r"sdtw_fill_kernelILi32ELb0ELb0ELb0ELb1EE", min_cells=128)}  # (its 32-row loops only)
    text = json.dumps(stats, indent=1)
    if out:
        with open(out, "w") as f:
            f.write(text + "\n")
    print(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
