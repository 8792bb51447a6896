"""The in-launch hand-over protocol, checked in the ISA of the library that ships (tools/isa_check.py).

Round 2 published a fill task's results behind a workgroup-scope release fence, believing it lowered to `s_waitcnt vmcnt(0)`;
on gfx950 / ROCm 7.2 it lowers to nothing, so the completion counter could overtake the write-through stores.  The wait is
now inline asm (sdtw_kernels.hpp: drain_stores) and THIS is the test that fails if a compiler or an edit removes it again:
the code objects are taken out of sigfish_amd/lib/libsigfish_amd.so and disassembled, no GPU needed."""
import os
import re

import pytest

from tests.util import ROOT

import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_check as I  # noqa: E402

SO = os.path.join(ROOT, "sigfish_amd", "lib", "libsigfish_amd.so")

pytestmark = pytest.mark.skipif(I.objdump() is None, reason="llvm-objdump not available")


@pytest.fixture(scope="module")
def funcs():
    assert os.path.exists(SO), "build the library first (python __graft_entry__.py)"
    return I.disassemble(SO)


def test_shipped_kernels_follow_the_publish_protocol(funcs):
    bad, seen = I.protocol_violations(funcs)
    # fused fill: MAXR 4 / 8 / 16 (+ the std_dtw and 32-row variants once they exist), their pass-2 functions, both strip kernels
    assert seen["fused"] >= 3 and seen["pass2"] >= 3 and seen["pipe"] == 2, seen
    assert bad == [], "\n".join(bad)


def _one(funcs, pattern):
    for name, ins in funcs.items():
        if re.search(pattern, name):
            return name, ins
    raise AssertionError(pattern)


def test_the_checker_notices_a_missing_wait_a_plain_store_and_a_missing_acquire(funcs):
    name, ins = _one(funcs, r"sdtw_fill_kernelILi16ELb0ELb0ELb1ELb1EE")
    assert I.check_fused_fill(name, ins) == []
    # (a) the wait in front of the completion counter gone (what round 2 shipped)
    no_wait = [(i, a) for i, a in ins if not (i.startswith("s_waitcnt") and "vmcnt(0)" in i)]
    assert any("without an s_waitcnt vmcnt(0)" in b for b in I.check_fused_fill(name, no_wait))
    # (b) one publish store not write-through
    k = next(j for j, (i, a) in enumerate(ins) if i.startswith("global_store_dword") and " sc1" in i)
    plain = list(ins)
    plain[k] = (plain[k][0].replace(" sc1", ""), plain[k][1])
    assert any("store without sc1" in b for b in I.check_fused_fill(name, plain))
    # (c) consumer without its acquire
    pname, pins = _one(funcs, r"fused_trace_dispatchILi16E")
    assert I.check_consumer(pname, pins) == []
    no_inv = [(i, a) for i, a in pins if not i.startswith("buffer_inv")]
    assert any("without a buffer_inv sc1" in b for b in I.check_consumer(pname, no_inv))
    # (d) pipelined strips: the wait in front of the progress word, the write-through boundary rows
    sname, sins = _one(funcs, r"sdtw_strip_pipe_kernelILb0E")
    assert I.check_strip_pipe(sname, sins) == []
    no_wait = [(i, a) for i, a in sins if not (i.startswith("s_waitcnt") and "vmcnt(0)" in i)]
    assert any("without an s_waitcnt vmcnt(0)" in b for b in I.check_strip_pipe(sname, no_wait))
    plain = [(i.replace(" sc1", "") if i.startswith("global_store_dwordx4") else i, a) for i, a in sins]
    assert any("boundary-row store without sc1" in b for b in I.check_strip_pipe(sname, plain))


def test_valu_per_cell_of_the_headline_loop_is_counted_from_the_build(funcs):
    st = I.fill_loop_stats(funcs, r"sdtw_fill_kernelILi16ELb0ELb0ELb1ELb1EE")
    assert st is not None and st["loops"] >= 16  # one steady-state loop per register of the last query row, two unroll depths
    # three arithmetic instructions per cell + the window minimum and loop overhead: 3.0 .. 3.2
    assert 3.0 <= st["valu_per_cell_min"] <= st["valu_per_cell"] <= st["valu_per_cell_max"] <= 3.2, st
    assert st["median_loop"]["vmem"] <= 2 and st["median_loop"]["lds"] == 2 * st["median_loop"]["cells"] // 16


def test_pass_2_of_the_fused_launch_keeps_its_step_loops_free_of_scratch(funcs):
    """A spilled register inside the step loop is memory traffic per step: the first --dtw-std build of the fused launch fetched
    19 GB and wrote 2.5 GB per launch that way (124 scratch reloads per four steps; found by the PMC pass).
    Pass 2 may spill around its loops (it is a function call with a 128-VGPR budget), not in them."""
    per = I.scratch_in_hot_loops(funcs, r"fused_trace_dispatch")
    assert len(per) >= 7  # MAXR 4 / 8 / 16, subsequence and std_dtw; MAXR 32, subsequence
    assert all(v == (0, 0) for k, v in per.items() if "ILi32E" not in k), per
    # the 32-row shapes: costs + start columns are 64 registers, the query rows live in LDS (LdsRows; in registers it was 85 reloads
    # per four steps); what is left of the bookkeeping spills a little: at most 2 % of a four-step block of ~1 580 instructions
    big = [v for k, v in per.items() if "ILi32E" in k]
    assert len(big) == 1 and big[0][0] <= 32 and big[0][1] <= 8, per
    # ... and the fill kernels of the fused launches not at all (the 32-row one spills around its sweeps like the unfused 32-row
    # fill does -- never inside a step loop: test_no_cost_only_fill_kernel_touches_scratch_inside_its_step_loops)
    for name, ins in funcs.items():
        if re.search(r"sdtw_fill_kernelILi(4|8|16)ELb[01]ELb0ELb[01]ELb1EE", name):
            assert not any(i.startswith("scratch_") for i, _ in ins), name


def test_no_cost_only_fill_kernel_touches_scratch_inside_its_step_loops(funcs):
    """Every cost-only fill kernel (any rows per lane, subsequence / std_dtw / segments / LDS checkpoints / fused): registers may
    spill around the sweep (the 32-row shapes do, a handful), never inside the unrolled step loops.  A wrapper lambda around
    the step call was enough to push the headline kernel's state into scratch once (3 GB fetched + 2.6 GB written per launch,
    27 + 15 GB on the 160-contig workload) with the step time unchanged -- only the counters and this check saw it."""
    per = I.scratch_in_hot_loops(funcs, r"sdtw_fill_kernelILi\d+E")
    assert len(per) >= 20, len(per)
    bad = {k: v for k, v in per.items() if v != (0, 0)}
    assert not bad, bad
