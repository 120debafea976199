"""Compile-time guards on the gfx950 ISA of the hot kernels (CPU-only: hipcc cross-compiles).
They pin two regressions DESIGN.md describes - the streaming (`nt`) hint silently lost when written as a
run-time select, and hand-written carry-chain asm breaking the compiler's register/SCC bookkeeping -
plus the occupancy and no-scratch assumptions the launch code relies on."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "thaler-study_amd", "csrc")


@pytest.fixture(scope="module")
def isa():
    subprocess.check_call(["make", "-C", CSRC, "isa"], stdout=subprocess.DEVNULL)
    text = open(os.path.join(CSRC, "build", "sumcheck_hip.s")).read()
    usage = open(os.path.join(CSRC, "build", "resource_usage.txt")).read()
    return text, usage


def kernel_usage(usage, mangled_fragment):
    """{field: int} of the first kernel whose mangled name contains the fragment"""
    blocks = usage.split("remark: Function Name: ")[1:]
    for b in blocks:
        name = b.split(" ")[0]
        if mangled_fragment in name:
            out = {}
            for key in ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]",
                        "VGPRs Spill", "SGPRs Spill"):
                m = re.search(re.escape(key) + r": (\d+)", b)
                out[key] = int(m.group(1))
            return out
    raise AssertionError("no kernel matching %s" % mangled_fragment)


def kernel_body(text, mangled_fragment):
    m = re.search(r"^(_ZN2sc\S*%s\S*):[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(mangled_fragment), text, flags=re.S | re.M)
    assert m, mangled_fragment
    return m.group(2)


GOLD = "INS_14GoldilocksMontE"


def test_first_pass_occupancy_and_no_scratch(isa):
    _, usage = isa
    u = kernel_usage(usage, "pass_kernel%sLi0ELi3ELi1E" % GOLD)           # 27-cell first pass, nt loads
    assert u["Occupancy [waves/SIMD]"] >= 2, u
    assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0 and u["SGPRs Spill"] == 0, u
    assert u["VGPRs"] <= 256
    for frag in ("pass_kernel%sLi3ELi2ELi3E" % GOLD, "pass_kernel%sLi2ELi2ELi1E" % GOLD,
                 "evaluate_kernel%sLb1E" % GOLD, "fix_low_kernel%sLb1E" % GOLD, "wgrid_pass_kernel%sLi5E" % GOLD):
        u = kernel_usage(usage, frag)
        assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0, (frag, u)
        assert u["Occupancy [waves/SIMD]"] >= 2, (frag, u)
    # every kernel of the library: no scratch, no spills
    for b in usage.split("remark: Function Name: ")[1:]:
        name = b.split(" ")[0]
        assert re.search(r"VGPRs Spill: 0\b", b), name
        if "wgrid_pass_kernelINS_11MontGeneric" in name:
            # the generic-modulus instances carry the field constants, 32 fold weights and the exchange descriptor in
            # SGPRs and spill a few of them (outside the loops); no vector register is spilled
            assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) <= 128, name
            continue
        assert re.search(r"ScratchSize \[bytes/lane\]: 0\b", b), name


def test_gram_pass_is_what_the_design_says(isa):
    """kernels/gram.hpp: the first pass on the int8 matrix cores - four MFMAs and twelve transposed LDS reads per stage and
    operand set, its tiles brought in by LDS-DMA, no scratch, registers for two waves per SIMD; the fold behind it keeps
    its loads out of the way of its LDS traffic (no vmcnt(0) behind an LDS wait inside the tile loop)"""
    text, usage = isa
    # (round 5: the kernel carries the field - its epilogue reduces the accumulators to the 81 cells mod p - and there is no
    # second kernel behind it)
    assert "gram_finish_kernel" not in text
    for fld in (GOLD, "INS_11MontGenericE"):
        u = kernel_usage(usage, "gram_pass_kernel%sLi4ELb1E" % fld)
        assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0 and u["VGPRs"] <= 256 and u["Occupancy [waves/SIMD]"] >= 2, u
    m = re.search(r"^(_ZN2sc16gram_pass_kernel%sLi4ELb1E\S*):[^\n]*\n(.*?)\n\.Lfunc_end" % GOLD, text, flags=re.S | re.M)
    assert m
    body = m.group(2)
    assert body.count("v_mfma_i32_32x32x32_i8") == 16, body.count("v_mfma_i32_32x32x32_i8")      # four stages x four
    assert body.count("ds_read_b64_tr_b8") == 12 * 5, body.count("ds_read_b64_tr_b8")            # prologue + four stages
    assert body.count("global_load_lds_dwordx4") >= 8 and "scratch_" not in body
    u = kernel_usage(usage, "pass_kernel%sLi4ELi2ELi1E" % GOLD)
    assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0 and u["Occupancy [waves/SIMD]"] >= 2, u


def test_streaming_variants_carry_nt_and_cached_variants_do_not(isa):
    text, _ = isa
    nt1 = kernel_body(text, "pass_kernel%sLi0ELi3ELi1E" % GOLD)
    loads = re.findall(r"global_load_dwordx4[^\n]*", nt1)
    # the full-tile path streams (8 nt loads per tile and table pair, twice: prologue + prefetch); only the
    # ragged last tile uses plain predicated loads
    assert sum(" nt" in l for l in loads) >= 8, "first pass lost its nontemporal loads"
    nt3 = kernel_body(text, "pass_kernel%sLi3ELi2ELi3E" % GOLD)
    assert any(" nt" in l for l in re.findall(r"global_load_dwordx4[^\n]*", nt3))
    assert any(" nt" in l for l in re.findall(r"global_store_dwordx4[^\n]*", nt3)), "streamed outputs lost their nt stores"
    nt0 = kernel_body(text, "pass_kernel%sLi2ELi2ELi0E" % GOLD)
    assert not any(" nt" in l for l in re.findall(r"global_(?:load|store)_dwordx4[^\n]*", nt0))
    ev = kernel_body(text, "evaluate_kernel%sLb1E" % GOLD)
    assert any(" nt" in l for l in re.findall(r"global_load_dwordx4[^\n]*", ev))


def test_carry_chain_asm_is_intact(isa):
    """the hand-scheduled Goldilocks sequences (field.hpp sub4/sub2/acc_mac) appear as written: carries on
    the pinned SGPR pairs s[72:85], the borrow mask formed by s_andn2_b64 on the scalar unit, and no
    compiler-inserted s_nop inside a block"""
    text, usage = isa
    body = kernel_body(text, "pass_kernel%sLi0ELi3ELi1E" % GOLD)
    assert len(re.findall(r"s_andn2_b64 s\[72:73\], s\[72:73\], s\[80:81\]", body)) >= 1
    assert body.count("v_mad_u64_u32") >= 27 * 4
    # the pinned pairs are clobbers the register allocator must respect: nothing else may be live in them
    # across a block, which shows up as the kernel needing at least s85
    u = kernel_usage(usage, "pass_kernel%sLi0ELi3ELi1E" % GOLD)
    assert u["TotalSGPRs"] >= 86 and u["TotalSGPRs"] <= 102, u
    # one sub4 block = 16 VALU + 4 SALU in a row
    blk = re.search(r"v_sub_co_u32_e64 v\d+, vcc, v\d+, v\d+\n(?:\s+[^\n]+\n){18}\s+v_subb_co_u32_e64 v\d+, s\[84:85\], v\d+, 0, s\[76:77\]", body)
    assert blk, "sub4 block not found as written"
    assert "s_nop" not in blk.group(0)


def test_block_per_cu_instantiations_fit_one_cu(isa):
    """the heavy passes hold ALL waves of a CU in one block (512 / 768 threads, kernels.hpp pass_block_threads) so that the
    waves of a SIMD can share their tiles through an LDS counter: the block must fit a CU - registers for 2 / 3 waves per
    SIMD, LDS below the 160 KiB of a CU - and the tile counter must be an LDS atomic, not a global one"""
    text, usage = isa
    for frag, waves in (("pass_kernel%sLi0ELi3ELi1E" % GOLD, 2), ("pass_kernel%sLi3ELi2ELi3E" % GOLD, 2), ("pass_kernel%sLi2ELi2ELi1E" % GOLD, 3),
                        ("pass_kernelINS_11MontGenericELi0ELi3ELi1E", 2), ("pass_kernelINS_11MontGenericELi3ELi2ELi3E", 2),
                        ("pass_kernelINS_11MontGenericELi2ELi2ELi1E", 3)):
        u = kernel_usage(usage, frag)
        assert u["Occupancy [waves/SIMD]"] >= waves, (frag, u)
        assert u["LDS Size [bytes/block]"] <= 160 * 1024, (frag, u)
        assert u["ScratchSize [bytes/lane]"] == 0, (frag, u)
        body = kernel_body(text, frag)
        assert re.search(r"ds_add_rtn_u32", body), frag               # the tile counter


def test_wfold_pass_is_what_the_design_says(isa):
    """kernels/grid_pass.hpp, wfold_pass_kernel: two waves per SIMD and one block per CU (131 KB of LDS), nothing in scratch, the
    nontemporal hint on the NT instances' loads and on none of the others', and the sub-step's eight loads issued as one batch
    (no load waits for an LDS operation of the same sub-step in between: the batch is what keeps 16 KiB per wave in flight)"""
    text, usage = isa
    for fld in (GOLD, "INS_11MontGenericE"):
        for kf, ks in ((4, 5), (5, 3), (5, 4), (5, 5)):
            u = kernel_usage(usage, "wfold_pass_kernel%sLi%dELi%dELb1E" % (fld, kf, ks))
            # (the generic field's constants + sixteen weights + the exchange descriptor: a couple of SGPRs go to VGPR lanes, never to memory)
            assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0 and u["SGPRs Spill"] <= (0 if fld == GOLD else 4), (fld, kf, ks, u)
            assert u["Occupancy [waves/SIMD]"] == 2 and u["VGPRs"] <= 256, (fld, kf, ks, u)
            assert 128 * 1024 <= u["LDS Size [bytes/block]"] <= 160 * 1024, (fld, kf, ks, u)
    nt = kernel_body(text, "wfold_pass_kernel%sLi4ELi5ELb1E" % GOLD)
    plain = kernel_body(text, "wfold_pass_kernel%sLi4ELi5ELb0E" % GOLD)
    loads_nt = re.findall(r"global_load_dwordx4[^\n]*", nt)
    table_loads = [l for l in loads_nt if "sc0 sc1" not in l]        # (the system-scope poll of the in-kernel exchange is no table load)
    assert len(table_loads) >= 64 and all(" nt" in l for l in table_loads), "the streaming fold lost its nontemporal loads"
    assert not any(" nt" in l for l in re.findall(r"global_(?:load|store)_dwordx4[^\n]*", plain))
    # batches: runs of >= 8 consecutive table loads with nothing but address arithmetic between them
    lines = [l.strip() for l in nt.split("\n")]
    best = run = 0
    for l in lines:
        if l.startswith("global_load_dwordx4"):
            run += 1
            best = max(best, run)
        elif l.startswith(("ds_", "s_waitcnt", "s_barrier", "v_mad", "v_mul")):
            run = 0
    assert best >= 8, "the eight loads of a sub-step are no longer issued as one batch (%d)" % best

