"""Guard of a residency cliff measured on gfx950: a kernel that is allocated more than 80 scalar
registers gets 7 wave slots per SIMD, not the 8 the compiler's occupancy remark claims, and the streaming kernels of the
fused solvers run 1024-thread workgroups TWO per CU, which needs all 8 -- above 80 the second half of their grid only
starts when the first half has finished (k_cg_update: +3.7 us of 15).  The single-GPU instantiations of the hot loop sit
at exactly 80, so one more kernel argument or one more 64-bit division would silently cost 5 % of the bench line.  This
test compiles the sources with hipcc's resource-usage remarks (no GPU needed) and pins the counts."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optimization_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resource_usage(src):
    """{demangled kernel name: (TotalSGPRs, VGPRs, compiler's waves/SIMD, scratch bytes per lane)} for one .hip file"""
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-I", os.path.join(ROOT, "optimization_amd", "include"),
           "-c", os.path.join(CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    pat = re.compile(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                     r"Occupancy \[waves/SIMD\]: (\d+)", re.S)
    rows = pat.findall(r.stderr)
    assert rows, "no resource-usage remarks in the compiler's output"
    names = subprocess.run(["c++filt"] + [m[0] for m in rows], capture_output=True, text=True).stdout.splitlines()
    out = {}
    for (_, sg, vg, scr, occ), n in zip(rows, names):
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[n] = (int(sg), int(vg), int(occ), int(scr))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_streaming_kernels_that_need_eight_waves_per_simd_stay_within_80_sgprs():
    cg = _resource_usage("stpcg.hip")
    must = [n for n in cg if n.startswith("k_cg_update_s80<") or n.startswith("k_cg_pupdate_s80<")]
    # the single-GPU hot loop: unpreconditioned recurrence form at p = 1, 2, 3 (4, 6, 9 components), the diagonal /
    # block-Jacobi forms, and the direction kernel
    must += ["k_cg_update<0, false, %d, mi::NoFold>" % kc for kc in (3, 4, 6, 9)]
    must += ["k_cg_update<%d, false, 3, mi::NoFold>" % pre for pre in (1, 2, 3)]
    must += ["k_cg_pupdate<false, 0, mi::NoFold>"]
    assert len(must) > 12
    for n in must:
        assert n in cg, (n, sorted(cg)[:5])
        sg, vg, _, scratch = cg[n]
        assert sg <= 80, "%s: %d SGPRs (residency cliff above 80)" % (n, sg)
        if ", 16," not in n:  # (16 components = p = 4: 20 bytes of scratch per lane buy the second workgroup per CU)
            assert scratch == 0, "%s spills to scratch memory" % n
        if "<false, 0," in n or "k_cg_update" in n:
            assert vg <= 64, "%s: %d VGPRs (two 1024-thread workgroups per CU need <= 64)" % (n, vg)
    # the instantiations that need more than 80 (sharded, slot-reading, 16 components) exist as capped twins ONLY
    assert not [n for n in cg if n.startswith(("k_cg_update<", "k_cg_pupdate<")) and ("Fold" in n and "NoFold" not in n)]
    assert "k_cg_update_s80<0, false, 9, mi::FoldArgs>" in cg and "k_cg_pupdate_s80<false, 0, mi::FoldPush>" in cg
    ls = _resource_usage("lsqr.hip")
    for n, (sg, vg, _, scratch) in ls.items():
        if n.startswith(("k_lsqr_xw<", "k_lsqr_unorm<", "k_lsqr_vnorm<")):
            assert sg <= 80 and vg <= 64 and scratch == 0, (n, sg, vg, scratch)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_asm_loads_of_the_wide_window_pass_are_left_alone_until_their_wait(tmp_path):
    """k_st_hess_widewin<4 | 6, ...> loads its own X / Y rows by P `global_load_dwordx4` ASM statements (stiefel.hip
    Epi::request: there is no ordered 16-byte load to be had from the compiler) and waits for them by an ASM
    `s_waitcnt vmcnt(0)` (Epi::arrive).  hipcc does not know that the registers of an asm load are filled later: a copy or
    a spill of one of them between the load and the wait would move garbage.  This test reads the generated code of every
    P = 4 and P = 6 instantiation: P asm loads each, no scratch memory, and no instruction between the loads and the wait that
    names one of their destination registers."""
    asm = tmp_path / "stiefel.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-I", os.path.join(ROOT, "optimization_amd", "include"), "-S", "--cuda-device-only",
           os.path.join(CSRC, "stiefel.hip"), "-o", str(asm)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = asm.read_text().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*k_st_hess_widewinILi[46].*:\s*; @", l)]
    assert len(starts) == 8, len(starts)   # P = 4, 6 x head widths 7 / 8 x computed / loaded far columns
    for st in starts:
        regs, waiting, nload, nwait, touched = set(), False, 0, 0, []
        i = st + 1
        while i < len(lines) and not re.match(r"^_ZN.*:\s*; @", lines[i]):
            l, in_asm = lines[i], "ASMSTART" in lines[i - 1]
            m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
            if m and in_asm:
                regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                waiting, nload = True, nload + 1
            elif waiting and in_asm and "s_waitcnt vmcnt(0)" in l:
                waiting, regs, nwait = False, set(), nwait + 1
            elif waiting and not l.strip().startswith(";"):
                assert "scratch_" not in l, l
                for m2 in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", l):
                    r_ = {int(m2.group(3))} if m2.group(3) else set(range(int(m2.group(1)), int(m2.group(2)) + 1))
                    if r_ & regs:
                        touched.append(l.strip())
            i += 1
        width = int(re.search(r"widewinILi(\d)", lines[st]).group(1))
        assert nload == width and nwait == 1, (lines[st][:80], nload, nwait)   # P / 2 pieces per field, two fields
        assert not touched, (lines[st][:80], touched[:4])
