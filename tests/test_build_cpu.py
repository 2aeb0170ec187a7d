"""Build-time invariants of the HIP library that a GPU-less host can check.

The pipeline tile kernel places its s_waitcnt vmcnt(N) by hand around in-flight LDS-DMA requests.  Scratch (spill) traffic
counts in vmcnt too, so a spilling build would silently break that arithmetic: the build keeps the compiler's
kernel-resource-usage remarks (gtsam_points_amd/csrc/Makefile -> gp_vgicp.resources.txt) and this test holds every
instantiation of the kernel to zero scratch, zero spills and the occupancy its __launch_bounds__ asks for."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "gtsam_points_amd", "csrc", "gp_vgicp.resources.txt")


def _kernels():
    txt = open(RES).read()
    out = {}
    for block in txt.split("remark: Function Name: ")[1:]:
        name = block.split()[0]
        get = lambda key: int(re.search(re.escape(key) + r":\s+(\d+)", block).group(1))
        out[name] = dict(vgprs=get("VGPRs"), scratch=get("ScratchSize [bytes/lane]"), sspill=get("SGPRs Spill"), vspill=get("VGPRs Spill"),
                         occupancy=get("Occupancy [waves/SIMD]"), lds=get("LDS Size [bytes/block]"))
    return out


def test_pipeline_kernels_do_not_spill():
    assert os.path.exists(RES), "build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')"
    ks = {k: v for k, v in _kernels().items() if "vgicp_pipeline_kernel" in k}
    assert len(ks) == 6  # the fallback families: HASHED, GRID_F64 (MODE_LIN / MODE_ERR each), LOOKAHEAD (linearise with the look-ahead lookup; error evaluation)
                         # (round 6 dropped LOOKAHEAD's 512- / 256-point tiles: the family serves maps of >= 2^26 voxels only -- VERDICT r05 #8)
    ks3 = {k: v for k, v in _kernels().items() if "vgicp_stream_kernel" in k}
    assert len(ks3) >= 16  # MODE x source policy x descriptor source x surface validation (gp_vgicp_stream.hpp)
    for name, r in ks3.items():
        # 4 waves x (2 x 1 KB points + 2 x 3 KB covariances [+ 2 x 1 KB normals]); four workgroups of the normals build fill the CU's 160 KB exactly
        assert r["scratch"] == 0 and r["vspill"] == 0 and r["lds"] in (34816, 40960) and r["occupancy"] >= 4 and r["vgprs"] <= 128, (name, r)
    for name, r in ks.items():
        assert r["scratch"] == 0 and r["vspill"] == 0, (name, r)
        assert r["lds"] == 36864, (name, r)  # 4 waves x 3 stages x 3 KB
        mode_lin = "ILi0E" in name
        f32_outer = "ILi0ELb1E" in name or "ILi1ELb1E" in name
        want = 4 if (f32_outer or not mode_lin) else 3  # __launch_bounds__(256, ...) in gp_vgicp_tile.hpp
        assert r["occupancy"] >= want, (name, r)


def test_stream_kernel_has_no_compiler_vmcnt_waits():
    """gp_vgicp_stream.hpp (with the building blocks of gp_vgicp_tile2.hpp) counts its vector-memory requests by hand; a `s_waitcnt vmcnt` inserted by hipcc (for a load it tracks itself)
    would drain the source requests in flight.  The Makefile summarises the device assembly (csrc/count_waits.py)."""
    path = os.path.join(ROOT, "gtsam_points_amd", "csrc", "gp_vgicp.waits.txt")
    assert os.path.exists(path), "build the HIP library first"
    rows = [l.split() for l in open(path) if l.strip()]
    assert len(rows) >= 16  # MODE x policy x descriptor source x normals
    assert all("vgicp_stream_kernel" in r[0] for r in rows)
    for row in rows:
        name, n, touches = row[0], row[2], row[4]
        assert int(n) == 0, name  # (compiler waits with none of the asm's requests in flight -- the fused finalize tail -- are listed as idle_vmcnt_waits)
        # no instruction outside the asm blocks may read or write the destination registers of an asm-issued load that is still in flight
        # (csrc/count_waits.py): the round-3 memory fault was a phi copy of such registers at a loop back-edge
        assert row[3] == "inflight_reg_touches" and int(touches) == 0, row


def test_count_waits_fails_the_build_on_a_violation(tmp_path):
    """the vmcnt schedule's guard is part of the BUILD (count_waits.py exits non-zero, the Makefile stops), not only of this test file: an instruction
    the compiler places on a register of a load still in flight, a compiler-tracked vmcnt wait inside the schedule, or an assembly without the kernel each fail it"""
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(RES), "count_waits.py")

    def run(text):
        f = tmp_path / "k.s"
        f.write_text(text)
        return subprocess.run([sys.executable, script, str(f)], capture_output=True, text=True)

    head = "_ZN2gp19vgicp_stream_kernelILi0EEEvv:\n#ASMSTART\nglobal_load_dwordx4 v[1:4], v5, s[0:1]\n#ASMEND\n"
    ok = run(head + "v_add_f32 v9, v9, v8\n#ASMSTART\ns_waitcnt vmcnt(0)\n#ASMEND\nv_add_f32 v1, v1, v2\n.Lfunc_end0:\n")
    assert ok.returncode == 0 and "inflight_reg_touches 0" in ok.stdout, ok.stderr
    touch = run(head + "v_add_f32 v1, v1, v2\n.Lfunc_end0:\n")
    assert touch.returncode == 1 and "inflight_reg_touches 1" in touch.stdout and "does not hold" in touch.stderr
    wait = run(head + "s_waitcnt vmcnt(0)\n.Lfunc_end0:\n")
    assert wait.returncode == 1 and "compiler_vmcnt_waits 1" in wait.stdout
    assert run("some_other_kernel:\ns_endpgm\n").returncode == 2
    mk = open(os.path.join(os.path.dirname(RES), "Makefile")).read()
    assert "count_waits.py $(HERE)gp_vgicp.device.s > $@ ||" in mk  # the recipe stops on a non-zero exit
