"""A guard on the hand-written kernels' resource figures.

`shade_body.h` pins v56-v62 in inline assembly, sets `exec` by hand and is laid out for 64 VGPRs / no scratch / 8 waves per SIMD; `k1_tile_cull`
relies on a small register file footprint for its eight blocks per CU.  A toolchain bump or an innocent edit can silently spill or halve the
occupancy, and nothing would fail.  This test reads the AMDGPU metadata notes of the code objects inside the built `.o` files
(objcopy .hip_fatbin -> clang-offload-bundler -> llvm-readelf --notes) and asserts the budgets the design rests on.  No GPU needed."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "sailor_amd" / "csrc"
LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernel_resources(obj: Path, tmp: Path) -> dict:
    fat, co = tmp / (obj.stem + ".fatbin"), tmp / (obj.stem + ".co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)], check=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}", "--unbundle"],
                   check=True)
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        fields = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|"
                                                   r"max_flat_workgroup_size):\s+(\d+)", block)}
        out[name] = fields
    return out


def find(res: dict, key: str) -> dict:
    """the kernel whose mangled name contains <len><key> (Itanium: the unqualified name prefixed by its length)"""
    hits = [v for k, v in res.items() if k.startswith(f"_Z{len(key)}{key}")]
    assert len(hits) == 1, (key, list(res))
    return hits[0]


def waves_per_simd(vgprs: int) -> int:
    # gfx950: 512 VGPRs per SIMD lane, allocated in blocks of 8, at most 8 waves
    return min(8, 512 // (-(-vgprs // 8) * 8))


def blocks_per_cu_by_sgprs(sgprs: int) -> int:
    # gfx950: 800 scalar registers per SIMD, allocated in sixteens plus sixteen for the trap handler / VCC; one wave of a 256-thread block per SIMD
    return min(8, 800 // (-(-sgprs // 16) * 16 + 16))


@pytest.fixture(scope="module")
def resources(tmp_path_factory):
    if not (LLVM / "clang-offload-bundler").exists() or not shutil.which("objcopy"):
        pytest.skip("no ROCm LLVM tools here")
    tmp = tmp_path_factory.mktemp("co")
    res = {}
    for name in ("shade", "light_cull", "ecs_sweep"):
        obj = CSRC / f"{name}.o"
        assert obj.exists(), f"{obj} is missing: run __graft_entry__.build()"
        res[name] = kernel_resources(obj, tmp)
    return res


@pytest.mark.parametrize("name", ["k2_shade", "k2_shade_p", "k2_shade_t", "k2_shade_pt"])
def test_k2_shade_fits_64_registers_without_scratch(resources, name):
    k = find(resources["shade"], name)
    assert k["vgpr_count"] <= 64 and waves_per_simd(k["vgpr_count"]) == 8
    assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0
    # eight 256-thread blocks per CU must fit the 160 KB of LDS
    assert 8 * k["group_segment_fixed_size"] <= 160 * 1024


@pytest.mark.parametrize("name", ["k2_shade_csm", "k2_shade_csm_p", "k2_shade_csm_t", "k2_shade_csm_pt"])
def test_k2_shade_csm_fits_64_registers_without_scratch(resources, name):
    """The K3 kernels run their shadow look-ups before the view / material terms exist ("K3 first", shade_body.h) and so fit 8 waves per SIMD; with
    24 bytes of scratch the same kernel was 310 us instead of 262 on C4 -- so no scratch at all."""
    k = find(resources["shade"], name)
    assert k["vgpr_count"] <= 64 and waves_per_simd(k["vgpr_count"]) == 8
    assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0


@pytest.mark.parametrize("name", ["k2_shade_csm_ibl", "k2_shade_csm_ibl_p", "k2_shade_csm_ibl_pt"])
def test_k2_shade_csm_ibl_keeps_five_waves_per_simd(resources, name):
    k = find(resources["shade"], name)
    assert k["vgpr_count"] <= 96 and waves_per_simd(k["vgpr_count"]) >= 5
    assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0


@pytest.mark.parametrize("name", ["k2_shade_band", "k2_shade_band_p", "k2_shade_band_pt"])
def test_k2_shade_band_stays_at_eight_waves(resources, name):
    k = find(resources["shade"], name)
    assert k["vgpr_count"] <= 64
    assert k["private_segment_fixed_size"] <= 16   # (two copies of the body at 64 registers each: a handful of bytes is what it has always had)


@pytest.mark.parametrize("name", ["k2_shade_band_csm", "k2_shade_band_csm_p", "k2_shade_band_csm_t", "k2_shade_band_csm_pt"])
def test_the_shadowed_band_kernels_fit_six_waves_without_scratch(resources, name):
    # (round 4: two copies of the K3 body spill at 64 registers -- 24 bytes of scratch cost the whole-frame K3 kernel a third of its speed in round 3 --
    # so these are pinned to six waves per SIMD = 80 registers, what the band kernels' wave-slot reserve leaves a CU anyway)
    k = find(resources["shade"], name)
    assert waves_per_simd(k["vgpr_count"]) >= 6 and k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0


@pytest.mark.parametrize("name", ["k0_band_count", "k0_band_scatter"])
def test_the_band_selection_kernels_have_no_scratch(resources, name):
    # (round 5: two launches, no block waits for another -- no residency requirement any more; what is left to guard is the four lights per thread
    # staying in registers)
    k = find(resources["light_cull"], name)
    assert waves_per_simd(k["vgpr_count"]) >= 4 and k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and k["group_segment_fixed_size"] <= 4096


def test_cull_kernels_keep_their_occupancy(resources):
    res = resources["light_cull"]
    tile = [v for n, v in res.items() if "k1_tile_cull" in n]
    assert tile, list(res)
    for k in tile:
        # (eight 256-thread blocks per CU = eight waves per SIMD: 64 registers; the cluster tiles' path of round 4 holds eight list entries and four
        # records per lane, the kernel went from 32 to ~55)
        assert k["vgpr_count"] <= 64 and waves_per_simd(k["vgpr_count"]) == 8 and k["private_segment_fixed_size"] <= 32, k   # (pinned to eight waves: a few bytes of scratch on the cluster tiles' path)
        assert 8 * k["group_segment_fixed_size"] <= 160 * 1024
        # (round 5: a CU admits min(8, 800 / (ceil(sgprs / 16) 16 + 16)) 256-thread blocks -- MI355X_MICROARCH.md; at 93-95 scalar registers that was
        # seven. Capped at 80: the frame pipeline's step 165.7 -> 161.5 us, profiles/r05/ab_sgpr_cap.txt)
        assert blocks_per_cu_by_sgprs(k["sgpr_count"]) == 8, k
    for key in ("k01_prepare", "k1_pack", "k1_group_lists"):
        k = find(res, key)
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (key, k)


def test_ecs_sweep_has_no_scratch(resources):
    for name, k in resources["ecs_sweep"].items():
        assert k["private_segment_fixed_size"] == 0, (name, k)


@pytest.mark.parametrize("name", ["k2_shade_h_p", "k2_shade_h_pt"])
def test_the_two_wave_shade_blocks_fit_sixteen_per_cu(resources, name):
    """Round 6: one 128-thread block per half tile, the records staged 64 at a time -- the point of the form is sixteen blocks (32 waves) per CU, so 64
    registers without scratch and at most 10 KB of LDS a block (off by default: measured neutral)."""
    k = find(resources["shade"], name)
    assert k["max_flat_workgroup_size"] == 128 and k["vgpr_count"] <= 64 and k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0
    assert 16 * k["group_segment_fixed_size"] <= 160 * 1024


def test_the_wide_list_builder_of_4x4_patches_fits_two_blocks_per_cu(resources):
    """Round 6: k1_group_lists_wide16 -- sixteen waves a block, 56 KB of LDS (sixteen mask rows double-buffered + sixteen queues of 128 entries): two blocks =
    32 waves per CU, and no more than the 64 KB a kernel may claim statically."""
    ks = [v for n, v in resources["light_cull"].items() if "k1_group_lists_wide16" in n]
    assert len(ks) == 2
    for k in ks:
        assert k["max_flat_workgroup_size"] == 1024 and k["group_segment_fixed_size"] <= 64 * 1024 and 2 * k["group_segment_fixed_size"] <= 160 * 1024
        assert k["vgpr_count"] <= 32 and k["private_segment_fixed_size"] == 0   # (sixteen waves of a block on four SIMDs: four each, at eight blocks' worth of registers)
