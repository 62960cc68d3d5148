"""CPU-only checks of the host side: C-ABI symbols, state_dict contract, scene encoders, schedule tables."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from ramp_amd import _lib, synth
from ramp_amd.spec import SCHEDULE_BUFFERS, make_unet_spec, unet_param_shapes
from util import GOLDEN, rel, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function declared in include/ramp_hip.h is exported by the built .so and bound in _lib."""
    hdr = open(os.path.join(ROOT, "include", "ramp_hip.h")).read()
    declared = set(re.findall(r"\b(ramp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ramp_ctx"}
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} not exported"
        assert name in _lib.PROTOTYPES, f"{name} not bound in ramp_amd/_lib.py"
    assert lib.ramp_version() >= 1
    # round 6: the micro-benchmark / stress harness is a tools-only library (csrc/bench.hip, include/ramp_hip_tools.h); the product
    # library exports none of it, the tools library exports everything
    import subprocess
    exported = set(re.findall(r" T (ramp_[a-z0-9_]+)", subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout))
    assert not [n for n in exported if n.startswith(("ramp_bench", "ramp_stress"))], "diagnostics leaked into the product library"
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))      # nothing undeclared either
    thdr = open(os.path.join(ROOT, "include", "ramp_hip_tools.h")).read()
    tdecl = set(re.findall(r"\b(ramp_[a-z0-9_]+)\s*\(", thdr)) - {"ramp_ctx"}
    assert tdecl == set(_lib.TOOL_PROTOTYPES), (tdecl, set(_lib.TOOL_PROTOTYPES))
    tools = _lib.load_tools()
    for name in sorted(tdecl | declared):
        assert hasattr(tools, name), f"{name} not exported by the tools library"


def test_no_cpu_fallback():
    """Without a HIP device context creation fails loudly instead of silently computing on the host."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    cfg = _lib.RampConfig(4, 48, 32, 4, 320, 16, 0, 0)
    h = C.c_void_p()
    assert lib.ramp_create(C.byref(cfg), C.byref(h)) != 0
    assert b"device" in lib.ramp_last_error().lower()
    from ramp_amd.apf import ObstacleField, avoidance
    with pytest.raises(_lib.RampHipError):
        avoidance(torch.zeros(1, 48, 4), ObstacleField(np.zeros((4, 2), np.float32)))


def test_struct_layout_matches_header():
    """ctypes mirrors of the C structs have the sizes AND field offsets the C compiler gives them."""
    import subprocess, tempfile
    pairs = [("ramp_config", _lib.RampConfig), ("ramp_apf_params", _lib.RampApfParams), ("ramp_sample_params", _lib.RampSampleParams),
             ("ramp_replan_params", _lib.RampReplanParams), ("ramp_replan_state", _lib.RampReplanState),
             ("ramp_replan_result", _lib.RampReplanResult)]
    body = []
    for cname, cls in pairs:
        body.append(f'printf("%zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            body.append(f'printf(" %zu", offsetof({cname}, {fname}));')
        body.append('printf("\\n");')
    src = '#include "ramp_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){' + "".join(body) + 'return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        lines = subprocess.check_output([os.path.join(d, "s")]).decode().strip().splitlines()
    for (cname, cls), line in zip(pairs, lines):
        got = [int(v) for v in line.split()]
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert got == want, (cname, got, want)


@pytest.mark.parametrize("S,o3,n", [(4, False, 684), (6, True, 608)])
def test_state_dict_contract(S, o3, n):
    """Key names/shapes follow the reference checkpoint contract; the wrapper adds the
    12 schedule buffers and the 'model.' prefix."""
    from ramp_amd.models import StaticGaussianDiffusionModel, TemporalUnetInference
    from ramp_amd.unet import load_numpy_state_dict
    sp = make_unet_spec(S, 48, obstacle_3d=o3)
    shapes = unet_param_shapes(sp)
    assert len(shapes) == n
    sd = weights(S, 48, o3)
    u = TemporalUnetInference(n_support_points=48, state_dim=S, obstacle_3d=o3)
    load_numpy_state_dict(u, sd)
    back = u.state_dict()
    assert set(back) == set(shapes)
    for k in shapes:
        assert tuple(back[k].shape) == tuple(shapes[k]), k
        assert np.array_equal(back[k].numpy(), sd[k]), k
    dm = StaticGaussianDiffusionModel(model=u, n_diffusion_steps=25, predict_epsilon=True)
    full = dm.state_dict()
    assert set(full) == set(SCHEDULE_BUFFERS) | {"model." + k for k in shapes}
    dm2 = StaticGaussianDiffusionModel(model=TemporalUnetInference(n_support_points=48, state_dim=S, obstacle_3d=o3),
                                       n_diffusion_steps=25, predict_epsilon=True)
    dm2.load_state_dict(full)                      # wrapper-level load with 'model.' prefix
    assert np.array_equal(dm2.model.state_dict()["time_mlp.encoder.1.weight"].numpy(), sd["time_mlp.encoder.1.weight"])
    with pytest.raises(RuntimeError):
        bad = dict(full); bad.pop("model.downs.0.0.cond_mlp.1.bias")
        dm2.load_state_dict(bad)


@pytest.mark.parametrize("T", [25, 50, 100])
def test_schedule_buffers_bitwise(T):
    """The wrapper computes its tables with the reference's own torch expressions: bitwise equal."""
    from ramp_amd.models import StaticGaussianDiffusionModel, TemporalUnetInference
    g = np.load(f"{GOLDEN}/schedule_T{T}.npz")
    dm = StaticGaussianDiffusionModel(model=TemporalUnetInference(n_support_points=48, state_dim=4),
                                      n_diffusion_steps=T, predict_epsilon=True)
    for k in SCHEDULE_BUFFERS:
        assert np.array_equal(getattr(dm, k).numpy(), g[k]), k


def test_scene_encoders_match_golden():
    """The scene encoders' parameter containers mean what the reference's modules mean: a torch restatement over them
    (tests/torch_scene_encoders.py; the product computes the latents in HIP) reproduces the reference's latents, and the
    containers themselves refuse to compute."""
    import torch_scene_encoders as tse
    from ramp_amd.models import TemporalUnetInference
    from ramp_amd.unet import load_numpy_state_dict
    g = np.load(f"{GOLDEN}/scene_latents.npz")
    u2 = load_numpy_state_dict(TemporalUnetInference(n_support_points=48, state_dim=4), weights(4, 48, False)).eval()
    u3 = load_numpy_state_dict(TemporalUnetInference(n_support_points=48, state_dim=6, obstacle_3d=True),
                               weights(6, 48, True)).eval()
    with torch.no_grad():
        for k in ("2d_6x64", "2d_16x64"):
            lat = tse.encode_2d(u2.scene_encoder, torch.from_numpy(g["cloud" + k])[None])[0].numpy()
            assert rel(lat, g["lat" + k]) < 2e-6
        for k in ("3d_5x50", "3d_20x200"):
            lat = tse.encode_3d(u3.scene_encoder, torch.from_numpy(g["cloud" + k])[None])[0].numpy()
            assert rel(lat, g["lat" + k]) < 2e-6
        with pytest.raises(RuntimeError):
            u2.scene_encoder(torch.from_numpy(g["cloud2d_6x64"])[None])


def test_synthetic_inputs_are_deterministic():
    a = synth.make_cloud(16, 64, 2, seed=43)
    b = synth.make_cloud(16, 64, 2, seed=43)
    assert np.array_equal(a, b) and a.shape == (16, 64, 2) and a.dtype == np.float32
    assert np.abs(a).max() < 0.9
    n1 = synth.make_noise((3, 2, 48, 4), seed=7)
    assert np.array_equal(n1, synth.make_noise((3, 2, 48, 4), seed=7))
    g = np.load(f"{GOLDEN}/scene_latents.npz")
    assert np.array_equal(g["cloud2d_16x64"], a)      # fixtures were generated from the same generator


def test_compat_layer_against_reference_fixture(tmp_path):
    """Pursuer dynamics, LimitsNormalizer and hard-condition helper vs the reference's outputs; checkpoint / context /
    environment-directory round trips in the reference's on-disk layout."""
    import torch
    import yaml
    from ramp_amd import compat
    from ramp_amd.models import StaticGaussianDiffusionModel, TemporalUnetInference
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "compat_cases.npz"))
    fn, vel = compat.DynamicsGenerator.create_pursuit_dynamics(0.5)
    assert np.array_equal(vel, g["dyn_vel"])
    for i, t in enumerate(g["dyn_t"]):
        assert np.abs(fn(int(t), g["dyn_prev"][i], g["dyn_robot"][i], vel) - g["dyn_out"][i]).max() < 1e-15
    n = compat.LimitsNormalizer(g["norm_mins"], g["norm_maxs"])
    x = torch.from_numpy(g["norm_x"])
    z = n.normalize(x)
    assert np.array_equal(z.numpy(), g["norm_z"]) and np.array_equal(n.unnormalize(z * 1.2).numpy(), g["norm_back"])
    hc = compat.StateGenerator.get_hard_cond_custom(torch.tensor([[0.1, -0.2], [0.5, 0.6], [0.7, 0.8]]), horizon=48)
    assert np.array_equal(hc[0].numpy(), g["hc0"]) and np.array_equal(hc[47].numpy(), g["hc47"])
    # pursuer field: same update path as MultiSphereFieldDynamics.update_centers
    ds = compat.make_pursuit_env(np.zeros((6, 2)), np.full((6, 2), 0.2), [0.5, 0.5])
    sphere = ds.env.obj_extra_list[0].fields[0]
    sphere.update_centers(3, torch.tensor([[-0.5, -0.5], [9.0, 9.0]]))
    want = fn(3, np.array([[0.5, 0.5]], np.float32), np.array([[-0.5, -0.5]]), vel.astype(np.float32))
    assert np.abs(sphere.centers.numpy() - want).max() < 1e-6
    # context + environment directory + checkpoint in the reference layout
    p = compat.ContextManager.save_context(torch.tensor([0.1, 0.2]), torch.tensor([0.3, 0.4]), str(tmp_path), "d", 7)
    assert p.endswith("contexts/context_007.pt")
    s0, g0 = compat.ContextManager.load_context(str(tmp_path / "contexts"), 7)
    assert torch.equal(s0, torch.tensor([0.1, 0.2])) and torch.equal(g0, torch.tensor([0.3, 0.4]))
    torch.save(torch.ones(3, 8, 2), tmp_path / "obstacle_points.pt")
    np.save(tmp_path / "box_centers.npy", np.zeros((3, 2), np.float32))
    (tmp_path / "metadata.yaml").write_text(yaml.safe_dump({"box_sizes": [[0.2, 0.2]] * 3}))
    env = compat.load_environment_dir(str(tmp_path))
    assert env["obstacle_points"].shape == (3, 8, 2) and env["box_sizes"].shape == (3, 2)
    sp = make_unet_spec(4, 48, obstacle_3d=False)
    sd = synth.make_unet_state_dict(sp, seed=3)
    dm = StaticGaussianDiffusionModel(model=TemporalUnetInference(n_support_points=48, state_dim=4), n_diffusion_steps=25,
                                      predict_epsilon=True)
    full = dm.state_dict()
    for k, v in sd.items():
        full["model." + k] = torch.from_numpy(np.asarray(v))
    ck = tmp_path / "models" / "m1" / "checkpoints"
    ck.mkdir(parents=True)
    torch.save(full, ck / "ema_model_current_state_dict.pth")
    dm2 = StaticGaussianDiffusionModel(model=TemporalUnetInference(n_support_points=48, state_dim=4), n_diffusion_steps=25,
                                       predict_epsilon=True)
    compat.load_checkpoint(dm2, str(tmp_path / "models"), "m1", use_ema=True)
    got = dm2.state_dict()
    assert all(torch.equal(got[k], full[k]) for k in full)


def test_gemm_kernels_keep_their_occupancy():
    """The GEMM kernels are tuned for a fixed number of resident blocks per CU (2, or 3 for the fp16x3 main variants);
    one careless change pushes a variant past the register budget and silently halves its occupancy.  Read the
    register counts back from the compiled code object."""
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "ramp_amd", "lib", "obj", "gemm.o")
    if not (os.path.exists(obj) and all(os.path.exists(f"{llvm}/{t}") for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))):
        pytest.skip("no built object / no llvm tools")
    d = tempfile.mkdtemp()
    try:
        subprocess.check_call([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, f"{d}/fat.bin"])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={d}/fat.bin", f"--output={d}/k.co", "--unbundle"])
        notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", f"{d}/k.co"], text=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    names = re.findall(r"\.name:\s+(\S+)", notes)
    vgprs = [int(v) for v in re.findall(r"\.vgpr_count:\s+(\d+)", notes)]
    assert len(names) == len(vgprs) and names
    seen = {"x6p": 0, "x6p3": 0}
    for n, v in zip(names, vgprs):
        if "gemm_x6p3_kernel" in n:
            assert v <= 168, (n, v); seen["x6p3"] += 1
        elif "gemm_x6p_kernel" in n or "gemm_x6_kernel" in n:
            assert v <= 256, (n, v); seen["x6p"] += 1
        elif "gemm_kernelILi4ELi1ELi1ELi1" in n:
            assert v <= 168, (n, v)                      # 128 x 32 tiles: three blocks per CU
        elif "gemm_kernel" in n:
            assert v <= 256, (n, v)
    assert seen["x6p"] >= 8 and seen["x6p3"] >= 1


def _kernel_notes(obj_name):
    """{kernel name: {vgpr_count, vgpr_spill_count, private_segment_fixed_size}} of a built object's gfx950 code object."""
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "ramp_amd", "lib", "obj", obj_name)
    if not (os.path.exists(obj) and all(os.path.exists(f"{llvm}/{t}") for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))):
        pytest.skip("no built object / no llvm tools")
    d = tempfile.mkdtemp()
    try:
        subprocess.check_call([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, f"{d}/fat.bin"])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={d}/fat.bin", f"--output={d}/k.co", "--unbundle"])
        notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", f"{d}/k.co"], text=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for entry in notes.split(".agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", entry).group(1)
        out[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", entry).group(1))
                     for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
    return out


def test_token_owning_kernels_do_not_spill():
    """The token-owning kernels run ONE wave per SIMD with inline-asm LDS-DMA in flight: a scratch reload there is an
    s_waitcnt vmcnt(0), i.e. a full memory round trip in the middle of the MFMA stream (round 3 removed 30 of them per tile from
    the fused feed-forward).  The shipped forward feed-forward, LN1 -> QKV, d(o) and d(ln1) kernels must
    stay free of scratch; the backward feed-forward and the residual variants may only spill a handful of per-tile values."""
    ffx = _kernel_notes("ffx.o")
    tkl = _kernel_notes("tkl.o")
    def one(d, pat):
        hits = [v for k, v in d.items() if pat in k]
        assert len(hits) == 1, (pat, [k for k in d if pat in k])
        return hits[0]
    assert one(ffx, "ffx_kernelILb0ELi0E")["vgpr_spill_count"] == 0          # forward, product variant
    assert one(ffx, "ffx_kernelILb1ELi0E")["vgpr_spill_count"] <= 48         # backward: tile prologue / epilogue only (41 today)
    assert one(tkl, "tkl_kernelILb1ELi0ELi0E")["vgpr_spill_count"] == 0      # LN1 -> QKV
    assert one(tkl, "tkl_kernelILb0ELi0ELi0E")["vgpr_spill_count"] == 0      # d(o)
    assert one(tkl, "tklb_kernel")["vgpr_spill_count"] == 0                  # d(ln1) + LayerNorm-1 backward
    for pat in ("tkl_kernelILb0ELi1ELi0E", "tkl_kernelILb0ELi3ELi0E"):      # out-projection (residual, + row-variant constant)
        assert one(tkl, pat)["vgpr_spill_count"] <= 8
    # atk.hip (self-attention + output projection): the raw q rows / k, v LDS regions of the next head step are requested while the
    # projection's MFMAs run; a scratch reload there drains them all.  Every product variant must be spill-free (the nested tile / head
    # loops make it so at T = 48: as ONE flat loop hipcc spilled 150 values)
    atk = _kernel_notes("atk.o")
    prod = [k for k in atk if "ato_kernel" in k and k.split("ato_kernelILi")[1].split("E")[2].startswith("Li0")]
    assert len(prod) == 4, list(atk)
    for k in prod:
        assert atk[k]["vgpr_spill_count"] == 0 and atk[k]["private_segment_fixed_size"] == 0, (k, atk[k])
    # atl.hip (attention backward + d(ln1) + LayerNorm-1 backward): a handful of kernel-lifetime values may live in scratch (stored in the
    # prologue, reloaded at a tile's / the kernel's end), nothing more -- a reload inside a head would drain every LDS-DMA piece in flight
    atl = _kernel_notes("atl.o")
    # (round 6: <NG, STAMP, PK>; the product variant parks k's planes (PK = 1) and, with the small MFMA result tiles in VGPRs, spills nothing)
    for pat, most in (("abl_kernelILi3ELb0ELb1E", 0), ("abl_kernelILi2ELb0ELb1E", 0), ("abl_kernelILi3ELb0ELb0E", 4), ("abl_kernelILi2ELb0ELb0E", 0)):
        assert one(atl, pat)["vgpr_spill_count"] <= most, (pat, one(atl, pat))
        assert one(atl, pat)["private_segment_fixed_size"] <= 16 * (most > 0), (pat, one(atl, pat))
    # round 6: the 16-wide token-owning kernels (what the product runs): no scratch in the fused feed-forward pair, LN1 -> QKV and d(o)
    ffx16 = _kernel_notes("ffx16.o")
    tkl16 = _kernel_notes("tkl16.o")
    for pat in ("ffx16_kernelILb0ELi0E", "ffx16_kernelILb1ELi0E"):
        assert one(ffx16, pat)["vgpr_spill_count"] == 0 and one(ffx16, pat)["private_segment_fixed_size"] == 0, (pat, one(ffx16, pat))
    for pat in ("ffx16h_kernelILb0E", "ffx16h_kernelILb1E"):      # the half-tile twins (a wave owns 16 tokens): the same, with room to spare
        assert one(ffx16, pat)["vgpr_spill_count"] == 0 and one(ffx16, pat)["private_segment_fixed_size"] == 0, (pat, one(ffx16, pat))
    for pat in ("tkl16_kernelILb1ELi0E", "tkl16_kernelILb0ELi0E"):
        assert one(tkl16, pat)["private_segment_fixed_size"] == 0, (pat, one(tkl16, pat))       # (LN variant: 8 values parked in spare AGPRs, no scratch)
    for d in (ffx, tkl, atk, atl, ffx16, tkl16):
        for k, v in d.items():
            assert v["vgpr_count"] <= 512, (k, v)


def test_touching_a_shared_header_marks_its_objects_stale():
    """build(force=False) must recompile what a header edit reaches (round-3 review: tokmma.h was missing from the dependency list,
    so an edit of the token-owning kernels' helpers left ffx.o / tkl.o stale) -- and, since round 6, ONLY that: build.py follows the
    `#include "..."` lines of every translation unit, a kernel file includes core.h + its family's argument header, so an edit to the
    token-owning kernels' helpers leaves the sampler, the row kernels and the tile GEMMs alone."""
    from ramp_amd import build as B
    deps = [os.path.basename(p) for p in B.header_deps()]
    for h in ("common.h", "core.h", "tokmma.h", "gemm_x6p_body.inc", "ramp_hip.h", "ramp_hip_tools.h", "build.py"):
        assert h in deps, (h, deps)
    csrc = os.path.join(ROOT, "ramp_amd", "csrc")
    on_disk = {f for f in os.listdir(csrc) if f.endswith((".h", ".inc"))}
    assert on_disk <= set(deps), on_disk - set(deps)
    # what one translation unit reaches
    per_tu = {src: {os.path.basename(d) for d in B.source_deps(os.path.join(csrc, src))} for src in B.SOURCES + B.TOOLS_SOURCES}
    assert {"core.h", "args_token.h", "pack.h", "tokmma.h", "build.py"} <= per_tu["ffx.hip"] and "args_sampler.h" not in per_tu["ffx.hip"]
    assert {"gemm_x6p_body.inc", "args_gemm.h", "pack.h"} <= per_tu["gemm.hip"] and "tokmma.h" not in per_tu["gemm.hip"]
    assert per_tu["sampler.hip"] == {"sampler.hip", "args_sampler.h", "core.h", "build.py"}
    for src in ("engine.hip", "ops.hip", "bench.hip"):      # the files that see every family
        assert {"common.h", "engine_util.h", "ramp_hip.h"} | {h for h in on_disk if h.startswith("args_")} <= per_tu[src], src
    assert "ramp_hip_tools.h" in per_tu["bench.hip"]
    for src, d in per_tu.items():                           # every header on disk is reached by somebody; nobody includes a file that is not there
        assert src in d
    assert on_disk <= set().union(*per_tu.values()), on_disk - set().union(*per_tu.values())
    B.build(force=False, verbose=False)                     # everything current first
    assert B.stale_sources() == []
    hdr = os.path.join(csrc, "tokmma.h")
    st = os.stat(hdr)
    try:
        newest = max(os.path.getmtime(op) for _, op in B.stale_sources(force=True))
        os.utime(hdr, (newest + 10, newest + 10))           # "edited after the last build"
        stale = {os.path.basename(sp) for sp, _ in B.stale_sources()}
        assert stale == {src for src, d in per_tu.items() if "tokmma.h" in d}, stale
        assert {"ffx.hip", "ffx16.hip", "tkl.hip", "tkl16.hip", "atk.hip", "atl.hip", "tkc.hip", "tkw.hip"} <= stale, stale
        assert not ({"sampler.hip", "rowops.hip", "gemm.hip", "scene.hip", "metrics.hip", "engine.hip"} & stale), stale
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
    assert B.stale_sources() == []


def test_philox_replica_known_answer():
    """tests/util.philox_normal (the host replica the GPU test compares ramp_philox_normal with) against the published
    known-answer vector of Philox4x32-10 (Random123 kat_vectors: counter 0, key 0) and the moments of its normals."""
    import util
    z, r = util.philox_normal(0, 0, 4)
    assert [int(v) for v in r[:4]] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    z, _ = util.philox_normal(5, 0, 1 << 20)
    assert abs(float(z.mean())) < 4e-3 and abs(float(z.std()) - 1.0) < 4e-3 and np.isfinite(z).all()
    a, _ = util.philox_normal(5, 3, 8)           # offset = groups of four elements
    assert np.array_equal(a, z[12:20])


def test_sampler_helper_methods_against_reference_fixture():
    """predict_start_from_noise / predict_noise_from_start / q_posterior / deep_repeat_tensor of the sampler classes
    (diffusion_model_static.py:96-147; the blocked ``repeat`` of the 3-D and dynamic classes, diffusion_model_3d.py:124-142)
    against outputs of the imported reference, for predict_epsilon True and False (the constructor default)."""
    from ramp_amd.models import (DynamicGaussianDiffusionModel, GaussianDiffusionModel3d, StaticGaussianDiffusionModel,
                                 TemporalUnetInference)
    g = np.load(f"{GOLDEN}/boundary_cases.npz")
    x, z = torch.from_numpy(g["x"]), torch.from_numpy(g["z"])
    t = torch.full((3,), int(g["t"]), dtype=torch.long)
    net = TemporalUnetInference(n_support_points=48, state_dim=4)
    for pe, tag in ((True, "eps"), (False, "x0")):
        dm = StaticGaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=pe)
        assert np.array_equal(dm.predict_start_from_noise(x, t, z).numpy(), g[f"psn_{tag}"])
        assert np.array_equal(dm.predict_noise_from_start(x, t, z).numpy(), g[f"pns_{tag}"])
        qm, qv, qlv = dm.q_posterior(x_start=z, x_t=x, t=t)
        assert np.array_equal(qm.numpy(), g[f"q_mean_{tag}"]) and np.array_equal(qv.numpy(), g[f"q_var_{tag}"])
        assert np.array_equal(qlv.numpy(), g[f"q_logvar_{tag}"])
    assert StaticGaussianDiffusionModel(model=net, n_diffusion_steps=25).predict_epsilon is False      # the reference's default
    dm = StaticGaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True)
    pts = torch.from_numpy(g["cloud"]).unsqueeze(0)
    xr, tr, trj, obr = dm.deep_repeat_tensor(x, torch.arange(3), z, pts, 2)
    assert np.array_equal(xr.numpy(), g["rep_x"]) and np.array_equal(tr.numpy(), g["rep_t"])
    assert np.array_equal(trj.numpy(), g["rep_traj"]) and list(obr.shape) == list(g["rep_obst_shape"])
    for cls, kw in ((GaussianDiffusionModel3d, {}), (DynamicGaussianDiffusionModel, {})):
        d3 = cls(model=net, n_diffusion_steps=25, predict_epsilon=True, **kw)
        xr, tr, _, obr = d3.deep_repeat_tensor(x, torch.arange(3), z, pts, 2)
        assert torch.equal(xr, torch.cat([x, x])) and tr.tolist() == [0, 1, 2, 0, 1, 2] and obr.shape[0] == 2
