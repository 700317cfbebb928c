"""No-GPU checks of the boundary: the C-ABI library loads, exports every symbol include/freepose_hip.h declares, the
ctypes table mirrors the header, the product fails loudly without its extension, and the product never imports the
oracle."""
import ctypes
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "freepose_hip.h"


def _declared():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(fp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from freepose_amd import _lib, build
    build.build_hip(verbose=False)
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in freepose_hip.h but not exported"


def test_product_library_carries_no_lab():
    """The shipped library reads no environment variable and has no process-global option: every FP_* measurement toggle, the
    alternative GEMM / attention kernels and the wrong-numerics hooks exist only in the lab build (-DFP_LAB).  One GEMM kernel per
    (epilogue, tile tier): 9 epilogues x 3 tiers, the big tier in two store policies (streaming / cached output, round 5) + the two GELU
    helper kernels."""
    from freepose_amd import _lib, build
    build.build_hip(verbose=False)
    blob = _lib.LIB_PATH.read_bytes()
    for needle in (b"FP_GEMM", b"FP_ATTN", b"FP_LN_FUSED", b"FP_TOPK_SELECT", b"FP_RASTER_TILED", b"gemm_dbg", b"gemm_variant"):
        assert needle not in blob, needle
    nm = subprocess.run(["nm", "-D", "--undefined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in nm, "the product library must not read the environment"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    assert not hasattr(lib, "fp_set_option") and not hasattr(lib, "fp_lab_set_option") and hasattr(lib, "fp_ctx_set_option")
    obj = ROOT / "freepose_amd" / "lib" / "obj" / "gemm_bf16.o"
    stubs = [ln for ln in subprocess.run(["nm", str(obj)], capture_output=True, text=True, check=True).stdout.splitlines()
             if "__device_stub__" in ln]
    assert 0 < len(stubs) <= 39, len(stubs)
    asm_obj = ROOT / "freepose_amd" / "lib" / "obj" / "gemm_asm.o"     # the hand-scheduled tier: one kernel per epilogue it can be dispatched for
    asm_stubs = [ln for ln in subprocess.run(["nm", str(asm_obj)], capture_output=True, text=True, check=True).stdout.splitlines()
                 if "__device_stub__" in ln]
    assert 0 < len(asm_stubs) <= 4, len(asm_stubs)
    for src in (ROOT / "freepose_amd" / "csrc").glob("*"):
        text = src.read_text()
        depth, bad = 0, []
        for i, ln in enumerate(text.splitlines(), 1):     # every getenv / fp_opt_get sits inside an #ifdef FP_LAB block
            t = ln.strip()
            if t.startswith("#ifdef FP_LAB"):
                depth += 1
            elif depth and t.startswith(("#if ", "#ifdef ", "#ifndef ")):
                depth += 1
            elif depth and t.startswith("#endif"):
                depth -= 1
            elif depth == 0 and re.search(r"\bgetenv\s*\(|\bfp_opt_get\s*\(", ln) and not t.startswith("//"):
                bad.append((src.name, i))
        assert not bad, bad


def test_attention_kernels_have_no_scratch(tmp_path):
    """every attention instantiation the product can dispatch (short tail first / plain, pre-scaled q / unscaled) compiles to 128
    registers and ZERO scratch: a spill inside the K/V loop shares vmcnt with the LDS-DMA ring and drains it every tile (710 against
    1014 TFLOP/s when it happened, profiles/r04_ab.md §5).  Read from the code object's metadata notes."""
    from freepose_amd import build
    build.build_hip(verbose=False)
    obj = ROOT / "freepose_amd" / "lib" / "obj" / "attention.o"
    llvm = Path("/opt/rocm/lib/llvm/bin")
    if not (llvm / "clang-offload-bundler").exists() or not (llvm / "llvm-readelf").exists():
        import pytest
        pytest.skip("ROCm LLVM tools not found")
    fat, co = tmp_path / "attention.fatbin", tmp_path / "attention.co"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)], check=True)
    subprocess.run([str(llvm / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950:sramecc+", f"--input={fat}",
                    f"--output={co}", "--unbundle"], check=True)
    notes = subprocess.run([str(llvm / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    kernels = {}
    name = None
    for ln in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
            kernels[name] = {}
        for key in ("private_segment_fixed_size", "vgpr_spill_count", "vgpr_count"):
            m = re.search(r"\.%s:\s+(\d+)" % key, ln)
            if m and name:
                kernels[name][key] = int(m.group(1))
    attn = {k: v for k, v in kernels.items() if "attn_fwd_kernel" in k}
    assert len(attn) == 4, sorted(attn)
    for k, v in attn.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["vgpr_count"] <= 128, (k, v)


def test_ctypes_table_mirrors_header():
    from freepose_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} parameters, ctypes table {len(args)}"


def test_no_compute_entry_points_work_without_gpu():
    from freepose_amd import _lib
    lib = _lib.load()
    assert lib.fp_version() >= 100
    import numpy as np
    out = np.empty((4, 3, 3))
    assert lib.fp_generate_rotations(4, out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.allclose(out @ out.transpose(0, 2, 1), np.eye(3), atol=1e-12)
    # error convention: non-zero status + message, no exception across the ABI
    assert lib.fp_generate_rotations(0, None) != 0
    assert b"generate_rotations" in lib.fp_last_error()
    h = ctypes.c_void_p()
    import torch
    if not torch.cuda.is_available():
        assert lib.fp_ctx_create(0, ctypes.byref(h)) != 0  # no device: loud failure, not a fallback


def test_comm_entry_points_refuse_bad_arguments_without_a_gpu():
    """fp_comm_init never hangs on a call it can refuse: null context / id, nranks < 1, rank outside [0, nranks) -> status + message
    (the rendezvous timeout itself needs a device: tests/test_gpu_comm.py)"""
    import ctypes as C
    from freepose_amd import _lib
    lib = _lib.load()
    uid = (C.c_char * 128)()
    for ctx, nranks, rank, ident in ((None, 2, 0, uid), (None, 0, 0, uid), (None, 2, 5, uid), (None, 2, 0, None)):
        assert lib.fp_comm_init(ctx, nranks, rank, ident) == 1 and b"comm_init" in lib.fp_last_error()
    assert lib.fp_comm_size(None) == 1 and lib.fp_comm_rank(None) == 0 and lib.fp_comm_destroy(None) == 0
    assert lib.fp_comm_unique_id(None) != 0


def test_missing_extension_fails_loudly(tmp_path):
    from freepose_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(tmp_path / "libfreepose_hip.so")


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    offenders = []
    for p in list((ROOT / "freepose_amd").rglob("*.py")) + list((ROOT / "scripts").glob("*.py")) + list((ROOT / "src").glob("*.py")):
        t = p.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", t, flags=re.M) or "fp_oracle" in t and "build_oracle" not in t and p.name != "build.py":
            offenders.append(str(p))
    assert not offenders, offenders
    code = "import sys; import freepose_amd.ops, freepose_amd.retrieval, freepose_amd.parallel; " \
           "import freepose_amd.src.pipeline.estimators.online_pose_estimator; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for f in (ROOT / "freepose_amd" / "csrc").glob("*"):
        assert "oracle/" not in f.read_text() or f.name in ("retrieval.hip", "raster.hip"), f  # comments citing the checker only


def test_cli_flags_match_reference():
    """flag names / defaults of the three drivers"""
    from freepose_amd.scripts import dino_inference_video as v
    ns = v.build_parser().parse_args(["--video", "x", "--proposals", "p.json"])
    assert (ns.layer, ns.depth_method, ns.bbox_extend, ns.batch_size, ns.template_cache_size, ns.cache_size) == (22, "zoedepth", 0.05, 128, 21, 50)
    assert ns.viz is False and ns.no_rescore is False and ns.save_all_cache is False
    import inspect
    from freepose_amd.scripts import dino_inference as d, extract_retrieval_features as e
    src = inspect.getsource(d.build_parser)
    for flag in ("--dataset", "--split", "--proposals", "--layer", "--depth_method", "--bbox_extend", "--batch_size", "--cache_size", "--save_all_cache"):
        assert flag in src
    src = inspect.getsource(e.build_parser)
    for flag in ("--shards_folder", "--filelist", "--feature", "--layer", "--mesh_per_job", "--batch_size"):
        assert flag in src
    ne = e.build_parser().parse_args([])
    assert (ne.shards_folder, ne.filelist, ne.feature, ne.layer, ne.mesh_per_job, ne.batch_size) == ("objaverse_shards", "mesh_cache.csv", "ffa", 22, 100, 128)
    assert d.CSV_COLUMNS == ["scene_id", "im_id", "obj_id", "score", "R", "t", "bbox_visib", "scale", "time"]


def test_missing_checkpoint_fails_closed_and_self_launch_command(tmp_path, monkeypatch):
    """no checkpoint and no explicit request for random weights -> FileNotFoundError before anything touches the GPU
    (reference: torch.hub.load would have raised, src/pipeline/retrieval/dino.py:10); `--gpus N` without a launcher re-execs
    under torch.distributed.run on 127.0.0.1."""
    import subprocess
    import torch
    from freepose_amd import parallel
    from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    monkeypatch.setenv("FREEPOSE_DINOV2_WEIGHTS", str(tmp_path / "nowhere.pth"))
    monkeypatch.delenv("FREEPOSE_ALLOW_RANDOM_WEIGHTS", raising=False)
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path))
    with pytest.raises(FileNotFoundError, match="allow_random_weights"):
        DINOv2FeatureExtractor("dinov2_vitl14_reg")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    parallel.self_launch(1, ["bench.py"], ["--gpus", "1"])                     # one rank: no launcher, returns
    assert not seen
    with pytest.raises(SystemExit) as e:
        parallel.self_launch(4, ["bench.py"], ["--gpus", "4", "--steps", "2"])
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "2"]
    monkeypatch.setenv("WORLD_SIZE", "4")                                       # already a rank: never re-launch
    seen.clear()
    parallel.self_launch(4, ["bench.py"], [])
    assert not seen


def test_more_ranks_than_gpus_is_refused_unless_overridden(monkeypatch):
    """A box with fewer GPUs than ranks must not produce a plausible N-"GPU" result: init_from_env / self_launch exit non-zero
    unless FP_ALLOW_SHARED_GPU=1 (then the launcher falls back to gloo and the result line is stamped shared_devices)."""
    import subprocess
    import torch
    from freepose_amd import parallel
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FP_ALLOW_SHARED_GPU", "FP_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    called = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: called.update(cmd=cmd, env=env) or 0)
    with pytest.raises(SystemExit) as e:
        parallel.self_launch(2, ["bench.py"], ["--gpus", "2"])
    assert "FP_ALLOW_SHARED_GPU" in str(e.value.code) and not called            # a message = exit status 1, nothing launched
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        parallel.init_from_env("nccl")                                          # a rank under somebody else's launcher: same refusal
    assert "2 ranks on this node but only 1 visible GPU" in str(e.value.code)
    # two nodes x one GPU (LOCAL_WORLD_SIZE from the launcher): 2 ranks in the world, ONE on this node's one GPU -> not refused
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    parallel._require_own_devices(2)
    assert parallel.local_world(2) == 1
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setenv("FP_ALLOW_SHARED_GPU", "1")
    with pytest.raises(SystemExit) as e:
        parallel.self_launch(2, ["bench.py"], ["--gpus", "2"])
    assert e.value.code == 0 and called["env"]["FP_DIST_BACKEND"] == "gloo" and "--nproc-per-node=2" in called["cmd"]
    rep = parallel.rank_report.__doc__
    assert "shared_devices" in rep


def test_reference_import_paths_resolve_in_both_forms():
    """`import src.pipeline.X as m` and `from src.pipeline.X import name` for every reference module on the path"""
    code = (
        "import src.utils.bbox_utils as a, src.pipeline.utils as b, src.pipeline.retrieval.dino as c, src.pipeline.retrieval.renderer as d\n"
        "import src.pipeline.estimators.pose_estimator as e, src.pipeline.estimators.online_pose_estimator as f\n"
        "import src.pipeline.estimators.tracking_refiner as g, src.pipeline.refiner_utils as h, src.dataloader.template as i, src.dataloader.bop as j\n"
        "from src.pipeline.estimators.tracking_refiner import TrackingRefiner\n"
        "from src.pipeline.refiner_utils import crop_image, update_K_with_crop\n"
        "assert all(m.__name__.startswith('freepose_amd.src.') for m in (a, b, c, d, e, f, g, h, i, j))\n")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
