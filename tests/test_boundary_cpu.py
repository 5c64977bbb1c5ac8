"""CPU: the drop-in boundary.  The C-ABI library loads and exports every symbol include/nr3d_hip.h declares; the
product path never touches the oracle and has no CPU fallback; host-side logic (meta, wrappers, dispatch errors)."""
import ctypes
import os
import pickle
import re
import subprocess

import pytest
import torch

from nr3d_lib_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol(hiplib):
    header = open(os.path.join(ROOT, "include", "nr3d_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(nr3d_[A-Za-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 30, declared
    missing = [s for s in declared if not hasattr(hiplib, s)]
    assert not missing, f"declared in include/nr3d_hip.h but not exported: {missing}"
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "nr3d_lib_amd", "libnr3d_hip.so")],
                        capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (nr3d_[A-Za-z0-9_]+)", nm)))
    assert exported == declared, (set(exported) ^ set(declared))
    # 2: map_col; 3: forest entry points take param_dtype; 4: option table, bwd_fused removed; 5: nr3d_sort_pairs_u32;
    # 6: NR3D_ABI_VERSION in the header, signature table generated into the package (round 6)
    m = re.search(r"#define\s+NR3D_ABI_VERSION\s+(\d+)", header)
    assert hiplib.nr3d_abi_version() == _hip.ABI_VERSION == int(m.group(1)) >= 6


def test_signature_table_is_the_header():
    """nr3d_lib_amd/_abi.py (what the loader declares argtypes from) is exactly what tools/gen_abi.py makes of the header, covers every
    declared entry point, and the loader has applied it"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_abi", os.path.join(ROOT, "tools", "gen_abi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    version, sigs = gen.parse(open(gen.HEADER).read())
    assert open(gen.OUT).read() == gen.render(version, sigs), "nr3d_lib_amd/_abi.py is stale: python tools/gen_abi.py"
    from nr3d_lib_amd import _abi
    assert _abi.ABI_VERSION == version and _abi.SIGNATURES == sigs
    header = re.sub(r"/\*.*?\*/", "", open(gen.HEADER).read(), flags=re.S)
    assert sorted(sigs) == sorted(set(re.findall(r"\b(nr3d_[A-Za-z0-9_]+)\s*\(", header)))
    l = _hip.lib()
    for name, (ret, args) in sigs.items():
        fn = getattr(l, name)
        assert len(fn.argtypes) == len(args), name
        assert all((t is ctypes.c_void_p) == (a == "ptr") for t, a in zip(fn.argtypes, args)), name


def test_vendored_package_needs_no_header(tmp_path):
    """a copy of nr3d_lib_amd/ ALONE (no include/, no tools/, no csrc/) imports, loads the library and resolves every entry point"""
    import shutil
    import sys
    dst = tmp_path / "site" / "nr3d_lib_amd"
    shutil.copytree(os.path.join(ROOT, "nr3d_lib_amd"), dst, ignore=shutil.ignore_patterns("csrc", "__pycache__"))
    code = ("import nr3d_lib_amd, nr3d_lib_amd._hip as H, os; l = H.lib(); "
            "assert os.path.dirname(H.LIB_PATH) == os.path.dirname(nr3d_lib_amd.__file__); "
            "import nr3d_lib_amd.bindings._lotd, nr3d_lib_amd.bindings._pack_ops, nr3d_lib_amd.bindings._occ_grid; "
            "print('OK', l.nr3d_abi_version(), nr3d_lib_amd.__file__)")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=str(tmp_path / "site")))
    assert r.returncode == 0 and "OK" in r.stdout and str(tmp_path) in r.stdout, r.stdout + r.stderr


def test_library_is_gfx950_only():
    """the fat binary embeds code objects for exactly one target"""
    blob = open(os.path.join(ROOT, "nr3d_lib_amd", "libnr3d_hip.so"), "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "nr3d_lib_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "liboracle" in txt:
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"\bimport oracle\b", bench)]
    assert len(uses) == 1 and bench.rfind("def cpu_baseline", 0, uses[0]) > bench.rfind("\ndef ", 0, bench.rfind("def cpu_baseline", 0, uses[0]))


def test_no_cpu_fallback(hiplib):
    from nr3d_lib_amd.bindings import _lotd, _pack_ops, _occ_grid
    m = _lotd.LoDMeta(3, [8, 16], [2, 2], ["Dense", "Hash"], 1024)
    x, p = torch.rand(4, 3), torch.zeros(m.n_params)
    with pytest.raises(RuntimeError, match="GPU only|CPU tensor"):
        _lotd.lod_fwd(m, x, p)
    with pytest.raises(RuntimeError, match="GPU only|CPU tensor"):
        _pack_ops.packed_sum(torch.rand(5), torch.tensor([[0, 5]]))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _occ_grid.ray_marching(torch.rand(2, 3), torch.rand(2, 3), torch.zeros(2), torch.ones(2), torch.zeros(6),
                               torch.zeros(4, 4, 4, dtype=torch.bool), _occ_grid.ContractionType.AABB, 0.1, 1e10, 0., 8, True)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nr3d_lib_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.lib()


def test_meta_surface(hiplib):
    from nr3d_lib_amd.bindings import _lotd
    m = _lotd.LoDMeta(3, [[8, 6, 5], [12, 9, 7]], [4, 8], ["vm", "CP"], None, True)
    assert m.level_types == [int(_lotd.LoDType.VectorMatrix), int(_lotd.LoDType.CP)] and m.level_types_str == ["vm", "CP"]
    assert m.level_res == [0, 0] and m.level_res_multidim == [[8, 6, 5], [12, 9, 7]]
    assert m.level_sizes == [6 * 5 + 8 * 5 + 8 * 6 + 8 + 6 + 5, 12 + 9 + 7]
    assert m.n_feat_per_pseudo_lvl == 4 and m.n_pseudo_levels == 3 and m.map_levels == [0, 1, 1] and m.map_cnt == [0, 0, 1]
    assert m.n_encoded_dims == 12 and m.interpolation_type == _lotd.InterpolationType.Smoothstep and not m.c_hash_only
    assert (m.c_profile, m.c_bmm_backend, m.c_prefetch, m.c_permute_dydx) == (False, True, True, True)
    assert int(_lotd.LoDType.Hash) == 7 and _lotd.Dense == _lotd.LoDType.Dense      # export_values()
    assert _lotd.string_to_lod_type("NPlane") == _lotd.LoDType.NPlaneSum
    with pytest.raises(RuntimeError, match="Invalid lod type"):
        _lotd.LoDMeta(3, [8], [2], ["Octree"])
    with pytest.raises(RuntimeError, match="exceeds maximum level"):
        _lotd.LoDMeta(3, [8] * 33, [2] * 33, ["Dense"] * 33)
    with pytest.raises(RuntimeError, match="too large"):
        _lotd.LoDMeta(3, [2049], [2], ["Dense"])
    with pytest.raises(RuntimeError, match="<= 1024"):
        _lotd.LoDMeta(3, [8] * 17, [64] * 17, ["Dense"] * 17)


def test_module_surface(hiplib):
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTD, LoDType, generate_meta, get_lotd_cfg
    enc = LoTD(3, [8, 16, 32], 2, ["Dense", "Dense", "Hash"], log2_hashmap_size=10, dtype=torch.float)
    assert (enc.in_features, enc.out_features, enc.n_levels) == (3, 6, 3)
    assert enc.level_res == [8, 16, 32] and enc.level_types[-1] == LoDType.Hash and enc.loss_scale == 1.0
    assert enc.n_params == 8 ** 3 * 2 + 16 ** 3 * 2 + 1024 * 2 and "num_params" in enc.extra_repr()
    enc2 = pickle.loads(pickle.dumps(enc))
    assert enc2.meta.level_offsets == enc.meta.level_offsets
    assert LoTD(3, [8], 2, "Dense").loss_scale == 128.0                    # default dtype is half, like the reference
    cfg = get_lotd_cfg("gen_ngp", 3, num_levels=4)
    assert generate_meta(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], cfg["hashmap_size"]).n_levels == 4
    from nr3d_lib_amd.profile import profile
    with profile("anything"):
        pass
    assert profile(lambda: 7)() == 7


def test_raymarch_records():
    from nr3d_lib_amd.graphics.raymarch import RaymarchRetSingle, RaymarchRetBatched
    r = RaymarchRetSingle(0, None, None, None, None, None, None, None, None)
    assert len(list(r)) == 9 and r["num_hit_rays"] == 0
    assert len(list(RaymarchRetBatched(0, *[None] * 9))) == 10


def test_lotd_encoding_module_cpu_surface():
    """LoTDEncoding: per-type init bounds, whole-level views, state_dict round trip (no kernel is launched)"""
    import torch
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDEncoding
    cfg = dict(lod_res=[8, 12, 10, 14, 9], lod_n_feats=[4, 4, 8, 2, 4], lod_types=["Dense", "VM", "CP", "NPlaneSum", "Hash"],
               hashmap_size=512)
    torch.manual_seed(0)
    e = LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float)
    assert e.flattened_params.dtype == torch.float32 and e.out_features == sum(cfg["lod_n_feats"])
    bounds = [1e-4, 1e-2, 1e-4 ** (1 / 3), 1e-4, 1e-4]
    for l, b in enumerate(bounds):
        p = e.get_level_param(l)
        assert p.shape[1] == cfg["lod_n_feats"][l] and 0.5 * b < float(p.detach().abs().max()) <= b * (1 + 1e-6)
    assert tuple(e.get_level_param(0, "vol").shape) == (8, 8, 8, 4)
    assert tuple(e.get_level_param(1, "vec", 0).shape) == (cfg["lod_res"][1], cfg["lod_n_feats"][1])   # a VM level's x lines
    e.set_level_param(1, "vec", 0, value=torch.full((cfg["lod_res"][1], cfg["lod_n_feats"][1]), 0.5))
    assert float(e.get_level_param(1, "vec", 0).detach().min()) == 0.5 and float(e.get_level_param(1, "vec", 1).detach().abs().max()) <= 1e-2
    e.set_level_param(1, "vec", 0, value=e.get_level_param(1, "vec", 1).detach().clone())
    e.set_level_param(4, value=torch.ones(512, 4))
    assert float(e.get_level_param(4).detach().min()) == 1.0 and float(e.get_level_param(3).detach().abs().max()) <= 1e-4
    e2 = LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float, param_init_cfg={"type": "normal", "std": 0.1})
    e2.load_state_dict(e.state_dict())
    assert torch.equal(e2.flattened_params, e.flattened_params) and e2.lotd_cfg == cfg
    auto = LoTDEncoding(3, lotd_auto_compute_cfg=dict(type="gen_ngp", num_levels=4), dtype=torch.half)
    assert auto.lotd.n_levels == 4 and auto.inference_param.dtype == torch.half
    from nr3d_lib_amd.models.spatial import BatchedBlockSpace
    assert isinstance(LoTDEncoding(3, lotd_cfg=cfg, space_cfg=dict(type="batched", bounding_size=4.0)).space, BatchedBlockSpace)
    an = LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float, anneal_cfg=dict(type="hardmask", stop_it=100, start_level=1))
    an.set_anneal_iter(0)
    assert an.max_level == 1 and an.window is None


def test_reference_call_sites_bind():
    """INTEGRATION.md section 1 (zero-code-change route): every attribute the reference's Python takes from its three compiled
    modules exists on the twin, and every call it makes binds against the twin's signature.  The call sites are DATA
    (tests/golden/ref_backend_uses.json: module, name, file:line, positional count, keyword names), extracted from the reference
    with ast by tests/golden/make_ref_backend_uses.py in the build container; nothing of the reference is read here."""
    import importlib
    import inspect
    import json
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_backend_uses.json")))
    assert d["n_uses"] == len(d["uses"]) >= 100 and d["n_names"] >= 40
    twins = {m: importlib.import_module(f"nr3d_lib_amd.bindings.{m}") for m in ("_lotd", "_pack_ops", "_occ_grid")}
    missing, mismatched, bound = [], [], 0
    for r in d["uses"]:
        where = f'{r["module"]}.{r["name"]} ({r["file"]}:{r["line"]})'
        obj = getattr(twins[r["module"]], r["name"], None)
        if obj is None:
            missing.append(where)
            continue
        c = r.get("call")
        if c is None or c["star_args"] or c["star_kwargs"]:
            continue
        try:
            sig = inspect.signature(obj)
        except (TypeError, ValueError):
            continue                              # enum classes etc.: attribute presence is the contract
        try:
            sig.bind(*([None] * c["n_positional"]), **{k: None for k in c["keywords"]})
            bound += 1
        except TypeError as e:
            mismatched.append(f"{where}: {e}")
    assert not missing, missing
    assert not mismatched, mismatched
    assert bound >= 80


def test_option_table_round_trip(hiplib):
    """the library's selectable code paths (include/nr3d_hip.h NR3D_OPT_*): one table behind nr3d_set_option / nr3d_get_option, no
    environment variable; Python names cover every id; a negative value restores the default; unknown ids fail"""
    import ctypes as C
    import re
    from nr3d_lib_amd import _hip
    header = open(os.path.join(ROOT, "include", "nr3d_hip.h")).read()
    ids = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"NR3D_OPT_([A-Z_0-9]+) = (\d+)", header))
    count = ids.pop("COUNT")
    assert sorted(ids.values()) == list(range(count))
    assert sorted(_hip.OPTION_IDS.values()) == list(range(count)), "nr3d_lib_amd._hip.OPTION_IDS must name every option of the header"
    for name, i in _hip.OPTION_IDS.items():
        assert ids[name.upper()] == i, name
        d = _hip.get_option(name)
        _hip.set_option(name, 7)
        assert _hip.get_option(name) == 7
        _hip.set_option(name, -1)
        assert _hip.get_option(name) == d
        with _hip.options(**{name: 3}):
            assert _hip.get_option(name) == 3
        assert _hip.get_option(name) == d
    hiplib.nr3d_get_option.restype = C.c_int64
    assert hiplib.nr3d_get_option(C.c_int(count)) == -1
    assert hiplib.nr3d_set_option(C.c_int(count), C.c_int64(1)) != 0
    # nothing in the shipped library reads the environment
    blob = open(_hip.LIB_PATH, "rb").read()
    assert b"getenv" not in blob or b"NR3D_FWD_DBG" not in blob
    assert blob.count(b"NR3D_LOTD_") == 0 and blob.count(b"NR3D_PAIR_") == 0 and blob.count(b"NR3D_PACK_") == 0
