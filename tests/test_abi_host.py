"""CPU-side checks: the C-ABI library loads and exports every symbol include/gmf_hip.h declares, fails
loudly without a device, and the host logic (weight packing, state_dict surface) is right.  No GPU compute."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gmf_amd import _lib
    lib = _lib.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gmf_hip.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    assert lib.gmf_abi_version() == 5


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device error path")
def test_no_device_fails_loudly():
    from gmf_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.Handle(0)
    import gmf_amd
    m = gmf_amd.FusionLayer(depth=0, dim=128, latent_dim=128, cross_heads=1, cross_dim_head=64).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, 128), queries_encoder=torch.zeros(1, 8, 128))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gmf_amd.rigid_transform_3d(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gmf_amd.global_registration_batched(torch.zeros(8, 3), torch.zeros(8, 3), torch.ones(8, 1), [0, 8])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gmf_amd.nn_match(torch.zeros(4, 32), torch.zeros(5, 32))


def test_state_dict_surface_matches_reference(golden_dir):
    """Key names and shapes of the drop-in modules equal the reference's (captured by oracle/gen_fixtures.py)."""
    import gmf_amd
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    m = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128)
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    hot_ref = {k: v for k, v in ref["pointdsc"].items() if not k.startswith("encoder.image_encoder.")}
    hot_mine = {k: v for k, v in mine.items() if not k.startswith("encoder.image_encoder.")}
    assert hot_mine == hot_ref
    img_mine = {k: v for k, v in mine.items() if k.startswith("encoder.image_encoder.")}
    assert img_mine and all(ref["pointdsc"].get(k) == v for k, v in img_mine.items())
    f = gmf_amd.FusionLayer(depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8, cross_dim_head=64,
                            latent_dim_head=64, pe=True)
    assert {k: list(v.shape) for k, v in f.state_dict().items()} == ref["fusion_layer_pe"]
    p = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8, cross_dim_head=64,
                            latent_dim_head=64)
    assert {k: list(v.shape) for k, v in p.state_dict().items()} == ref["perceiver_io_128"]


def test_wide_fusion_packing():
    """DGR bottleneck widths (256/128/128): blob sizes match the stage counts of fusion_wide.hip."""
    from gmf_amd import packing, synthetic
    sd = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, 256, 128, pe=True, out_to_query=True), seed=1)
    f = packing.pack_fusion(sd, "", pe=True)
    assert (f["latent_dim"], f["d_head"]) == (256, 128)
    assert f["ctx_wst"].numel() == 8 * 4096 and f["attn_wst"].numel() == 16 * 4096 and f["ff_wst"].numel() == 192 * 4096
    bad = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, 512, 64, pe=False), seed=1)
    with pytest.raises(NotImplementedError):
        packing.pack_fusion(bad, "", pe=False)


def test_unsupported_configurations_raise():
    import gmf_amd
    # [r5] every constructor argument of the reference's FusionLayer (golden F23) and NonLocalBlock (F24) is honoured
    fl = gmf_amd.FusionLayer(depth=2, dim=128, latent_dim=128)
    assert "layers.1.0.fn.to_kv.weight" in fl.state_dict()
    assert gmf_amd.NonLocalBlock(num_channels=64, num_heads=2).projection_q.weight.shape == (64, 64, 1)
    with pytest.raises(NotImplementedError):
        gmf_amd.NonLocalBlock(num_channels=64, num_heads=3)


def test_p32_image_definition():
    from gmf_amd import packing
    W = torch.arange(64 * 16, dtype=torch.float32).reshape(64, 16)
    img = packing.p32(W).reshape(2, 2, 64, 4)          # [out-block][g][lane][e]
    for mb, g, lane, e in [(0, 0, 0, 0), (1, 1, 37, 2), (0, 1, 63, 3), (1, 0, 31, 1)]:
        i, h = lane & 31, lane >> 5
        assert img[mb, g, lane, e] == W[32 * mb + i, 8 * g + 4 * h + e]


def test_bn_folding_and_blob_sizes():
    """Folded conv+BN equals conv followed by eval BatchNorm; blob sizes match the kernels' stage counts."""
    from gmf_amd import packing, synthetic
    from oracle import gmf_oracle as O
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 2, 128), seed=3)
    pc = "encoder.blocks.PointCN_layer_1."
    W, b = packing.fold_bn(sd[pc + "0.weight"][:, :, 0], sd[pc + "0.bias"], sd, pc + "1.")
    x = torch.randn(1, 50, 128)
    assert torch.allclose(torch.relu(x @ W.t() + b), O.point_cn(sd, pc, x), atol=1e-5)
    wst, vec = packing.pack_front(sd, 0, with_layer0=True)
    assert wst.numel() == 16 * 4096 and vec.numel() == packing.FRONT_VEC
    wst, vec = packing.pack_tail(sd, 1)
    assert wst.numel() == 5 * 4096
    f = packing.pack_fusion(sd, "encoder.blocks.NonLocal_layer_0.fusion_layer_2.", pe=True)
    assert f["ff_wst"].numel() == 48 * 4096 and f["attn_wst"].numel() == 4 * 4096 and f["ctx_wst"].numel() == 4 * 4096
    wst, vec = packing.pack_head(sd)
    assert wst.numel() == 2 * 4096


def _c_blob(ptr, n):
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(n,)).view(np.uint32)


@pytest.mark.parametrize("standalone", [False, True])
def test_c_weight_packing_equals_python_packing_bit_for_bit(standalone):
    """gmf_encoder_pack_weights (C ABI, gmf_amd/csrc/gmf_pack.cpp) builds the encoder's blobs from the state_dict tensors by name:
    BatchNorm folding, softmax-scale folding, P32 images and split-fp16 planes.  Every blob equals the pure-Python packers'
    (gmf_amd/packing.py) bit for bit.  With h = NULL the blobs stay in host memory, so this runs without a GPU."""
    from gmf_amd import _lib, packing, synthetic
    L = 3
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, L, 128), seed=11)
    if standalone:
        sd = {k: v for k, v in sd.items() if k.startswith("encoder.blocks.NonLocal_layer_0.")}
        L = 1
    lib = _lib.load_library()
    arr, keep = _lib.tensor_list(sd)
    out = ctypes.c_void_p()
    flags = _lib.GMF_PACK_STANDALONE_BLOCK if standalone else 0
    assert lib.gmf_encoder_pack_weights(None, arr, len(arr), L, flags, ctypes.byref(out)) == 0
    try:
        w = lib.gmf_packed_encoder_weights(out).contents
        ref, split = packing.python_packed_encoder(sd, L, standalone_block=standalone)
        assert split and w.num_layers == L
        names = [f[0] for f in _lib.EncoderWeights._fields_]
        checked = 0
        for name, t in ref.items():
            assert name in names
            ptr = getattr(w, name)
            assert ptr, name
            assert np.array_equal(_c_blob(ptr, t.numel()), t.reshape(-1).numpy().view(np.uint32)), name
            checked += 1
        assert checked == (15 if standalone else 26)      # (standalone block: no Fusion-1, no classifier)
        assert (w.front_wst_stride, w.tail_wst_stride, w.ff_wst_stride) == (packing.FRONT_WST, packing.TAIL_WST, packing.FF_WST)
        sig, sig_d, sp, amax = ctypes.c_float(), ctypes.c_float(), ctypes.c_int(), ctypes.c_float()
        assert lib.gmf_packed_encoder_info(out, ctypes.byref(sig), ctypes.byref(sig_d), ctypes.byref(sp), ctypes.byref(amax)) == 0
        if not standalone:
            assert sig.value == float(sd["sigma"].reshape(-1)[0]) and sig_d.value == float(sd["sigma_spat"].reshape(-1)[0])
        assert sp.value == 1 and 0 < amax.value <= 65504
    finally:
        lib.gmf_packed_encoder_free(out)


def test_pv_guard_thresholds_of_the_packer():
    """gmf_encoder_weights::pv_guard (ABI 5): per layer the squared row norm of the layer input up to which the attention's P V
    cross products may run on the fp8 pipe, from the spectral norms of projection_q / projection_k (power iteration in the C
    packer; an SVD here).  Scaling a layer's projections moves its threshold the way the bound says; biases that pass the bound
    by themselves give -1 (always guarded)."""
    from gmf_amd import _lib, packing, synthetic
    L = 3
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, L, 128), seed=11)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["encoder.blocks.NonLocal_layer_1.projection_q.weight"] *= 7.0
    sd["encoder.blocks.NonLocal_layer_2.projection_q.bias"] = torch.full_like(sd["encoder.blocks.NonLocal_layer_2.projection_q.bias"], 400.0)
    sd["encoder.blocks.NonLocal_layer_2.projection_k.bias"] = torch.full_like(sd["encoder.blocks.NonLocal_layer_2.projection_k.bias"], 400.0)
    lib = _lib.load_library()
    arr, keep = _lib.tensor_list(sd)
    out = ctypes.c_void_p()
    assert lib.gmf_encoder_pack_weights(None, arr, len(arr), L, 0, ctypes.byref(out)) == 0
    try:
        w = lib.gmf_packed_encoder_weights(out).contents
        assert w.pv_guard
        got = _c_blob(w.pv_guard, L).view(np.float32)
        ref = packing.pv_guard_thresholds(sd, L).numpy()
        assert got[2] == -1.0 and ref[2] == -1.0          # |bq| |bk| = 400^2 * 128 > 1024 sqrt(128)
        assert np.allclose(got[:2], ref[:2], rtol=2e-3), (got, ref)
        assert 20.0 < got[0] ** 0.5 < 200.0               # seeded weights: the guard trips at row norms of a few tens
        assert 2.0 < got[0] / got[1] < 49.0               # a 7 x larger Wq: a lower threshold, by at most 7^2
    finally:
        lib.gmf_packed_encoder_free(out)


def test_c_weight_packing_rejects_and_falls_back():
    """A missing tensor is named; a weight outside the fp16 range drops the split-fp16 images (the fp32-MFMA path remains), as the
    Python packers do; unsupported widths are GMF_ERR_UNSUPPORTED_SHAPE."""
    from gmf_amd import _lib, packing, synthetic
    lib = _lib.load_library()
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 2, 128), seed=3)
    out = ctypes.c_void_p()
    miss = {k: v for k, v in sd.items() if k != "encoder.blocks.NonLocal_layer_1.projection_k.bias"}
    arr, keep = _lib.tensor_list(miss)
    assert lib.gmf_encoder_pack_weights(None, arr, len(arr), 2, 0, ctypes.byref(out)) == -1 and not out.value
    big = dict(sd)
    big["encoder.blocks.PointCN_layer_1.1.weight"] = sd["encoder.blocks.PointCN_layer_1.1.weight"] * 1e5     # folds to |256 w| > 65504
    arr, keep = _lib.tensor_list(big)
    assert lib.gmf_encoder_pack_weights(None, arr, len(arr), 2, 0, ctypes.byref(out)) == 0
    try:
        w = lib.gmf_packed_encoder_weights(out).contents
        assert not w.front_wst_h2 and not w.ff_wst_h2 and not w.tail_wst_h2 and w.front_wst and w.ff_wst
        sp = ctypes.c_int(7)
        lib.gmf_packed_encoder_info(out, None, None, ctypes.byref(sp), None)
        assert sp.value == 0
        ref, split = packing.python_packed_encoder(big, 2)
        assert not split
        assert np.array_equal(_c_blob(w.front_wst, ref["front_wst"].numel()), ref["front_wst"].reshape(-1).numpy().view(np.uint32))
    finally:
        lib.gmf_packed_encoder_free(out)
    # one FusionLayer: both widths, with and without LCPE, against packing.pack_fusion
    for lat, dh, pe in ((128, 64, True), (256, 128, True), (256, 128, False)):
        fsd = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, lat, dh, pe=pe, out_to_query=True), seed=5)
        arr, keep = _lib.tensor_list(fsd)
        pf = ctypes.c_void_p()
        assert lib.gmf_fusion_pack_weights(None, arr, len(arr), b"", int(pe), 0, ctypes.byref(pf)) == 0
        try:
            fw = lib.gmf_packed_fusion_weights(pf).contents
            assert (fw.latent_dim, fw.d_head, fw.split_fp16) == (lat, dh, 1)
            ref = packing.pack_fusion(fsd, "", pe)
            refh = packing.pack_fusion(fsd, "", pe, img=packing.p32_h2s)
            for k in ("ctx_wst", "ctx_vec", "attn_wst", "attn_vec", "ff_wst", "ff_vec"):
                assert np.array_equal(_c_blob(getattr(fw, k), ref[k].numel()), ref[k].numpy().view(np.uint32)), (lat, pe, k)
            for k in ("ctx_wst", "attn_wst", "ff_wst"):
                assert np.array_equal(_c_blob(getattr(fw, k + "_h2"), refh[k].numel()), refh[k].numpy().view(np.uint32)), (lat, pe, k)
        finally:
            lib.gmf_packed_fusion_free(pf)
    bad = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, 512, 64, pe=False), seed=1)
    arr, keep = _lib.tensor_list(bad)
    pf = ctypes.c_void_p()
    assert lib.gmf_fusion_pack_weights(None, arr, len(arr), b"", 0, 0, ctypes.byref(pf)) == -2 and not pf.value


def test_synthetic_scene_is_consistent():
    from gmf_amd import synthetic
    p = synthetic.synthetic_pair(5, 500)
    inl = p["gt_labels"] > 0
    assert abs(inl.mean() - 0.25) < 0.01
    R, t = p["gt_trans"][:3, :3], p["gt_trans"][:3, 3]
    err = np.linalg.norm(p["src_keypts"][inl] @ R.T + t - p["tgt_keypts"][inl], axis=1)
    assert err.max() < 0.08 and abs(np.linalg.det(R) - 1) < 1e-5
    assert np.abs(p["corr_pos"].mean(0)).max() < 1e-5


def test_se3_helpers():
    from gmf_amd import SE3
    T1 = torch.eye(4).repeat(2, 1, 1)
    T1[:, :3, 3] = torch.tensor([1.0, 2.0, 3.0])
    pts = torch.randn(2, 5, 3)
    assert torch.allclose(SE3.transform(pts, T1), pts + torch.tensor([1.0, 2.0, 3.0]))
    R, t = SE3.decompose_trans(T1)
    assert torch.equal(SE3.integrate_trans(R, t), T1)
    assert torch.allclose(SE3.concatenate(T1, T1)[:, :3, 3], torch.tensor([2.0, 4.0, 6.0]).repeat(2, 1))


def test_image_encoder_matches_reference(golden_dir):
    """the ResNet-34 -> layer2 image encoder (plain PyTorch module, MIOpen on the GPU) produces
    the reference's tokens from the same seeded weights (reference keys; its unused layer3/layer4/fc are ignored)."""
    import gmf_amd
    from gmf_amd import synthetic
    g = np.load(os.path.join(golden_dir, "f11_image_encoder.npz"))
    enc = gmf_amd.ImageEncoder().eval()
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(synthetic.seeded_state_dict(shapes, seed=int(g["seed"]), gain=1.0))
    r = np.random.default_rng([111])
    img = torch.from_numpy(r.uniform(0, 1, (2, 3, 120, 160)).astype(np.float32))
    with torch.no_grad():
        f = enc(img)
    assert f.shape == (2, 128, 15, 20)
    tok = f.view(2, 128, -1).permute(0, 2, 1)
    assert np.abs(tok.numpy() - g["tokens"]).max() < 1e-4 * max(1.0, np.abs(g["tokens"]).max())


@pytest.mark.parametrize("tag", ["64x120x160", "32x96x128"])
def test_image_encoder_matches_reference_at_batch_size(golden_dir, tag):
    """Golden F15: the reference's ImageEncoder on 64 images of 120 x 160 (32 pairs) and 32 of 96 x 128 - sampled token
    rows and per-image fp64 checksums.  Here the stock module on the CPU; tests/test_gpu_parity.py holds the fused HIP
    encoder to the same fixture."""
    import gmf_amd
    from gmf_amd import synthetic
    g = np.load(os.path.join(golden_dir, "f15_image_encoder_batch.npz"))
    enc = gmf_amd.ImageEncoder().eval()
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(synthetic.seeded_state_dict(shapes, seed=int(g["seed"]), gain=1.0))
    nimg, H, W = (int(v) for v in g[f"shape_{tag}"])
    img = synthetic.seeded_images(nimg, H, W)
    with torch.no_grad():
        f = torch.cat([enc(img[i:i + 8]) for i in range(0, nimg, 8)])
    tok = f.view(nimg, 128, -1).permute(0, 2, 1)
    scale = max(1.0, float(np.abs(g[f"rows_{tag}"]).max()))
    assert np.abs(tok[::8, ::7].numpy() - g[f"rows_{tag}"]).max() < 1e-4 * scale
    assert np.abs(tok.double().sum((1, 2)).numpy() / g[f"sum_{tag}"] - 1).max() < 1e-5
    assert np.abs((tok.double() ** 2).sum((1, 2)).numpy() / g[f"sumsq_{tag}"] - 1).max() < 1e-5


def test_oracle_is_only_reachable_from_the_checkers():
    """The oracle is test infrastructure: nothing in the product package, the dev tools or the
    C/HIP sources may import or reference it; only tests/, bench.py's cpu_baseline leg and
    __graft_entry__.smoke()/build() do."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|gmf_oracle", re.M)
    offenders = []
    for sub in ("gmf_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".sh")):
                    p = os.path.join(dirpath, f)
                    if pat.search(open(p, errors="ignore").read()):
                        offenders.append(os.path.relpath(p, root))
    assert offenders == []


def test_unsupported_configurations_are_refused_at_construction():
    """Configurations the kernels are not built for raise NotImplementedError when the module is constructed (on any
    host), not an assertion somewhere inside a forward."""
    import pytest
    import gmf_amd
    for kw in (dict(num_channels=64), dict(in_dim=12), dict(k=100), dict(num_iterations=0)):
        with pytest.raises(NotImplementedError):
            gmf_amd.PointDSC(**kw)
    gmf_amd.PointDSC(in_dim=6, num_layers=3, k=40)        # what GMF instantiates (with 12 layers) constructs anywhere


def test_bench_launches_its_own_ranks_and_refuses_without_devices():
    """`python bench.py --gpus N` without a launcher starts its ranks itself (bench.launch_ranks: children of a process that made
    no GPU call).  On a host with fewer devices than ranks it refuses up front with exit code 2; with --rehearsal (shared
    devices, gloo) the children are started and - on a host without any HIP device - each of them fails loudly, which the
    parent reports as a non-zero exit code instead of hanging."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU host: the refusal path does not apply")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-sweep", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "one rank per GPU needs 2" in r.stderr and not r.stdout.strip()
    if torch.cuda.device_count() == 0:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--no-sweep", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and r.stderr.count("needs a HIP device") == 2 and not r.stdout.strip()
