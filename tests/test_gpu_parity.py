"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden vectors and the oracle.

Tolerances: fp32 path, 1e-4 absolute on logits and poses (BASELINE.json north_star); the stress case
states its tolerance as a multiple of the measured fp32 noise floor of the computation itself.
"""
import os

import numpy as np
import pytest
import torch

import gmf_amd
from gmf_amd import synthetic
from oracle import gmf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def _gpu(t):
    return t.to(DEV)


@pytest.fixture(scope="module")
def sd_full():
    return synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7)


@pytest.fixture(scope="module")
def model(sd_full):
    m = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                         inlier_threshold=0.10, sigma_d=0.10, k=40, nms_radius=0.10)
    missing, unexpected = m.load_state_dict(sd_full, strict=False)
    assert not unexpected
    return m.to(DEV).eval()


def test_native_library_loaded():
    from gmf_amd import _lib
    lib = _lib.load_library()
    assert lib.gmf_abi_version() == 5
    assert _lib.handle_for(0).h


def test_pack_unpack_roundtrip():
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    st = torch.cuda.current_stream().cuda_stream
    for (B, N, K, chan_major) in [(2, 70, 128, False), (1, 33, 64, False), (3, 257, 128, True)]:
        x = torch.randn(B, N, K, device=DEV)
        src = x.permute(0, 2, 1).contiguous().permute(0, 2, 1) if chan_major else x
        tiles = (N + 31) // 32
        img = torch.empty(B * tiles * 32 * K, device=DEV)
        h.call("gmf_pack_rows_p32", src.data_ptr(), src.stride(0), src.stride(1), src.stride(2), B, N, K, img.data_ptr(), st)
        # image definition: float4 ((tile*(K/8)+g)*64 + lane) = X[32*tile + (lane&31)][8g + 4(lane>>5) ..]
        ref = torch.zeros(B, tiles * 32, K, device=DEV)
        ref[:, :N] = x
        ref = ref.reshape(B, tiles, 32, K // 8, 2, 4).permute(0, 1, 3, 4, 2, 5).reshape(-1)
        assert torch.equal(img, ref)
        y = torch.empty_like(x)
        h.call("gmf_unpack_rows_p32", img.data_ptr(), B, N, K, y.data_ptr(), y.stride(0), y.stride(1), y.stride(2), st)
        assert torch.equal(x, y)


def test_f1_fusion1(golden_dir):
    g = _load(golden_dir, "f1_fusion1.npz")
    sd = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, 128, 64, pe=False), seed=int(g["seed"]))
    m = gmf_amd.FusionLayer(depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8, cross_dim_head=64, latent_dim_head=64)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    for T in (12, 196, 300):
        b = synthetic.synthetic_batch(list(g["pair_seeds"]), N=8, T=T)
        for h2_attn in (True, False):          # context preparation + cross-attention on split-fp16 operands / on the fp32 MFMA
            m.split_fp16_attn = h2_attn
            y = m(_gpu(b["p_tokens"]), queries_encoder=_gpu(b["q_tokens"]))
            assert _maxerr(y.cpu(), g[f"out_T{T}"]) < 1e-4, (T, h2_attn)


@pytest.mark.parametrize("N,T", [(64, 12), (257, 196), (1000, 196), (33, 1), (1, 7)])
def test_f2_fusion2(golden_dir, N, T):
    g = _load(golden_dir, "f2_fusion2.npz")
    sd = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, 128, 64, pe=True), seed=int(g["seed"]))
    m = gmf_amd.FusionLayer(depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8, cross_dim_head=64,
                            latent_dim_head=64, pe=True)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    r = np.random.default_rng([102, N, T])
    x = torch.from_numpy(r.normal(0, 1, (1, N, 128)).astype(np.float32))
    ctx = torch.from_numpy(r.normal(0, 1, (1, T, 128)).astype(np.float32))
    y = m(_gpu(ctx), queries_encoder=_gpu(x))
    assert _maxerr(y.cpu(), g[f"out_N{N}_T{T}"]) < 1e-4
    # the reference feeds a transposed [B,C,N] view (PointDSC.py:70): strided queries must give the same result
    xt = _gpu(x).permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert torch.equal(m(_gpu(ctx), queries_encoder=xt), y)
    m.split_fp16_attn = False                  # the fp32-MFMA form of context preparation + cross-attention
    assert _maxerr(m(_gpu(ctx), queries_encoder=_gpu(x)).cpu(), g[f"out_N{N}_T{T}"]) < 1e-4


@pytest.mark.parametrize("name", ["c64_h2", "c128_h4", "c32_h1"])
def test_f24_general_nonlocal_block(golden_dir, name):
    """Golden F24 [r5]: `NonLocalBlock(num_channels, num_heads)` for values GMF never passes (PointDSC.py:11; VERDICT r4 missing 3) -
    the forward-only composition of HIP primitives (`NonLocalBlock._forward_general`) against the reference's own output."""
    C, H, B, N, T = synthetic.F24_CASES[name]
    blk = gmf_amd.NonLocalBlock(num_channels=C, num_heads=H)
    blk.load_state_dict(synthetic.seeded_state_dict({k: tuple(v.shape) for k, v in blk.state_dict().items()}, seed=124))
    blk = blk.to(DEV).eval()
    feat, src, tgt, img = synthetic.f24_inputs(name)
    compat, _ = O.compat_matrix(src, tgt, 0.1)
    out = blk(_gpu(feat), _gpu(compat), _gpu(img))
    ref = _load(golden_dir, "f24_nonlocal_block_general.npz")[f"out_{name}"]
    print(f"F24 {name}: max |out - reference| {_maxerr(out.cpu(), ref):.2e}")
    assert out.shape == (B, C, N) and _maxerr(out.cpu(), ref) < 1e-4


@pytest.mark.parametrize("name", ["fl_d2_h2", "fl_tied", "fl_w96", "pio_d1"])
def test_f23_general_fusion_layer(golden_dir, name):
    """Golden F23 [r5]: every constructor argument of the reference's FusionLayer / PerceiverIO is honoured - latent self-attention
    layers (depth > 0, weight_tie_layers), several cross / latent heads, widths other than the two the fused kernels cover
    (fusion_layer.py:131-201, perceiver_io.py:139-221; VERDICT r4 missing 2).  These run `FusionLayer._forward_general`: the library's
    HIP primitives (fp32 GEMM, LayerNorm, softmax, GEGLU, LCPE) through the C ABI, forward only.  Against the reference's own
    output from the same seeded weights."""
    cls, depth, dim, lat, ch, lh, cdh, ldh, tie, pe, B, N, T = synthetic.F23_CASES[name]
    mod = (gmf_amd.FusionLayer if cls == "fl" else gmf_amd.PerceiverIO)(depth=depth, dim=dim, latent_dim=lat, cross_heads=ch, latent_heads=lh,
                                                                         cross_dim_head=cdh, latent_dim_head=ldh, weight_tie_layers=tie, pe=pe)
    mod.load_state_dict(synthetic.f23_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()}, tie))
    mod = mod.to(DEV).eval()
    x, ctx = synthetic.f23_inputs(name)
    g = _load(golden_dir, "f23_fusion_layer_general.npz")
    out = mod(_gpu(ctx), queries_encoder=_gpu(x))
    ref = g[f"out_{name}"]
    print(f"F23 {name}: max |out - reference| {_maxerr(out.cpu(), ref):.2e} on outputs up to {np.abs(ref).max():.1f}")
    assert _maxerr(out.cpu(), ref) < 1e-4
    # a strided view of the queries (the reference is fed `feat.permute(0, 2, 1)`, PointDSC.py:70)
    xt = _gpu(x).permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert torch.equal(mod(_gpu(ctx), queries_encoder=xt), out)


@pytest.mark.parametrize("h2_attn", [True, False])
@pytest.mark.parametrize("M,T", [(100, 12), (515, 300)])
def test_f9_dgr_perceiver_256(golden_dir, M, T, h2_attn):
    """DGR bottleneck instance: latent 256, context 128, one head of 128 (resunet_new.py:516-525); context preparation and
    cross-attention on split-fp16 operands (default) and on the fp32 MFMA."""
    g = _load(golden_dir, "f9_dgr_perceiver.npz")
    sd = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, 256, 128, pe=True, out_to_query=True), seed=int(g["seed"]))
    m = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8, cross_dim_head=128,
                            latent_dim_head=128, pe=True)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.split_fp16_attn = h2_attn
    r = np.random.default_rng([109, M, T])
    x = torch.from_numpy(r.normal(0, 1, (1, M, 256)).astype(np.float32))
    ctx = torch.from_numpy(r.normal(0, 1, (1, T, 128)).astype(np.float32))
    # the DGR call site feeds F [M,256] unsqueezed to [1,M,256] (resunet_new.py:696-699)
    y = m(_gpu(ctx), queries_encoder=_gpu(x))
    assert y.shape == (1, M, 256)
    err = _maxerr(y.cpu(), g[f"out_M{M}_T{T}"])
    print(f"F9 M={M} T={T} split_fp16_attn={h2_attn}: max err {err:.3e}")
    assert err < 1e-4
    # small grids split the feed-forward's hidden chunks over up to 8 workgroups per row block: same sums, another order
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    try:
        for hs in (1, 2, 8):
            h.call("gmf_set_tuning", b"ff_hidden_splits", hs)
            assert _maxerr(m(_gpu(ctx), queries_encoder=_gpu(x)).cpu(), g[f"out_M{M}_T{T}"]) < 1e-4, hs
        h.call("gmf_set_tuning", b"ff_hidden_splits", 0)
        h.call("gmf_set_tuning", b"wide_attn_tile", 0)     # the cross-attention with one workgroup per four tiles
        assert _maxerr(m(_gpu(ctx), queries_encoder=_gpu(x)).cpu(), g[f"out_M{M}_T{T}"]) < 1e-4
    finally:
        h.call("gmf_set_tuning", b"ff_hidden_splits", 0)
        h.call("gmf_set_tuning", b"wide_attn_tile", 1)


@pytest.mark.parametrize("h2_attn", [True, False])
@pytest.mark.parametrize("M,T", [(100, 12), (515, 300)])
def test_f9b_dgr_perceiver_256_fpfh_twin(golden_dir, M, T, h2_attn):
    """The fpfh twin of the DGR bottleneck layer: GMF_DeepGlobalRegistration_fpfh/model/perceiver_io.py:112-200 has no `cpe`
    (no LCPE) and is built at the same widths (latent 256, head 128; ..._fpfh/model/resunet_new.py:516-525).  Golden F9b is
    that module's own output; split-fp16 and fp32-MFMA forms."""
    g = _load(golden_dir, "f9b_dgr_perceiver_fpfh.npz")
    sd = synthetic.seeded_state_dict(synthetic.fusion_layer_shapes("", 128, 256, 128, pe=False, out_to_query=True), seed=int(g["seed"]))
    m = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8, cross_dim_head=128,
                            latent_dim_head=128)          # the fpfh constructor has no `pe` argument: the default is False
    assert not any(k.startswith("cpe.") for k in m.state_dict())
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.split_fp16_attn = h2_attn
    r = np.random.default_rng([119, M, T])
    x = torch.from_numpy(r.normal(0, 1, (1, M, 256)).astype(np.float32))
    ctx = torch.from_numpy(r.normal(0, 1, (1, T, 128)).astype(np.float32))
    y = m(_gpu(ctx), queries_encoder=_gpu(x))
    err = _maxerr(y.cpu(), g[f"out_M{M}_T{T}"])
    print(f"F9b M={M} T={T} split_fp16_attn={h2_attn}: max err {err:.3e}")
    assert err < 1e-4
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    try:
        for hs in (1, 8):
            h.call("gmf_set_tuning", b"ff_hidden_splits", hs)
            assert _maxerr(m(_gpu(ctx), queries_encoder=_gpu(x)).cpu(), g[f"out_M{M}_T{T}"]) < 1e-4, hs
        h.call("gmf_set_tuning", b"ff_hidden_splits", 0)
        h.call("gmf_set_tuning", b"wide_attn_tile", 0)
        assert _maxerr(m(_gpu(ctx), queries_encoder=_gpu(x)).cpu(), g[f"out_M{M}_T{T}"]) < 1e-4
    finally:
        h.call("gmf_set_tuning", b"ff_hidden_splits", 0)
        h.call("gmf_set_tuning", b"wide_attn_tile", 1)


def test_non_finite_results_set_the_status_word(model):
    """Activations beyond the range of the split-fp16 MFMA operands (|x| < 65504) turn into inf / NaN.  The reference has no such
    limit; here the overflow must not pass silently: the classifier kernel (and the last kernel of a FusionLayer) set a sticky
    status bit in host-mapped memory, `gmf_amd.check_status()` raises after a synchronisation, and so does the NEXT forward."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch([5, 6], N=300, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    model(data)
    gmf_amd.check_status()                                   # a normal forward leaves the word clear
    assert h.status() == 0
    bad = dict(data)
    bad["corr_pos"] = data["corr_pos"] * 1e30                # layer0's output leaves the fp16 range by far
    model(bad)
    with pytest.raises(RuntimeError, match="non-finite"):
        gmf_amd.check_status()
    assert h.status() == 0                                   # reading it through the wrapper clears it
    model(bad)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="non-finite"):    # ... and the next forward on the device reports it too
        model(data)
    res = model(data)                                        # the flag was cleared by the raise: the handle is usable again
    gmf_amd.check_status()
    assert torch.isfinite(res["final_trans"]).all()
    # FusionLayer / PerceiverIO surface: NaN in the queries reaches the output and the word
    m = gmf_amd.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8, cross_dim_head=128,
                            latent_dim_head=128, pe=True).to(DEV).eval()
    x = torch.randn(1, 200, 256, device=DEV)
    ctx = torch.randn(1, 50, 128, device=DEV)
    m(ctx, queries_encoder=x)
    gmf_amd.check_status()
    x[0, 7, 3] = float("nan")
    m(ctx, queries_encoder=x)
    with pytest.raises(RuntimeError, match="non-finite"):
        gmf_amd.check_status()


def test_workspace_is_a_torch_tensor_and_the_device_is_restored(model):
    """gmf_set_workspace: the Python host hands the library a torch tensor (the memory belongs to torch's allocator), grows it
    when a call reports GMF_ERR_WORKSPACE, and results do not depend on who owns the block.  Every call leaves the caller's
    current device as it found it."""
    import ctypes as C
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch([11], N=500, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    r1 = model(data)
    lg1 = model.last_logits.clone()
    assert h._ws is not None and h.lib.gmf_workspace_bytes(h.h) == h._ws.numel()
    # a too-small caller block: the raw call reports GMF_ERR_WORKSPACE and what it wants, and launches nothing
    small = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    h.check(h.lib.gmf_set_workspace(h.h, small.data_ptr(), small.numel()), "gmf_set_workspace")
    h._ws = small
    assert h.lib.gmf_workspace_wanted(h.h) >= 0
    r2 = model(data)                                         # Handle.call grows the tensor and repeats the call
    assert h._ws.numel() > (1 << 20) and h.lib.gmf_workspace_bytes(h.h) == h._ws.numel()
    assert torch.equal(model.last_logits, lg1) and torch.equal(r2["final_trans"], r1["final_trans"])
    # library-owned block (what a host without an allocator of its own gets): same bits
    h.check(h.lib.gmf_set_workspace(h.h, None, 0), "gmf_set_workspace")
    h._ws, h.torch_workspace = None, False
    try:
        r3 = model(data)
        assert h.lib.gmf_workspace_bytes(h.h) > 0
        assert torch.equal(model.last_logits, lg1) and torch.equal(r3["final_trans"], r1["final_trans"])
    finally:
        h.torch_workspace = True
    model(data)
    assert h._ws is not None
    # misuse is rejected
    assert h.lib.gmf_set_workspace(h.h, C.c_void_p(h._ws.data_ptr() + 4), 1024) == -1      # not 256-byte aligned
    assert h.lib.gmf_set_workspace(h.h, None, 1024) == -1
    # the caller's current device survives a call (one visible device here: the call must at least not change it)
    before = torch.cuda.current_device()
    model(data)
    assert torch.cuda.current_device() == before


def test_f3_nonlocal_block(golden_dir, sd_full):
    g = _load(golden_dir, "f3_nonlocal_block.npz")
    layer = int(g["layer"])
    pre = f"encoder.blocks.NonLocal_layer_{layer}."
    blk = gmf_amd.NonLocalBlock(128)
    blk.load_state_dict({k[len(pre):]: v for k, v in sd_full.items() if k.startswith(pre)})
    blk = blk.to(DEV).eval()
    b = synthetic.synthetic_batch(list(g["pair_seeds"]), N=257, T=196)
    compat, _ = O.compat_matrix(b["src_keypts"], b["tgt_keypts"], 0.1)
    feat = torch.from_numpy(g["feat"])
    y = blk(_gpu(feat).permute(0, 2, 1).contiguous(), _gpu(compat), _gpu(torch.from_numpy(g["img"])))
    assert y.shape == (2, 128, 257)
    assert _maxerr(y.permute(0, 2, 1).cpu(), g["out"]) < 1e-4


@pytest.mark.parametrize("N", [64, 257, 1000])
def test_f4_f10_pointdsc(golden_dir, model, N):
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    seeds = list(g[f"pair_seeds_N{N}"])
    b = synthetic.synthetic_batch(seeds, N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = model(data)
    logits = model.last_logits.cpu().numpy()
    assert _maxerr(logits, g[f"logits_N{N}"]) < 1e-4
    assert _maxerr(res["final_trans"].cpu(), g[f"final_trans_N{N}"]) < 1e-4
    assert (res["final_labels"].cpu().numpy() == g[f"final_labels_N{N}"]).mean() > 0.999
    _, _, feat = model.encode(data["corr_pos"], data["src_keypts"], data["tgt_keypts"], data["p_tokens"], data["q_tokens"], True)
    if N <= 257:
        assert _maxerr(feat.cpu(), g[f"feat_N{N}"]) < 1e-4
    else:
        assert _maxerr(feat[:, ::50].cpu(), g[f"feat_rows_N{N}"]) < 1e-4
    if len(seeds) == 2:      # train mode: logits are the labels, B=2 in one call
        del data["testing"]
        tr = model(data)
        assert _maxerr(tr["final_labels"].cpu(), g[f"train_logits_N{N}"]) < 1e-4
        assert _maxerr(tr["final_trans"].cpu(), g[f"train_final_trans_N{N}"]) < 1e-4
        assert tr["M"].shape == (2, N, N)


def test_batched_equals_per_pair(model):
    """B>1 in one launch == B independent B=1 calls, bit for bit (pairs never interact)."""
    b = synthetic.synthetic_batch([61, 62, 63], N=333, T=50)
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    full = model.encode(*[_gpu(b[k]) for k in keys])[0]
    for i in range(3):
        one = model.encode(*[_gpu(b[k][i:i + 1]) for k in keys])[0]
        assert torch.equal(full[i:i + 1], one)


def test_ragged_batch_equals_per_pair_calls(model):
    """gmf_encoder_forward_ragged + gmf_pose_head_ragged: five pairs with their OWN N in one launch (what the reference's
    evaluation loop feeds one pair at a time, evaluation/test_3DMatch.py:69) give each pair the result of its own B = 1 call -
    logits to 5e-5 (the ragged launch runs the large-grid kernels, a B = 1 call the small-grid ones: other summation orders),
    poses to 1e-4, identical inlier labels up to the points a 5e-5 logit change can flip."""
    sizes = [700, 1531, 5000, 257, 45]             # (45: barely more than k = 40 neighbours, two tiles, four seeds)
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    pairs = [synthetic.synthetic_batch([300 + i], N=n, T=196) for i, n in enumerate(sizes)]
    rag = {k: [_gpu(b[k][0]) for b in pairs] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag["p_tokens"] = torch.cat([_gpu(b["p_tokens"]) for b in pairs])
    rag["q_tokens"] = torch.cat([_gpu(b["q_tokens"]) for b in pairs])
    rag["testing"] = True
    out = model(rag)                                           # lists -> forward_ragged
    assert out["final_trans"].shape == (len(sizes), 4, 4) and [t.shape[0] for t in out["final_labels"]] == sizes
    gmf_amd.check_status()
    for i, b in enumerate(pairs):
        one = {k: _gpu(b[k]) for k in keys}
        one["testing"] = True
        r1 = model(one)
        dl = _maxerr(out["logits"][i].cpu(), model.last_logits[0].cpu())
        dT = _maxerr(out["final_trans"][i].cpu(), r1["final_trans"][0].cpu())
        same = float((out["final_labels"][i] == r1["final_labels"][0]).float().mean())
        print(f"ragged pair {i} (N = {sizes[i]}): max |dlogit| {dl:.2e}, max |dT| {dT:.2e}, labels equal {same:.4f}")
        assert dl < 5e-5, (i, dl)
        assert dT < 1e-4, (i, dT)
        assert same > 0.999
    # the packed form with "n_points" is the same call
    packed = {k: torch.cat(rag[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    packed.update(p_tokens=rag["p_tokens"], q_tokens=rag["q_tokens"], n_points=sizes, testing=True)
    out2 = model(packed)
    assert torch.equal(out2["final_trans"], out["final_trans"]) and all(torch.equal(a, b) for a, b in zip(out2["logits"], out["logits"]))


def test_ragged_batch_of_equal_sizes_is_the_uniform_batch(model):
    """With every n_b equal the ragged entry points run the very kernels of the uniform large-grid batch: logits bit for bit.
    (The pose differs in ONE documented respect: a ragged batch resolves the power iteration's allclose exit per pair - B
    independent B = 1 calls - where the uniform batch resolves it over the batch as the reference's tensor-wide test does.)"""
    B, N = 33, 1000                                            # 33 x 8 = 264 row blocks: the large-grid (two-launch) form
    b = synthetic.synthetic_batch(list(range(500, 500 + B)), N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    ru = model(data)
    lg_u = model.last_logits.clone()
    rag = {k: [data[k][i] for i in range(B)] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=data["p_tokens"], q_tokens=data["q_tokens"], testing=True)
    rr = model(rag)
    assert torch.equal(torch.stack(rr["logits"]), lg_u)
    assert _maxerr(rr["final_trans"].cpu(), ru["final_trans"].cpu()) < 1e-4
    # misuse: a pair too small for its k neighbours is refused, not mis-computed
    bad = dict(rag)
    for k in ("corr_pos", "src_keypts", "tgt_keypts"):
        bad[k] = [t[:30] if i == 2 else t for i, t in enumerate(rag[k])]
    with pytest.raises(RuntimeError, match="more than k"):
        model(bad)


@pytest.mark.parametrize("kind", ["3dmatch", "kitti"])
def test_pv_fp8_form_against_the_oracle(kind):
    """The large-grid attention multiplies the two cross products of O += P V on the block-scaled fp8 matrix pipe ("pv_fp8" = 1, the
    default, there under a device-side guard, "pv_fp8" = 2 unconditionally).  Scenes travel as a ragged batch (ragged batches always take the large-grid path) and every scene
    is held to the fp32 oracle (PointDSC.py:56-64) AND to an fp64 evaluation, in both forms; the two forms agree to 5e-5.  The
    KITTI shape (coordinates of +-40 m, attention logits two orders larger) is the case a 16-bit compat cache failed."""
    from gmf_amd import _lib
    sigma_d = 0.1 if kind == "3dmatch" else 1.2
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=sigma_d)
    m = gmf_amd.PointDSC(num_layers=12) if kind == "3dmatch" else gmf_amd.PointDSC(
        in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=1.2, sigma_d=1.2, k=40, nms_radius=1.2)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).eval()
    sizes = [777, 1500, 2048] if kind == "3dmatch" else [700, 1500, 2000]
    seeds = [1003, 1011, 1017] if kind == "3dmatch" else [83, 84, 84]       # (2000, 84): the F22 stress pair the unguarded form misses
    pairs = [synthetic.synthetic_batch([sc], N=n, T=196, kind=kind) for sc, n in zip(seeds, sizes)]
    kw = {} if kind == "3dmatch" else {"inlier_threshold": 1.2, "nms_radius": 1.2}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    refs, truths = [], []
    for b in pairs:
        refs.append(O.pointdsc_forward(sd, b, testing=True, **kw)["logits"][0])
        b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        c64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], sigma_d)
        truths.append(O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], c64, b64["p_tokens"], b64["q_tokens"], 12))[0])
    rag = {k: [_gpu(b[k][0]) for b in pairs] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=torch.cat([_gpu(b["p_tokens"]) for b in pairs]), q_tokens=torch.cat([_gpu(b["q_tokens"]) for b in pairs]), testing=True)
    h = _lib.handle_for(0)
    got = {}
    try:
        for form in (0, 1, 2):
            h.call("gmf_set_tuning", b"pv_fp8", form)
            got[form] = [lg.cpu() for lg in m(rag)["logits"]]
    finally:
        h.call("gmf_set_tuning", b"pv_fp8", 1)
    for i in range(len(pairs)):
        floor = float((refs[i].double() - truths[i]).abs().max())        # the reference's own fp32 evaluation against fp64
        for form in (0, 1, 2):
            e32, e64 = _maxerr(got[form][i], refs[i]), float((got[form][i].double() - truths[i]).abs().max())
            print(f"{kind} N={sizes[i]} pv_fp8={form}: vs fp32 oracle {e32:.2e}, vs fp64 {e64:.2e} (fp32 oracle vs fp64 {floor:.2e})")
            # 0 = three f16 products, 1 = the default (fp8 cross products under the device-side guard): ONE floor-relative contract;
            # 2 = the fp8 forms unconditionally: the seeded KITTI-shape network is where their e4m3 rounding shows
            # (unconditional, KITTI shape: with the cross products of Q' K^T on e4m3 as well - round 5 - the seeded network's first
            # layers, whose scores reach 10^3 .. 10^4, put it at 1.7e-3: the guard exists for exactly this; "pv_fp8" = 2 is an
            # A/B setting for well-conditioned networks)
            if form < 2 or kind == "3dmatch":
                assert e64 < 1.5 * floor + 2e-5, (kind, i, form, e64, floor)
            else:
                assert e64 < 1e-2, (kind, i, form, e64, floor)
            if kind == "3dmatch":
                assert e32 < 1e-4, (i, form, e32)
        assert _maxerr(got[0][i], got[2][i]) < (5e-5 if kind == "3dmatch" else 1e-2), (kind, i)     # (unguarded on the stress set: 1.4e-3)
        assert _maxerr(got[0][i], got[1][i]) < (5e-5 if kind == "3dmatch" else 2e-4), (kind, i)
        if kind == "3dmatch":
            assert torch.equal(got[1][i], got[2][i]), i          # a well-conditioned network never trips the guard
    with pytest.raises(RuntimeError):
        h.call("gmf_set_tuning", b"pv_fp8", 3)


@pytest.mark.parametrize("grid", ["small", "large"])
def test_pv_fp8_guard_decides_per_pair_and_layer(model, sd_full, grid):
    """The default "pv_fp8" = 1 decides ON THE DEVICE, per pair and per layer, whether the P V cross products run on the fp8 pipe
    (PointDSC.py:56-64): the statistic is the largest row norm of the layer's input features, the threshold
    gmf_encoder_weights::pv_guard.  Mechanics, without any oracle: (a) an ordinary pair never trips it - bit-identical to the
    unconditional fp8 form ("pv_fp8" = 2), status bit clear; (b) a pair whose inputs are 40 x larger trips its first layers -
    different bits from the unconditional form, status bit set - and does so WITHOUT touching the ordinary pairs of the same batch
    (no dependence on batch composition); (c) weights whose projection biases alone pass the score bound (threshold -1) send every
    layer to the three-product form: bit-identical to "pv_fp8" = 0."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    B, N = (2, 1000) if grid == "small" else (9, 3970)          # 9 x 32 row blocks >= 256: the two-launch form of large grids
    b = synthetic.synthetic_batch(list(range(520, 520 + B)), N=N, T=196)
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    data = {k: _gpu(b[k]) for k in keys}
    hot = {k: v.clone() for k, v in data.items()}
    hot["corr_pos"][1] *= 40.0                                   # (the compat matrix comes from the key points: unchanged)

    def run(m, d, pv):
        h.call("gmf_set_tuning", b"pv_fp8", pv)
        h.status(clear=True)
        lg = m.encode(*[d[k] for k in keys])[0].clone()
        torch.cuda.synchronize()
        return lg, bool(h.status() & _lib.GMF_STATUS_PV_GUARDED)
    try:
        lg1, t1 = run(model, data, 1)
        lg2, t2 = run(model, data, 2)
        assert torch.equal(lg1, lg2) and not t1 and not t2                       # (a)
        hg1, ht1 = run(model, hot, 1)
        hg2, ht2 = run(model, hot, 2)
        hg0, _ = run(model, hot, 0)
        assert ht1 and not ht2                                                   # (b)
        assert torch.isfinite(hg1).all()
        assert not torch.equal(hg1[1], hg2[1])
        others = [i for i in range(B) if i != 1]
        assert torch.equal(hg1[others], hg2[others]) and torch.equal(hg1[others], lg1[others])
        print(f"guard, {grid} grid: hot pair, guarded vs unconditional fp8 {_maxerr(hg1[1].cpu(), hg2[1].cpu()):.2e}, "
              f"vs three products {_maxerr(hg1[1].cpu(), hg0[1].cpu()):.2e}")
        sd = dict(sd_full)                                                       # (c)
        for layer in range(12):
            for proj in ("projection_q", "projection_k"):
                key = f"encoder.blocks.NonLocal_layer_{layer}.{proj}.bias"
                sd[key] = torch.full_like(sd[key], 20.0)
        m = gmf_amd.PointDSC(num_layers=12)
        m.load_state_dict(sd, strict=False)
        m = m.to(DEV).eval()
        g1, gt1 = run(m, data, 1)
        g0, _ = run(m, data, 0)
        g2, _ = run(m, data, 2)
        assert gt1 and torch.isfinite(g1).all()
        assert torch.equal(g1, g0) and not torch.equal(g1, g2)
        # (d) a weights block WITHOUT thresholds (filled in by hand, pv_guard = NULL): the guarded default runs three products -
        # the unguarded fp8 form is only ever reached by asking for it ("pv_fp8" = 2)
        pw = model._weights(data["corr_pos"].device)           # (the packed block the forward itself uses: same cache key)
        saved = pw.struct.contents.pv_guard
        lg0, _ = run(model, data, 0)
        try:
            pw.struct.contents.pv_guard = None
            n1, nt1 = run(model, data, 1)
            n2, _ = run(model, data, 2)
        finally:
            pw.struct.contents.pv_guard = saved
        assert torch.equal(n1, lg0) and not nt1 and torch.equal(n2, lg2)
    finally:
        h.call("gmf_set_tuning", b"pv_fp8", 1)
        h.status(clear=True)


def test_pv_fp8_guard_in_a_small_ragged_batch(model):
    """[r5] The guard where the newest paths meet: a SMALL ragged batch (key-split attention items planned on the smallest pair, the
    two-role linear kernel and k_scattn_merge reading the pair table, the slots in work-balanced order - PairTab::ord) with one pair
    whose inputs are 40 x larger (its first layers take the three-product form).  Every pair, hot or not, must equal its own B = 1
    call to fp32 rounding - the decision belongs to the pair, not to its slot or its neighbours (PointDSC.py:56-64)."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    keys = ("corr_pos", "src_keypts", "tgt_keypts")
    sizes = [1000, 2311, 640, 1500]
    pairs = [synthetic.synthetic_batch([880 + i], N=n, T=196) for i, n in enumerate(sizes)]
    one = [{k: _gpu(p[k]) for k in keys + ("p_tokens", "q_tokens")} for p in pairs]
    one[2]["corr_pos"] = one[2]["corr_pos"] * 40.0
    rag = {k: [o[k][0] for o in one] for k in keys}
    rag.update(p_tokens=torch.cat([o["p_tokens"] for o in one]), q_tokens=torch.cat([o["q_tokens"] for o in one]), testing=True)
    h.status(clear=True)
    lg = [x.clone() for x in model(rag)["logits"]]
    torch.cuda.synchronize()                     # (the status word is written by the forward's last kernels)
    tripped = bool(h.status() & _lib.GMF_STATUS_PV_GUARDED)
    h.status(clear=True)
    assert tripped and all(torch.isfinite(x).all() for x in lg)
    for i, o in enumerate(one):
        o["testing"] = True
        model(o)
        ref = model.last_logits[0]
        scale = max(1.0, float(ref.abs().max()))
        d = _maxerr(lg[i].cpu(), ref.cpu())
        print(f"pair {i} (n = {sizes[i]}{', hot' if i == 2 else ''}): ragged vs own B = 1 call {d:.2e} (largest logit {scale:.1f})")
        # (the hot pair's scores are 40 x larger: the two calls split its keys differently, and that order shows - 1.3e-4 measured)
        assert d < (3e-4 if i == 2 else 1e-4) * scale, (i, d, scale)
    h.status(clear=True)


def test_outlier_correspondence_with_huge_coordinates(model, sd_full):
    """One correspondence whose coordinates are 300 x larger than the scene: its V row dominates the per-(feature, key tile) scale
    of the e4m3 cross planes of its tile (the other 31 keys of that tile then lose bits of their CROSS terms
    only), its Q' / K rows produce attention logits far outside the others'.  Both attention forms, the small-grid path (B = 1)
    and the large-grid path (the same scene in a ragged batch), stay within the contract against an fp64 evaluation."""
    from gmf_amd import _lib
    b = synthetic.synthetic_batch([321], N=1000, T=196)
    for k in ("src_keypts", "tgt_keypts"):
        b[k] = b[k].clone()
        b[k][0, 17] *= 300.0
    b["corr_pos"] = torch.cat([b["src_keypts"], b["tgt_keypts"]], dim=-1)
    ref = O.pointdsc_forward(sd_full, b, testing=True)["logits"][0]
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_full.items()}
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
    c64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 0.1)
    truth = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], c64, b64["p_tokens"], b64["q_tokens"], 12))[0]
    floor = float((ref.double() - truth).abs().max())
    one = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    one["testing"] = True
    rag = {k: [one[k][0], one[k][0]] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=torch.cat([one["p_tokens"]] * 2), q_tokens=torch.cat([one["q_tokens"]] * 2), testing=True)
    h = _lib.handle_for(0)
    try:
        for pv in (0, 1, 2):      # three products | guarded (the default: this scene trips it) | fp8 unconditionally
            h.call("gmf_set_tuning", b"pv_fp8", pv)
            model(one)
            small = model.last_logits[0].cpu()
            large = model(rag)["logits"][0].cpu()
            for name, lg in (("small grid", small), ("large grid", large)):
                e64 = float((lg.double() - truth).abs().max())
                print(f"outlier scene, pv_fp8 {pv}, {name}: vs fp64 {e64:.2e}, vs fp32 oracle {_maxerr(lg, ref):.2e} (fp32 oracle vs fp64 {floor:.2e})")
                assert torch.isfinite(lg).all()
                # (unconditionally on the fp8 pipe - 2 - the outlier's scores cost 1.2e-3: this scene is what the guard is for)
                assert e64 < (1.5 * floor + 2e-5 if pv < 2 else 1e-2), (pv, name, e64, floor)
    finally:
        h.call("gmf_set_tuning", b"pv_fp8", 1)


def test_pv_fp8_near_dead_v_channel(sd_full):
    """a V channel whose values are all tiny - one row of projection_v and its bias scaled to 1e-6, in every
    layer - puts the tile maximum of that feature below 2^-15, where fp16(v) is subnormal and the residual v - hi is absolute
    (up to 2^-25).  With the e4m3 scale following the tile maximum all the way down, residual / scale left e4m3's range and the
    conversion wrote NaN bytes (store_block_v8, enc_common.hpp); the scale now has a floor of 2^-22.  Both grid paths: finite
    logits, equal to the three-product form (pv_fp8 = 0) to 5e-5 and within 1e-4 of the oracle."""
    from gmf_amd import _lib
    sd = dict(sd_full)
    for layer in range(12):
        pre = f"encoder.blocks.NonLocal_layer_{layer}.projection_v."
        w, bv = sd[pre + "weight"].clone(), sd[pre + "bias"].clone()
        for ch in (3, 77, 127):
            w[ch] *= 1e-6
            bv[ch] *= 1e-6
        sd[pre + "weight"], sd[pre + "bias"] = w, bv
    m = gmf_amd.PointDSC(num_layers=12)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).eval()
    b = synthetic.synthetic_batch([411], N=1000, T=196)
    ref = O.pointdsc_forward(sd, b, testing=True)["logits"][0]
    one = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    one["testing"] = True
    rag = {k: [one[k][0], one[k][0]] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=torch.cat([one["p_tokens"]] * 2), q_tokens=torch.cat([one["q_tokens"]] * 2), testing=True)
    h = _lib.handle_for(0)
    got = {}
    try:
        for pv in (0, 2):
            h.call("gmf_set_tuning", b"pv_fp8", pv)
            m(one)
            got[pv, "small grid"] = m.last_logits[0].cpu()
            got[pv, "large grid"] = m(rag)["logits"][0].cpu()
    finally:
        h.call("gmf_set_tuning", b"pv_fp8", 1)
    gmf_amd.check_status()
    for name in ("small grid", "large grid"):
        assert torch.isfinite(got[2, name]).all(), name
        print(f"near-dead V channels, {name}: pv_fp8 1 vs 0 {_maxerr(got[2, name], got[0, name]):.2e}, vs oracle {_maxerr(got[2, name], ref):.2e}")
        assert _maxerr(got[2, name], got[0, name]) < 5e-5, name
        assert _maxerr(got[2, name], ref) < 1e-4, name


def _pv_fp8_adversarial_case(ratio, N=1000, T=196, seed=5):
    """The worst case of the shared e4m3 scale (one power of two per feature and 32-key tile; INTEGRATION
    "Supported value range"), built through a whole 2-layer model by making layer 0 positively homogeneous in corr_pos:
    layer0 / PointCN biases, the BatchNorm shift and projection_v's bias are zero, so a row whose corr_pos is `ratio` times
    larger has f and V exactly `ratio` times larger IN EVERY FEATURE.  Even rows are the big ones: every 32-key tile holds 16
    keys whose V is `ratio` above the other 16.  The softmax is as peaked as it gets: key points sit on a grid with the
    target a 3x scaled copy (c_ij = 0 exactly for i != j, c_ii = 1) and the q / k biases give s_ii ~ 30, so a query's
    probability mass is on ITSELF (p_ij / p_ii = e^-30) - a small row's output is its own V row, whose cross terms the
    big keys of its tile have pushed below e4m3's resolution: that term falls back to single-plane fp16 accuracy (2^-12
    relative), the bound the analysis predicts.  The LCPE taps of the query side are zero so that Fusion-2 does not mix
    big and small neighbour rows; layer 1 (the default path needs two layers) attends to itself the same way, so a small
    row never sees a big one, and its own V tiles (big rows ~1e4, small rows ~1) repeat the case."""
    import math
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 2, 128), seed=7)
    n = "encoder.blocks.NonLocal_layer_0."
    for k in ("encoder.layer0.bias", "encoder.blocks.PointCN_layer_0.0.bias", "encoder.blocks.PointCN_layer_0.1.bias",
              "encoder.blocks.PointCN_layer_0.1.running_mean", n + "projection_v.bias"):
        sd[k] = torch.zeros_like(sd[k])
    r = np.random.default_rng([seed, 77])
    u = r.normal(size=128)
    u /= np.linalg.norm(u)
    amp = math.sqrt(30.0 * math.sqrt(128.0))
    for layer, gain in ((0, 0.1), (1, 1e-5)):       # (the default path needs two layers; the second keeps the rows apart the same way)
        nl = f"encoder.blocks.NonLocal_layer_{layer}."
        sd[nl + "fusion_layer_2.cpe.proj_q.weight"] = torch.zeros_like(sd[nl + "fusion_layer_2.cpe.proj_q.weight"])
        sd[nl + "projection_q.bias"] = torch.from_numpy((amp * u).astype(np.float32))
        sd[nl + "projection_k.bias"] = torch.from_numpy((amp * u).astype(np.float32))
        sd[nl + "projection_q.weight"] = sd[nl + "projection_q.weight"] * gain
        sd[nl + "projection_k.weight"] = sd[nl + "projection_k.weight"] * gain
    b = synthetic.synthetic_batch([seed], N=N, T=T)
    big = torch.arange(N) % 2 == 0
    x = b["corr_pos"].clone() * 0.25
    x[0, big] *= ratio
    grid = torch.stack(torch.meshgrid(*[torch.arange(10.)] * 3, indexing="ij"), -1).reshape(-1, 3)[r.permutation(1000)[:N]] * 0.3
    b["corr_pos"], b["src_keypts"], b["tgt_keypts"] = x, grid[None].clone(), 3.0 * grid[None]
    return sd, b, big


@pytest.mark.parametrize("log2_ratio", [14, 15])
def test_pv_fp8_shared_scale_worst_case(log2_ratio):
    """the range contract of the pv_fp8 form, tested where it is weakest (`_pv_fp8_adversarial_case`: half
    the keys of EVERY tile 2^-14 / 2^-15 below the others in every feature of V, softmax peaked on one small key).  Both
    grid paths.  Asserted: finite; on the small rows (logits of O(1)) the HIP logits are no further from the fp64 evaluation
    than 1.5 x the reference's own fp32 evaluation + 2e-5 in BOTH forms (2^15, default form: + 4e-5 - that case is past
    the documented range), and the two forms agree to 4e-5 - the analysis
    predicts <= 2^-11 |v| of the attended key per lost cross term, i.e. ~1e-5 on these logits (CPU emulation of the scheme,
    tests/tools/mx_cross_emulation.py: 5.4e-6 / 1.1e-5 against 7e-8 / 1e-7 for three f16 products and 2.9e-6 for the
    reference's fp32); on the big rows (logits of 1e4) the same relative to the largest logit."""
    from gmf_amd import _lib
    sd, b, big = _pv_fp8_adversarial_case(2.0 ** log2_ratio)
    m = gmf_amd.PointDSC(num_layers=2)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).eval()
    with torch.no_grad():
        c32, _ = O.compat_matrix(b["src_keypts"], b["tgt_keypts"], 0.1)
        ref = O.classifier(sd, O.encoder(sd, b["corr_pos"], c32, b["p_tokens"], b["q_tokens"], 2))[0]
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        b64 = {k: v.double() for k, v in b.items()}
        c64, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 0.1)
        truth = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], c64, b64["p_tokens"], b64["q_tokens"], 2))[0]
    one = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    rag = {k: [one[k][0], one[k][0]] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=torch.cat([one["p_tokens"]] * 2), q_tokens=torch.cat([one["q_tokens"]] * 2), testing=True)
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    h = _lib.handle_for(0)
    got = {}
    try:
        for pv in (0, 2):
            h.call("gmf_set_tuning", b"pv_fp8", pv)
            got[pv, "small grid"] = m.encode(*[one[k] for k in keys])[0][0].cpu()
            got[pv, "large grid"] = m(rag)["logits"][0].cpu()
    finally:
        h.call("gmf_set_tuning", b"pv_fp8", 1)
    gmf_amd.check_status()
    small = ~big
    floor_s = float((ref[small].double() - truth[small]).abs().max())
    top = float(truth[big].abs().max())
    floor_b = float((ref[big].double() - truth[big]).abs().max()) / top
    for name in ("small grid", "large grid"):
        for pv in (0, 2):
            lg = got[pv, name]
            assert torch.isfinite(lg).all(), (name, pv)
            e_s = float((lg[small].double() - truth[small]).abs().max())
            e_b = float((lg[big].double() - truth[big]).abs().max()) / top
            print(f"ratio 2^{log2_ratio}, {name}, pv_fp8 {pv}: small rows vs fp64 {e_s:.2e} (fp32 oracle {floor_s:.2e}); "
                  f"big rows relative {e_b:.2e} (fp32 oracle {floor_b:.2e})")
            slack = 2e-5 if (pv == 0 or log2_ratio == 14) else 4e-5     # (measured at 2^15: 2.2e-5 with a floor of 2.8e-6)
            assert e_s < 1.5 * floor_s + slack, (name, pv, e_s, floor_s)
            assert e_b < 1.5 * floor_b + 2e-5, (name, pv, e_b, floor_b)
        d = _maxerr(got[2, name][small], got[0, name][small])
        print(f"ratio 2^{log2_ratio}, {name}: pv_fp8 1 vs 0 on the small rows {d:.2e}")
        assert d < 4e-5, (name, d)          # measured 1.4e-5 (2^14) / 2.3e-5 (2^15); the bound is 2^-10 |v| of the attended key


def test_packed_weights_live_in_a_torch_tensor(model, sd_full):
    """[r4] The packed weights of a module are a torch tensor: gmf_encoder_pack_weights with GMF_PACK_HOST_BLOCK packs on the
    host, gmf_packed_encoder_place copies the block into caller-owned device memory (here: torch's caching allocator) and re-bases
    the gmf_encoder_weights - no hipMalloc per pack, no device-wide hipFree from __del__.  Placing the same object a second time
    moves the weights: the forward from the new block is bit-identical, and the C entry point refuses a short or misaligned block."""
    import ctypes as C
    from gmf_amd import _lib
    b = synthetic.synthetic_batch([71, 72], N=300, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res0 = model(data)
    lg0, T0 = model.last_logits.clone(), res0["final_trans"].clone()
    pw = model._weights(torch.device(DEV))
    assert torch.is_tensor(pw._block) and pw._block.is_cuda
    lib, h = _lib.load_library(), _lib.handle_for(0)
    nbytes = int(lib.gmf_packed_encoder_bytes(pw._p))
    assert nbytes == pw._block.numel() * 4 and nbytes > 10_000_000
    old = pw._block
    new = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.gmf_packed_encoder_place(h.h, pw._p, C.c_void_p(new.data_ptr() + 4), nbytes, st) == -1            # misaligned
    assert lib.gmf_packed_encoder_place(h.h, pw._p, C.c_void_p(new.data_ptr()), nbytes - 4, st) == -1           # too small
    assert lib.gmf_packed_encoder_place(h.h, pw._p, C.c_void_p(new.data_ptr()), nbytes, st) == 0
    pw._block = new
    old.fill_(float("nan"))                      # the old block is no longer referenced by the weights
    res1 = model(data)
    assert torch.equal(model.last_logits, lg0) and torch.equal(res1["final_trans"], T0)
    gmf_amd.check_status()


def test_q_in_attention_is_bit_identical(model):
    """[r4] gmf_set_tuning("q_in_attention", 1) (default): on large grids k_linear_h2 no longer writes a Q' image - every attention
    workgroup projects the Q' of its own 128 query rows in its prologue (PointDSC.py:56; the same weight stages, the same MFMA
    order, the same bias add and fp16 split as k_linear_h2).  Logits, features and poses are bit-identical to the form with the
    image, on a uniform large-grid batch (sizes that leave padding waves), in both attention forms, and on a ragged batch."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch(list(range(40, 40 + 9)), N=3970, T=196)        # 9 x 32 row blocks >= 256: the two-launch form
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    sizes = [700, 1531, 5000, 257]
    pairs = [synthetic.synthetic_batch([300 + i], N=n, T=196) for i, n in enumerate(sizes)]
    rag = {k: [_gpu(p[k][0]) for p in pairs] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=torch.cat([_gpu(p["p_tokens"]) for p in pairs]), q_tokens=torch.cat([_gpu(p["q_tokens"]) for p in pairs]), testing=True)
    got = {}
    try:
        for pv in (1, 0):
            h.call("gmf_set_tuning", b"pv_fp8", pv)
            for knob in (1, 0):
                h.call("gmf_set_tuning", b"q_in_attention", knob)
                res = model(data)
                r2 = model(rag)
                got[pv, knob] = (model.encode(*[data[k] for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")], True),
                                 res["final_trans"].clone(), torch.cat(r2["logits"]).clone(), r2["final_trans"].clone())
    finally:
        h.call("gmf_set_tuning", b"q_in_attention", 1)
        h.call("gmf_set_tuning", b"pv_fp8", 1)
    for pv in (1, 0):
        (lg1, fn1, f1), T1, rl1, rT1 = got[pv, 1]
        (lg0, fn0, f0), T0, rl0, rT0 = got[pv, 0]
        assert torch.equal(lg1, lg0) and torch.equal(fn1, fn0) and torch.equal(f1, f0) and torch.equal(T1, T0), pv
        assert torch.equal(rl1, rl0) and torch.equal(rT1, rT0), pv
    ref = O.pointdsc_forward(synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7), {k: v[:1] for k, v in b.items()}, testing=True)
    assert _maxerr(got[1, 1][0][0][:1].cpu(), ref["logits"]) < 1e-4


def test_compat_format_16bit_is_an_opt_in_within_the_gate_on_3dmatch_shape(model, sd_full):
    """gmf_set_tuning("compat_format", 2): the compat cache as 16-bit fixed point (half the cache and its stream; DESIGN section 4b
    has why it is not the default: KITTI-shape inputs).  On a 3DMatch-shape large-grid batch both attention forms (pv_fp8 0 / 1)
    stay within 1e-4 of the fp32 oracle with it, and within 1e-4 of the fp32-cache logits."""
    from gmf_amd import _lib
    B, N = 33, 1000                                            # 264 row blocks: the large-grid (two-launch) form
    b = synthetic.synthetic_batch(list(range(700, 700 + B)), N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    one = {k: v[:1] for k, v in b.items() if torch.is_tensor(v)}
    ref0 = O.pointdsc_forward(sd_full, one, testing=True)["logits"][0]
    h = _lib.handle_for(0)
    model(data)
    base = model.last_logits.clone()
    try:
        for pv in (0, 1):
            h.call("gmf_set_tuning", b"pv_fp8", pv)
            h.call("gmf_set_tuning", b"compat_format", 2)
            model(data)
            lg = model.last_logits.clone()
            d, e = _maxerr(lg.cpu(), base.cpu()), _maxerr(lg[0].cpu(), ref0)
            print(f"compat_format 2, pv_fp8 {pv}: vs fp32 cache {d:.2e}, pair 0 vs oracle {e:.2e}")
            assert d < 1e-4 and e < 1e-4, (pv, d, e)
        with pytest.raises(RuntimeError):
            h.call("gmf_set_tuning", b"compat_format", 3)
    finally:
        h.call("gmf_set_tuning", b"compat_format", 0)
        h.call("gmf_set_tuning", b"pv_fp8", 1)


def test_stress_conditioning(golden_dir):
    """gain 0.9 weights amplify rounding by ~1.4x per block: two fp32 evaluations that only differ in summation
    order disagree by the noise floor |oracle32 - oracle64|.  The HIP path must stay within 4x that floor."""
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, gain=0.9)
    b = synthetic.synthetic_batch([42], N=257, T=196)
    o32 = O.pointdsc_forward(sd, b, testing=False)["logits"]
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: v.double() for k, v in b.items()}
    compat, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 0.1)
    o64 = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat, b64["p_tokens"], b64["q_tokens"], 12))
    floor = float((o32.double() - o64).abs().max())
    m = gmf_amd.PointDSC(num_layers=12)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).eval()
    lg = m.encode(*[_gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")])[0]
    err = float((lg.cpu().double() - o64).abs().max())
    assert err < 4 * max(floor, 2e-5), (err, floor)


@pytest.mark.parametrize("scene,N,T", [(95, 3000, 40), (97, 3000, 196), (3, 777, 196)])
def test_random_scenes_against_the_fp32_noise_floor(model, sd_full, scene, N, T):
    """Scenes of the randomised sweep (tests/tools/parity_sweep.py, seeds 1000 + scene).  Scene 95 is the one of its first 100 where
    fp32 arithmetic itself is short of the 1e-4 gate: the reference's own fp32 evaluation is 6.3e-5 from an fp64 evaluation of the
    same network, and two fp32 evaluations that differ only in summation order then differ by up to about twice that.  The
    contract tested [r3, tightened]: the HIP logits are no further from the EXACT network (the fp64 evaluation) than 1.5 x the
    reference's own fp32 evaluation is, plus 2e-5 - "about as accurate as the reference", stated against the truth rather than
    against the other fp32 result (measured over the 100-scene sweep: HIP median 1.3e-5 / oracle 1.5e-5; worst ratio scene 95:
    9.7e-5 against 6.3e-5) - and within 1e-4 of the fp32 oracle wherever that bound leaves room for it."""
    b = synthetic.synthetic_batch([1000 + scene], N=N, T=T)
    o32 = O.pointdsc_forward(sd_full, b, testing=True)["logits"]
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_full.items()}
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
    compat, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 0.1)
    o64 = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat, b64["p_tokens"], b64["q_tokens"], 12))
    floor = float((o32.double() - o64).abs().max())
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    model(data)
    lg = model.last_logits.cpu()
    e32, e64 = _maxerr(lg, o32), float((lg.double() - o64).abs().max())
    print(f"scene {scene}: HIP vs fp32 oracle {e32:.2e}, vs fp64 {e64:.2e}; fp32 oracle vs fp64 {floor:.2e}")
    assert e64 < 1.5 * floor + 2e-5, (e64, floor)
    assert e32 < max(1e-4, 2.5 * floor + 2e-5), (e32, floor)       # (follows from the line above by the triangle inequality)


def test_f5_f7_pose_head(golden_dir, model):
    g = _load(golden_dir, "f5_f7_pose_head.npz")
    N = int(g["N"])
    b = synthetic.synthetic_batch([int(g["pair_seed"])], N=N, T=12)
    feat_n, scores = _gpu(torch.from_numpy(g["feat_n"])), _gpu(torch.from_numpy(g["scores"]))
    src, tgt = _gpu(b["src_keypts"]), _gpu(b["tgt_keypts"])
    seeds = model.pick_seeds(None, scores, R=0.10, max_num=int(N * 0.1), src_keypts=src)
    assert (seeds.cpu().numpy() == g["seeds"]).all()
    fT, labels, aux = model.pose_head(feat_n, src, tgt, scores, testing=True, return_aux=True)
    assert (aux["seeds"].cpu().numpy() == g["seeds"]).all()
    assert (np.sort(aux["knn_idx"].cpu().numpy(), -1) == np.sort(g["knn_idx"], -1)).mean() > 0.999
    assert _maxerr(aux["seed_trans"].cpu(), g["seed_trans"]) < 1e-3
    assert _maxerr(aux["fitness"].cpu(), g["fitness"]) < 1e-6
    assert (labels.cpu().numpy() == g["labels"]).all()
    assert _maxerr(fT.cpu(), g["refined"]) < 1e-4
    fT0, _, _ = model.pose_head(feat_n, src, tgt, scores, testing=False, seeds=_gpu(torch.from_numpy(g["seeds"])))
    assert _maxerr(fT0.cpu(), g["final_trans"]) < 1e-4
    ref = model.post_refinement(_gpu(torch.from_numpy(g["final_trans"])), src, tgt)
    assert _maxerr(ref.cpu(), g["refined"]) < 1e-4


def test_power_iteration_early_exit_is_resolved_per_pair(model):
    """PointDSC.py:444: the power iteration stops as soon as EVERY seed passes allclose.  k_seed_power sums the weighted Kabsch
    problem of the last iterate itself; when the pair stopped earlier, k_seed_kabsch redoes the seeds from the iterate the
    reference stopped at.  A scene of exact inliers with near-identical features (M close to all-ones: convergence in a few
    steps) takes the early path; the same kind of scene with noise and 3 iterations allowed runs them all.  Both match the
    oracle's seed_weights + rigid_transform_3d (the oracle's power iteration has the same exit)."""
    gen = torch.Generator().manual_seed(77)
    N, k = 200, 40
    sigma, sigma_d = float(model.sigma.detach()), float(model.sigma_spat)
    iters_default = model.num_iterations
    try:
        for early in (True, False):
            model.num_iterations = iters_default if early else 3
            src = torch.rand(1, N, 3, generator=gen) * 2
            A = torch.linalg.qr(torch.randn(3, 3, generator=gen))[0]
            A = A * torch.sign(torch.det(A))
            tgt = src @ A.T + torch.tensor([0.3, -0.2, 0.1])
            u = torch.randn(1, 1, 128, generator=gen)
            if not early:
                tgt[:, ::3] += 0.5 * torch.randn(1, (N + 2) // 3, 3, generator=gen)
            feat_n = torch.nn.functional.normalize(u + (0.02 if early else 0.6) * torch.randn(1, N, 128, generator=gen), dim=-1)
            logits = torch.randn(1, N, generator=gen)
            fT, _, aux = model.pose_head(_gpu(feat_n), _gpu(src), _gpu(tgt), _gpu(logits), testing=False, return_aux=True)
            knn_idx = aux["knn_idx"].cpu().long()
            w, sk, tk = O.seed_weights(feat_n, src, tgt, knn_idx, sigma, sigma_d, model.num_iterations)
            Ts = O.rigid_transform_3d(sk, tk, w).reshape(1, -1, 4, 4)
            assert _maxerr(aux["seed_trans"].cpu(), Ts) < 1e-4
            # which path ran: replay the oracle's iteration and note where it stops
            bi = torch.arange(1)[:, None, None]
            f = feat_n[bi, knn_idx]
            Mf = torch.clamp(1 - (1 - f @ f.transpose(2, 3)) / sigma ** 2, min=0)
            d = torch.cdist(src[bi, knn_idx], src[bi, knn_idx]) - torch.cdist(tgt[bi, knn_idx], tgt[bi, knn_idx])
            M = (Mf * torch.clamp(1 - d * d / sigma_d ** 2, min=0)).reshape(-1, k, k).clone()
            M[:, torch.arange(k), torch.arange(k)] = 0
            v = torch.ones(M.shape[0], k, 1)
            stop = model.num_iterations - 1
            for it in range(model.num_iterations):
                nv = torch.bmm(M, v)
                nv = nv / (torch.norm(nv, dim=1, keepdim=True) + 1e-6)
                if torch.allclose(nv, v):
                    stop = it
                    break
                v = nv
            assert (stop < model.num_iterations - 1) == early, stop
    finally:
        model.num_iterations = iters_default


def test_throughput_precision_mode(model):
    """gmf_set_tuning("precision", 1) - the throughput numerics mode of SURVEY section 7 step 8: on large grids the
    spatial-consistency attention multiplies plain fp16 operands (one product instead of three, fp32 accumulation) and streams
    the compat matrix as fp16.  It is NOT within the 1e-4 parity gate and says so; its measured deviation from the parity mode
    is bounded here (logits 5e-3, > 99.9 % identical inlier labels, poses 5e-3), the default mode is unchanged by switching
    back (bitwise), small grids keep the parity numerics, and other values are rejected."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch(list(range(40, 72)), N=1000, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    r0 = model(data)
    lg0, T0, lab0 = model.last_logits.clone(), r0["final_trans"].clone(), r0["final_labels"].clone()
    try:
        h.call("gmf_set_tuning", b"precision", 1)
        r1 = model(data)
        lg1, T1, lab1 = model.last_logits.clone(), r1["final_trans"].clone(), r1["final_labels"].clone()
        small = {k: v[:2] if torch.is_tensor(v) else v for k, v in data.items()}     # 2 pairs x 1000: a small grid
        rs1 = model(small)["final_trans"].clone()
        with pytest.raises(RuntimeError):
            h.call("gmf_set_tuning", b"precision", 3)
    finally:
        h.call("gmf_set_tuning", b"precision", 0)
    d = float((lg1 - lg0).abs().max())
    assert 0.0 < d < 5e-3, d                                  # a different arithmetic, and a bounded one
    assert float((lab1 == lab0).float().mean()) > 0.999
    assert float((T1 - T0).abs().max()) < 5e-3
    r2 = model(data)
    assert torch.equal(model.last_logits, lg0) and torch.equal(r2["final_trans"], T0)
    assert torch.equal(model(small)["final_trans"], rs1)      # the small grid never left the parity numerics
    # PointDSC.set_precision belongs to the MODULE: its own forwards run in that mode, the handle's knob is back at parity after
    # each of them, so another module on the same device keeps the parity numerics
    import copy
    other = copy.deepcopy(model)                              # (the packed weights are never shared: the copy packs its own)
    try:
        model.set_precision("throughput")
        model(data)
        assert torch.equal(model.last_logits, lg1)
        other(data)
        assert torch.equal(other.last_logits, lg0)
    finally:
        model.set_precision("parity")
    model(data)
    assert torch.equal(model.last_logits, lg0)


def test_throughput_precision_level_2(model):
    """gmf_set_tuning("precision", 2): level 1 plus the layer's linear stages on the high fp16 planes only (grids of at least 512
    base workgroups).  A coarser arithmetic with a measured, bounded deviation - logits 0.2, > 97 % identical labels - that
    leaves the default mode untouched (bitwise) when switched back."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch(list(range(40, 72)), N=2048, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    r0 = model(data)
    lg0, T0, lab0 = model.last_logits.clone(), r0["final_trans"].clone(), r0["final_labels"].clone()
    out = {}
    try:
        for level in (1, 2):
            h.call("gmf_set_tuning", b"precision", level)
            r = model(data)
            out[level] = (model.last_logits.clone(), r["final_labels"].clone())
    finally:
        h.call("gmf_set_tuning", b"precision", 0)
    d1, d2 = float((out[1][0] - lg0).abs().max()), float((out[2][0] - lg0).abs().max())
    print(f"max |d logit| vs parity: level 1 {d1:.2e}, level 2 {d2:.2e}")
    assert torch.isfinite(out[2][0]).all() and d1 < d2 < 0.2
    assert float((out[2][1] == lab0).float().mean()) > 0.97
    r2 = model(data)
    assert torch.equal(model.last_logits, lg0) and torch.equal(r2["final_trans"], T0)


def test_power_iteration_exit_spans_the_batch(model):
    """PointDSC.py:444 tests allclose over the whole [bs * S, k] tensor: in a batch, a pair whose seeds have all converged keeps
    iterating until every pair has.  A fast-converging pair next to a slow one: the batch result matches the oracle run on the
    batch (which has the reference's exit), and the fast pair's hypotheses differ - within the allclose tolerance, but not
    bitwise - from what the same pair gives alone, where it stops early."""
    gen = torch.Generator().manual_seed(91)
    N = 200
    src, tgt, feat = [], [], []
    for early in (True, False):
        s_ = torch.rand(N, 3, generator=gen) * 2
        A = torch.linalg.qr(torch.randn(3, 3, generator=gen))[0]
        A = A * torch.sign(torch.det(A))
        t_ = s_ @ A.T + torch.tensor([0.3, -0.2, 0.1])
        if not early:
            t_[::3] += 0.5 * torch.randn((N + 2) // 3, 3, generator=gen)
        u = torch.randn(1, 128, generator=gen)
        feat.append(torch.nn.functional.normalize(u + (0.02 if early else 0.6) * torch.randn(N, 128, generator=gen), dim=-1))
        src.append(s_); tgt.append(t_)
    src, tgt, feat = torch.stack(src), torch.stack(tgt), torch.stack(feat)
    logits = torch.randn(2, N, generator=gen)
    sigma, sigma_d = float(model.sigma.detach()), float(model.sigma_spat)
    _, _, aux = model.pose_head(_gpu(feat), _gpu(src), _gpu(tgt), _gpu(logits), testing=False, return_aux=True)
    knn_idx = aux["knn_idx"].cpu().long()
    w, sk, tk = O.seed_weights(feat, src, tgt, knn_idx, sigma, sigma_d, model.num_iterations)      # the exit spans both pairs
    Ts = O.rigid_transform_3d(sk, tk, w).reshape(2, -1, 4, 4)
    assert _maxerr(aux["seed_trans"].cpu(), Ts) < 1e-4
    _, _, alone = model.pose_head(_gpu(feat[:1]), _gpu(src[:1]), _gpu(tgt[:1]), _gpu(logits[:1]), testing=False, return_aux=True)
    assert torch.equal(alone["knn_idx"], aux["knn_idx"][:1])
    d = (alone["seed_trans"] - aux["seed_trans"][:1]).abs().max()
    assert 0.0 < float(d) < 1e-4, float(d)


def test_f6_rigid_transform(golden_dir):
    g = _load(golden_dir, "f6_rigid_transform.npz")
    A, B, w = (_gpu(torch.from_numpy(g[k])) for k in ("A", "B", "w"))
    wc = w.clone()
    T = gmf_amd.rigid_transform_3d(A, B, wc)
    assert _maxerr(T.cpu(), g["T"]) < 1e-4
    assert float(wc.min()) >= 0          # negative weights are zeroed in place, as in the reference
    assert _maxerr(gmf_amd.rigid_transform_3d(A, B).cpu(), g["T_noweight"]) < 1e-4
    R = T[:, :3, :3].cpu().numpy()
    assert np.abs(np.linalg.det(R) - 1).max() < 1e-5
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    # degenerate inputs must not produce NaN: all-zero weights, a single repeated point
    Tz = gmf_amd.rigid_transform_3d(A[:2], B[:2], torch.zeros_like(w[:2]))
    assert torch.isfinite(Tz).all()
    Tp = gmf_amd.rigid_transform_3d(A[:1, :1].repeat(1, 5, 1), B[:1, :1].repeat(1, 5, 1))
    assert torch.isfinite(Tp).all()


@pytest.mark.parametrize("N", [10, 1000, 8000])
def test_f8_weighted_procrustes(golden_dir, N):
    g = _load(golden_dir, "f8_weighted_procrustes.npz")
    X, Y, w = (_gpu(torch.from_numpy(g[f"{k}_{N}"])) for k in ("X", "Y", "w"))
    R, t = gmf_amd.weighted_procrustes(X, Y, w, np.finfo(np.float32).eps)
    assert _maxerr(R.cpu(), g[f"R_{N}"]) < 1e-4
    assert _maxerr(t.cpu(), g[f"t_{N}"]) < 1e-4


@pytest.mark.parametrize("adam", ["foreach", "fused"])
def test_graph_captured_training_step_equals_eager_steps(sd_full, adam):
    """the reference's default training step (libs/trainer.py:131-166: train-mode forward, ClassificationLoss +
    SpectralMatchingLoss, backward, Adam) captured ONCE as a HIP graph (`gmf_amd.train.GraphedTrainingStep`) and replayed.  What made
    it capturable: sigma read by the kernels from the parameter's own device memory (`gmf_set_sigma_device`, ABI 5: the eager step
    reads it to the host once per step), loss statistics left on the device, a capturable Adam.  Two copies of the same model, same
    data: one takes 3 + 4 eager steps, the other 3 warm-up steps and 4 replays - the same parameters afterwards, bit for bit
    (every kernel is deterministic and both run the same optimizer code), and the per-step losses agree.  `adam`: torch's
    multi-tensor implementation and its fused one - the one to use: under capture the multi-tensor form falls back to three
    broadcast divisions PER PARAMETER (~950 extra launches, 4 ms of a 27 ms step at 16 x 1000)."""
    from gmf_amd import train as T
    B, N = 4, 500
    b = synthetic.synthetic_batch(list(range(600, 600 + B)), N=N, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["gt"] = _gpu(b["gt_labels"])
    cl_fn, sm_fn = gmf_amd.ClassificationLoss(balanced=False, host_stats=False), gmf_amd.SpectralMatchingLoss(balanced=False)

    def loss_fn(res, batch):
        return cl_fn(res["final_labels"], batch["gt"])["loss"] + sm_fn(res["M"], batch["gt"])

    def make():
        m = gmf_amd.PointDSC(num_layers=3)
        m.load_state_dict({k: v for k, v in synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 3, 128), seed=7).items()}, strict=False)
        m = m.to(DEV).train()
        return m, torch.optim.Adam([p for n, p in m.named_parameters() if not n.startswith("encoder.image_encoder.")], lr=1e-3, capturable=True,
                                   fused=(adam == "fused"))
    m_e, opt_e = make()
    m_e.sigma_on_device = True
    eager_losses = []
    for _ in range(7):
        opt_e.zero_grad(set_to_none=True)
        loss = loss_fn(m_e(data), data)
        loss.backward()
        opt_e.step()
        eager_losses.append(float(loss.detach()))
    m_g, opt_g = make()
    step = T.GraphedTrainingStep(m_g, opt_g, loss_fn, data, warmup=3)
    graph_losses = [float(step(data).detach()) for _ in range(4)]
    torch.cuda.synchronize()
    print("eager losses", [f"{x:.6f}" for x in eager_losses[3:]], "graph losses", [f"{x:.6f}" for x in graph_losses])
    assert graph_losses == eager_losses[3:]
    assert float(m_e.sigma.detach()) != 1.0                      # sigma is being trained ...
    for (n1, p1), (n2, p2) in zip(m_e.named_parameters(), m_g.named_parameters()):
        if not n1.startswith("encoder.image_encoder."):          # (not part of the token-fed step; randomly initialised per copy)
            assert n1 == n2 and torch.equal(p1, p2), n1          # ... and every parameter, sigma included, took the same seven steps
    for (n1, b1), (n2, b2) in zip(m_e.named_buffers(), m_g.named_buffers()):
        if "num_batches_tracked" not in n1 and not n1.startswith("encoder.image_encoder."):
            assert torch.equal(b1, b2), n1                       # BatchNorm running statistics
    # the by-value form (sigma read to the host every step) computes the same step
    m_v, opt_v = make()
    for _ in range(2):
        opt_v.zero_grad(set_to_none=True)
        lv = loss_fn(m_v(data), data)
        lv.backward()
        opt_v.step()
    assert float(lv.detach()) == eager_losses[1]


def test_dgr_argmin_se3_and_transformation():
    """The two remaining helper names of the DGR registration module (core/registration.py:67-88, 116-132; VERDICT r4 missing 6):
    `argmin_se3_squared_dist` (unweighted Kabsch: weighted_procrustes with unit weights, eps 0) against the oracle's solve and
    the ground truth; `Transformation` (6-D rotation parameters, first two columns of R) reproduces its initial pose and is
    differentiable."""
    from gmf_amd import synthetic as syn
    X, Y, _, Rgt, tgt = syn.dgr_scene(3000, 77, inlier_ratio=1.0)
    R, t = gmf_amd.argmin_se3_squared_dist(_gpu(X), _gpu(Y))
    assert _maxerr(R.cpu(), Rgt) < 2e-3 and _maxerr(t.cpu(), tgt) < 5e-3            # (noise 0.01 on 3000 points)
    Ro, to = O.weighted_procrustes(X, Y, torch.ones(X.shape[0], 1), 0.0)
    assert _maxerr(R.cpu(), Ro) < 1e-5 and _maxerr(t.cpu(), to) < 1e-5
    assert abs(float(torch.linalg.det(R.cpu())) - 1.0) < 1e-5
    Tm = gmf_amd.Transformation(R.cpu(), t.cpu())
    pts = torch.randn(50, 3)
    assert _maxerr(Tm(pts).detach(), pts @ R.cpu().t() + t.cpu()) < 1e-5
    assert _maxerr(gmf_amd.ortho2rotation(Tm.rot6d.detach()), O.ortho2rotation(Tm.rot6d.detach())) < 1e-6
    Tm(pts).sum().backward()
    assert Tm.rot6d.grad is not None and torch.isfinite(Tm.rot6d.grad).all() and Tm.trans.grad is not None


def test_weighted_procrustes_ragged_batch(golden_dir):
    g = _load(golden_dir, "f8_weighted_procrustes.npz")
    Ns = [10, 1000, 8000, 1000]
    X = _gpu(torch.from_numpy(np.concatenate([g[f"X_{n}"] for n in Ns])))
    Y = _gpu(torch.from_numpy(np.concatenate([g[f"Y_{n}"] for n in Ns])))
    w = _gpu(torch.from_numpy(np.concatenate([g[f"w_{n}"] for n in Ns])))
    R, t = gmf_amd.weighted_procrustes_batched(X, Y, w, np.cumsum([0] + Ns).tolist(), np.finfo(np.float32).eps)
    for i, n in enumerate(Ns):
        assert _maxerr(R[i].cpu(), g[f"R_{n}"]) < 1e-4 and _maxerr(t[i].cpu(), g[f"t_{n}"]) < 1e-4
    # [r4] the offsets as an int32 device tensor (taken as it is), the same list again (its cached device copy), and what
    # the host-side check refuses: not from 0, not to N, not increasing, a single entry
    off = torch.tensor(np.cumsum([0] + Ns), dtype=torch.int32, device=X.device)
    R2, t2 = gmf_amd.weighted_procrustes_batched(X, Y, w, off, np.finfo(np.float32).eps)
    R3, t3 = gmf_amd.weighted_procrustes_batched(X, Y, w, np.cumsum([0] + Ns).tolist(), np.finfo(np.float32).eps)
    assert torch.equal(R2, R) and torch.equal(t2, t) and torch.equal(R3, R) and torch.equal(t3, t)
    total = int(sum(Ns))
    for bad in ([1, total], [0, total - 1], [0, 500, 500, total], [0, 600, 500, total], [0]):
        with pytest.raises(RuntimeError, match="offsets"):
            gmf_amd.weighted_procrustes_batched(X, Y, w, bad, 1e-7)
    with pytest.raises(RuntimeError, match="offsets"):
        gmf_amd.weighted_procrustes_batched(X, Y, w, off.long(), 1e-7)
    # a scene far from the origin: the one-pass raw moments of k_weighted_procrustes against the centred fp64 sums
    Xf, Yf = X[:1000] + 250.0, Y[:1000] - 180.0
    Rf, tf = gmf_amd.weighted_procrustes(Xf, Yf, w[10:1010], 1e-7)
    Xd, Yd, wd = Xf.double().cpu(), Yf.double().cpu(), w[10:1010].reshape(-1).double().cpu()
    wn = (wd / (wd.abs().sum() + 1e-7))[:, None]
    mx, my = (wn * Xd).sum(0), (wn * Yd).sum(0)
    U, _, Vt = torch.linalg.svd((Yd - my).T @ (wn * (Xd - mx)))
    D = torch.diag(torch.tensor([1.0, 1.0, float(torch.det(U) * torch.det(Vt))], dtype=torch.float64))
    Rr = U @ D @ Vt
    assert _maxerr(Rf.cpu().double(), Rr) < 1e-5 and _maxerr(tf.cpu().double(), my - Rr @ mx) < 1e-3


def test_full_size_properties(model):
    """BASELINE config sizes, checked through size-independent properties (the oracle would take minutes):
    finite outputs, rigid poses, pose close to ground truth, logits invariant to the rigid frame of tgt."""
    b = synthetic.synthetic_batch([71, 72], N=5000, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = model(data)
    T = res["final_trans"].cpu().numpy()
    assert np.isfinite(T).all() and torch.isfinite(model.last_logits).all()
    R = T[:, :3, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    assert np.abs(T - b["gt_trans"].numpy()).max() < 5e-2
    lab = res["final_labels"].cpu().numpy()
    assert ((lab > 0.5) == (b["gt_labels"].numpy() > 0.5)).mean() > 0.97
    # one pair against the oracle at full size (about 2 s of CPU)
    one = {k: v[:1] for k, v in b.items()}
    ref = O.pointdsc_forward(synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7), one, testing=False)
    assert _maxerr(model.last_logits[:1].cpu(), ref["logits"]) < 1e-4


def test_config2_full_batch(model, sd_full):
    """BASELINE config 2 at FULL size - 32 pairs x 5000 correspondences, 196 tokens (the benchmark's own batch: the fused
    two-launch layers, the compat cache at 3.2 GB, 1 280 workgroups per launch): finite rigid poses close to the ground truth
    and > 97 % correct labels for all 32 pairs; pairs 0, 15 and 31 against the oracle (logits 1e-4, pose 1e-3 - about 6 s of
    CPU); and the same three pairs run alone (B = 1: the small-grid three-launch path) give the batch's logits to 5e-5."""
    B, N = 32, 5000
    b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = model(data)
    logits = model.last_logits.clone()
    T = res["final_trans"].cpu().numpy()
    assert np.isfinite(T).all() and torch.isfinite(logits).all()
    R = T[:, :3, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    assert np.abs(T - b["gt_trans"].numpy()).max() < 5e-2
    lab = res["final_labels"].cpu().numpy()
    assert ((lab > 0.5) == (b["gt_labels"].numpy() > 0.5)).mean(axis=1).min() > 0.97
    for p in (0, 15, 31):
        one = {k: v[p:p + 1] for k, v in b.items()}
        ref = O.pointdsc_forward(sd_full, one, testing=True)
        assert _maxerr(logits[p:p + 1].cpu(), ref["logits"]) < 1e-4
        assert _maxerr(T[p:p + 1], ref["final_trans"].numpy()) < 1e-3
        d1 = {k: (v[p:p + 1] if torch.is_tensor(v) else v) for k, v in data.items()}
        r1 = model(d1)
        assert _maxerr(model.last_logits.cpu(), logits[p:p + 1].cpu()) < 5e-5     # (split sums in another order: fp32 noise floor 2e-5)
        assert _maxerr(r1["final_trans"].cpu(), T[p:p + 1]) < 1e-4


def test_compat_cache_beyond_2_31_elements(model, sd_full):
    """the largest row bench.py times - 32 pairs x 10 000 correspondences - puts 32 x 313^2 x 1024 = 3.2e9
    elements (12.8 GB) in the compat cache (PointDSC.py:216-221), past 2^31: every tile offset on the path is 64-bit or this
    test fails.  Properties for all 32 pairs (finite, rigid, close to the ground truth, labels), the LAST pair - the highest
    addresses - and pair 0 against the CPU oracle at the literal 1e-4 (logits; pose: 3e-3, the tie contract of F16), and the last pair run alone (B = 1:
    a cache of 1e8 elements) reproduces its row of the batch to 5e-5.  Then the ragged entry with sum n_i^2 tiles past 2^31 as
    well: 24 pairs of 9 400 ... 10 000 rows in one launch, two of them against their own B = 1 runs."""
    B, N = 32, 10000
    assert B * ((N + 31) // 32) ** 2 * 1024 > 2 ** 31
    b = synthetic.synthetic_batch(list(range(2000, 2000 + B)), N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = model(data)
    logits = model.last_logits.clone()
    T = res["final_trans"].cpu().numpy()
    assert np.isfinite(T).all() and torch.isfinite(logits).all()
    R = T[:, :3, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    assert np.abs(T - b["gt_trans"].numpy()).max() < 5e-2
    lab = res["final_labels"].cpu().numpy()
    assert ((lab > 0.5) == (b["gt_labels"].numpy() > 0.5)).mean(axis=1).min() > 0.97
    for p in (B - 1, 0):
        one = {k: v[p:p + 1] for k, v in b.items()}
        ref = O.pointdsc_forward(sd_full, one, testing=True)
        e = _maxerr(logits[p:p + 1].cpu(), ref["logits"])
        print(f"32 x 10000, pair {p}: HIP vs oracle {e:.2e}, pose {_maxerr(T[p:p + 1], ref['final_trans'].numpy()):.2e}")
        assert e < 1e-4, (p, e)
        # (pose: the contract of scenes whose seed list involves zero-key ties, ordered by index here - golden F16; pair 0 of this
        # batch is one: 1.4e-3 from the oracle's pose, 6.6e-7 on pair 31, both as close to the ground truth as the oracle's)
        assert _maxerr(T[p:p + 1], ref["final_trans"].numpy()) < 3e-3
        gt = b["gt_trans"][p:p + 1].numpy()
        assert _maxerr(T[p:p + 1], gt) <= _maxerr(ref["final_trans"].numpy(), gt) + 5e-4
    d1 = {k: (v[B - 1:B] if torch.is_tensor(v) else v) for k, v in data.items()}
    r1 = model(d1)
    assert _maxerr(model.last_logits.cpu(), logits[B - 1:B].cpu()) < 5e-5
    assert _maxerr(r1["final_trans"].cpu(), T[B - 1:B]) < 1e-4
    del data, res, logits
    torch.cuda.empty_cache()
    # the ragged entry: every pair keeps a slot of tiles(max n)^2 tiles
    sizes = [10000 - 25 * i for i in range(24)]
    assert len(sizes) * ((max(sizes) + 31) // 32) ** 2 * 1024 > 2 ** 31
    pairs = [synthetic.synthetic_batch([2100 + i], N=n, T=196) for i, n in enumerate(sizes)]
    rag = {k: [_gpu(q[k][0]) for q in pairs] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=torch.cat([_gpu(q["p_tokens"]) for q in pairs]), q_tokens=torch.cat([_gpu(q["q_tokens"]) for q in pairs]), testing=True)
    rr = model(rag)
    for i, q in enumerate(pairs):
        lg = rr["logits"][i]
        assert lg.shape == (sizes[i],) and torch.isfinite(lg).all()
        assert np.abs(rr["final_trans"][i].cpu().numpy() - q["gt_trans"][0].numpy()).max() < 5e-2
    for i in (len(sizes) - 1, 0):
        d1 = {k: _gpu(pairs[i][k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
        d1["testing"] = True
        r1 = model(d1)
        assert _maxerr(model.last_logits[0].cpu(), rr["logits"][i].cpu()) < 5e-5, i
        assert _maxerr(r1["final_trans"][0].cpu(), rr["final_trans"][i].cpu()) < 1e-4, i


def test_kitti_shape_long_sequence():
    """BASELINE config 3 at full size: 16 pairs x N = 10000, sigma_d = tau = 1.2 (KITTI).  One pair against the oracle is ~10 s
    of CPU, so the full batch is checked through properties, plus a smaller KITTI-shape pair against the oracle."""
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=1.2)
    m = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                         inlier_threshold=1.2, sigma_d=1.2, k=40, nms_radius=1.2)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).eval()
    b = synthetic.synthetic_batch(list(range(81, 97)), N=10000, T=196, kind="kitti")      # config 3 at full size: 16 pairs
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = m(data)
    T = res["final_trans"].cpu().numpy()
    assert np.isfinite(T).all() and torch.isfinite(m.last_logits).all()
    R = T[:, :3, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    assert np.abs(T[:, :3, :3] - b["gt_trans"].numpy()[:, :3, :3]).max() < 2e-2
    assert np.abs(T[:, :3, 3] - b["gt_trans"].numpy()[:, :3, 3]).max() < 0.5
    # smaller KITTI-shape pair against the oracle (sigma_d = 1.2 path of the compat term)
    b2 = synthetic.synthetic_batch([83], N=700, T=50, kind="kitti")
    ref = O.pointdsc_forward(sd, b2, inlier_threshold=1.2, nms_radius=1.2, testing=True)
    d2 = {k: _gpu(b2[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    d2["testing"] = True
    r2 = m(d2)
    # KITTI coordinates are ~40x larger than 3DMatch ones, so with the same weights the fp32 noise floor of the
    # computation itself (oracle fp32 vs oracle fp64) is ~1.4e-4 here.  [r3, tightened] the HIP logits are no further from the
    # fp64 evaluation than 1.5 x the reference's own fp32 evaluation is, plus 2e-5 (measured: 1.27e-4 against 1.39e-4).
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: v.double() for k, v in b2.items()}
    compat, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], 1.2)
    o64 = O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat, b64["p_tokens"], b64["q_tokens"], 12))
    floor = float((ref["logits"].double() - o64).abs().max())
    e64 = float((m.last_logits.cpu().double() - o64).abs().max())
    print(f"kitti N=700: HIP vs fp64 {e64:.2e}, fp32 oracle vs fp64 {floor:.2e}, HIP vs fp32 oracle {_maxerr(m.last_logits.cpu(), ref['logits']):.2e}")
    assert e64 < 1.5 * floor + 2e-5, (e64, floor)
    assert _maxerr(m.last_logits.cpu(), ref["logits"]) < max(1e-4, 2.5 * floor + 2e-5), floor
    assert _maxerr(r2["final_trans"].cpu(), ref["final_trans"]) < 1e-3


def _kitti_model(sd):
    m = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                         inlier_threshold=1.2, sigma_d=1.2, k=40, nms_radius=1.2)
    m.load_state_dict(sd, strict=False)
    return m.to(DEV).eval()


def _fp64_logits(sd, b, sigma_d):
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: v.double() for k, v in b.items()}
    compat, _ = O.compat_matrix(b64["src_keypts"], b64["tgt_keypts"], sigma_d)
    return O.classifier(sd64, O.encoder(sd64, b64["corr_pos"], compat, b64["p_tokens"], b64["q_tokens"], 12))


@pytest.mark.parametrize("wset", ["stress", "cond"])
@pytest.mark.parametrize("case", range(2))
def test_f22_kitti_branch(golden_dir, wset, case):
    """Golden F22: the reference's own PointDSC in its KITTI configuration (sigma_d = tau = nms_radius = 1.2,
    evaluation/test_KITTI.py:219; the `[1.2] * 20` refinement list of PointDSC.py:505-508) on KITTI-shape scenes of +-40 m,
    N = 700 / 2000.  On the CONDITIONED weight set (`synthetic.kitti_conditioned`: layer0.weight / 13, so that activations are
    at the 3DMatch scale and the reference's own fp32 evaluation is 1.7e-5 from the exact network) the HIP logits are held to
    the LITERAL 1e-4 against the reference's, the pose to the F16 contract (seed ties by index) and to 1e-4 where the
    reference's seed list involves no tie.  On the STRESS set (the seeded weights as they are: the reference's fp32 is 3e-4
    from fp64 there) the gate is the floor-relative one: no further from the fp64 evaluation than 1.5 x the reference's own
    fp32 evaluation + 2e-5 - the same bound for three f16 products in P V ("pv_fp8" = 0) and for the DEFAULT form ("pv_fp8" = 1),
    whose device-side guard sends the layers of this ill-conditioned network that can put a query's whole softmax mass on one
    key (its first five) to the three-product form (round 4 held the default to 2.5 x)."""
    g = _load(golden_dir, "f22_kitti_branch.npz")
    N, seed = (int(v) for v in g["cases"][case])
    tag = f"{wset}_{N}_{seed}"
    tau = float(g["tau"])
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=float(g["sigma_d"]))
    if wset == "cond":
        sd = synthetic.kitti_conditioned(sd, float(g["layer0_div"]))
    from gmf_amd import _lib
    m = _kitti_model(sd)
    b = synthetic.synthetic_batch([seed], N=N, T=196, kind="kitti")
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    ref_lg = torch.from_numpy(g[f"logits_{tag}"])
    o64 = _fp64_logits(sd, b, 1.2)
    floor = float((ref_lg.double() - o64).abs().max())
    h = _lib.handle_for(0)
    try:
        h.call("gmf_set_tuning", b"pv_fp8", 0)
        m(data)
        lg3 = m.last_logits.cpu()
    finally:
        h.call("gmf_set_tuning", b"pv_fp8", 1)
    h.status(clear=True)
    res = m(data)
    lg = m.last_logits.cpu()
    dl, e64, e64_3 = _maxerr(lg, ref_lg), float((lg.double() - o64).abs().max()), float((lg3.double() - o64).abs().max())
    print(f"F22 {tag}: HIP vs reference {dl:.2e} (three f16 products: {_maxerr(lg3, ref_lg):.2e}); vs fp64: HIP {e64:.2e} "
          f"(three f16 products: {e64_3:.2e}), reference {floor:.2e}")
    if wset == "cond":
        assert dl < 1e-4, dl
        assert e64 < 1.5 * floor + 2e-5, (e64, floor)
        assert not (h.status() & _lib.GMF_STATUS_PV_GUARDED)        # the conditioned network never trips the guard
    else:
        # The stress set is where UNGUARDED e4m3 cross products of P V show (measured in round 4, N = 2000: 4.9e-4 / 6.1e-4 small /
        # large grid against 2.8e-4; three f16 products 3.4e-4): the guard is what holds the default to the same bound
        assert e64_3 < 1.5 * floor + 2e-5, (e64_3, floor)
        assert e64 < 1.5 * floor + 2e-5, (e64, floor)
        assert h.status() & _lib.GMF_STATUS_PV_GUARDED
    T_hip, T_ref, T_gt = res["final_trans"].cpu().numpy(), g[f"final_trans_{tag}"], g[f"gt_trans_{tag}"]

    def inliers(T):
        p = b["src_keypts"][0].numpy() @ T[0, :3, :3].T + T[0, :3, 3]
        return int((np.linalg.norm(p - b["tgt_keypts"][0].numpy(), axis=-1) < tau).sum())
    assert inliers(T_hip) >= inliers(T_ref)
    assert _maxerr(T_hip, T_ref) < 3e-3
    assert _maxerr(T_hip, T_gt) <= _maxerr(T_ref, T_gt) + 5e-4
    if _maxerr(T_hip, T_ref) < 1e-5:
        assert np.array_equal(res["final_labels"].cpu().numpy().astype(np.uint8), g[f"final_labels_{tag}"])


def test_config3_conditioned_weights_literal_gate():
    """BASELINE config 3 (16 pairs x N = 10000, sigma_d = tau = 1.2, KITTI-shape scenes) with the conditioned weight set
    (`synthetic.kitti_conditioned`; golden F22 pins the oracle on this branch against the reference itself): pair 5 of the
    FULL batch against the CPU oracle at the literal 1e-4 on the logits and 1e-3 on the pose (the pose contract of the
    tie scenes, F16), every pair against the ground truth."""
    sd = synthetic.kitti_conditioned(synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=1.2))
    m = _kitti_model(sd)
    b = synthetic.synthetic_batch(list(range(81, 97)), N=10000, T=196, kind="kitti")
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = m(data)
    T = res["final_trans"].cpu().numpy()
    assert np.isfinite(T).all() and torch.isfinite(m.last_logits).all()
    assert np.abs(T[:, :3, :3] - b["gt_trans"].numpy()[:, :3, :3]).max() < 2e-2
    assert np.abs(T[:, :3, 3] - b["gt_trans"].numpy()[:, :3, 3]).max() < 0.5
    p = 5
    one = {k: v[p:p + 1] for k, v in b.items()}
    with torch.no_grad():
        ref = O.pointdsc_forward(sd, one, inlier_threshold=1.2, nms_radius=1.2, testing=True)
    dl = _maxerr(m.last_logits[p:p + 1].cpu(), ref["logits"])
    print(f"config 3, conditioned weights, pair {p} of 16 x 10000: HIP vs fp32 oracle {dl:.2e}, "
          f"pose {_maxerr(T[p:p + 1], ref['final_trans'].numpy()):.2e}")
    assert dl < 1e-4, dl
    assert _maxerr(T[p:p + 1], ref["final_trans"].numpy()) < 1e-3


def test_dgr_config5_batched_procrustes():
    """BASELINE config 5: DGR surface, N = 8000 correspondences per pair, sigmoid weights clipped at 0.05,
    32 pairs in one launch; R, t within 1e-4 of the oracle (fp64 SVD) pair by pair."""
    B, N = 32, 8000
    Xs, Ys, ws, refs = [], [], [], []
    for i in range(B):
        r = np.random.default_rng([205, i])
        X = r.uniform(0, 3, (N, 3)).astype(np.float32)
        Rg = synthetic.random_rotation(r)
        Y = (X @ Rg.T + r.uniform(-0.5, 0.5, 3)).astype(np.float32)
        nout = int(N * 0.7)
        Y[:nout] = r.uniform(0, 3, (nout, 3)).astype(np.float32)
        logit = np.concatenate([r.normal(-3, 1, nout), r.normal(3, 1, N - nout)]).astype(np.float32)
        w = O.dgr_inlier_weights(torch.from_numpy(logit))
        Xs.append(torch.from_numpy(X)), Ys.append(torch.from_numpy(Y)), ws.append(w)
        refs.append(O.weighted_procrustes(Xs[-1], Ys[-1], w[:, None], np.finfo(np.float32).eps))
    R, t = gmf_amd.weighted_procrustes_batched(_gpu(torch.cat(Xs)), _gpu(torch.cat(Ys)), _gpu(torch.cat(ws)),
                                               [i * N for i in range(B + 1)], np.finfo(np.float32).eps)
    for i in range(B):
        assert _maxerr(R[i].cpu(), refs[i][0]) < 1e-4 and _maxerr(t[i].cpu(), refs[i][1]) < 1e-4


def test_forward_from_raw_images(model):
    """The reference's own input dict: p_image / q_image [B,3,120,160] through the ResNet-34 -> layer2 encoder (native HIP
    convolutions, both images of a pair in one batch), then the HIP path; tokens computed one image at a time and fed directly
    give the same logits bit for bit (resnet.py:195-216, PointDSC.py:129-137)."""
    b = synthetic.synthetic_batch([91], N=200, T=300)
    g = torch.Generator().manual_seed(5)
    p_img, q_img = torch.rand(1, 3, 120, 160, generator=g), torch.rand(1, 3, 120, 160, generator=g)
    data = {"corr_pos": _gpu(b["corr_pos"]), "src_keypts": _gpu(b["src_keypts"]), "tgt_keypts": _gpu(b["tgt_keypts"]),
            "p_image": _gpu(p_img), "q_image": _gpu(q_img), "testing": True}
    res = model(data)
    lg = model.last_logits.clone()
    with torch.no_grad():
        pt, qt = model.encoder.image_tokens(_gpu(p_img)), model.encoder.image_tokens(_gpu(q_img))
    assert pt.shape == (1, 300, 128)
    d2 = {k: data[k] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    d2.update(p_tokens=pt, q_tokens=qt, testing=True)
    model(d2)
    # [r5] every convolution of the eval path is a native kernel at every batch size (k_conv_small_h2 for a few images): an output
    # pixel's accumulation order does not depend on how many images share the launch, so the two routes - two images at once
    # against one at a time - give the SAME bits (until round 4 small batches ran MIOpen's kernels, which are not repeatable
    # across batch sizes: the bound was 1e-4)
    assert torch.equal(model.last_logits, lg)
    assert res["final_trans"].shape == (1, 4, 4)


def _image_model(seed):
    """gmf_amd.PointDSC whose image encoder carries the seeded ResNet weights of golden F11 / F15."""
    m = gmf_amd.PointDSC(in_dim=6, num_layers=1, num_channels=128)
    enc = m.encoder.image_encoder
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(synthetic.seeded_state_dict(shapes, seed=seed, gain=1.0))
    return m.to(DEV).eval()


@pytest.mark.parametrize("graph", [True, False])
def test_f11_image_tokens_small_batch(golden_dir, graph):
    """Golden F11 on the GPU: 2 images through NonLocalNet.image_tokens (below the native-convolution threshold: folded
    BatchNorms, NHWC, MIOpen convolutions + the HIP bias/ReLU pass), captured graph and eager."""
    g = _load(golden_dir, "f11_image_encoder.npz")
    m = _image_model(int(g["seed"]))
    m.encoder.graph_image_encoder = graph
    r = np.random.default_rng([111])
    img = torch.from_numpy(r.uniform(0, 1, (2, 3, 120, 160)).astype(np.float32))
    with torch.no_grad():
        tok = m.encoder.image_tokens(_gpu(img))
        tok2 = m.encoder.image_tokens(_gpu(img))          # second call: graph replay
    assert tok.shape == (2, 300, 128)
    scale = max(1.0, float(np.abs(g["tokens"]).max()))
    assert _maxerr(tok.cpu(), g["tokens"]) < 1e-4 * scale
    assert torch.equal(tok, tok2)                            # [r5] native kernels at this size too: repeatable bit for bit
    # [r4] the module on its own (`gmf_amd.ImageEncoder.forward`, resnet.py:195-216) takes the same fused path in eval mode - it
    # used to run the stock torch modules there; [B, 128, H', W'] as the reference returns it
    with torch.no_grad():
        fmap = m.encoder.image_encoder(_gpu(img))
    assert fmap.shape == (2, 128, 15, 20)
    assert _maxerr(fmap.flatten(2).permute(0, 2, 1).cpu(), g["tokens"]) < 1e-4 * scale
    # eval() with autograd on and trainable parameters (fine-tuning with frozen BatchNorm statistics): the differentiable stock
    # modules, not the graph-replayed pass whose output has no grad_fn
    fmap_g = m.encoder.image_encoder(_gpu(img))
    assert fmap_g.grad_fn is not None and _maxerr(fmap_g.detach().cpu(), fmap.cpu()) < 1e-4 * scale


@pytest.mark.parametrize("tag,graph,patch", [("64x120x160", True, 1), ("64x120x160", False, 1), ("64x120x160", False, 2),
                                             ("64x120x160", False, 0), ("32x96x128", True, 1)])
def test_f15_image_tokens_at_batch_size(golden_dir, tag, graph, patch):
    """Golden F15 (the reference's ImageEncoder on 64 images of 120 x 160 and 32 of 96 x 128) against the fused HIP encoder:
    at 64 images every layer1 / layer2 convolution runs on gmf_conv_nhwc - the 64 -> 64 shape in its three-workgroups-per-CU
    form (conv_lds_patch 1), its two-workgroup form (2) and the gather form (0) - at 32 smaller images layer1 is native
    (two-workgroup form) and layer2 stays below the threshold.  Sampled rows within 1e-4 of the largest token, per-image
    fp64 checksums within 1e-5, graph replay identical to the first call."""
    from gmf_amd import _lib
    g = _load(golden_dir, "f15_image_encoder_batch.npz")
    m = _image_model(int(g["seed"]))
    m.encoder.graph_image_encoder = graph
    nimg, H, W = (int(v) for v in g[f"shape_{tag}"])
    img = _gpu(synthetic.seeded_images(nimg, H, W))
    h = _lib.handle_for(0)
    try:
        h.call("gmf_set_tuning", b"conv_lds_patch", patch)
        with torch.no_grad():
            tok = m.encoder.image_tokens(img)
            tok2 = m.encoder.image_tokens(img)
    finally:
        h.call("gmf_set_tuning", b"conv_lds_patch", 1)
    assert tok.shape == (nimg, (H // 8) * (W // 8), 128)
    scale = max(1.0, float(np.abs(g[f"rows_{tag}"]).max()))
    assert _maxerr(tok[::8, ::7].cpu(), g[f"rows_{tag}"]) < 1e-4 * scale
    assert np.abs(tok.double().sum((1, 2)).cpu().numpy() / g[f"sum_{tag}"] - 1).max() < 1e-5
    assert np.abs((tok.double() ** 2).sum((1, 2)).cpu().numpy() / g[f"sumsq_{tag}"] - 1).max() < 1e-5
    assert _maxerr(tok.cpu(), tok2.cpu()) < 1e-5 * scale


def test_image_encoder_accepts_nchw_strided_input(golden_dir):
    """The HIP passes read raw NHWC pointers: an NCHW-contiguous or strided input image must be converted, not misread."""
    g = _load(golden_dir, "f11_image_encoder.npz")
    m = _image_model(int(g["seed"]))
    r = np.random.default_rng([111])
    img = torch.from_numpy(r.uniform(0, 1, (2, 3, 120, 160)).astype(np.float32))
    big = torch.zeros(2, 3, 120, 200)
    big[..., :160] = img
    with torch.no_grad():
        tok = m.encoder.image_tokens(_gpu(big)[..., :160])
    assert _maxerr(tok.cpu(), g["tokens"]) < 1e-4 * max(1.0, float(np.abs(g["tokens"]).max()))


@pytest.mark.parametrize("d", [32, 33])
def test_f12_descriptor_matching(golden_dir, d):
    """Row f-2: fused distance GEMM + row argmin against the reference's matching on both plugin surfaces."""
    g = _load(golden_dir, "f12_descriptor_matching.npz")
    F0, F1 = _gpu(torch.from_numpy(g[f"F0_{d}"])), _gpu(torch.from_numpy(g[f"F1_{d}"]))
    idx, dis = gmf_amd.nn_match(F0, F1)
    assert (idx.cpu().numpy() == g[f"pdsc_idx_{d}"]).all()
    assert _maxerr(dis.cpu(), g[f"pdsc_dis_{d}"]) < 1e-4
    i1, d1 = gmf_amd.find_knn_gpu(F0, F1, nn_max_n=250, knn=1, return_distance=True)
    assert i1.shape == (F0.shape[0], 1) and (i1.cpu().numpy() == g[f"dgr_idx_chunk_{d}"]).all()
    assert _maxerr(d1.cpu(), g[f"dgr_dis_chunk_{d}"]) < 1e-5
    i2, d2 = gmf_amd.find_knn_gpu(F0, F1, nn_max_n=-1, knn=1, return_distance=True)
    assert (i2.cpu().numpy() == g[f"dgr_idx_{d}"]).all() and _maxerr(d2.cpu(), g[f"dgr_dis_{d}"]) < 1e-5


def test_descriptor_matching_full_size():
    """BASELINE config 5 shape: FCGF d = 32, 8000 x 8000; property: the match of a perturbed copy is its original."""
    r = np.random.default_rng(8000)
    F1 = r.normal(0, 1, (8000, 32)).astype(np.float32)
    F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    perm = r.permutation(8000)
    F0 = F1[perm] + 0.02 * r.normal(0, 1, (8000, 32)).astype(np.float32)
    F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
    idx, dis = gmf_amd.nn_match(_gpu(torch.from_numpy(F0.astype(np.float32))), _gpu(torch.from_numpy(F1)))
    assert (idx.cpu().numpy() == perm).all()
    assert float(dis.max()) < 0.3


def test_descriptor_matching_key_splits_and_edges():
    """[r4] `k_nn_match` splits the keys over workgroups and folds the winners with a 64-bit atomic minimum of (ordered score, index):
    against torch's own argmin of the reference formula (ThreeDMatch.py:164-166) over sizes that give 1 .. 16 key splits and every
    padded width; duplicated target rows (equal scores: the FIRST index wins, as torch.argmin's); a single source row; more source
    than target rows; a source row of NaN (no winner: index 0, distance NaN - and nothing out of range)."""
    g = torch.Generator().manual_seed(12)
    for N0, N1, d in ((1, 7, 32), (130, 64, 32), (777, 3000, 33), (3000, 500, 64), (2000, 9000, 128), (5000, 5000, 5)):
        F0 = torch.nn.functional.normalize(torch.randn(N0, d, generator=g), dim=1)
        F1 = torch.nn.functional.normalize(torch.randn(N1, d, generator=g), dim=1)
        idx, dis = gmf_amd.nn_match(_gpu(F0), _gpu(F1))
        dist = torch.sqrt(2 - 2 * (F0.double() @ F1.double().T) + 1e-6)
        ref = dist.argmin(1)
        got = idx.cpu()
        # a different winner is only acceptable where fp32 cannot tell the two distances apart
        diff = got != ref
        if diff.any():
            rows = diff.nonzero()[:, 0]
            assert float((dist[rows, got[rows]] - dist[rows, ref[rows]]).abs().max()) < 2e-6, (N0, N1, d)
        assert float(diff.float().mean()) < 2e-3
        assert _maxerr(dis.cpu().double(), dist[torch.arange(N0), got]) < 1e-5
    F1 = torch.nn.functional.normalize(torch.randn(900, 32, generator=g), dim=1)
    F1d = torch.cat([F1, F1, F1])                                   # every target row three times: rows j, j + 900, j + 1800 tie exactly
    F0 = F1[torch.randperm(900, generator=g)[:300]].clone()
    idx, _ = gmf_amd.nn_match(_gpu(F0), _gpu(F1d))
    assert int(idx.max()) < 900                                     # the first of the three
    assert torch.equal(F1d[idx.cpu()], F0)
    F0n = torch.nn.functional.normalize(torch.randn(40, 32, generator=g), dim=1)
    F0n[17] = float("nan")
    idx, dis = gmf_amd.nn_match(_gpu(F0n), _gpu(F1))
    assert int(idx[17]) == 0 and bool(torch.isnan(dis[17])) and int(idx.min()) >= 0 and int(idx.max()) < 900
    assert bool(torch.isfinite(dis[torch.arange(40) != 17]).all())


# ---- row f-3: DGR GlobalRegistration as one persistent HIP kernel --------------------------------------------------
def _f13_cases(g):
    return [(int(c[0]), int(c[1]), float(c[2]), float(c[3]), bool(c[4])) for c in g["cases"]]


# The reference's optimisation is not a continuous function of its inputs: the stopping counter compares fp32 losses of
# consecutive steps (registration.py:181), and the loss derivative jumps from 1/2 to 1/4 where a residual crosses d^2 = 1
# (core/loss.py:52-56).  Two correct fp32 evaluations that differ in the last bit of one sum can therefore stop a step
# apart or part ways late in the run - the reference's own answer depends on torch's vector width for `sum`.  Parity is
# asserted in two parts: the optimisation TRAJECTORY (stopping rule disabled) against the oracle at 1e-5, and the
# naturally stopped result against the reference's own outputs at a tolerance that covers a late split, with the cases
# whose stopping iteration is not on such a boundary held to 1e-5.
@pytest.mark.parametrize("case", range(6))
def test_f13_trajectory(golden_dir, case):
    """25 Adam steps with the stopping rule disabled: HIP kernel == oracle (== reference, test_oracle_golden) to 1e-5."""
    g = _load(golden_dir, "f13_global_registration.npz")
    N, seed, ratio, q, use_w = _f13_cases(g)[case]
    X, Y, w, _, _ = synthetic.dgr_scene(N, seed)
    Ro, to, oo = O.global_registration(X, Y, w if use_w else None, max_iter=25, break_threshold_ratio=0.0,
                                       quantization_size=q)
    R, t, o = gmf_amd.GlobalRegistration(_gpu(X), _gpu(Y), weights=_gpu(w) if use_w else None, max_iter=25,
                                         break_threshold_ratio=0.0, quantization_size=q)
    assert o["iterations"] == oo["iterations"] == 24 and o["break_count"] == 0
    assert _maxerr(R.cpu(), Ro) < 1e-5 and _maxerr(t.cpu(), to) < 1e-5
    assert abs(o["loss"] - oo["loss"]) < 1e-5 * oo["loss"]


@pytest.mark.parametrize("case", range(6))
def test_f13_global_registration(golden_dir, case):
    """gmf_amd.GlobalRegistration, naturally stopped, against the reference's own outputs (golden F13)."""
    g = _load(golden_dir, "f13_global_registration.npz")
    N, seed, ratio, q, use_w = _f13_cases(g)[case]
    X, Y, w, Rg, tg = synthetic.dgr_scene(N, seed)
    R, t, o = gmf_amd.GlobalRegistration(_gpu(X), _gpu(Y), weights=_gpu(w) if use_w else None,
                                         break_threshold_ratio=ratio, quantization_size=q)
    tag = f"{N}_{seed}"
    it_ref, loss_ref = int(g[f"stats_{tag}"][0]), float(g[f"stats_{tag}"][1])
    assert o["break_count"] == 20 and abs(float(torch.det(R.cpu().double())) - 1) < 1e-5
    assert abs(o["loss"] - loss_ref) < 1e-4 * loss_ref
    assert _maxerr(R.cpu(), g[f"R_{tag}"]) < 3e-3 and _maxerr(t.cpu(), g[f"t_{tag}"]) < 3e-3
    if case in (0, 3, 4):          # stopping iteration well away from a rounding boundary
        assert o["iterations"] == it_ref
        assert _maxerr(R.cpu(), g[f"R_{tag}"]) < 1e-5 and _maxerr(t.cpu(), g[f"t_{tag}"]) < 1e-5
    # never worse than the reference's own answer against the ground truth (by more than the split tolerance)
    assert _maxerr(R.cpu(), Rg) < _maxerr(g[f"R_{tag}"], Rg) + 3e-3


def test_global_registration_batched_config5():
    """BASELINE config 5 shape: 32 DGR problems of 8000 correspondences refined in ONE launch; each equals its own
    single-problem call bit for bit, recovers the ground-truth pose better than the Procrustes initialisation, and the
    oracle agrees on a sample of them."""
    B, N = 32, 8000
    scenes = [synthetic.dgr_scene(N, 100 + i) for i in range(B)]
    X = _gpu(torch.cat([s[0] for s in scenes])); Y = _gpu(torch.cat([s[1] for s in scenes]))
    w = _gpu(torch.cat([s[2] for s in scenes]))
    R, t, stats = gmf_amd.global_registration_batched(X, Y, w, [i * N for i in range(B + 1)], break_threshold_ratio=1e-4,
                                                      quantization_size=0.1)
    R0, t0 = gmf_amd.weighted_procrustes_batched(X, Y, w, [i * N for i in range(B + 1)], np.finfo(np.float32).eps)
    better = 0
    for i in range(B):
        Ri, ti, oi = gmf_amd.GlobalRegistration(_gpu(scenes[i][0]), _gpu(scenes[i][1]), weights=_gpu(scenes[i][2]),
                                                break_threshold_ratio=1e-4, quantization_size=0.1)
        assert torch.equal(Ri, R[i]) and torch.equal(ti, t[i]) and oi["iterations"] == int(stats[i, 0])
        better += _maxerr(R[i].cpu(), scenes[i][3]) < _maxerr(R0[i].cpu(), scenes[i][3])
        assert _maxerr(R[i].cpu(), scenes[i][3]) < 1e-2 and _maxerr(t[i].cpu(), scenes[i][4]) < 5e-2
    assert better >= B - 2
    for i in (0, 17):
        Ro, to, oo = O.global_registration(*scenes[i][:3], break_threshold_ratio=1e-4, quantization_size=0.1)
        assert _maxerr(R[i].cpu(), Ro) < 3e-3 and _maxerr(t[i].cpu(), to) < 3e-3      # see the note above test_f13_trajectory
        assert abs(float(stats[i, 1]) - oo["loss"]) < 1e-4 * oo["loss"]


def test_global_registration_edge_cases():
    """numpy inputs (registration.py:145-149), a ragged batch, max_iter = 0 (returns the initialisation) and bad arguments."""
    X, Y, w, Rg, tg = synthetic.dgr_scene(300, 7)
    R, t, o = gmf_amd.GlobalRegistration(X.numpy(), Y.numpy(), weights=w, break_threshold_ratio=1e-4, quantization_size=0.1)
    R2, t2, o2 = gmf_amd.GlobalRegistration(_gpu(X), _gpu(Y), weights=_gpu(w), break_threshold_ratio=1e-4, quantization_size=0.1)
    assert torch.equal(R, R2) and o == o2
    R0, t0, o0 = gmf_amd.GlobalRegistration(_gpu(X), _gpu(Y), weights=_gpu(w), max_iter=0, quantization_size=0.1)
    Rp, tp = gmf_amd.weighted_procrustes(_gpu(X), _gpu(Y), _gpu(w), np.finfo(np.float32).eps)
    assert _maxerr(R0.cpu(), Rp.cpu()) < 1e-6 and _maxerr(t0.cpu(), tp.cpu()) < 1e-6
    Xa, Ya, wa, _, _ = synthetic.dgr_scene(50, 8)
    Rb, tb, sb = gmf_amd.global_registration_batched(_gpu(torch.cat([X, Xa])), _gpu(torch.cat([Y, Ya])),
                                                     _gpu(torch.cat([w, wa])), [0, 300, 350], break_threshold_ratio=1e-4,
                                                     quantization_size=0.1)
    assert torch.equal(Rb[0], R2)
    with pytest.raises(RuntimeError):
        gmf_amd.global_registration_batched(_gpu(X), _gpu(Y), _gpu(w), [0, 100], quantization_size=0.1)
    with pytest.raises(RuntimeError):
        gmf_amd.GlobalRegistration(_gpu(X), _gpu(Y), weights=_gpu(w), quantization_size=0.0)
    with pytest.raises(NotImplementedError):
        gmf_amd.GlobalRegistration(_gpu(X), _gpu(Y), loss_fn=lambda a, b: 0)


# ---- every selectable form of the attention kernel gives the same logits ----------------------------------------------
@pytest.mark.parametrize("variant,cache", [(0, 0), (9, 0), (9, 1), (18, 1), (18, 0)])
def test_attention_kernel_variants(golden_dir, model, variant, cache):
    """fp32 MFMA on fp32 images (0), split-fp16 without / with the compat cache (9: the cache-less fallback), the cached,
    software-pipelined default (18) and the default with the cache switched off (falls back to the cache-less kernel): each
    within 1e-4 of the reference's golden logits (F4, N = 257: ragged last tile, 9 tiles -> pipelined loop, peeled tail and
    the padding waves all run)."""
    from gmf_amd import _lib
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    h = _lib.handle_for(0)
    try:
        h.call("gmf_set_tuning", b"scattn_variant", variant)
        h.call("gmf_set_tuning", b"compat_cache", cache)
        b = synthetic.synthetic_batch(list(g["pair_seeds_N257"]), N=257, T=196)
        data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
        logits = model.encode(*[data[k] for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")])[0]
        assert _maxerr(logits.cpu(), g["logits_N257"]) < 1e-4
    finally:
        h.call("gmf_set_tuning", b"scattn_variant", 18)
        h.call("gmf_set_tuning", b"compat_cache", 1)


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("B,N,T", [(32, 1000, 196), (2, 257, 196), (40, 700, 40)])
def test_two_launch_layers_match_oracle(sd_full, model, fused, B, N, T):
    """The default layer is two launches - k_linear_h2 (Q'/K/V + LCPE + cross-attention + GEGLU feed-forward in one pass, x'
    and x1 in registers) and the attention kernel whose epilogue applies the NEXT layer's PointCN; `fused_linear` = 0 is the
    four-launch sequence.  32 x 1000 and 40 x 700 (ragged last tile, padding waves) run the one-kernel form (>= 256
    workgroups), 2 x 257 the small-grid forms behind the same epilogue.  Logits within 1e-4 of the oracle for every pair."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch([7000 + i for i in range(B)], N=N, T=T)
    args = [_gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
    try:
        h.call("gmf_set_tuning", b"fused_linear", fused)
        logits, feat_n, _ = model.encode(*args)
    finally:
        h.call("gmf_set_tuning", b"fused_linear", 1)
    check = range(B) if B <= 4 else (0, B // 2, B - 1)
    for p in check:
        one = {k: v[p:p + 1] for k, v in b.items()}
        ref = O.pointdsc_forward(sd_full, one, testing=True)
        assert _maxerr(logits[p:p + 1].cpu(), ref["logits"]) < 1e-4
        fn = torch.nn.functional.normalize(ref["corr_features"], p=2, dim=-1)
        assert _maxerr(feat_n[p:p + 1].cpu(), fn) < 1e-4


@pytest.mark.parametrize("roles", [1, 0])
@pytest.mark.parametrize("N", [64, 257, 1000])
def test_small_grid_three_launch_layers(golden_dir, model, roles, N):
    """Small grids (the reference's B = 1 evaluation mode): a layer is three launches whose workgroups play different roles -
    {Q' | K | V | LCPE + cross-attention}, {key-split attention | hidden-split feed-forward}, merge (attention partials +
    feed-forward partials + fc_message + next PointCN) - or, with `small_grid_roles` = 0, one kernel per stage.  Both give
    the reference's golden logits and pose (F4 / F10; N = 64 is too small to split: it takes the per-stage path either way)."""
    from gmf_amd import _lib
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch(list(g[f"pair_seeds_N{N}"]), N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    try:
        h.call("gmf_set_tuning", b"small_grid_roles", roles)
        res = model(data)
    finally:
        h.call("gmf_set_tuning", b"small_grid_roles", 1)
    assert _maxerr(model.last_logits.cpu(), g[f"logits_N{N}"]) < 1e-4
    assert _maxerr(res["final_trans"].cpu(), g[f"final_trans_N{N}"]) < 1e-4


@pytest.mark.parametrize("B,N", [(1, 5000), (1, 1000), (2, 1531), (3, 257)])
def test_small_grid_merge_forms_are_bit_identical(model, B, N):
    """The merge step of the small-grid layer: one workgroup per query tile whose four waves split the feature blocks and run
    the blocks of a level in parallel (default) against one workgroup per four tiles - same arithmetic per element, same order."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch(list(range(40, 40 + B)), N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    out = {}
    try:
        for form in (1, 0):
            h.call("gmf_set_tuning", b"small_merge_tile", form)
            res = model(data)
            out[form] = (model.last_logits.clone(), res["final_trans"].clone())
    finally:
        h.call("gmf_set_tuning", b"small_merge_tile", 1)
    assert torch.isfinite(out[1][0]).all()
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1])


@pytest.mark.parametrize("B,N,T", [(1, 1000, 300), (1, 5000, 196), (2, 777, 33), (4, 3000, 300), (3, 200, 1), (1, 64, 700), (16, 1000, 300)])
def test_small_grid_prologue_roles_are_bit_identical(model, B, N, T):
    """[r5] Small grids: the forward's prologue - image side (Fusion-1 context, cross-attention, feed-forward; fusion_layer.py:172-201)
    and point side (key points, compat cache PointDSC.py:216-221, layer 0 + first PointCN :88,104-109) - as three launches that
    carry one link of each chain as two workgroup roles (k_pro_*), against the six kernels: the roles call the kernels' own bodies,
    so logits and poses are bit-identical, for a uniform batch and for the same pairs as one ragged call."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    keys = ("corr_pos", "src_keypts", "tgt_keypts")
    b = synthetic.synthetic_batch(list(range(70, 70 + B)), N=N, T=T)
    data = {k: _gpu(b[k]) for k in keys + ("p_tokens", "q_tokens")}
    data["testing"] = True
    lens = [N - 5 * i for i in range(B)]
    rag = {k: [data[k][i, :lens[i]] for i in range(B)] for k in keys}
    rag.update(p_tokens=data["p_tokens"], q_tokens=data["q_tokens"], testing=True)
    out = {}
    try:
        for form in (1, 0):
            h.call("gmf_set_tuning", b"small_prologue_roles", form)
            res = model(data)
            u = (model.last_logits.clone(), res["final_trans"].clone())
            rr = model(rag)
            out[form] = u + (model.last_logits.clone(), rr["final_trans"].clone())
    finally:
        h.call("gmf_set_tuning", b"small_prologue_roles", 1)
    assert torch.isfinite(out[1][0]).all() and torch.isfinite(out[1][2]).all()
    for a, c in zip(out[1], out[0]):
        assert torch.equal(a, c)


def test_random_shapes_against_the_oracle(model, sd_full):
    """Shape fuzz (the standing version of tests/tools/shape_fuzz.py): every grid plan of the encoder - one kernel per stage, mixed-role
    small grids, per-tile roles, key / hidden splits, prologue roles, the two-launch form - is picked by (B, N, T), and a plan that
    is wrong for some size only shows at that size.  Ten seeded random shapes with odd N and T, uniform and as one ragged call with
    per-pair lengths, against the CPU oracle (PointDSC.py:125-181) under the floor-relative contract of the fixed-size tests."""
    rng = np.random.default_rng(17)
    keys = ("corr_pos", "src_keypts", "tgt_keypts")
    for case in range(10):
        B = int(rng.integers(1, 7))
        N = int(rng.choice([int(rng.integers(64, 400)), int(rng.integers(400, 1400)), int(rng.integers(1400, 2300))]))
        T = int(rng.choice([1, 7, 31, 33, 100, 196, 257, 300]))
        if B * N > 6000:
            B = max(1, 6000 // N)
        b = synthetic.synthetic_batch([4000 + 10 * case + i for i in range(B)], N=N, T=T)
        ref = O.pointdsc_forward(sd_full, b, testing=True)
        data = {k: _gpu(b[k]) for k in keys + ("p_tokens", "q_tokens")}
        data["testing"] = True
        model(data)
        e = _maxerr(model.last_logits.cpu(), ref["logits"])
        lens = [N - 3 * i for i in range(B)]
        er = 0.0
        if B > 1 and min(lens) >= 64:
            rag = {k: [data[k][i, :lens[i]] for i in range(B)] for k in keys}
            rag.update(p_tokens=data["p_tokens"], q_tokens=data["q_tokens"], testing=True)
            lg = model(rag)["logits"]
            for i in range(B):
                bi = {k: (b[k][i:i + 1, :lens[i]] if k in keys + ("gt_labels",) else b[k][i:i + 1]) for k in b if torch.is_tensor(b[k])}
                er = max(er, _maxerr(lg[i].cpu(), O.pointdsc_forward(sd_full, bi, testing=True)["logits"][0]))
        print(f"fuzz case {case}: B={B} N={N} T={T}: uniform {e:.2e}, ragged {er:.2e}")
        assert e < 2.5e-4 and er < 2.5e-4, (B, N, T, e, er)       # (a wrong plan is off by 1e-2 and more; the parity gates proper are the fixed-size tests)


def test_alternate_forms_behind_knobs_agree(model):
    """The forms that are reachable on their own grids AND through a knob, each against the default on a grid where the default is
    the other form: same products, another order of partial sums - logits to fp32 rounding, poses to 1e-5.  `attn_tail_split` (the last
    partial round of a large attention grid split by keys, k_scattn_merge), `small_fattn_tile` = 0 (the cross-attention role per four
    query tiles, fusion_layer.py:84-94), `conv_small_grid` = 0 (the 128-pixel convolution kernels on a two-image grid, resnet.py:195-216)."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    # (18 x 3970: 576 attention items on 512 workgroup slots - a partial last round, which is what the tail split splits)
    cases = [("attn_tail_split", 1, 0, 18, 3970), ("small_fattn_tile", 0, 1, 1, 1500)]
    for knob, alt, default, B, N in cases:
        b = synthetic.synthetic_batch(list(range(900, 900 + B)), N=N, T=300)
        data = {k: _gpu(b[k]) for k in keys}
        data["testing"] = True
        out = {}
        try:
            for v in (default, alt):
                h.call("gmf_set_tuning", knob.encode(), v)
                r = model(data)
                out[v] = (model.last_logits.clone(), r["final_trans"].clone())
        finally:
            h.call("gmf_set_tuning", knob.encode(), default)
        dl, dT = _maxerr(out[alt][0].cpu(), out[default][0].cpu()), _maxerr(out[alt][1].cpu(), out[default][1].cpu())
        print(f"{knob} = {alt} against {default} at {B} x {N}: logits {dl:.2e}, poses {dT:.2e}")
        assert torch.isfinite(out[alt][0]).all() and dl < 3e-5 and dT < 1e-5, (knob, dl, dT)
    # the image encoder on two images, eager (a captured graph would replay whatever form it was captured with)
    m = _image_model(5)
    m.encoder.graph_image_encoder = False
    img = _gpu(synthetic.seeded_images(2, 120, 160))
    tok = {}
    try:
        for v in (1, 0):
            h.call("gmf_set_tuning", b"conv_small_grid", v)
            with torch.no_grad():
                tok[v] = m.encoder.image_tokens(img).clone()
    finally:
        h.call("gmf_set_tuning", b"conv_small_grid", 1)
    dt = _maxerr(tok[0].cpu(), tok[1].cpu())
    print(f"conv_small_grid = 0 against 1 on two images: tokens {dt:.2e} (largest {float(tok[1].abs().max()):.2f})")
    assert torch.isfinite(tok[0]).all() and dt < 1e-5 * max(1.0, float(tok[1].abs().max()))
    # the stem alone: its small-grid form (4 x 4 pooled pixels per workgroup) multiplies the same products in the same order per pixel
    # as the 8 x 8 form - bit-identical, odd image sizes included (resnet.py:198-204)
    fused = m.encoder._fused_image_encoder()
    for n, H, W in ((2, 120, 160), (1, 97, 131)):
        im = _gpu(synthetic.seeded_images(n, H, W))
        st = {}
        try:
            for v in (1, 0):
                h.call("gmf_set_tuning", b"conv_small_grid", v)
                st[v] = fused._stem(im).clone()
        finally:
            h.call("gmf_set_tuning", b"conv_small_grid", 1)
        assert torch.equal(st[0], st[1]) and torch.isfinite(st[1]).all()
    gmf_amd.check_status()


def test_launch_counts_of_a_forward(model):
    """Structure, not numbers: how many kernels one test-mode forward launches (torch profiler, device activities).  Small grids are
    launch-bound (B = 1 is the reference's evaluation mode, evaluation/test_3DMatch.py:69): three launches per layer and a prologue of
    five; large grids run two launches per layer.  A change that silently adds launches per layer shows here before it shows in a timing."""
    from torch.profiler import profile, ProfilerActivity
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    for B, N, per_layer, budget in ((1, 1000, 3, 52), (32, 1000, 2, 44)):
        b = synthetic.synthetic_batch(list(range(B)), N=N, T=196)
        data = {k: _gpu(b[k]) for k in keys}
        data["testing"] = True
        for _ in range(2):
            model(data)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            model(data)
            torch.cuda.synchronize()
        kern = [e for e in prof.key_averages() if e.device_time_total > 0 and ("gmf::" in e.key or "Memcpy" in e.key or "Memset" in e.key or "at::" in e.key)]
        total = sum(e.count for e in kern)
        layer = sum(e.count for e in kern if e.count % 12 == 0 and "gmf::" in e.key)
        print(f"B={B} N={N}: {total} device operations per forward, {layer} of them in per-layer kernels:", sorted((e.count, e.key[:40]) for e in kern if e.count >= 12))
        assert layer == 12 * per_layer, (B, N, layer)
        assert total <= budget, (B, N, total, sorted((e.count, e.key[:60]) for e in kern))


def test_handle_per_stream_overlaps_and_agrees(model):
    """[r5] `gmf_amd.set_handle_per_stream(True)` (serving): forwards on different torch streams take handles - and workspaces - of their
    own and may overlap on the device; the results are those of the default stream bit for bit, the default stream keeps the base
    handle, and switching it off returns every stream to the base handle."""
    from gmf_amd import _lib, _util
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    datas = []
    for i in range(3):
        b = synthetic.synthetic_batch([760 + i], N=700 + 100 * i, T=196)
        d = {k: _gpu(b[k]) for k in keys}
        d["testing"] = True
        datas.append(d)
    ref = []
    for d in datas:
        r = model(d)
        ref.append((model.last_logits.clone(), r["final_trans"].clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    base = _lib.handle_for(0)
    try:
        gmf_amd.set_handle_per_stream(True)
        hs = []
        for s in streams:
            with torch.cuda.stream(s):
                hs.append(_util.handle_and_stream(datas[0]["corr_pos"])[0])
        assert len({id(x) for x in hs}) == 3 and all(x is not base for x in hs)
        assert _util.handle_and_stream(datas[0]["corr_pos"])[0] is base                    # the default stream
        outs = [None] * 3
        for rep in range(4):                       # interleaved: three forwards in flight at a time
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    r = model(datas[i])
                    outs[i] = (model.last_logits.clone(), r["final_trans"].clone())
        torch.cuda.synchronize()
        for (lg, T), (lg0, T0) in zip(outs, ref):
            assert torch.equal(lg, lg0) and torch.equal(T, T0)
        gmf_amd.check_status()
    finally:
        gmf_amd.set_handle_per_stream(False)
    with torch.cuda.stream(streams[0]):
        assert _util.handle_and_stream(datas[0]["corr_pos"])[0] is base
    torch.cuda.synchronize()


def test_tuning_rejects_unknown_and_removed_settings():
    """The round-1 timing-only ablations (scattn_variant 11..15: wrong results) and the measured-and-rejected forms are no
    longer part of the library: gmf_set_tuning refuses them, out-of-range values and unknown knobs with GMF_ERR_BAD_ARG (-1)
    and leaves the handle's setting untouched."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    for v in (1, 3, 10, 11, 12, 13, 14, 15, 16, 17, 19, -1, 20):
        assert h.lib.gmf_set_tuning(h.h, b"scattn_variant", v) == -1
        assert b"scattn_variant" in h.lib.gmf_last_error_string(h.h)
    assert h.lib.gmf_set_tuning(h.h, b"no_such_knob", 1) == -1
    assert h.lib.gmf_set_tuning(h.h, b"compat_cache", 7) == -1
    assert h.lib.gmf_set_tuning(h.h, b"ff_hidden_splits", 3) == -1
    assert h.lib.gmf_set_tuning(h.h, b"scattn_variant", 18) == 0
    # [r4] gmf_get_tuning: what a caller that changes a knob for one call puts back afterwards (the Python module's numerics mode)
    import ctypes as C
    v = C.c_int(-5)
    for name, default in ((b"scattn_variant", 18), (b"pv_fp8", 1), (b"compat_format", 0), (b"precision", 0), (b"q_in_attention", 1),
                          (b"mid_grid_roles", 512), (b"topk_select", 1)):
        assert h.lib.gmf_get_tuning(h.h, name, C.byref(v)) == 0 and v.value == default, (name, v.value)
    assert h.lib.gmf_set_tuning(h.h, b"compat_format", 2) == 0
    assert h.lib.gmf_get_tuning(h.h, b"compat_format", C.byref(v)) == 0 and v.value == 2
    assert h.lib.gmf_set_tuning(h.h, b"compat_format", 0) == 0
    assert h.lib.gmf_get_tuning(h.h, b"no_such_knob", C.byref(v)) == -1
    assert h.lib.gmf_get_tuning(h.h, b"pv_fp8", None) == -1


def test_module_precision_restores_the_handles_setting(model):
    """PointDSC.set_precision is module-local - its forward applies the mode for the call and puts back what the HANDLE
    had (gmf_get_tuning), so a handle-level gmf_set_tuning("precision", ...) survives a forward of a module in another mode; a
    ragged batch refuses the throughput modes instead of silently running the parity numerics."""
    import ctypes as C
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch([5, 6], N=300, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    v = C.c_int(-1)
    try:
        h.call("gmf_set_tuning", b"precision", 2)              # a handle-level choice made through the C knob
        model.set_precision("throughput")                      # the module's own mode (level 1)
        model(data)
        h.call("gmf_get_tuning", b"precision", C.byref(v))
        assert v.value == 2
        with pytest.raises(RuntimeError, match="parity numerics only"):
            model({k: ([t[0], t[1]] if k in ("corr_pos", "src_keypts", "tgt_keypts") else t) for k, t in data.items()})
        model.set_precision("parity")
        model(data)
        h.call("gmf_get_tuning", b"precision", C.byref(v))
        assert v.value == 2
    finally:
        model.set_precision("parity")
        h.call("gmf_set_tuning", b"precision", 0)


def test_backward_entry_points_reject_bad_arguments():
    """The round-2 training entry points validate their arguments like the rest of the C ABI: null pointers, empty shapes, a
    test-mode parameter block (the post-refinement is not differentiable) and non-positive bandwidths return an error code and
    a message instead of launching."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    L = h.lib
    x = torch.zeros(2, 64, 128, device=DEV)
    pts = torch.zeros(2, 64, 3, device=DEV)
    knn = torch.zeros(2, 6, 40, device=DEV, dtype=torch.int32)
    fit = torch.zeros(2, 6, device=DEV)
    gT = torch.zeros(2, 4, 4, device=DEV)
    dF, ds = torch.zeros_like(x), torch.zeros(2, device=DEV)
    pp = _lib.PoseParams()
    pp.num_seeds, pp.k, pp.num_iterations, pp.use_nms, pp.refine_iters = 6, 40, 10, 0, 0
    pp.sigma, pp.sigma_d, pp.inlier_threshold, pp.nms_radius, pp.refine_threshold = 1.0, 0.1, 0.1, 0.1, 0.1
    args = [x.data_ptr(), pts.data_ptr(), pts.data_ptr(), knn.data_ptr(), fit.data_ptr(), gT.data_ptr(), 2, 64, dF.data_ptr(),
            ds.data_ptr(), None]
    assert L.gmf_pose_head_backward(h.h, pp, *args) == 0
    bad = list(args); bad[3] = None
    assert L.gmf_pose_head_backward(h.h, pp, *bad) == -1 and b"null" in L.gmf_last_error_string(h.h)
    pp.refine_iters = 20
    assert L.gmf_pose_head_backward(h.h, pp, *args) == -1 and b"not differentiable" in L.gmf_last_error_string(h.h)
    pp.refine_iters, pp.sigma = 0, 0.0
    assert L.gmf_pose_head_backward(h.h, pp, *args) != 0
    pp.sigma, pp.k = 1.0, 65
    assert L.gmf_pose_head_backward(h.h, pp, *args) != 0
    assert L.gmf_transformation_loss_backward(h.h, gT.data_ptr(), pts.data_ptr(), pts.data_ptr(), fit.data_ptr(), 0, 64,
                                              gT.data_ptr(), None) != 0
    assert L.gmf_transformation_loss_backward(h.h, None, pts.data_ptr(), pts.data_ptr(), fit.data_ptr(), 2, 64, gT.data_ptr(),
                                              None) == -1
    assert L.gmf_compat_dense(h.h, pts.data_ptr(), pts.data_ptr(), 2, 64, 0.0, x.data_ptr(), None) == -1
    assert L.gmf_weighted_procrustes_backward(h.h, pts.data_ptr(), pts.data_ptr(), fit.data_ptr(), None, 1, 1e-7, gT.data_ptr(),
                                              gT.data_ptr(), fit.data_ptr(), None) == -1
    # gmf_gemm_f32: a leading dimension smaller than the row it strides, overlapping batched outputs and a z-grid beyond 65535 are
    # refused on the host (they used to reach the device as out-of-bounds accesses or an opaque launch error)
    A, Bm, Cm = torch.zeros(64, 32, device=DEV), torch.zeros(32, 48, device=DEV), torch.zeros(64, 48, device=DEV)
    gemm = lambda **kw: L.gmf_gemm_f32(h.h, kw.get("ta", 0), kw.get("tb", 0), A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(), None, None,
                                       kw.get("M", 64), kw.get("N", 48), kw.get("K", 32), kw.get("lda", 32), kw.get("ldb", 48),
                                       kw.get("ldc", 48), kw.get("sa", 0), kw.get("sb", 0), kw.get("sc", 0), kw.get("batch", 1), 1.0, 0, None)
    assert gemm() == 0
    assert gemm(lda=31) == -1 and b"lda" in L.gmf_last_error_string(h.h)
    assert gemm(ldb=47) == -1 and gemm(ldc=47) == -1
    assert gemm(ta=1, lda=63) == -1                    # op(A) = A^T: rows of A are M long
    assert gemm(batch=2, sc=40) == -1                  # outputs of the two problems overlap
    assert gemm(batch=70000, sc=64 * 48) == -2 and b"65535" in L.gmf_last_error_string(h.h)
    torch.cuda.synchronize()


def test_mid_grid_linear_roles_are_bit_identical(model):
    """Grids of 256 .. 511 base workgroups run the fused linear kernel as two workgroup roles per row block (Q'/K/V |
    Fusion-2: the two independent halves of k_linear_h2's body, `mid_grid_roles`).  Same arithmetic in the same order: logits
    and poses bitwise equal to the single-kernel form."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    b = synthetic.synthetic_batch(list(range(200, 233)), N=1000, T=196)        # 33 pairs x 8 row blocks = 264 workgroups
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    try:
        h.call("gmf_set_tuning", b"mid_grid_roles", 0)
        r0 = model(data)
        lg0, T0 = model.last_logits.clone(), r0["final_trans"].clone()
        h.call("gmf_set_tuning", b"mid_grid_roles", 512)
        r1 = model(data)
        assert torch.equal(model.last_logits, lg0) and torch.equal(r1["final_trans"], T0)
        assert h.lib.gmf_set_tuning(h.h, b"mid_grid_roles", -1) == -1
    finally:
        h.call("gmf_set_tuning", b"mid_grid_roles", 512)
    ref = O.pointdsc_forward(synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 12, 128), seed=7),
                             synthetic.synthetic_batch([200], N=1000, T=196), testing=True)
    assert _maxerr(lg0[:1].cpu(), ref["logits"]) < 1e-4


def test_tuning_is_per_handle(golden_dir, model):
    """Tuning state lives in the handle: a second handle on the same device set to the fp32 path does not change what the
    first one runs (bitwise identical logits before and after)."""
    from gmf_amd import _lib
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    b = synthetic.synthetic_batch(list(g["pair_seeds_N257"]), N=257, T=196)
    args = [_gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
    before = model.encode(*args)[0].clone()
    other = _lib.Handle(0)
    other.call("gmf_set_tuning", b"scattn_variant", 0)
    other.call("gmf_set_tuning", b"compat_cache", 0)
    after = model.encode(*args)[0]
    assert torch.equal(before, after)
    del other


def test_weight_outside_fp16_range_falls_back_to_fp32_mfma(sd_full):
    """A BatchNorm-folded weight with |256 w| > 65504 cannot be split into fp16 planes (it would give inf / nan logits):
    packing detects it, warns once, and the encoder runs on the fp32-MFMA kernels - finite logits within 1e-4 of the oracle."""
    import warnings
    sd = {k: v.clone() for k, v in sd_full.items()}
    # fc_message of layer 3: BatchNorm 1 scaled by 4000 (gamma and beta) and the next convolution by 1 / 4000 - the same
    # function (ReLU is positively homogeneous), but the folded first weight is ~1e5 x 256 in the split-fp16 image
    p = "encoder.blocks.NonLocal_layer_3.fc_message."
    sd[p + "1.weight"] = sd[p + "1.weight"] * 4000.0
    sd[p + "1.bias"] = sd[p + "1.bias"] * 4000.0
    sd[p + "3.weight"] = sd[p + "3.weight"] / 4000.0
    m = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
                         sigma_d=0.10, k=40, nms_radius=0.10)
    m.load_state_dict(sd, strict=False)
    m = m.cuda().eval()
    b = synthetic.synthetic_batch([11], N=200, T=40)
    args = [_gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        logits = m.encode(*args)[0]
    assert any("fp16 range" in str(x.message) for x in w)
    assert not m._packed.split_fp16
    assert torch.isfinite(logits).all()
    ref = O.pointdsc_forward(sd, b, testing=True)
    assert _maxerr(logits.cpu(), ref["logits"]) < 1e-4


@pytest.mark.parametrize("N", [48, 96, 129, 160])
def test_ragged_tile_counts(model, sd_full, N):
    """Tile counts 2, 3, 5 (N > k + 1 = 41 as the reference's knn requires; all four residues mod 4 of the attention workgroup's wave count together with the other tests):
    1, 2 or 3 padding waves in a pair's last workgroup, an odd and an even number of pipelined tiles, a ragged last tile."""
    b = synthetic.synthetic_batch([300 + N, 301 + N], N=N, T=40)
    ref = O.pointdsc_forward(sd_full, b, testing=True)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = model(data)
    assert _maxerr(model.last_logits.cpu(), ref["logits"]) < 1e-4
    assert _maxerr(res["final_trans"].cpu(), ref["final_trans"]) < 1e-3


@pytest.mark.parametrize("on", [0, 1])
def test_front_output_split(golden_dir, model, on):
    """Small grids run k_front_h2 with one workgroup per output (Q' + f | K | V); on and off give the golden logits (F4)."""
    from gmf_amd import _lib
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    h = _lib.handle_for(0)
    try:
        h.call("gmf_set_tuning", b"front_output_split", on)
        b = synthetic.synthetic_batch(list(g["pair_seeds_N257"]), N=257, T=196)
        args = [_gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
        assert _maxerr(model.encode(*args)[0].cpu(), g["logits_N257"]) < 1e-4
    finally:
        h.call("gmf_set_tuning", b"front_output_split", 1)


@pytest.mark.parametrize("hs", [1, 0, 2, 4])
def test_hidden_split_feed_forward(golden_dir, model, hs):
    """Small grids divide the 16 GEGLU chunks of a row block over 2 / 4 / 8 workgroups (partials summed in a fixed order by
    k_ff_reduce): off (1), automatic (0 -> 8 at this size) and forced 2, 4 reproduce the reference's golden logits (F4)."""
    from gmf_amd import _lib
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    h = _lib.handle_for(0)
    try:
        h.call("gmf_set_tuning", b"ff_hidden_splits", hs)
        b = synthetic.synthetic_batch(list(g["pair_seeds_N1000"]), N=1000, T=196)
        args = [_gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
        logits = model.encode(*args)[0]
        assert _maxerr(logits.cpu(), g["logits_N1000"]) < 1e-4
    finally:
        h.call("gmf_set_tuning", b"ff_hidden_splits", 0)


@pytest.mark.parametrize("N", [257, 1000])
@pytest.mark.parametrize("splits", [1, 0, 3, 8])
def test_key_split_attention(golden_dir, model, N, splits):
    """Small grids divide the keys of a query block over several workgroups (k_scattn_h2p<.., KSPLIT> + k_scattn_merge):
    off (1), automatic (0) and forced (3, 8; capped at tiles / 4) all reproduce the reference's golden logits (F4)."""
    from gmf_amd import _lib
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    h = _lib.handle_for(0)
    try:
        h.call("gmf_set_tuning", b"attn_key_splits", splits)
        b = synthetic.synthetic_batch(list(g[f"pair_seeds_N{N}"]), N=N, T=196)
        args = [_gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")]
        logits = model.encode(*args)[0]
        assert _maxerr(logits.cpu(), g[f"logits_N{N}"]) < 1e-4
    finally:
        h.call("gmf_set_tuning", b"attn_key_splits", 0)


def _write_dump(path, tensors):
    """name -> array as the "GMFD" dump tests/abi_cpp/abi_host.cpp reads."""
    import struct
    with open(path, "wb") as f:
        f.write(b"GMFD" + struct.pack("<i", len(tensors)))
        for name, a in tensors.items():
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
            shp = list(a.shape) + [1] * (4 - a.ndim)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", a.ndim) + struct.pack("<4q", *shp))
            f.write(a.tobytes())


def test_cpp_host_of_the_c_abi(tmp_path, golden_dir, sd_full):
    """A C++ program (tests/abi_cpp/abi_host.cpp) drives libgmf_hip.so through include/gmf_hip.h with hipMalloc'd buffers -
    no Python, no torch in that process.  The hot path: the state_dict tensors by name -> gmf_encoder_pack_weights ->
    gmf_encoder_forward -> gmf_pose_head reproduce the reference's logits and pose (golden F4, N = 257) to 1e-4; then weighted
    Procrustes and the robust refinement recover known rigid motions, and a bad call returns a status code with a message."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_host")
    libdir = os.path.join(root, "gmf_amd")
    build = subprocess.run([hipcc, "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi_cpp", "abi_host.cpp"),
                            "-L", libdir, "-lgmf_hip", f"-Wl,-rpath,{libdir}", "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    g = _load(golden_dir, "f4_f10_pointdsc.npz")
    N = 257
    b = synthetic.synthetic_batch(list(g[f"pair_seeds_N{N}"]), N=N, T=196)
    dump = {k: v.numpy() for k, v in sd_full.items() if v.is_floating_point()}
    for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens"):
        dump["input." + k] = b[k].numpy()
    dump["expect.logits"], dump["expect.final_trans"] = g[f"logits_N{N}"], g[f"final_trans_N{N}"]
    dump["param.num_layers"] = np.array([12.0], np.float32)
    path = str(tmp_path / "f4.gmfd")
    _write_dump(path, dump)
    run = subprocess.run([exe, path], capture_output=True, text=True, timeout=120)
    print(run.stdout[-1500:])
    assert run.returncode == 0 and "ABI host OK" in run.stdout and "hot path OK" in run.stdout, run.stdout[-2000:] + run.stderr[-1000:]


@pytest.mark.parametrize("N", [1500, 3000])
def test_pose_parity_beyond_golden_sizes(sd_full, N):
    """Whole forward at N beyond the golden fixtures, on weights whose classifier bias is raised so that more than S local
    maxima have a positive score.  (With fewer, the reference's seed list runs into the zero-key tie group, whose order
    torch's unstable argsort leaves to its sorting algorithm - DESIGN.md section 2 - and no two implementations need agree.)
    Seeds identical, final transform within 1e-4 of the oracle."""
    sd = dict(sd_full)
    sd["classification.4.bias"] = sd_full["classification.4.bias"] + 4.0
    m = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                         inlier_threshold=0.10, sigma_d=0.10, k=40, nms_radius=0.10)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).eval()
    b = synthetic.synthetic_batch([4000 + N], N=N, T=196)
    ref = O.pointdsc_forward(sd, b, testing=True)
    src = b["src_keypts"]
    sdist = torch.norm(src[:, :, None, :] - src[:, None, :, :], dim=-1)
    is_max = torch.all((ref["logits"][:, :, None] >= ref["logits"][:, None, :]) | (sdist >= 0.10), dim=-1)
    assert int(((ref["logits"] > 0) & is_max).sum()) >= int(N * 0.1), "scene does not have S positive local maxima"
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = m(data)
    assert _maxerr(m.last_logits.cpu(), ref["logits"]) < 1e-4
    _, _, aux = m.pose_head(m.last_features, data["src_keypts"], data["tgt_keypts"], m.last_logits, True, return_aux=True)
    assert np.array_equal(aux["seeds"][0].cpu().numpy(), ref["seeds"][0].numpy())
    assert _maxerr(res["final_trans"].cpu(), ref["final_trans"]) < 1e-4


def test_sharded_driver_one_rank_rccl(model):
    """The multi-GPU step on the hardware there is: a ONE-rank RCCL process group, the real PointDSC, the packed
    all-gather (logits | pose in one buffer, one collective) - gathered logits and poses bitwise equal to the local ones.
    (The 2-rank logic, uneven shards included, is covered on gloo in tests/test_dist_gloo.py.)"""
    import torch.distributed as dist
    from gmf_amd.dist import ShardedBatchDriver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    b = synthetic.synthetic_batch([5, 6, 7], N=300, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    drv = ShardedBatchDriver(model, 1, 0, torch.device(DEV), always_collective=True)
    try:
        assert dist.is_initialized() and dist.get_backend() == "nccl"
        drv.time_steps = True
        out = drv.step(data)
        assert out["all_logits"].shape == (3, 300) and out["all_trans"].shape == (3, 4, 4)
        assert torch.equal(out["all_logits"], model.last_logits)
        assert torch.equal(out["all_trans"], out["final_trans"])
        model_ms, gather_ms = drv.read_timings()
        assert model_ms > 0 and gather_ms > 0
        out2 = drv.run(data)                      # global batch of 3 pairs over 1 rank
        assert torch.equal(out2["all_logits"], out["all_logits"])
    finally:
        drv.close()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _two_rank_worker(rank, world, port, q):
    """One of two processes sharing GPU 0: the real PointDSC on its shard of a 5-pair batch, gloo gather through the host."""
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch as _t
    import gmf_amd as _g
    from gmf_amd import synthetic as _s
    from gmf_amd.dist import ShardedBatchDriver
    dev = _t.device("cuda:0")
    m = _g.PointDSC(num_layers=12)
    m.load_state_dict(_s.seeded_state_dict(_s.pointdsc_shapes(6, 12, 128), seed=7), strict=False)
    m = m.to(dev).eval()
    b = _s.synthetic_batch([11, 12, 13, 14, 15], N=257, T=40)
    data = {k: b[k].to(dev) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    drv = ShardedBatchDriver(m, world, rank, dev, backend="gloo")
    out = drv.run(data)                                   # 3 + 2 pairs: padded inside the one packed buffer
    drv.barrier()
    t = drv.max_over_ranks(float(rank + 1))
    q.put((rank, out["all_logits"].cpu(), out["all_trans"].cpu(), out["logits"].shape[0], t))
    drv.close()


def test_two_process_sharded_forward_on_one_gpu(model):
    """The N > 1 step rehearsed on the one GPU there is: TWO processes, each with its own library handle and the real PointDSC
    on its shard (3 + 2 pairs of a 5-pair batch), exchanging the packed logits | poses with one collective (gloo through the
    host here - the RCCL form of the same call runs in test_sharded_driver_one_rank_rccl).  Every rank ends with all five
    pairs, equal to the single-process forward of the whole batch (5e-5: the shard sizes take different small-grid splits)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    b = synthetic.synthetic_batch([11, 12, 13, 14, 15], N=257, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    ref = model(data)
    ref_logits, ref_T = model.last_logits.cpu(), ref["final_trans"].cpu()
    sizes = {}
    for rank, all_logits, all_trans, local_b, t in got:
        sizes[rank] = local_b
        assert all_logits.shape == (5, 257) and all_trans.shape == (5, 4, 4)
        assert _maxerr(all_logits, ref_logits) < 5e-5
        assert _maxerr(all_trans, ref_T) < 1e-4
        assert t == 2.0
    assert sizes == {0: 3, 1: 2}
    assert torch.equal(got[0][1], got[1][1]) and torch.equal(got[0][2], got[1][2])      # both ranks hold the same gathered result


def _f16_cases(g):
    return [(int(n), int(s)) for n, s in g["cases"]]


@pytest.mark.parametrize("case", range(5))
def test_f16_pose_in_tie_scenes(golden_dir, model, case):
    """Golden F16: the reference's own test-mode forward at N = 1500 / 3000 with the UNMODIFIED seeded weights.  In four of
    the five scenes fewer than S local maxima have a positive score, so the reference's seed list
    `argsort(scores * is_local_max, descending)[:S]` (PointDSC.py:284-286) is filled from the tie group of zero keys in the
    order torch's unstable CPU sort happens to leave.  The contract of the HIP path there ("Seed ties"):
      * logits within 1e-4 of the reference's;
      * seeds = (key descending, index ascending) - exactly what a STABLE sort of the reference's keys gives;
      * the seeds with a positive key (where no tie is involved) are the reference's own, in its order;
      * the final pose has at least the reference's inlier count, equals the reference's to 3e-3 and is no worse against the
        ground truth by more than 5e-4 (see the note at the assertion)."""
    g = _load(golden_dir, "f16_pose_tie_scenes.npz")
    N, seed = _f16_cases(g)[case]
    tag = f"{N}_{seed}"
    b = synthetic.synthetic_batch([seed], N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    data["testing"] = True
    res = model(data)
    logits = model.last_logits
    assert _maxerr(logits.cpu(), g[f"logits_{tag}"]) < 1e-4
    _, _, aux = model.pose_head(model.last_features, data["src_keypts"], data["tgt_keypts"], logits, True, return_aux=True)
    seeds = aux["seeds"][0].cpu().numpy()
    # expected order from the HIP path's own logits: the reference's NMS keys, stable sort (descending key, ascending index)
    lg = logits.cpu()
    src = b["src_keypts"]
    sdist = torch.norm(src[:, :, None, :] - src[:, None, :, :], dim=-1)
    is_max = torch.all((lg[:, :, None] >= lg[:, None, :]) | (sdist >= 0.10), dim=-1).float()
    keys = (lg * is_max)[0]
    expect = torch.sort(keys, descending=True, stable=True)[1][: N // 10].numpy()
    assert np.array_equal(seeds, expect)
    # the untied part (positive keys) is the reference's own seed list
    ref_seeds = g[f"seeds_{tag}"][0]
    rl = torch.from_numpy(g[f"logits_{tag}"])
    ref_is_max = torch.all((rl[:, :, None] >= rl[:, None, :]) | (sdist >= 0.10), dim=-1).float()
    ref_keys = (rl * ref_is_max)[0]
    n_pos = int((ref_keys[torch.from_numpy(ref_seeds.astype(np.int64))] > 0).sum())
    assert n_pos == int((ref_keys > 0).sum()) or n_pos == N // 10
    assert np.array_equal(seeds[:n_pos], ref_seeds[:n_pos])
    if n_pos < N // 10:
        assert float(keys[seeds[n_pos]]) == 0.0 and np.all(np.diff(seeds[n_pos:][keys[seeds[n_pos:]].numpy() == 0]) > 0)
    # The two seed lists lead to different starting hypotheses for post_refinement, whose early exit ("inlier count unchanged",
    # PointDSC.py:516) then stops on slightly different IRLS iterates.  Measured on these scenes: identical inlier counts;
    # |T_hip - T_ref| 1.6e-3 (3000_84) and 5.1e-4 (3000_85); against the ground truth the HIP pose is 2.9e-4 worse in the
    # first and 3.9e-4 better in the second - both inside the registration's own noise (1 cm on the inliers).  Contract:
    # at least the reference's inlier count (the quantity the method maximises, PointDSC.py:413-425), the same pose to 3e-3,
    # and no worse against the ground truth than the reference by more than 5e-4.
    def inliers(T):
        T = torch.as_tensor(np.asarray(T), dtype=torch.float32)
        p = b["src_keypts"][0] @ T[0, :3, :3].T + T[0, :3, 3]
        return int(((p - b["tgt_keypts"][0]).norm(dim=-1) < 0.10).sum())
    T_hip, T_ref = res["final_trans"].cpu().numpy(), g[f"final_trans_{tag}"]
    assert inliers(T_hip) >= inliers(T_ref)
    assert _maxerr(T_hip, T_ref) < 3e-3
    err_hip, err_ref = _maxerr(T_hip, g[f"gt_trans_{tag}"]), _maxerr(T_ref, g[f"gt_trans_{tag}"])
    assert err_hip <= err_ref + 5e-4, (err_hip, err_ref)
    if n_pos >= N // 10:                       # no tie involved: the reference's pose itself
        assert _maxerr(res["final_trans"].cpu(), g[f"final_trans_{tag}"]) < 1e-4


# ---- validation step (row f-4, forward half): golden F14 from the reference's forward and libs/loss.py ---------------
_CLS = ("loss", "precision", "recall", "f1", "logit_true", "logit_false")


def _rel(a, b):
    a, b = np.asarray([float(v) for v in a], np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


@pytest.mark.parametrize("N", [96, 257])
def test_f14_validation_step(golden_dir, model, N):
    """eval-mode forward without 'testing' (libs/trainer.py:233) -> M, logits, pose; then the three metrics."""
    g = _load(golden_dir, "f14_validation_step.npz")
    b = synthetic.synthetic_batch(list(g[f"pair_seeds_N{N}"]), N=N, T=196)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    res = model(data)
    M, logits, T = res["M"], res["final_labels"], res["final_trans"]
    assert _maxerr(logits.cpu(), g[f"logits_N{N}"]) < 1e-4
    assert _maxerr(T.cpu(), g[f"final_trans_N{N}"]) < 1e-4
    assert M.shape == (len(b["corr_pos"]), N, N)
    if N <= 96:
        assert _maxerr(M.cpu(), g[f"M_N{N}"]) < 2e-5
    else:
        assert _maxerr(M[:, ::16].cpu(), g[f"M_rows_N{N}"]) < 2e-5
    assert _rel(M.double().sum((1, 2)).cpu(), g[f"M_sum_N{N}"]) < 1e-5
    assert _rel((M.double() ** 2).sum((1, 2)).cpu(), g[f"M_sumsq_N{N}"]) < 1e-5
    assert bool((torch.diagonal(M, dim1=1, dim2=2) == 0).all()) and float(M.min()) >= 0 and float(M.max()) <= 1
    gt = _gpu(b["gt_labels"])
    # metrics on the reference's own outputs: isolates each metric kernel
    rl, rT = _gpu(torch.from_numpy(g[f"logits_N{N}"])), _gpu(torch.from_numpy(g[f"final_trans_N{N}"]))
    cs = gmf_amd.ClassificationLoss(balanced=True)(rl, gt)
    assert np.abs(np.array([float(cs[k]) for k in _CLS]) - g[f"class_N{N}"]).max() < 1e-5
    cu = gmf_amd.ClassificationLoss(balanced=False)(rl, gt)
    assert abs(float(cu["loss"]) - g[f"class_unbalanced_N{N}"][0]) < 1e-5
    sm_b, sm_u = gmf_amd.SpectralMatchingLoss(balanced=True), gmf_amd.SpectralMatchingLoss(balanced=False)
    if N <= 96:
        Mr = _gpu(torch.from_numpy(g[f"M_N{N}"]))
        assert abs(float(sm_b(Mr, gt)) - g[f"sm_N{N}"][0]) < 1e-6
        assert abs(float(sm_u(Mr, gt)) - g[f"sm_N{N}"][1]) < 1e-6
    # ... and end to end on the HIP path's own M, both forms
    assert abs(float(sm_b(M, gt)) - g[f"sm_N{N}"][0]) < 1e-5
    assert abs(float(sm_u(M, gt)) - g[f"sm_N{N}"][1]) < 1e-5
    feat_n = model.last_features
    assert abs(float(sm_b.from_features(feat_n, model._weights(feat_n.device).sigma, gt)) - g[f"sm_N{N}"][0]) < 1e-5
    assert abs(float(sm_u.from_features(feat_n, model._weights(feat_n.device).sigma, gt)) - g[f"sm_N{N}"][1]) < 1e-5
    tl = gmf_amd.TransformationLoss(re_thre=15, te_thre=30)(rT, _gpu(b["gt_trans"]), data["src_keypts"], data["tgt_keypts"], rl)
    ref = g[f"trans_N{N}"]
    assert abs(float(tl[0]) - ref[0]) < 1e-5 * abs(ref[0]) and tl[1] == ref[1]
    assert abs(float(tl[2]) - ref[2]) < 2e-3          # RE: acos of an fp32 trace, ill-conditioned near 0 deg
    assert abs(float(tl[3]) - ref[3]) < 1e-4 * abs(ref[3]) and abs(float(tl[4]) - ref[4]) < 1e-5 * abs(ref[4])


def test_f14_metrics_corner_cases(golden_dir):
    """No predicted inlier in a pair (zero loss term), a pair without ground-truth inliers, recall below 100 %."""
    g = _load(golden_dir, "f14_validation_step.npz")
    pred, gt, M, T = (_gpu(torch.from_numpy(g[k])) for k in ("alone_pred", "alone_gt", "alone_M", "alone_T"))
    cs = gmf_amd.ClassificationLoss()(pred, gt)
    assert np.abs(np.array([float(cs[k]) for k in _CLS]) - g["alone_class"]).max() < 1e-5
    assert abs(float(gmf_amd.ClassificationLoss(balanced=False)(pred, gt)["loss"]) - g["alone_class_unbalanced"][0]) < 1e-5
    assert abs(float(gmf_amd.SpectralMatchingLoss()(M, gt)) - g["alone_sm"][0]) < 1e-6
    assert abs(float(gmf_amd.SpectralMatchingLoss(balanced=False)(M, gt)) - g["alone_sm"][1]) < 1e-6
    bb = synthetic.synthetic_batch(list(g["alone_seeds"]), N=pred.shape[1], T=12)
    tl = gmf_amd.TransformationLoss()(T, _gpu(bb["gt_trans"]), _gpu(bb["src_keypts"]), _gpu(bb["tgt_keypts"]), pred)
    ref = g["alone_trans"]
    assert _rel(tl, ref) < 1e-5
    # weighted form (loss.py:88-90) against the oracle
    w = torch.rand_like(pred)
    want = O.classification_loss(pred.cpu(), gt.cpu(), weight=w.cpu())["loss"]
    assert abs(float(gmf_amd.ClassificationLoss()(pred, gt, weight=w)["loss"]) - want) < 1e-5


@pytest.mark.parametrize("B,N", [(1, 31), (2, 1000), (3, 2049)])
def test_similarity_matrix_sizes(B, N):
    """Ragged sizes, several column chunks and row groups; against the oracle on the same unit features; the fused
    loss against the loss of the written matrix."""
    gen = torch.Generator().manual_seed(N)
    f = torch.nn.functional.normalize(torch.randn(B, N, 128, generator=gen), dim=-1)
    f[:, 1::2] = torch.nn.functional.normalize(f[:, 1::2] + 2.0 * f[:, 0::2][:, : f[:, 1::2].shape[1]], dim=-1)   # correlated rows: M not all zero
    gt = (torch.rand(B, N, generator=gen) < 0.3).float()
    for sigma in (1.0, 0.6):
        want = O.similarity_matrix(f, sigma)
        got = gmf_amd.similarity_matrix(_gpu(f), sigma)
        assert _maxerr(got.cpu(), want) < 2e-5
        assert float(want.max()) > 0.3
        dense = gmf_amd.similarity_matrix(_gpu(f), sigma, contiguous=True)     # ldm = N: same values, dense layout
        assert dense.is_contiguous() and got.stride(1) % 32 == 0 and torch.equal(dense, got)
        for balanced in (True, False):
            sm = gmf_amd.SpectralMatchingLoss(balanced=balanced)
            ref = O.spectral_matching_loss(want, gt, balanced=balanced)
            assert abs(float(sm(got, _gpu(gt))) - ref) < 1e-5 * max(1.0, abs(ref))
            assert abs(float(sm.from_features(_gpu(f), sigma, _gpu(gt))) - ref) < 1e-5 * max(1.0, abs(ref))


@pytest.mark.parametrize("tag", ["N96_bal", "N150_bal", "N150_mse"])
def test_f17_sm_loss_backward(golden_dir, tag):
    """Golden F17: the reference's autograd gradients of SpectralMatchingLoss(M(F.normalize(corr_features), sigma), gt) with
    respect to the encoder output and to sigma.  Here: torch's F.normalize on the device, then ONE HIP backward launch for
    the whole N x N x 128 part (gmf_spectral_matching_backward; M and dL/dM are never written).  Tolerance: 2e-5 of the
    largest gradient entry (split-fp16 products, fp32 accumulation), dsigma 1e-4 relative."""
    g = _load(golden_dir, "f17_sm_loss_backward.npz")
    N = int(tag[1:].split("_")[0])
    feat = _gpu(torch.from_numpy(g[f"corr_features_N{N}"])).requires_grad_(True)
    sigma = torch.tensor([float(g["sigma"])], device=DEV, requires_grad=True)
    gt = _gpu(synthetic.synthetic_batch(list(g[f"pair_seeds_{tag}"]), N=N, T=196)["gt_labels"])
    fn = torch.nn.functional.normalize(feat, p=2, dim=-1)
    loss = gmf_amd.SpectralMatchingLoss(balanced=tag.endswith("bal")).from_features(fn, sigma, gt)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[f"loss_{tag}"])) < 1e-5 * abs(float(g[f"loss_{tag}"]))
    ref = g[f"d_corr_features_{tag}"]
    assert _maxerr(feat.grad.cpu(), ref) < 2e-5 * np.abs(ref).max()
    assert abs(float(sigma.grad) - float(g[f"d_sigma_{tag}"][0])) < 1e-4 * abs(float(g[f"d_sigma_{tag}"][0]))


@pytest.mark.parametrize("tag", ["fl128", "pio256"])
def test_f18_fusion_layer_backward(golden_dir, tag):
    """Golden F18: the reference's autograd through one FusionLayer (pe = True, 128 / 128 / head 64 - the Fusion-2 form) and the
    DGR bottleneck PerceiverIO (256 / head 128).  gmf_amd's module in train() mode runs the HIP training primitives behind a
    torch.autograd.Function: output, d queries, d context and all 18 parameter gradients within 2e-5 of each tensor's largest
    entry (fp32 MFMA products; sums in another order than the reference's)."""
    g = _load(golden_dir, "f18_fusion_layer_backward.npz")
    B, N, T, lat, dh = (int(v) for v in g[f"{tag}_dims"])
    cls = gmf_amd.PerceiverIO if tag == "pio256" else gmf_amd.FusionLayer
    m = cls(depth=0, dim=128, latent_dim=lat, cross_heads=1, latent_heads=8, cross_dim_head=dh, latent_dim_head=dh, pe=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synthetic.seeded_state_dict(shapes, seed=int(g["seed"])))
    m = m.to(DEV).train()
    r = np.random.default_rng([118, N, T])
    x = _gpu(torch.from_numpy(r.normal(0, 1, (B, N, lat)).astype(np.float32))).requires_grad_(True)
    ctx = _gpu(torch.from_numpy(r.normal(0, 1, (B, T, 128)).astype(np.float32))).requires_grad_(True)
    up = _gpu(torch.from_numpy(r.normal(0, 1, (B, N, lat)).astype(np.float32)))
    y = m(ctx, queries_encoder=x)
    y.backward(up)

    def close(a, b, what):
        err, ref = _maxerr(a.detach().cpu(), b), float(np.abs(b).max())
        assert err < 2e-5 * max(ref, 1e-12), (what, err, ref)
    close(y, g[f"{tag}_out"], "out")
    close(x.grad, g[f"{tag}_dx"], "dx")
    close(ctx.grad, g[f"{tag}_dctx"], "dctx")
    checked = 0
    for name, p in m.named_parameters():
        if f"{tag}_grad::{name}" in g.files:
            close(p.grad, g[f"{tag}_grad::{name}"], name)
        else:
            close(p.grad[::8], g[f"{tag}_gradrows::{name}"], name)
            s = g[f"{tag}_gradsum::{name}"]
            assert abs(float(p.grad.double().sum()) - s[0]) < 1e-4 * np.sqrt(s[1]), name
        checked += 1
    assert checked == 18
    # eval() mode keeps the fused inference kernels and agrees with the training forward
    with torch.no_grad():
        ye = m.eval()(ctx, queries_encoder=x)
    assert _maxerr(ye.cpu(), y.detach().cpu()) < 1e-4


@pytest.mark.parametrize("tag", ["def", "bal"])
def test_f19_training_step(golden_dir, tag):
    """Golden F19: one training step of the reference in its default configuration (libs/trainer.py:121-166: train() mode with
    BatchNorm batch statistics, non-test forward, ClassificationLoss + SpectralMatchingLoss, loss.backward(); the MSE / plain-BCE
    forms of config_3DMatch.py:49 and the balanced ones) on a 3-layer model (the 12-layer reference is chaotic in train mode
    with seeded weights - see gen_f19).  gmf_amd's PointDSC in train() mode runs HIP training primitives behind
    torch.autograd.Functions: logits within 3e-4 (fp32 noise of the reference itself: 8e-5), both losses to 1e-5 relative, M
    checksums, the BatchNorm running statistics, and the gradient of EVERY parameter (137 tensors): each within
    2e-4 of its own largest entry + 3e-6 of the largest gradient in the model (the biases in front of a BatchNorm have
    mathematically zero gradients: the reference's and ours are both rounding noise there)."""
    g = _load(golden_dir, "f19_training_step.npz")
    cfg = g[f"{tag}_cfg"]
    balanced, N, seeds = bool(cfg[0]), int(cfg[1]), [int(v) for v in cfg[2:]]
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 3, 128), seed=7)
    m = gmf_amd.PointDSC(in_dim=6, num_layers=3, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
                         sigma_d=0.10, k=40, nms_radius=0.10)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).train()
    b = synthetic.synthetic_batch(seeds, N=N, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    gt = _gpu(b["gt_labels"])
    res = m(data)
    cl = gmf_amd.ClassificationLoss(balanced=balanced)(res["final_labels"], gt)
    sm = gmf_amd.SpectralMatchingLoss(balanced=balanced)(res["M"], gt)
    loss = 1.0 * cl["loss"] + 1.0 * sm
    loss.backward()
    assert _maxerr(res["final_labels"].detach().cpu(), g[f"{tag}_logits"]) < 3e-4
    ref_losses = g[f"{tag}_losses"]
    assert abs(float(cl["loss"].detach()) - ref_losses[0]) < 1e-5 * ref_losses[0]
    assert abs(float(sm.detach()) - ref_losses[1]) < 1e-5 * ref_losses[1]
    M = res["M"].detach()
    assert np.abs(M.double().sum((1, 2)).cpu().numpy() / g[f"{tag}_M_sum"] - 1).max() < 1e-5
    assert _maxerr(M[:, ::25].cpu(), g[f"{tag}_M_rows"]) < 1e-4
    assert res["final_trans"].shape == (len(seeds), 4, 4) and res["final_trans"].requires_grad
    bn = m.encoder.blocks["NonLocal_layer_2"].fc_message[1]
    assert _maxerr(torch.stack([bn.running_mean, bn.running_var]).cpu(), g[f"{tag}_bn_running"]) < 1e-5
    pc = m.encoder.blocks["PointCN_layer_0"][1]
    assert _maxerr(torch.stack([pc.running_mean, pc.running_var]).cpu(), g[f"{tag}_pcn_running"]) < 1e-5
    names, stats, heads = list(g[f"{tag}_grad_names"]), g[f"{tag}_grad_stats"], g[f"{tag}_grad_heads"]
    params = dict(m.named_parameters())
    gmax = float(stats[:, 2].max())
    assert len(names) == 137
    for i, n in enumerate(names):
        gr = params[n].grad
        assert gr is not None, n
        gr = gr.double().reshape(-1).cpu()
        k = min(16, gr.numel())
        tol = 2e-4 * stats[i, 2] + 3e-6 * gmax
        assert np.abs(gr[:k].numpy() - heads[i, :k]).max() < tol, (n, stats[i])
        assert abs(float(gr.norm()) - stats[i, 1]) < 2e-4 * stats[i, 1] + 3e-6 * gmax * np.sqrt(gr.numel()), (n, stats[i])
    # an optimiser step on the module's own parameters works as with the reference (torch.optim over nn.Parameters)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    before = m.classification[0].weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m.classification[0].weight.detach())


@pytest.mark.parametrize("tag", ["N200", "N150"])
def test_f20_pose_head_backward(golden_dir, tag):
    """Golden F20, part A: the reference's autograd through its pose head in the non-test forward (top-S seeds, kNN, feature
    compatibility + power iteration + weighted SVD of every seed, the best hypothesis; PointDSC.py:246-252,304-425) and
    TransformationLoss, from the encoder output.  HIP: gmf_normalize_rows -> gmf_pose_head (forward) ->
    gmf_transformation_loss, then gmf_transformation_loss_backward -> gmf_pose_head_backward (closed-form derivative of the 3x3
    SVD, the power iterations in reverse) -> normalize backward: final_trans 1e-4, the five loss outputs, and the gradients
    with respect to the encoder output (the same k neighbour rows non-zero) and sigma within 2e-3 of their largest entry."""
    from gmf_amd import train as T_
    g = _load(golden_dir, "f20_pose_head_backward.npz")
    seeds = [int(v) for v in g[f"pair_seeds_{tag}"]]
    N = int(tag[1:])
    b = synthetic.synthetic_batch(seeds, N=N, T=196)
    m = gmf_amd.PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
                         sigma_d=0.10, k=40, nms_radius=0.10).to(DEV)
    with torch.no_grad():
        m.sigma.fill_(float(g["sigma"]))
    cf = _gpu(torch.from_numpy(g[f"corr_features_{tag}"])).requires_grad_(True)
    logits = _gpu(torch.from_numpy(g[f"logits_{tag}"]))
    src, tgt = _gpu(b["src_keypts"]), _gpu(b["tgt_keypts"])
    B = len(seeds)
    feat_n = T_.normalize_rows(cf.reshape(B * N, -1)).reshape(B, N, -1)
    final_T = T_.pose_head_train(m, feat_n, m.sigma, src, tgt, logits, (float(g["sigma"]), 0.10))
    tl = gmf_amd.TransformationLoss(re_thre=15, te_thre=30)(final_T, _gpu(b["gt_trans"]), src, tgt, logits)
    tl[0].backward()
    assert _maxerr(final_T.detach().cpu(), g[f"final_trans_{tag}"]) < 1e-4
    ref_l = g[f"loss_{tag}"]
    got_l = np.array([float(tl[0].detach()), float(tl[1]), float(tl[2]), float(tl[3]), float(tl[4])])
    assert np.abs(got_l - ref_l).max() < 1e-3 * max(1.0, float(np.abs(ref_l).max())), (got_l, ref_l)
    ref = g[f"d_corr_features_{tag}"]
    got = cf.grad.cpu().numpy()
    assert ((np.abs(ref).sum(-1) > 0) == (np.abs(got).sum(-1) > 0)).all()
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())
    ds = float(g[f"d_sigma_{tag}"].reshape(-1)[0])
    assert abs(float(m.sigma.grad) - ds) < 2e-3 * abs(ds) + 1e-9, (float(m.sigma.grad), ds)


def test_f20_training_step_with_transformation_loss(golden_dir):
    """Golden F20, part B: the training step of F19 with weight_transformation = 1 (loss = Classification + SpectralMatching +
    Transformation, libs/trainer.py:141-143): the gradient through the pose head joins the other two at the encoder output.
    Every parameter's gradient within 5e-4 of its largest entry + 1e-5 of the model's largest gradient."""
    g = _load(golden_dir, "f20_pose_head_backward.npz")
    cfg = g["step_cfg"]
    N, seeds = int(cfg[0]), [int(v) for v in cfg[1:]]
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 3, 128), seed=7)
    m = gmf_amd.PointDSC(in_dim=6, num_layers=3, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
                         sigma_d=0.10, k=40, nms_radius=0.10)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).train()
    b = synthetic.synthetic_batch(seeds, N=N, T=40)
    data = {k: _gpu(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    gt = _gpu(b["gt_labels"])
    res = m(data)
    cl = gmf_amd.ClassificationLoss(balanced=False)(res["final_labels"], gt)
    sm = gmf_amd.SpectralMatchingLoss(balanced=False)(res["M"], gt)
    tl = gmf_amd.TransformationLoss(re_thre=15, te_thre=30)(res["final_trans"], _gpu(b["gt_trans"]), data["src_keypts"],
                                                            data["tgt_keypts"], res["final_labels"])
    (1.0 * cl["loss"] + 1.0 * sm + 1.0 * tl[0]).backward()
    ref_losses = g["step_losses"]
    got = [float(cl["loss"].detach()), float(sm.detach()), float(tl[0].detach())]
    assert np.abs(np.array(got) / ref_losses - 1).max() < 1e-4, (got, ref_losses)
    assert _maxerr(res["final_trans"].detach().cpu(), g["step_final_trans"]) < 1e-4
    names, stats, heads = list(g["step_grad_names"]), g["step_grad_stats"], g["step_grad_heads"]
    params = dict(m.named_parameters())
    gmax = float(stats[:, 2].max())
    assert len(names) == 137
    for i, n in enumerate(names):
        gr = params[n].grad
        assert gr is not None, n
        gr = gr.double().reshape(-1).cpu()
        k = min(16, gr.numel())
        tol = 5e-4 * stats[i, 2] + 1e-5 * gmax
        assert np.abs(gr[:k].numpy() - heads[i, :k]).max() < tol, (n, stats[i])
        assert abs(float(gr.norm()) - stats[i, 1]) < 5e-4 * stats[i, 1] + 1e-5 * gmax * np.sqrt(gr.numel()), (n, stats[i])


@pytest.mark.parametrize("tag", ["n10", "n1000", "n8000"])
def test_f21_weighted_procrustes_backward(golden_dir, tag):
    """Golden F21: DGR's weighted_procrustes differentiated with respect to the weights by the reference's own autograd
    (core/registration.py:91-113; the DGR trainer trains its inlier network through this solve, core/trainer.py:594-614).
    gmf_amd.weighted_procrustes carries the same gradient (gmf_weighted_procrustes_backward)."""
    g = _load(golden_dir, "f21_weighted_procrustes_backward.npz")
    X, Y = _gpu(torch.from_numpy(g[f"X_{tag}"])), _gpu(torch.from_numpy(g[f"Y_{tag}"]))
    w = _gpu(torch.from_numpy(g[f"w_{tag}"])).requires_grad_(True)
    R, t = gmf_amd.weighted_procrustes(X, Y, w, float(np.finfo(np.float32).eps))
    ((_gpu(torch.from_numpy(g[f"gR_{tag}"])) * R).sum() + (_gpu(torch.from_numpy(g[f"gt_{tag}"])) * t).sum()).backward()
    assert _maxerr(R.detach().cpu(), g[f"R_{tag}"]) < 1e-5 and _maxerr(t.detach().cpu(), g[f"t_{tag}"]) < 1e-5
    ref = g[f"dw_{tag}"]
    assert w.grad.shape == w.shape
    assert np.abs(w.grad.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
    # no gradient asked for: the plain forward, and X / Y requiring grad is refused
    with torch.no_grad():
        R2, _ = gmf_amd.weighted_procrustes(X, Y, w, float(np.finfo(np.float32).eps))
    assert torch.equal(R2, R.detach())
    with pytest.raises(RuntimeError):
        gmf_amd.weighted_procrustes(X.clone().requires_grad_(True), Y, w, 1e-7)


def test_compat_dense_matches_formula():
    """gmf_compat_dense (the trainable path's [B,N,N] compat matrix, PointDSC.py:216-221) against the formula in torch."""
    from gmf_amd import train as T_
    b = synthetic.synthetic_batch([5, 6], N=333, T=8)
    src, tgt = _gpu(b["src_keypts"]), _gpu(b["tgt_keypts"])
    c = T_.compat_dense(src, tgt, 0.1)
    d = torch.norm(src[:, :, None] - src[:, None], dim=-1) - torch.norm(tgt[:, :, None] - tgt[:, None], dim=-1)
    ref = torch.clamp(1.0 - d ** 2 / 0.1 ** 2, min=0)
    assert (c - ref).abs().max() < 2e-5 and torch.equal(c, c.transpose(1, 2))


def test_training_primitives_against_torch():
    """Each HIP training primitive on odd sizes against torch on the device: gmf_gemm_f32 in its four transpose forms, as a
    batched sub-matrix product and with a long contraction (split-K path), LCPE, LayerNorm, softmax, GEGLU, column sums."""
    from gmf_amd import train as T_
    gen = torch.Generator().manual_seed(11)
    rn = lambda *s: _gpu(torch.randn(*s, generator=gen))
    A, Bm = rn(70, 45), rn(45, 133)
    assert _maxerr(T_.gemm(A, Bm).cpu(), (A @ Bm).cpu()) < 1e-4
    assert _maxerr(T_.gemm(A.t().contiguous(), Bm, ta=True).cpu(), (A @ Bm).cpu()) < 1e-4
    assert _maxerr(T_.gemm(A, Bm.t().contiguous(), tb=True).cpu(), (A @ Bm).cpu()) < 1e-4
    assert _maxerr(T_.gemm(A.t().contiguous(), Bm.t().contiguous(), ta=True, tb=True).cpu(), (A @ Bm).cpu()) < 1e-4
    bias, R = rn(133), rn(70, 133)
    assert _maxerr(T_.gemm(A, Bm, bias=bias, residual=R, alpha=0.5).cpu(), (0.5 * (A @ Bm) + bias + R).cpu()) < 1e-4
    X, Y = rn(20000, 96), rn(20000, 160)                       # K = 20000 rows: split-K
    ref = (X.double().t() @ Y.double()).float()
    assert _maxerr(T_.gemm(X, Y, ta=True).cpu(), ref.cpu()) < 2e-5 * float(ref.abs().max())
    # 16-byte friendly sizes take the LDS-staged kernel (k_gemm_lds): ragged tiles (260 = 2 x 128 + 4 rows, 132 columns), a
    # contraction that is not a multiple of the 16-deep k-step, all four transpose forms, bias + residual + ReLU, a batch
    A2, B2 = rn(260, 200), rn(200, 132)
    ref2 = (A2.double() @ B2.double()).float()
    tol2 = 2e-5 * float(ref2.abs().max())
    assert _maxerr(T_.gemm(A2, B2).cpu(), ref2.cpu()) < tol2
    assert _maxerr(T_.gemm(A2.t().contiguous(), B2, ta=True).cpu(), ref2.cpu()) < tol2
    assert _maxerr(T_.gemm(A2, B2.t().contiguous(), tb=True).cpu(), ref2.cpu()) < tol2
    assert _maxerr(T_.gemm(A2.t().contiguous(), B2.t().contiguous(), ta=True, tb=True).cpu(), ref2.cpu()) < tol2
    bias2, R2 = rn(132), rn(260, 132)
    assert _maxerr(T_.gemm(A2, B2, bias=bias2, residual=R2, relu=True).cpu(), torch.relu(ref2 + bias2 + R2).cpu()) < tol2
    A3, B3 = rn(5, 300, 64), rn(5, 264, 64)                    # batched Q K^T shape
    out3 = torch.empty(5, 300, 264, device=DEV)
    T_.gemm(A3, B3, tb=True, out=out3, m=300, n=264, k=64, lda=64, ldb=64, ldc=264, batch=5, sa=300 * 64, sb=264 * 64, sc=300 * 264)
    assert _maxerr(out3.cpu(), torch.matmul(A3, B3.transpose(1, 2)).cpu()) < 1e-4
    P3 = rn(3, 1000, 1000)                                     # long contraction with few tiles: split-K through the LDS kernel
    V3 = rn(3, 1000, 128)
    out4 = torch.empty(3, 1000, 128, device=DEV)
    T_.gemm(P3, V3, out=out4, m=1000, n=128, k=1000, lda=1000, ldb=128, ldc=128, batch=3, sa=10 ** 6, sb=128000, sc=128000)
    ref4 = torch.matmul(P3.double(), V3.double()).float()
    assert _maxerr(out4.cpu(), ref4.cpu()) < 2e-5 * float(ref4.abs().max())
    x = rn(3 * 37, 64)
    w, b = rn(64, 1, 3), rn(64)
    ref = torch.nn.functional.conv1d(x.reshape(3, 37, 64).permute(0, 2, 1), w, b, padding=1, groups=64).permute(0, 2, 1) + x.reshape(3, 37, 64)
    assert _maxerr(T_.lcpe_fwd(x, w, b, 37).cpu(), ref.reshape(-1, 64).cpu()) < 1e-5
    gam, bet = rn(64), rn(64)
    y, mu, rs = T_.layernorm_fwd(x, gam, bet)
    assert _maxerr(y.cpu(), torch.nn.functional.layer_norm(x, (64,), gam, bet).cpu()) < 1e-5
    S = rn(50, 196)
    assert _maxerr(T_.softmax_rows(S, 0.3).cpu(), torch.softmax(S * 0.3, -1).cpu()) < 1e-6
    hd = rn(33, 256)
    assert _maxerr(T_.geglu_fwd(hd).cpu(), (hd[:, :128] * torch.nn.functional.gelu(hd[:, 128:])).cpu()) < 1e-5
    assert _maxerr(T_.colsum(x).cpu(), x.sum(0).cpu()) < 1e-4
    xs = x.reshape(3, 37, 64)
    ref = (xs[:, 1:] * xs[:, :-1]).sum((0, 1))               # sum_l x[l] * x[l - 1] within each sequence
    assert _maxerr(T_.colsum(x, y=x, shift=-1, L=37).cpu(), ref.cpu()) < 1e-4
    # one pass, two sums (a LayerNorm's dgamma / dbeta), with the gradient masked by a ReLU output
    yr = rn(111, 64)
    prod, plain = T_.colsum(x, y=y, mean=None, rstd=None, dual=True, relu_y=yr)
    xm = x * (yr > 0)
    assert _maxerr(prod.cpu(), (xm * y).sum(0).cpu()) < 1e-4 and _maxerr(plain.cpu(), xm.sum(0).cpu()) < 1e-4
    xh = (x - mu[:, None]) * rs[:, None]
    dg, db = T_.colsum(y, y=x, mean=mu, rstd=rs, dual=True)
    assert _maxerr(dg.cpu(), (y * xh).sum(0).cpu()) < 1e-4 and _maxerr(db.cpu(), y.sum(0).cpu()) < 1e-4


def test_sm_loss_backward_full_size():
    """32 pairs x 5000 (the benchmark shape): the HIP backward against torch autograd on the device over the dense
    formulation for one pair (M materialised: 100 MB), and a finite, symmetric-consistent result for all."""
    B, N = 4, 5000
    b = synthetic.synthetic_batch([900 + i for i in range(B)], N=N, T=12)
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(B, N, 128, generator=g)
    feat[b["gt_labels"] > 0] += 1.5 * torch.randn(1, 128, generator=g)
    fn = torch.nn.functional.normalize(_gpu(feat), p=2, dim=-1).requires_grad_(True)
    sigma = torch.tensor([0.8], device=DEV, requires_grad=True)
    gt = _gpu(b["gt_labels"])
    loss = gmf_amd.SpectralMatchingLoss().from_features(fn, sigma, gt)
    loss.backward()
    assert torch.isfinite(fn.grad).all() and torch.isfinite(sigma.grad).all()
    fr = fn.detach().clone().requires_grad_(True)
    sr = sigma.detach().clone().requires_grad_(True)
    M = torch.clamp(1 - (1 - fr @ fr.permute(0, 2, 1)) / sr ** 2, min=0, max=1)
    idx = torch.arange(N, device=DEV)
    M[:, idx, idx] = 0
    gtM = ((gt[:, None, :] + gt[:, :, None]) == 2).float()
    gtM[:, idx, idx] = 0
    lp = ((M - 1) ** 2 * gtM).sum((-1, -2)) / (torch.relu(gtM.sum((-1, -2)) - 1.0) + 1.0)
    ln = (M ** 2 * (1 - gtM)).sum((-1, -2)) / (torch.relu((1 - gtM).sum((-1, -2)) - 1.0) + 1.0)
    ref = torch.mean(lp * 0.5 + ln * 0.5)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * abs(float(ref.detach()))
    assert _maxerr(fn.grad.cpu(), fr.grad.cpu()) < 5e-5 * float(fr.grad.abs().max())
    assert abs(float(sigma.grad) - float(sr.grad)) < 1e-4 * abs(float(sr.grad))


def test_loss_modules_gradients_and_forward_only_rest():
    """ClassificationLoss, SpectralMatchingLoss(M, gt) and TransformationLoss carry their gradients (checked against torch
    autograd over the same formulas: balanced and not, with a per-element weight; the transformation loss with the reference's
    broadcast over the batch and a pair without positive logits); inputs on the CPU are refused."""
    gen = torch.Generator().manual_seed(5)
    pred = _gpu(torch.randn(3, 50, generator=gen)).requires_grad_(True)
    gt = _gpu((torch.rand(3, 50, generator=gen) < 0.3).float())
    w = _gpu(torch.rand(3, 50, generator=gen))
    for balanced, weight in ((True, None), (False, None), (True, w)):
        pred.grad = None
        out = gmf_amd.ClassificationLoss(balanced=balanced)(pred, gt, weight)
        out["loss"].backward()
        pr = pred.detach().clone().requires_grad_(True)
        num_pos, num_neg = torch.relu(gt.sum() - 1) + 1, torch.relu((1 - gt).sum() - 1) + 1
        if weight is not None:
            ref = (torch.nn.functional.binary_cross_entropy_with_logits(pr, gt, reduction="none") * weight).mean()
        elif balanced:
            ref = torch.nn.functional.binary_cross_entropy_with_logits(pr, gt, pos_weight=num_neg / num_pos)
        else:
            ref = torch.nn.functional.binary_cross_entropy_with_logits(pr, gt)
        ref.backward()
        assert abs(float(out["loss"].detach()) - float(ref.detach())) < 1e-5
        assert _maxerr(pred.grad.cpu(), pr.grad.cpu()) < 1e-6
    f = torch.nn.functional.normalize(_gpu(torch.randn(2, 70, 128, generator=gen)), dim=-1)
    for balanced in (True, False):
        M = gmf_amd.similarity_matrix(f, 0.9, contiguous=True).requires_grad_(True)
        gt2 = _gpu((torch.rand(2, 70, generator=gen) < 0.4).float())
        loss = gmf_amd.SpectralMatchingLoss(balanced=balanced)(M, gt2)
        loss.backward()
        Mr = M.detach().clone().requires_grad_(True)
        gtM = ((gt2[:, None, :] + gt2[:, :, None]) == 2).float()
        idx = torch.arange(70, device=DEV)
        gtM[:, idx, idx] = 0
        if balanced:
            lp = ((Mr - 1) ** 2 * gtM).sum((-1, -2)) / (torch.relu(gtM.sum((-1, -2)) - 1.0) + 1.0)
            ln = (Mr ** 2 * (1 - gtM)).sum((-1, -2)) / (torch.relu((1 - gtM).sum((-1, -2)) - 1.0) + 1.0)
            ref = torch.mean(lp * 0.5 + ln * 0.5)
        else:
            ref = ((Mr - gtM) ** 2).mean()
        ref.backward()
        assert abs(float(loss.detach()) - float(ref.detach())) < 1e-6
        assert _maxerr(M.grad.cpu(), Mr.grad.cpu()) < 1e-5 * float(Mr.grad.abs().max())
    bs, n = 3, 41
    T = torch.eye(4).repeat(bs, 1, 1)
    T[:, :3, :] += 0.1 * torch.randn(bs, 3, 4, generator=gen)
    T = _gpu(T).requires_grad_(True)
    src3, tgt3 = _gpu(torch.rand(bs, n, 3, generator=gen)), _gpu(torch.rand(bs, n, 3, generator=gen))
    probs = _gpu(torch.randn(bs, n, generator=gen))
    probs[1] = -probs[1].abs()                           # pair 1: no positive logit -> contributes nothing (loss.py:57-59)
    out = gmf_amd.TransformationLoss()(T, T.detach(), src3, tgt3, probs)
    (2.5 * out[0]).backward()
    Tr = T.detach().clone().requires_grad_(True)
    ref = 0
    for i in (0, 2):
        warp = src3[i] @ Tr[i, :3, :3].T + Tr[i, :3, 3]
        ref = ref + ((warp[None] - tgt3) ** 2).sum(-1).mean()
    (2.5 * ref / bs).backward()
    assert abs(float(out[0].detach()) - float(ref.detach() / bs)) < 1e-6 * float(ref.detach())
    assert _maxerr(T.grad.cpu(), Tr.grad.cpu()) < 1e-5 * float(Tr.grad.abs().max())
    assert float(T.grad[1].abs().max()) == 0.0 and float(T.grad[:, 3].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="HIP device"):
        gmf_amd.ClassificationLoss()(torch.zeros(1, 8), torch.zeros(1, 8))


@pytest.mark.parametrize("B,N,S", [(3, 5000, 500), (2, 37, 3), (1, 16384, 1638), (2, 1000, 1000), (4, 2049, 1),
                                   (1, 16385, 1638), (2, 20000, 2000), (1, 40000, 4000), (2, 70001, 700), (1, 33000, 19000)])
def test_topk_select_equals_full_sort(B, N, S):
    """The radix-select form of argsort(descending)[:S] against the full bitonic sort and against a stable host sort:
    many exact ties (zeros of both signs, repeated values), negative keys, S = 1 and S = N.  [r4] N > 16 384 (the reference has
    no limit: PointDSC.py:268-286): the keys stay in global memory (k_select_topk<false>; above 65 535 rows the counters of
    the ordered compaction need more than 16 bits), both knob settings then run that form."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(N + S)
    keys = torch.randn(B, N, generator=gen)
    keys[:, ::3] = 0.0
    keys[:, 1::7] = -0.0
    keys[:, 2::5] = torch.round(keys[:, 2::5] * 4) / 4          # repeated values
    if B > 1:
        keys[1] = -keys[1].abs()                                 # no positive key at all
    dsrc = _gpu(torch.zeros(B, N, 3))
    dk = _gpu(keys)
    outs = []
    for knob in (1, 0):
        h.call("gmf_set_tuning", b"topk_select", knob)
        out = torch.empty((B, S), device=DEV, dtype=torch.int32)
        h.call("gmf_pick_seeds", dsrc.data_ptr(), dk.data_ptr(), B, N, 0.1, 0, S, out.data_ptr(), st)   # use_nms = 0: plain top-S
        outs.append(out.cpu())
    h.call("gmf_set_tuning", b"topk_select", 1)
    want = torch.sort(keys.double() + 0.0, dim=1, descending=True, stable=True)[1][:, :S]
    assert torch.equal(outs[0].long(), want)
    assert torch.equal(outs[1].long(), want)


def _nms_keys_chunked(src, scores, R, chunk=1000):
    """PointDSC.pick_seeds' keys (PointDSC.py:276-285) without the [N, N] matrices: score_i * [all_j (score_i >= score_j or
    ||src_i - src_j|| >= R)], the same element-wise arithmetic as the oracle's pick_seeds, row blocks at a time."""
    N = src.shape[1]
    keys = torch.empty(N)
    for i0 in range(0, N, chunk):
        d = torch.norm(src[0, i0:i0 + chunk, None, :] - src[0, None, :, :], dim=-1)
        ge = scores[0, i0:i0 + chunk, None] >= scores[0, None, :]
        keys[i0:i0 + chunk] = scores[0, i0:i0 + chunk] * (ge | (d >= R)).all(dim=-1).float()
    return keys


@pytest.mark.parametrize("N", [3000, 20000, 40000])
def test_pose_head_beyond_16384(model, N):
    """[r4] The test-mode pose head has no limit on N any more (the reference has none, PointDSC.py:268-286 / common.py:53-75, and its
    3DMatch evaluation feeds every correspondence: evaluation/test_3DMatch.py:143): N = 20 000 and 40 000 (seed selection with the
    keys in global memory, streamed kNN selection) against the oracle's functions, F5-style - seeds (the oracle's keys, stable
    order), the k neighbours of every seed row, the per-seed hypotheses from those neighbours, their fitness, the chosen pose and
    the refinement.  N = 3000 runs the same comparison on the in-LDS path and checks the chunked key helper against the oracle's
    own pick_seeds."""
    torch.set_num_threads(16)
    b = synthetic.synthetic_batch([900 + N % 97], N=N, T=12)
    r = np.random.default_rng([205, N])
    feat = torch.from_numpy(r.normal(0, 1, (1, N, 128)).astype(np.float32))
    inl = b["gt_labels"][0] > 0
    feat[0, inl] += 2.5 * torch.from_numpy(r.normal(0, 1, (1, 128)).astype(np.float32))
    feat_n = torch.nn.functional.normalize(feat, p=2, dim=-1)
    scores = torch.from_numpy(r.normal(0, 1, (1, N)).astype(np.float32)) + 2.0 * b["gt_labels"]
    src, tgt = b["src_keypts"], b["tgt_keypts"]
    S, k = N // 10, 40
    fT, labels, aux = model.pose_head(_gpu(feat_n), _gpu(src), _gpu(tgt), _gpu(scores), testing=True, return_aux=True)
    # seeds: the reference's keys, ordered (key descending, index ascending)
    keys = _nms_keys_chunked(src, scores, 0.10)
    if N <= 3000:
        sdist = torch.norm(src[:, :, None, :] - src[:, None, :, :], dim=-1)
        ref_seeds = O.pick_seeds(sdist, scores, 0.10, S)
        n_pos = int((keys > 0).sum())
        assert torch.equal(ref_seeds[0, :min(S, n_pos)], torch.sort(keys, descending=True, stable=True)[1][:min(S, n_pos)])
    want = torch.sort(keys.double() + 0.0, descending=True, stable=True)[1][:S]
    seeds = aux["seeds"][0].cpu().long()
    assert torch.equal(seeds, want)
    # the k neighbours of every seed row under 2 - 2 <f_s, f_j> (common.py:70-74; rank 0 dropped)
    d = 2 - 2 * (feat_n[0, seeds] @ feat_n[0].T)
    ref_knn = d.topk(k + 1, dim=-1, largest=False)[1][:, 1:]
    got_knn = aux["knn_idx"][0].cpu().long()
    same = (torch.sort(got_knn, -1)[0] == torch.sort(ref_knn, -1)[0]).float().mean()
    assert float(same) > 0.999, float(same)             # (two fp32 evaluations of d order near-ties differently)
    assert int(got_knn.min()) >= 0 and int(got_knn.max()) < N
    # per-seed hypotheses from the HIP path's own neighbour lists, fitness, best hypothesis, refinement
    sigma, sigma_d = float(model.sigma.detach()), float(model.sigma_spat)
    w, sk, tk = O.seed_weights(feat_n, src, tgt, got_knn[None], sigma, sigma_d, model.num_iterations)
    Ts = O.rigid_transform_3d(sk, tk, w).reshape(1, -1, 4, 4)
    assert _maxerr(aux["seed_trans"].cpu(), Ts) < 1e-3
    fit, best_T, lab = O.score_hypotheses(Ts, src, tgt, 0.10)
    assert _maxerr(aux["fitness"].cpu(), fit) < 2e-4     # (a hypothesis 1e-4 away moves a few of N points across the threshold)
    ref_T = O.post_refinement(best_T, src, tgt, 0.10)
    assert _maxerr(fT.cpu(), ref_T) < 1e-4
    assert _maxerr(fT.cpu(), b["gt_trans"]) < 5e-3
    assert float((labels.cpu() == lab).float().mean()) > 0.999


def test_public_knn_op():
    """[r4] `gmf_amd.knn` (models/common.py:53-75; the op behind PointDSC.py:327) on its own: since round 4 it runs the pose head's path
    (MFMA distance rows + threshold selection) instead of one workgroup per row.  Against the oracle's 2 - 2 x x^T top-(k + 1) on
    unit rows: the same neighbour SETS (ties and rounding at the k-th distance aside) and the same order for nearly every row; the
    sliced form (distance rows above 4 GiB go through pair by pair) against the whole form, exactly."""
    g = torch.Generator().manual_seed(5)
    for B, N, k in ((2, 777, 40), (1, 5000, 40), (3, 100, 10), (1, 17000, 40)):
        x = torch.nn.functional.normalize(torch.randn(B, N, 128, generator=g), dim=-1)
        got = gmf_amd.knn(_gpu(x), k, ignore_self=True, normalized=True).cpu()
        ref = O.knn_indices(x, k)
        assert got.shape == ref.shape and int(got.min()) >= 0 and int(got.max()) < N
        same_set = (torch.sort(got, -1)[0] == torch.sort(ref, -1)[0]).float().mean()
        same_seq = (got == ref).float().mean()
        assert same_set > 0.999 and same_seq > 0.995, (B, N, float(same_set), float(same_seq))
        assert not (got == torch.arange(N)[None, :, None]).any()          # the row itself is dropped
        # [r5] ignore_self=False (common.py:70-71): top-k INCLUDING rank 0 = the row itself followed by its k - 1 nearest
        got0 = gmf_amd.knn(_gpu(x), k, ignore_self=False, normalized=True).cpu()
        assert torch.equal(got0[:, :, 0], torch.arange(N).repeat(B, 1)) and torch.equal(got0[:, :, 1:], got[:, :, :k - 1])
    # [r5] normalized=False (xx - 2 x x^T + xx^T, common.py:66-68) and other feature widths: fp32-MFMA distance rows + the same
    # selection.  Unnormalised rows of very different lengths, against the oracle's knn with the same flags
    for B, N, Cw, k, norm in ((2, 777, 128, 10, False), (1, 3000, 64, 20, False), (3, 100, 33, 9, False), (2, 500, 32, 8, True)):
        x = torch.randn(B, N, Cw, generator=g) * (0.2 + 3.0 * torch.rand(B, N, 1, generator=g))
        if norm:
            x = torch.nn.functional.normalize(x, dim=-1)
        for ign in (True, False):
            got = gmf_amd.knn(_gpu(x), k, ignore_self=ign, normalized=norm).cpu()
            ref = O.knn(x, k, ignore_self=ign, normalized=norm)
            assert got.shape == ref.shape and int(got.min()) >= 0 and int(got.max()) < N
            same_set = (torch.sort(got, -1)[0] == torch.sort(ref, -1)[0]).float().mean()
            same_seq = (got == ref).float().mean()
            assert same_set > 0.999 and same_seq > 0.99, (B, N, Cw, norm, ign, float(same_set), float(same_seq))
    x = _gpu(torch.nn.functional.normalize(torch.randn(3, 20000, 128, generator=g), dim=-1))
    sliced = gmf_amd.knn(x, 40, ignore_self=True, normalized=True)       # 3 x 20000 x 20000 floats > 4 GiB: slices
    for b in range(3):
        assert torch.equal(sliced[b:b + 1], gmf_amd.knn(x[b:b + 1], 40, ignore_self=True, normalized=True))
    gmf_amd.check_status()


def test_full_forward_beyond_16384(model):
    """[r4] `PointDSC.forward` in test mode with more than 16 384 correspondences per pair (the reference's 3DMatch evaluation feeds every
    correspondence, evaluation/test_3DMatch.py:143) - a uniform B = 1 call at N = 20 000 and a ragged batch of 16 500 + 17 000: through
    properties (the oracle's N x N matrices at this size are minutes of CPU; the pose head's new paths have their own oracle comparison,
    test_pose_head_beyond_16384): finite logits, the pose within 2e-3 of the ground truth, the 25 % inliers found."""
    b = synthetic.synthetic_batch([77, 78], N=20000, T=196)
    keys = ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")
    one = {k: _gpu(b[k][:1]) for k in keys}
    one["testing"] = True
    r = model(one)
    assert torch.isfinite(model.last_logits).all()
    assert _maxerr(r["final_trans"].cpu(), b["gt_trans"][:1]) < 2e-3
    assert float(r["final_labels"].sum()) > 0.24 * 20000
    sizes = [16500, 17000]
    rag = {k: [_gpu(b[k][i, :n]) for i, n in enumerate(sizes)] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    rag.update(p_tokens=_gpu(b["p_tokens"]), q_tokens=_gpu(b["q_tokens"]), testing=True)
    r2 = model(rag)
    assert all(torch.isfinite(lg).all() for lg in r2["logits"])
    assert _maxerr(r2["final_trans"].cpu(), b["gt_trans"]) < 2e-3
    assert all(float(lab.sum()) > 0.2 * n for lab, n in zip(r2["final_labels"], sizes))
    gmf_amd.check_status()


@pytest.mark.parametrize("case", ["3dmatch", "kitti", "clustered", "wrapped", "far", "tiny_radius"])
def test_binned_nms_equals_all_pairs(case):
    """pick_seeds with the grid-binned candidate lists against the all-pairs kernel: the full order (S = N) must be
    identical - uniform clouds, dense clusters inside one cell, clouds spanning many wraps of the 16-cell grid, coordinates
    far from the origin (the flagged all-candidates path) and a radius below the point spacing."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(len(case))
    B, N, R = 3, 3000, 0.1
    pts = torch.rand(B, N, 3, generator=gen) * 3.0
    if case == "kitti":
        N, R = 6000, 1.2
        pts = torch.rand(B, N, 3, generator=gen) * torch.tensor([80.0, 80.0, 4.0]) - torch.tensor([40.0, 40.0, 2.0])
    elif case == "clustered":
        pts = torch.randn(B, N, 3, generator=gen) * 0.03 + torch.randint(0, 4, (B, N, 1), generator=gen).float()
    elif case == "wrapped":
        pts = torch.rand(B, N, 3, generator=gen) * 40.0 - 20.0            # 400 cells per axis: 25 wraps
        pts[:, 1::2] = pts[:, 0::2] + 0.05 * torch.randn(B, N // 2, 3, generator=gen)   # close partners across the cloud
    elif case == "far":
        pts = pts + torch.tensor([9.0e3, -7.0e3, 0.0])                    # 9e4 cells from the origin: flagged pair
    elif case == "tiny_radius":
        R = 1e-3
    scores = torch.randn(B, N, generator=gen)
    scores[:, ::11] = scores[:, 1::11][:, : scores[:, ::11].shape[1]]     # equal scores: the >= test matters
    dp, ds = _gpu(pts.contiguous()), _gpu(scores.contiguous())
    outs = []
    for knob in (2, 0):                                                   # 2 = binned whatever the grid size
        h.call("gmf_set_tuning", b"nms_binned", knob)
        out = torch.empty((B, N), device=DEV, dtype=torch.int32)
        h.call("gmf_pick_seeds", dp.data_ptr(), ds.data_ptr(), B, N, R, 1, N, out.data_ptr(), st)
        outs.append(out.cpu())
    h.call("gmf_set_tuning", b"nms_binned", 1)
    assert torch.equal(outs[0], outs[1])
    # and against the reference formula on pair 0 (oracle.pick_seeds works on the N x N distance matrix)
    d = torch.norm(pts[0][:, None] - pts[0][None], dim=-1)
    want = O.pick_seeds(d[None], scores[:1], R, N)
    n_pos = int((want[0] >= 0).sum())
    keys_ref = scores[0] * ((scores[0][:, None] >= scores[0][None]) | (d >= R)).all(1).float()
    n_lead = int((keys_ref > 0).sum())                                    # the order of the positive keys is unique
    assert torch.equal(outs[0][0, :n_lead].long(), torch.sort(keys_ref, descending=True, stable=True)[1][:n_lead]) and n_pos == N


def test_bias_relu_nhwc_epilogue():
    """gmf_bias_relu_nhwc (the fused bias + residual + ReLU pass of the image encoder) against the same torch expression."""
    from gmf_amd import _lib
    h = _lib.handle_for(0)
    st = torch.cuda.current_stream().cuda_stream
    for (B, C, H, W, with_res) in [(3, 64, 30, 40, False), (2, 128, 15, 20, True), (1, 4, 1, 1, True)]:
        y = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, device=DEV)
        r = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last) if with_res else None
        want = torch.relu(y + b[None, :, None, None] + (r if with_res else 0))
        h.call("gmf_bias_relu_nhwc", y.data_ptr(), b.data_ptr(), r.data_ptr() if with_res else None, B * H * W, C, st)
        assert torch.equal(y, want)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        h.call("gmf_bias_relu_nhwc", y.data_ptr(), b.data_ptr(), None, 4, 6, st)


@pytest.mark.parametrize("cin,cout,ks,stride,H,W", [(64, 64, 3, 1, 30, 40), (64, 128, 3, 2, 30, 40), (128, 128, 3, 1, 15, 20),
                                                      (64, 128, 1, 2, 30, 40), (64, 64, 3, 1, 7, 5), (64, 128, 3, 2, 9, 11),
                                                      (128, 128, 3, 1, 5, 48), (64, 64, 3, 1, 3, 50)])
def test_conv_nhwc_matches_torch(cin, cout, ks, stride, H, W):
    """gmf_conv_nhwc (implicit GEMM, split-fp16 MFMA) against torch's fp64 convolution of the same fp32 inputs: ResNet-34
    layer1 / layer2 shapes, odd sizes, residual and ReLU on and off."""
    from gmf_amd import _lib, packing
    h = _lib.handle_for(0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(cin + cout + ks + H)
    B = 3
    x = torch.randn(B, cin, H, W, generator=gen)
    Wt = torch.randn(cout, cin, ks, ks, generator=gen) / (cin * ks * ks) ** 0.5
    b = torch.randn(cout, generator=gen)
    pad = ks // 2
    ref = torch.nn.functional.conv2d(x.double(), Wt.double(), b.double(), stride, pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = torch.randn(B, cout, Ho, Wo, generator=gen)
    xg = _gpu(x).contiguous(memory_format=torch.channels_last)
    rg = _gpu(res).contiguous(memory_format=torch.channels_last)
    wimg, bg = _gpu(packing.conv_image(Wt, stride)), _gpu(b)
    for with_res, relu, patch in ((False, 0, 1), (True, 1, 1), (True, 1, 0)):
        h.call("gmf_set_tuning", b"conv_lds_patch", patch)      # stride-1 3x3 shapes: LDS-patch kernel / gather kernel
        y = torch.empty((B, cout, Ho, Wo), device=DEV).contiguous(memory_format=torch.channels_last)
        h.call("gmf_conv_nhwc", xg.data_ptr(), wimg.data_ptr(), bg.data_ptr(), rg.data_ptr() if with_res else None, y.data_ptr(),
               B, H, W, cin, cout, ks, stride, relu, st)
        want = ref + (res.double() if with_res else 0)
        if relu:
            want = torch.relu(want)
        f32 = torch.nn.functional.conv2d(x, Wt, b, stride, pad).double() + (res.double() if with_res else 0)
        f32 = torch.relu(f32) if relu else f32
        err, floor = _maxerr(y.cpu().double(), want), _maxerr(f32, want)
        assert err < max(4.0 * floor, 4e-6), (err, floor)        # floor = error of torch's own fp32 convolution
    h.call("gmf_set_tuning", b"conv_lds_patch", 1)


def test_conv_three_workgroup_form():
    """The 49.5 KiB form of the 64 -> 64 convolution (three workgroups per CU; taken when the grid has between 513 and 768
    workgroups - 56 ... 81 images of 30 x 40) against the two-workgroup form (bit-identical: same arithmetic in the same
    order) and against torch's convolution."""
    from gmf_amd import _lib, packing
    h = _lib.handle_for(0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(56)
    B, C, H, W = 56, 64, 30, 40
    x = torch.randn(B, C, H, W, generator=gen)
    Wt = torch.randn(C, C, 3, 3, generator=gen) / (C * 9) ** 0.5
    b = torch.randn(C, generator=gen)
    xg = _gpu(x).contiguous(memory_format=torch.channels_last)
    wimg, bg = _gpu(packing.conv_image(Wt, 1)), _gpu(b)
    outs = []
    for knob in (1, 2):
        h.call("gmf_set_tuning", b"conv_lds_patch", knob)
        y = torch.empty((B, C, H, W), device=DEV).contiguous(memory_format=torch.channels_last)
        h.call("gmf_conv_nhwc", xg.data_ptr(), wimg.data_ptr(), bg.data_ptr(), None, y.data_ptr(), B, H, W, C, C, 3, 1, 1, st)
        outs.append(y)
    h.call("gmf_set_tuning", b"conv_lds_patch", 1)
    assert torch.equal(outs[0], outs[1])
    want = torch.relu(torch.nn.functional.conv2d(_gpu(x), _gpu(Wt), bg, 1, 1))
    assert _maxerr(outs[0].cpu(), want.cpu()) < 1e-5
