"""Generate golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference, read-only).  The reference's
Python never leaves that container: this script stores seeds, small inputs and the
reference's OUTPUTS as .npz fixtures.  Re-run with:  python oracle/gen_fixtures.py

Import notes: `models.PointDSC` imports torchvision only to
download ImageNet weights for the ResNet image encoder, which is upstream of the hot
path.  torchvision is not installed, so a stub module that returns a random-init
state dict is registered before the import, and the model's `image_encoder` is then
replaced by nn.Identity so that image TOKENS are fed directly (the [B,128,H,W] view the
reference flattens at PointDSC.py:129-135).
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gmf_oracle as O  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvu = types.ModuleType("torchvision.models.utils")
    tvu.load_state_dict_from_url = lambda *a, **k: _STUB_SD[0]
    tv.models, tvm.utils = tvm, tvu
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.utils": tvu})
    sys.path.insert(0, os.path.join(REF, "GMF_PointDSC"))
    import models.resnet as resnet
    _STUB_SD[0] = resnet.ResNet(3, resnet.BasicBlock, [3, 4, 6, 3]).state_dict()
    import models.PointDSC as pdsc
    import models.fusion_layer as fl
    import models.common as common
    spec = importlib.util.spec_from_file_location(
        "dgr_perceiver_io", os.path.join(REF, "GMF_DeepGlobalRegistration/GMF_DeepGlobalRegistration_fcgf/model/perceiver_io.py"))
    pio = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pio)
    # core/registration.py imports core.knn / core.loss; load it with its package dir on the path
    dgr_root = os.path.join(REF, "GMF_DeepGlobalRegistration/GMF_DeepGlobalRegistration_fcgf")
    sys.path.insert(0, dgr_root)
    spec = importlib.util.spec_from_file_location("dgr_registration", os.path.join(dgr_root, "core/registration.py"))
    reg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(reg)
    return pdsc, fl, common, pio, reg


_STUB_SD = [None]


def _np(t):
    return t.detach().cpu().numpy()


def _strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _tok_to_image(tok):
    """[B,T,C] tokens -> the [B,C,1,T] map whose .view(B,C,H*W).permute(0,2,1) is `tok`."""
    return tok.permute(0, 2, 1)[:, :, None, :].contiguous()


def build_ref_pointdsc(pdsc, sd, num_layers=12, sigma_d=0.1, tau=0.10, nms=0.10, k=40, ratio=0.1):
    m = pdsc.PointDSC(in_dim=6, num_layers=num_layers, num_channels=128, num_iterations=10, ratio=ratio,
                      inlier_threshold=tau, sigma_d=sigma_d, k=k, nms_radius=nms)
    m.encoder.image_encoder = torch.nn.Identity()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return m.eval()


def gen_f13(reg):
    """F13: DGR GlobalRegistration (row f-3) - the reference's own function on seeded scenes (gmf_amd.synthetic.dgr_scene).
    The fixture holds the seeds, the call arguments and the reference's outputs; inputs are regenerated from the seeds."""
    from gmf_amd import synthetic
    out, cases = {}, []
    for N, seed, ratio, q, use_w in ((200, 0, 1e-4, 0.1, True), (1000, 1, 1e-4, 0.1, True), (8000, 2, 1e-4, 0.1, True),
                                     (1000, 3, 1e-5, 1.0, True), (500, 4, 1e-4, 0.1, False), (3000, 5, 1e-4, 0.05, True)):
        X, Y, w, _, _ = synthetic.dgr_scene(N, seed)
        R, t, o = reg.GlobalRegistration(X.clone(), Y.clone(), weights=w.clone() if use_w else None,
                                         break_threshold_ratio=ratio, quantization_size=q)
        tag = f"{N}_{seed}"
        cases.append([N, seed, ratio, q, int(use_w)])
        out[f"R_{tag}"], out[f"t_{tag}"] = _np(R), _np(t)
        out[f"stats_{tag}"] = np.array([o["iterations"], o["loss"], o["break_count"]], np.float64)
        Rp, tp = (reg.weighted_procrustes(X, Y, w, np.finfo(np.float32).eps) if use_w
                  else reg.argmin_se3_squared_dist(X, Y))
        out[f"Rinit_{tag}"], out[f"tinit_{tag}"] = _np(Rp), _np(tp)
        print("F13", tag, o)
    out["cases"] = np.array(cases, np.float64)
    # ortho2rotation and the loss on their own
    r = np.random.default_rng([113])
    poses = torch.from_numpy(r.normal(size=(16, 6)).astype(np.float32))
    out["o2r_in"], out["o2r_out"] = _np(poses), _np(reg.ortho2rotation(poses))
    A = torch.from_numpy(r.normal(size=(300, 3)).astype(np.float32))
    Bm = A + torch.from_numpy(r.normal(scale=0.7, size=(300, 3)).astype(np.float32))
    wl = torch.from_numpy(r.uniform(0, 1, (300, 1)).astype(np.float32))
    from core.loss import HighDimSmoothL1Loss
    out["loss_A"], out["loss_B"], out["loss_w"] = _np(A), _np(Bm), _np(wl)
    out["loss_val"] = np.array([float(HighDimSmoothL1Loss(wl, 0.5)(A, Bm)), float(HighDimSmoothL1Loss(None, 0.5)(A, Bm))])
    np.savez_compressed(os.path.join(GOLD, "f13_global_registration.npz"), **out)



def gen_f14(pdsc):
    """F14: validation step (row f-4, forward half) - the reference's non-test forward (M, logits, pose) and its three
    metrics (libs/loss.py) on seeded scenes.  Inputs are regenerated from the seeds; M is stored whole at N = 96 and
    as sampled rows + fp64 checksums at N = 257."""
    import libs.loss as L
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    model = build_ref_pointdsc(pdsc, O.seeded_state_dict(O.pointdsc_shapes(6, 12, 128), seed=7))
    out = {}
    for N, seeds in ((96, [61, 62]), (257, [63, 64, 65])):
        b = O.synthetic_batch(seeds, N=N, T=196)
        data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
                "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"])}
        res = model(data)
        M, logits, T = res["M"], res["final_labels"], res["final_trans"]
        gt = b["gt_labels"]
        out[f"pair_seeds_N{N}"] = np.array(seeds)
        out[f"logits_N{N}"], out[f"final_trans_N{N}"] = _np(logits), _np(T)
        if N <= 96:
            out[f"M_N{N}"] = _np(M)
        else:
            out[f"M_rows_N{N}"] = _np(M[:, ::16])
        out[f"M_sum_N{N}"] = _np(M.double().sum((1, 2)))
        out[f"M_sumsq_N{N}"] = _np((M.double() ** 2).sum((1, 2)))
        cs = L.ClassificationLoss(balanced=True)(logits, gt)
        out[f"class_N{N}"] = np.array([float(cs[k]) for k in ("loss", "precision", "recall", "f1", "logit_true", "logit_false")])
        cu = L.ClassificationLoss(balanced=False)(logits, gt)
        out[f"class_unbalanced_N{N}"] = np.array([float(cu["loss"])])
        out[f"sm_N{N}"] = np.array([float(L.SpectralMatchingLoss(balanced=True)(M, gt)),
                                    float(L.SpectralMatchingLoss(balanced=False)(M, gt))])
        tl = L.TransformationLoss(re_thre=15, te_thre=30)(T, b["gt_trans"], b["src_keypts"], b["tgt_keypts"], logits)
        out[f"trans_N{N}"] = np.array([float(v) for v in tl])
        print("F14", N, out[f"class_N{N}"], out[f"sm_N{N}"], out[f"trans_N{N}"])
    # the metrics on their own, on inputs that reach the corner cases: no predicted inlier in pair 1 (loss term 0,
    # loss.py:57-59), a pair without any ground-truth inlier, poses far from the ground truth (recall < 100 %)
    r = np.random.default_rng([114])
    Bm, Nm = 3, 200
    pred = torch.from_numpy(r.normal(0, 2, (Bm, Nm)).astype(np.float32))
    pred[1] = -pred[1].abs() - 0.1
    gt = torch.from_numpy((r.uniform(size=(Bm, Nm)) < 0.3).astype(np.float32))
    gt[2] = 0
    f = torch.from_numpy(r.normal(size=(Bm, Nm, 16)).astype(np.float32))
    f = f / f.norm(dim=-1, keepdim=True)
    Mm = torch.clamp(1 - (1 - f @ f.permute(0, 2, 1)) / 0.8 ** 2, min=0, max=1)
    bb = O.synthetic_batch([71, 72, 73], N=Nm, T=12)
    T = bb["gt_trans"].clone()
    T[1, :3, 3] += 0.5
    T[2, :3, :3] = T[2, :3, :3] @ torch.tensor([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    out["alone_pred"], out["alone_gt"], out["alone_M"], out["alone_T"] = _np(pred), _np(gt), _np(Mm), _np(T)
    out["alone_seeds"] = np.array([71, 72, 73])
    cs = L.ClassificationLoss(balanced=True)(pred, gt)
    out["alone_class"] = np.array([float(cs[k]) for k in ("loss", "precision", "recall", "f1", "logit_true", "logit_false")])
    out["alone_class_unbalanced"] = np.array([float(L.ClassificationLoss(balanced=False)(pred, gt)["loss"])])
    out["alone_sm"] = np.array([float(L.SpectralMatchingLoss(balanced=True)(Mm, gt)),
                                float(L.SpectralMatchingLoss(balanced=False)(Mm, gt))])
    out["alone_trans"] = np.array([float(v) for v in L.TransformationLoss()(T, bb["gt_trans"], bb["src_keypts"], bb["tgt_keypts"], pred)])
    print("F14 alone", out["alone_class"], out["alone_sm"], out["alone_trans"])
    np.savez_compressed(os.path.join(GOLD, "f14_validation_step.npz"), **out)


def gen_f15():
    """F15: the image encoder at batch size (row f-1) - the reference's ImageEncoder (models/Img_Encoder.py:9-19,
    models/resnet.py:195-216) on 64 seeded 120 x 160 images (32 pairs x 2: every layer1 / layer2 convolution of the HIP
    encoder then takes its native kernel, and the 64 -> 64 shape its three-workgroups-per-CU form) and on 32 images of
    96 x 128.  Stored: the tokens [B, H'W', 128] of every 8th image row-sampled (::7), plus fp64 sum and sum of squares
    of every image's tokens."""
    import models.Img_Encoder as ie
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    enc = ie.ImageEncoder().eval()
    shapes_ie = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(O.seeded_state_dict(shapes_ie, seed=111, gain=1.0))
    out = {"seed": 111}
    for tag, nimg, H, W in (("64x120x160", 64, 120, 160), ("32x96x128", 32, 96, 128)):
        r = np.random.default_rng([115, nimg, H, W])
        img = torch.from_numpy(r.uniform(0, 1, (nimg, 3, H, W)).astype(np.float32))
        feat = torch.cat([enc(img[i:i + 8]) for i in range(0, nimg, 8)])
        tok = feat.view(nimg, 128, -1).permute(0, 2, 1)           # PointDSC.py:129-131
        out[f"shape_{tag}"] = np.array([nimg, H, W])
        out[f"rows_{tag}"] = _np(tok[::8, ::7]).astype(np.float32)
        out[f"sum_{tag}"] = _np(tok.double().sum((1, 2)))
        out[f"sumsq_{tag}"] = _np((tok.double() ** 2).sum((1, 2)))
        print("F15", tag, tuple(tok.shape), float(tok.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "f15_image_encoder_batch.npz"), **out)


def gen_f16(pdsc):
    """F16: the whole test-mode forward where the seed list runs into the zero-key tie group - N = 1500 / 3000 with the
    UNMODIFIED seeded weights (mostly negative logits: fewer than S local maxima have a positive score, so
    `argsort(scores * is_local_max, descending)[:S]` (PointDSC.py:284-286) is filled up from the suppressed points, whose
    key is +-0, in whatever order this torch build's unstable CPU sort leaves them).  Stored per scene: the reference's
    logits, seeds (captured from pick_seeds), final_trans, final_labels and the ground truth."""
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    model = build_ref_pointdsc(pdsc, O.seeded_state_dict(O.pointdsc_shapes(6, 12, 128), seed=7))
    caps = {}
    model.classification.register_forward_hook(lambda m, i, o: caps.__setitem__("logits", o))
    orig_pick = model.pick_seeds

    def pick(*a, **k):
        caps["seeds"] = orig_pick(*a, **k)
        return caps["seeds"]
    model.pick_seeds = pick
    out, cases = {}, []
    for N, seeds in ((1500, [81, 82, 83]), (3000, [84, 85])):
        for sd_ in seeds:
            b = O.synthetic_batch([sd_], N=N, T=196)
            data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
                    "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"]), "testing": True}
            res = model(data)
            tag = f"{N}_{sd_}"
            cases.append([N, sd_])
            lg = caps["logits"].squeeze(1)
            out[f"logits_{tag}"] = _np(lg)
            out[f"seeds_{tag}"] = _np(caps["seeds"]).astype(np.int32)
            out[f"final_trans_{tag}"] = _np(res["final_trans"])
            out[f"final_labels_{tag}"] = _np(res["final_labels"]).astype(np.uint8)
            out[f"gt_trans_{tag}"] = _np(b["gt_trans"])
            err = float((res["final_trans"] - b["gt_trans"]).abs().max())
            print("F16", tag, "positive logits:", int((lg > 0).sum()), "S:", N // 10, "max|T - T_gt|:", err)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLD, "f16_pose_tie_scenes.npz"), **out)


def gen_f17(pdsc):
    """F17: first backward slice of the training path (row f-4) - the reference's own autograd through
    M = clamp(1 - (1 - Fn Fn^T) / sigma^2, 0, 1) (PointDSC.py:229-234, inside its non-test forward) and
    libs/loss.py SpectralMatchingLoss: gradients with respect to the encoder output `corr_features` (i.e. through
    F.normalize as well) and to the learned `sigma`.  The encoder output is captured by a hook and stored, so the HIP
    slice is driven from the same features; sigma is set to 0.8 so that the clamp has both active bounds."""
    import libs.loss as L
    torch.manual_seed(0)
    sd = O.seeded_state_dict(O.pointdsc_shapes(6, 12, 128), seed=7)
    sd["sigma"] = torch.tensor([0.8])
    model = build_ref_pointdsc(pdsc, sd)
    caps = {}

    def hook(m, i, o):
        o.retain_grad()
        caps["enc_out"] = o
    model.encoder.register_forward_hook(hook)
    out = {"sigma": np.float32(0.8)}
    with torch.enable_grad():
        for N, seeds, balanced in ((96, [91, 92], True), (150, [93, 94], True), (150, [93, 94], False)):
            b = O.synthetic_batch(seeds, N=N, T=196)
            data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
                    "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"])}
            model.zero_grad()
            res = model(data)
            loss = L.SpectralMatchingLoss(balanced=balanced)(res["M"], b["gt_labels"])
            loss.backward()
            tag = f"N{N}_{'bal' if balanced else 'mse'}"
            feat = caps["enc_out"].permute(0, 2, 1)                       # [B,N,128] = corr_features (PointDSC.py:223-228)
            out[f"pair_seeds_{tag}"] = np.array(seeds)
            out[f"corr_features_N{N}"] = _np(feat)                       # (the same for both forms of the loss)
            out[f"d_corr_features_{tag}"] = _np(caps["enc_out"].grad.permute(0, 2, 1))
            out[f"d_sigma_{tag}"] = _np(model.sigma.grad)
            out[f"loss_{tag}"] = np.float32(float(loss))
            print("F17", tag, "loss", float(loss), "|dF|max", float(caps["enc_out"].grad.abs().max()), "dsigma", float(model.sigma.grad))
    np.savez_compressed(os.path.join(GOLD, "f17_sm_loss_backward.npz"), **out)


def gen_f18(fl, pio):
    """F18: second backward slice (row f-4) - the reference's own autograd through one FusionLayer (Fusion-2 form: pe = True,
    128 / 128 / head 64, fusion_layer.py:131-201) and through the DGR bottleneck PerceiverIO (latent 256, head 128): for seeded
    weights, inputs and upstream gradient, the output and the gradients with respect to the queries, the context and EVERY
    parameter.  Inputs and weights are regenerated from the seeds; the fixture stores outputs and gradients."""
    out = {}
    with torch.enable_grad():
        for tag, mod, lat, dh, N, T, B in (("fl128", fl.FusionLayer, 128, 64, 150, 40, 2), ("pio256", pio.PerceiverIO, 256, 128, 70, 45, 1)):
            torch.manual_seed(0)
            kw = dict(depth=0, dim=128, latent_dim=lat, cross_heads=1, latent_heads=8, cross_dim_head=dh, latent_dim_head=dh, pe=True)
            ref = mod(**kw).train()
            shapes = O.fusion_layer_shapes("", 128, lat, dh, pe=True, out_to_query=(tag == "pio256"))
            ref.load_state_dict(O.seeded_state_dict(shapes, seed=118))
            r = np.random.default_rng([118, N, T])
            x = torch.from_numpy(r.normal(0, 1, (B, N, lat)).astype(np.float32)).requires_grad_(True)
            ctxt = torch.from_numpy(r.normal(0, 1, (B, T, 128)).astype(np.float32)).requires_grad_(True)
            up = torch.from_numpy(r.normal(0, 1, (B, N, lat)).astype(np.float32))
            y = ref(ctxt, queries_encoder=x)
            y.backward(up)
            out[f"{tag}_dims"] = np.array([B, N, T, lat, dh])
            out[f"{tag}_out"] = _np(y)
            out[f"{tag}_dx"], out[f"{tag}_dctx"] = _np(x.grad), _np(ctxt.grad)
            for name, p in ref.named_parameters():
                gr = _np(p.grad)
                if gr.size > 40000:          # the wide feed-forward matrices: every 8th row + fp64 checksums
                    out[f"{tag}_gradrows::{name}"] = gr[::8]
                    out[f"{tag}_gradsum::{name}"] = np.array([gr.astype(np.float64).sum(), (gr.astype(np.float64) ** 2).sum()])
                else:
                    out[f"{tag}_grad::{name}"] = gr
            print("F18", tag, "|out|max", float(y.detach().abs().max()), "|dx|max", float(x.grad.abs().max()),
                  "params", len(list(ref.named_parameters())))
    np.savez_compressed(os.path.join(GOLD, "f18_fusion_layer_backward.npz"), seed=118, **out)


def gen_f19(pdsc):
    """F19: one training step of the reference in its DEFAULT configuration (libs/trainer.py:121-166; config_3DMatch.py:49-52:
    balanced = False, weights 1 / 1 / 0): the model in train() mode (BatchNorm batch statistics), the non-test forward, loss =
    ClassificationLoss + SpectralMatchingLoss, loss.backward().  Stored: logits, loss values, checksums + samples of M, the
    BatchNorm running statistics after the forward, and for EVERY parameter the gradient's fp64 sum, L2 norm, max and first 16
    entries; a second case uses the balanced losses.  Inputs and weights are regenerated from the seeds.

    The model has THREE layers (the class takes num_layers; GMF configures 12): with train-mode BatchNorm (division by the batch
    standard deviation of every channel) and the seeded weights, the 12-layer reference is chaotic - its own fp32 and fp64
    evaluations differ by 0.31 on the logits, the feature error growing 1.6-10x per layer - so its gradients pin nothing.  At
    three layers the fp32 noise on the logits is 8e-5 and every kind of module and gradient is still exercised."""
    import libs.loss as L
    out = {}
    for tag, balanced, N, seeds in (("def", False, 200, [121, 122]), ("bal", True, 150, [123, 124, 125])):
        torch.manual_seed(0)
        sd = O.seeded_state_dict(O.pointdsc_shapes(6, 3, 128), seed=7)
        model = build_ref_pointdsc(pdsc, sd, num_layers=3).train()
        b = O.synthetic_batch(seeds, N=N, T=40)
        data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
                "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"])}
        with torch.enable_grad():
            res = model(data)
            cl = L.ClassificationLoss(balanced=balanced)(res["final_labels"], b["gt_labels"])
            sm = L.SpectralMatchingLoss(balanced=balanced)(res["M"], b["gt_labels"])
            loss = 1.0 * cl["loss"] + 1.0 * sm
            loss.backward()
        M = res["M"].detach()
        out[f"{tag}_cfg"] = np.array([int(balanced), N] + seeds)
        out[f"{tag}_logits"] = _np(res["final_labels"])
        out[f"{tag}_losses"] = np.array([float(cl["loss"]), float(sm)])
        out[f"{tag}_M_sum"] = _np(M.double().sum((1, 2)))
        out[f"{tag}_M_sumsq"] = _np((M.double() ** 2).sum((1, 2)))
        out[f"{tag}_M_rows"] = _np(M[:, ::25])
        names, stats, heads = [], [], []
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.double().reshape(-1)
            names.append(name)
            stats.append([float(g.sum()), float(g.norm()), float(g.abs().max())])
            hd = np.zeros(16)
            hd[:min(16, g.numel())] = g[:16].numpy()
            heads.append(hd)
        out[f"{tag}_grad_names"] = np.array(names)
        out[f"{tag}_grad_stats"] = np.array(stats)
        out[f"{tag}_grad_heads"] = np.array(heads, dtype=np.float32)
        bn = model.encoder.blocks["NonLocal_layer_2"].fc_message[1]
        out[f"{tag}_bn_running"] = np.stack([_np(bn.running_mean), _np(bn.running_var)])
        pc = model.encoder.blocks["PointCN_layer_0"][1]
        out[f"{tag}_pcn_running"] = np.stack([_np(pc.running_mean), _np(pc.running_var)])
        print("F19", tag, "losses", out[f"{tag}_losses"], "params with grad", len(names),
              "max |grad|", float(np.array(stats)[:, 2].max()))
    np.savez_compressed(os.path.join(GOLD, "f19_training_step.npz"), **out)


def gen_f20(pdsc):
    """F20: the rest of the backward (row f-4) - the reference's own autograd through its pose head in the non-test forward
    (PointDSC.py:246-252,304-425: top-S seeds, kNN, feature compatibility, power iteration, weighted SVD of the best seed) and
    libs/loss.py TransformationLoss.

    Part A (slice): the 12-layer model in eval() mode with autograd enabled; loss = TransformationLoss alone.  Stored: the
    encoder output (captured by a hook, so the HIP slice is driven from the same features), the logits, final_trans, the
    loss, d loss / d encoder output (only the best seed's k neighbour rows are non-zero) and d loss / d sigma.
    Part B (whole step): the 3-layer model of F19 in train() mode, loss = Classification + SpectralMatching + 1.0 *
    Transformation (config weight_transformation = 1): the gradient statistics of every parameter."""
    import libs.loss as L
    out = {}
    torch.manual_seed(0)
    sd = O.seeded_state_dict(O.pointdsc_shapes(6, 12, 128), seed=7)
    sd["sigma"] = torch.tensor([0.8])
    model = build_ref_pointdsc(pdsc, sd)
    caps = {}

    def hook(m, i, o):
        o.retain_grad()
        caps["enc_out"] = o
    model.encoder.register_forward_hook(hook)
    out["sigma"] = np.float32(0.8)
    with torch.enable_grad():
        for N, seeds in ((200, [131, 132]), (150, [133, 134, 135])):
            b = O.synthetic_batch(seeds, N=N, T=196)
            data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
                    "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"])}
            model.zero_grad()
            res = model(data)
            tl = L.TransformationLoss(re_thre=15, te_thre=30)(res["final_trans"], b["gt_trans"], b["src_keypts"], b["tgt_keypts"],
                                                              res["final_labels"])
            tl[0].backward()
            tag = f"N{N}"
            g = caps["enc_out"].grad.permute(0, 2, 1)                     # [B,N,128]
            out[f"pair_seeds_{tag}"] = np.array(seeds)
            out[f"corr_features_{tag}"] = _np(caps["enc_out"].permute(0, 2, 1))
            out[f"logits_{tag}"] = _np(res["final_labels"])
            out[f"final_trans_{tag}"] = _np(res["final_trans"])
            out[f"loss_{tag}"] = np.array([float(tl[0]), float(tl[1]), float(tl[2]), float(tl[3]), float(tl[4])], dtype=np.float32)
            out[f"d_corr_features_{tag}"] = _np(g)
            out[f"d_sigma_{tag}"] = _np(model.sigma.grad)
            print("F20", tag, "loss", float(tl[0]), "|dF|max", float(g.abs().max()), "rows with grad",
                  [int((g[i].abs().sum(-1) > 0).sum()) for i in range(g.shape[0])], "dsigma", float(model.sigma.grad))
    # part B
    torch.manual_seed(0)
    sd = O.seeded_state_dict(O.pointdsc_shapes(6, 3, 128), seed=7)
    model = build_ref_pointdsc(pdsc, sd, num_layers=3).train()
    seeds, N = [121, 122], 200
    b = O.synthetic_batch(seeds, N=N, T=40)
    data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
            "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"])}
    with torch.enable_grad():
        res = model(data)
        cl = L.ClassificationLoss(balanced=False)(res["final_labels"], b["gt_labels"])
        sm = L.SpectralMatchingLoss(balanced=False)(res["M"], b["gt_labels"])
        tl = L.TransformationLoss(re_thre=15, te_thre=30)(res["final_trans"], b["gt_trans"], b["src_keypts"], b["tgt_keypts"],
                                                          res["final_labels"])
        loss = 1.0 * cl["loss"] + 1.0 * sm + 1.0 * tl[0]
        loss.backward()
    out["step_cfg"] = np.array([N] + seeds)
    out["step_losses"] = np.array([float(cl["loss"]), float(sm), float(tl[0])])
    out["step_final_trans"] = _np(res["final_trans"])
    names, stats, heads = [], [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.double().reshape(-1)
        names.append(name)
        stats.append([float(g.sum()), float(g.norm()), float(g.abs().max())])
        hd = np.zeros(16)
        hd[:min(16, g.numel())] = g[:16].numpy()
        heads.append(hd)
    out["step_grad_names"] = np.array(names)
    out["step_grad_stats"] = np.array(stats)
    out["step_grad_heads"] = np.array(heads, dtype=np.float32)
    print("F20 step losses", out["step_losses"], "params with grad", len(names), "max |grad|", float(np.array(stats)[:, 2].max()))
    np.savez_compressed(os.path.join(GOLD, "f20_pose_head_backward.npz"), **out)


def gen_f21(reg):
    """F21: DGR's weighted_procrustes (core/registration.py:91-113) differentiated by the reference's own autograd with respect
    to the weights, as its trainer does (core/trainer.py:594-614): L = sum(gR * R) + sum(gt * t) for seeded gR, gt."""
    out = {}
    g = torch.Generator().manual_seed(2100)
    for tag, n, neg in (("n10", 10, False), ("n1000", 1000, False), ("n8000", 8000, True)):
        X = torch.randn(n, 3, generator=g)
        A = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        if torch.det(A) < 0:
            A[:, 0] = -A[:, 0]
        Y = X @ A.t() + torch.randn(1, 3, generator=g) + 0.05 * torch.randn(n, 3, generator=g)
        n_out = n // 3
        Y[:n_out] = torch.randn(n_out, 3, generator=g) * 2.0              # outliers
        w = torch.rand(n, 1, generator=g)
        if neg:
            w[::17] = -0.1 * w[::17]                                      # (|w| in the normaliser: the sign term of the gradient)
        w.requires_grad_(True)
        gR, gt = torch.randn(3, 3, generator=g), torch.randn(3, generator=g)
        with torch.enable_grad():
            R, t = reg.weighted_procrustes(X, Y, w, np.finfo(np.float32).eps)
            ((gR * R).sum() + (gt * t).sum()).backward()
        out[f"X_{tag}"], out[f"Y_{tag}"], out[f"w_{tag}"] = _np(X), _np(Y), _np(w)
        out[f"gR_{tag}"], out[f"gt_{tag}"] = _np(gR), _np(gt)
        out[f"R_{tag}"], out[f"t_{tag}"], out[f"dw_{tag}"] = _np(R), _np(t), _np(w.grad)
        print("F21", tag, "|dw|max", float(w.grad.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "f21_weighted_procrustes_backward.npz"), **out)


def gen_f9b():
    """F9b: the fpfh twin of the DGR bottleneck layer - GMF_DeepGlobalRegistration_fpfh/model/perceiver_io.py:112-200 has NO
    `cpe` (no LCPE) and is instantiated at the same 256-wide bottleneck (…_fpfh/model/resunet_new.py:516-525: latent 256,
    head 128).  Same seeded weights as F9 minus the cpe tensors; M = 100 / 515 voxels."""
    spec = importlib.util.spec_from_file_location(
        "dgr_perceiver_io_fpfh", os.path.join(REF, "GMF_DeepGlobalRegistration/GMF_DeepGlobalRegistration_fpfh/model/perceiver_io.py"))
    pio = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pio)
    shapes = O.fusion_layer_shapes("", 128, 256, 128, pe=False, out_to_query=True)
    sd = O.seeded_state_dict(shapes, seed=119)
    ref = pio.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8,
                          cross_dim_head=128, latent_dim_head=128).eval()
    ref.load_state_dict(sd)
    out = {}
    for M, T in ((100, 12), (515, 300)):
        r = np.random.default_rng([119, M, T])
        x = torch.from_numpy(r.normal(0, 1, (1, M, 256)).astype(np.float32))
        ctx = torch.from_numpy(r.normal(0, 1, (1, T, 128)).astype(np.float32))
        with torch.no_grad():
            out[f"out_M{M}_T{T}"] = _np(ref(ctx, queries_encoder=x))
        print("F9b", M, T, "|out|max", float(np.abs(out[f"out_M{M}_T{T}"]).max()))
    np.savez_compressed(os.path.join(GOLD, "f9b_dgr_perceiver_fpfh.npz"), seed=119, **out)


def gen_f22(pdsc):
    """F22: the KITTI branch of the whole test-mode forward - the reference's own PointDSC built as its KITTI evaluation
    builds it: sigma_d = 1.2 and inlier_threshold = 1.2 (config_3DMatch.py:89-90 as overridden for KITTI,
    evaluation/test_KITTI.py:219: nms_radius = inlier_threshold), which also selects the `[1.2] * 20` refinement list
    (PointDSC.py:505-508).  Scenes are KITTI-shape (+-40 m, 40 % inliers, gmf_amd.synthetic kind="kitti"), N = 700 / 2000,
    T = 196, in TWO weight sets: "stress" = the seeded weights as every other fixture uses them (scaled for 3DMatch-size
    coordinates: the reference's own fp32 is 3e-4 from the exact network there) and "cond" = synthetic.kitti_conditioned
    (layer0.weight / 13: fp32 floor 1.7e-5), the set on which the literal 1e-4 gate is meaningful.  Stored per case: the
    reference's logits, seeds, final_trans, final_labels and the ground truth; inputs are regenerated from the seeds."""
    from gmf_amd import synthetic
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    base = O.seeded_state_dict(O.pointdsc_shapes(6, 12, 128), seed=7, sigma_d=1.2)
    out, cases = {}, []
    for wset, sd in (("stress", base), ("cond", synthetic.kitti_conditioned(base))):
        model = build_ref_pointdsc(pdsc, sd, sigma_d=1.2, tau=1.2, nms=1.2)
        assert abs(float(model.sigma_spat) - 1.2) < 1e-6 and model.inlier_threshold == 1.2 and model.nms_radius == 1.2
        caps = {}
        model.classification.register_forward_hook(lambda m, i, o: caps.__setitem__("logits", o))
        orig_pick = model.pick_seeds

        def pick(*a, _orig=orig_pick, **k):
            caps["seeds"] = _orig(*a, **k)
            return caps["seeds"]
        model.pick_seeds = pick
        for N, seed in ((700, 83), (2000, 84)):
            b = O.synthetic_batch([seed], N=N, T=196, kind="kitti")
            data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
                    "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"]), "testing": True}
            res = model(data)
            tag = f"{wset}_{N}_{seed}"
            lg = caps["logits"].squeeze(1)
            out[f"logits_{tag}"] = _np(lg)
            out[f"seeds_{tag}"] = _np(caps["seeds"]).astype(np.int32)
            out[f"final_trans_{tag}"] = _np(res["final_trans"])
            out[f"final_labels_{tag}"] = _np(res["final_labels"]).astype(np.uint8)
            out[f"gt_trans_{tag}"] = _np(b["gt_trans"])
            if wset == "stress":
                cases.append([N, seed])
            print("F22", tag, "logit range", float(lg.min()), float(lg.max()), "positive:", int((lg > 0).sum()),
                  "inliers:", int(res["final_labels"].sum()), "max|T - T_gt|:",
                  float((res["final_trans"] - b["gt_trans"]).abs().max()))
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLD, "f22_kitti_branch.npz"), sigma_d=1.2, tau=1.2, layer0_div=13.0, **out)


F23_CASES = {
    # name: (class, depth, dim, latent_dim, cross_heads, latent_heads, cross_dim_head, latent_dim_head, weight_tie_layers, pe, B, N, T)
    "fl_d2_h2": ("fl", 2, 128, 128, 2, 4, 32, 32, False, True, 2, 70, 33),
    "fl_tied": ("fl", 3, 128, 128, 1, 2, 64, 48, True, True, 1, 97, 40),
    "fl_w96": ("fl", 1, 96, 96, 3, 2, 16, 24, False, False, 2, 45, 12),
    "pio_d1": ("pio", 1, 128, 256, 2, 8, 64, 32, False, True, 1, 130, 50),
}


def gen_f23(fl, pio):
    """F23 [r5]: the reference's FusionLayer (fusion_layer.py:131-201) and DGR PerceiverIO (perceiver_io.py:139-221) in
    configurations GMF never instantiates - latent self-attention layers (depth 1 .. 3), several heads, tied layers, widths other
    than 128 - for the general forward of gmf_amd.FusionLayer / PerceiverIO.  Weights seeded by state_dict key from the MODULE'S
    OWN key list (tied layers: the shared tensors get the seed of their FIRST name); inputs N(0, 1)."""
    out, keys = {}, {}
    for name, (cls, depth, dim, lat, ch, lh, cdh, ldh, tie, pe, B, N, T) in F23_CASES.items():
        mod = (fl.FusionLayer if cls == "fl" else pio.PerceiverIO)(depth=depth, dim=dim, latent_dim=lat, cross_heads=ch, latent_heads=lh,
                                                                    cross_dim_head=cdh, latent_dim_head=ldh, weight_tie_layers=tie, pe=pe).eval()
        own = mod.state_dict()
        first = {}
        for k, v in own.items():                     # tied layers: one tensor under several names
            first.setdefault(v.data_ptr(), k)
        sd = {}
        for k, v in own.items():
            src = first[v.data_ptr()]
            sd[k] = O.seeded_state_dict({src: tuple(v.shape)}, seed=123)[src]
        mod.load_state_dict(sd)
        r = np.random.default_rng([123, N, T])
        x = torch.from_numpy(r.normal(0, 1, (B, N, lat)).astype(np.float32))
        ctx = torch.from_numpy(r.normal(0, 1, (B, T, dim)).astype(np.float32))
        with torch.no_grad():
            y = mod(ctx, queries_encoder=x)
        out[f"out_{name}"] = _np(y)
        keys[name] = sorted(own.keys())
        print("F23", name, tuple(y.shape), "|out|max", float(y.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "f23_fusion_layer_general.npz"), seed=123, **out)
    with open(os.path.join(GOLD, "f23_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)


F24_CASES = {"c64_h2": (64, 2, 2, 90, 20), "c128_h4": (128, 4, 1, 150, 33), "c32_h1": (32, 1, 2, 40, 7)}     # channels, heads, B, N, T


def gen_f24(pdsc):
    """F24 [r5]: the reference's NonLocalBlock (PointDSC.py:10-74) with widths / head counts GMF never instantiates (its constructor
    takes num_channels and num_heads).  eval() mode, seeded weights and BatchNorm statistics by key, compat matrix from seeded points."""
    out = {}
    for name, (C, H, B, N, T) in F24_CASES.items():
        blk = pdsc.NonLocalBlock(num_channels=C, num_heads=H).eval()
        sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in blk.state_dict().items()}, seed=124)
        blk.load_state_dict(sd)
        r = np.random.default_rng([124, C, N])
        feat = torch.from_numpy(r.normal(0, 1, (B, C, N)).astype(np.float32))
        img = torch.from_numpy(r.normal(0, 1, (B, T, C)).astype(np.float32))
        src = torch.from_numpy(r.uniform(0, 3, (B, N, 3)).astype(np.float32))
        tgt = src + torch.from_numpy(r.normal(0, 0.05, (B, N, 3)).astype(np.float32))
        compat, _ = O.compat_matrix(src, tgt, 0.1)
        with torch.no_grad():
            y = blk(feat, compat, img)
        out[f"out_{name}"] = _np(y)
        print("F24", name, tuple(y.shape), "|out|max", float(y.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "f24_nonlocal_block_general.npz"), seed=124, **out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f24":
        gen_f24(_import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f23":
        _, fl, _, pio, _ = _import_reference()
        gen_f23(fl, pio)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f22":
        gen_f22(_import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f9b":
        gen_f9b()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f20":
        gen_f20(_import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f21":
        gen_f21(_import_reference()[4])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f19":
        gen_f19(_import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f18":
        r_ = _import_reference()
        gen_f18(r_[1], r_[3])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f17":
        gen_f17(_import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f15":
        _import_reference()
        gen_f15()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f16":
        gen_f16(_import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f14":
        gen_f14(_import_reference()[0])
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "f13":
        gen_f13(_import_reference()[4])
        return
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    pdsc, fl, common, pio, reg = _import_reference()
    os.makedirs(GOLD, exist_ok=True)

    # ---------------- F1: Fusion-1 -------------------------------------------------
    shapes = O.fusion_layer_shapes("", 128, 128, 64, pe=False)
    sd = O.seeded_state_dict(shapes, seed=101)
    ref = fl.FusionLayer(depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8,
                         cross_dim_head=64, latent_dim_head=64).eval()
    ref.load_state_dict(sd)
    out = {}
    for T in (12, 196, 300):
        b = O.synthetic_batch([11, 12], N=8, T=T)
        out[f"out_T{T}"] = _np(ref(b["p_tokens"], queries_encoder=b["q_tokens"]))
    np.savez_compressed(os.path.join(GOLD, "f1_fusion1.npz"), seed=101, pair_seeds=[11, 12], **out)

    # ---------------- F2: one Fusion-2 layer (with LCPE) ---------------------------
    shapes = O.fusion_layer_shapes("", 128, 128, 64, pe=True)
    sd = O.seeded_state_dict(shapes, seed=102)
    ref = fl.FusionLayer(depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8,
                         cross_dim_head=64, latent_dim_head=64, pe=True).eval()
    ref.load_state_dict(sd)
    out = {}
    for N, T in ((64, 12), (257, 196), (1000, 196), (33, 1), (1, 7)):
        r = np.random.default_rng([102, N, T])
        x = torch.from_numpy(r.normal(0, 1, (1, N, 128)).astype(np.float32))
        ctx = torch.from_numpy(r.normal(0, 1, (1, T, 128)).astype(np.float32))
        caps = {}
        h1 = ref.cpe.register_forward_hook(lambda m, i, o: caps.__setitem__("cpe", o))
        h2 = ref.cross_attend_blocks[0].register_forward_hook(lambda m, i, o: caps.__setitem__("attn", o))
        y = ref(ctx, queries_encoder=x)
        h1.remove(), h2.remove()
        out[f"out_N{N}_T{T}"] = _np(y)
        if N <= 257:
            out[f"xpe_N{N}_T{T}"] = _np(caps["cpe"][0])
            out[f"ctxpe_N{N}_T{T}"] = _np(caps["cpe"][1])
            out[f"attn_N{N}_T{T}"] = _np(caps["attn"])   # Attention output before the residual add
    np.savez_compressed(os.path.join(GOLD, "f2_fusion2.npz"), seed=102, **out)

    # ---------------- F9: DGR PerceiverIO, latent 256 / dim 128 / head 128 ----------
    shapes = O.fusion_layer_shapes("", 128, 256, 128, pe=True, out_to_query=True)
    sd = O.seeded_state_dict(shapes, seed=109)
    ref = pio.PerceiverIO(depth=0, dim=128, latent_dim=256, cross_heads=1, latent_heads=8,
                          cross_dim_head=128, latent_dim_head=128, pe=True).eval()
    ref.load_state_dict(sd)
    out = {}
    for M, T in ((100, 12), (515, 300)):
        r = np.random.default_rng([109, M, T])
        x = torch.from_numpy(r.normal(0, 1, (1, M, 256)).astype(np.float32))
        ctx = torch.from_numpy(r.normal(0, 1, (1, T, 128)).astype(np.float32))
        out[f"out_M{M}_T{T}"] = _np(ref(ctx, queries_encoder=x))
    np.savez_compressed(os.path.join(GOLD, "f9_dgr_perceiver.npz"), seed=109, **out)

    # ---------------- full-model weights ------------------------------------------
    full_shapes = O.pointdsc_shapes(6, 12, 128)
    sd_full = O.seeded_state_dict(full_shapes, seed=7)
    model = build_ref_pointdsc(pdsc, sd_full)

    # ---------------- F3: one NonLocalBlock ----------------------------------------
    blk = model.encoder.blocks["NonLocal_layer_3"]
    b = O.synthetic_batch([31, 32], N=257, T=196)
    r = np.random.default_rng([103])
    feat = torch.from_numpy(r.normal(0, 1, (2, 257, 128)).astype(np.float32))   # token-major
    img = torch.from_numpy(r.normal(0, 1, (2, 196, 128)).astype(np.float32))
    src_d = torch.norm(b["src_keypts"][:, :, None] - b["src_keypts"][:, None], dim=-1)
    tgt_d = torch.norm(b["tgt_keypts"][:, :, None] - b["tgt_keypts"][:, None], dim=-1)
    compat = torch.clamp(1.0 - (src_d - tgt_d) ** 2 / 0.1 ** 2, min=0)
    y = blk(feat.permute(0, 2, 1).contiguous(), compat, img).permute(0, 2, 1)
    np.savez_compressed(os.path.join(GOLD, "f3_nonlocal_block.npz"), seed=7, layer=3, pair_seeds=[31, 32],
                        feat=_np(feat), img=_np(img), out=_np(y), compat_row5=_np(compat[0, 5]))

    # ---------------- F4: encoder features + logits; F10: whole forward ------------
    out = {}
    for N, seeds in ((64, [41]), (257, [42, 43]), (1000, [44])):
        b = O.synthetic_batch(seeds, N=N, T=196)
        caps = {}
        h1 = model.classification.register_forward_hook(lambda m, i, o: caps.__setitem__("logits", o))
        h2 = model.encoder.register_forward_hook(lambda m, i, o: caps.__setitem__("feat", o))
        res_list = []
        for bi in range(len(seeds)):       # test mode is B=1 (PointDSC.py:279,504)
            data = {"corr_pos": b["corr_pos"][bi:bi + 1], "src_keypts": b["src_keypts"][bi:bi + 1],
                    "tgt_keypts": b["tgt_keypts"][bi:bi + 1],
                    "p_image": _tok_to_image(b["p_tokens"][bi:bi + 1]),
                    "q_image": _tok_to_image(b["q_tokens"][bi:bi + 1]), "testing": True}
            res = model(data)
            res_list.append((res, caps["logits"].squeeze(1).clone(), caps["feat"].permute(0, 2, 1).clone()))
        h1.remove(), h2.remove()
        out[f"logits_N{N}"] = np.concatenate([_np(r[1]) for r in res_list])
        out[f"final_trans_N{N}"] = np.concatenate([_np(r[0]["final_trans"]) for r in res_list])
        out[f"final_labels_N{N}"] = np.concatenate([_np(r[0]["final_labels"]) for r in res_list])
        out[f"gt_trans_N{N}"] = _np(b["gt_trans"])
        feat = np.concatenate([_np(r[2]) for r in res_list])
        if N <= 257:
            out[f"feat_N{N}"] = feat
        else:
            out[f"feat_rows_N{N}"] = feat[:, ::50]
            out[f"feat_sum_N{N}"] = feat.astype(np.float64).sum(axis=(1, 2))
        out[f"pair_seeds_N{N}"] = np.array(seeds)
        # batched train-mode call (B=2): logits must be batch independent
        if len(seeds) == 2:
            data = {"corr_pos": b["corr_pos"], "src_keypts": b["src_keypts"], "tgt_keypts": b["tgt_keypts"],
                    "p_image": _tok_to_image(b["p_tokens"]), "q_image": _tok_to_image(b["q_tokens"])}
            res = model(data)
            out[f"train_logits_N{N}"] = _np(res["final_labels"])
            out[f"train_final_trans_N{N}"] = _np(res["final_trans"])
    np.savez_compressed(os.path.join(GOLD, "f4_f10_pointdsc.npz"), seed=7, **out)

    # ---------------- F5: cal_seed_trans with fixed seeds; F7 refinement trace ------
    out = {}
    N = 400
    b = O.synthetic_batch([51], N=N, T=12)
    r = np.random.default_rng([105])
    feat = torch.from_numpy(r.normal(0, 1, (1, N, 128)).astype(np.float32))
    # make inlier features cluster so the seed neighbourhoods are meaningful
    inl = b["gt_labels"][0] > 0
    feat[0, inl] += 2.5 * torch.from_numpy(r.normal(0, 1, (1, 128)).astype(np.float32))
    feat_n = torch.nn.functional.normalize(feat, p=2, dim=-1)
    scores = torch.from_numpy(r.normal(0, 1, (1, N)).astype(np.float32)) + 2.0 * b["gt_labels"]
    src_d = torch.norm(b["src_keypts"][:, :, None] - b["src_keypts"][:, None], dim=-1)
    seeds = model.pick_seeds(src_d, scores, R=0.10, max_num=int(N * 0.1))
    seed_T, fit, final_T, labels = model.cal_seed_trans(seeds, feat_n, b["src_keypts"], b["tgt_keypts"])
    k = 40
    knn_idx = common.knn(feat_n, k=k, ignore_self=True, normalized=True).gather(
        dim=1, index=seeds[:, :, None].expand(-1, -1, k))
    refined = model.post_refinement(final_T.clone(), b["src_keypts"], b["tgt_keypts"])
    np.savez_compressed(os.path.join(GOLD, "f5_f7_pose_head.npz"), pair_seed=51, N=N,
                        feat_n=_np(feat_n), scores=_np(scores), seeds=_np(seeds), knn_idx=_np(knn_idx),
                        seed_trans=_np(seed_T), fitness=_np(fit), final_trans=_np(final_T),
                        labels=_np(labels), refined=_np(refined), gt_trans=_np(b["gt_trans"]))

    # ---------------- F6: rigid_transform_3d ---------------------------------------
    r = np.random.default_rng([106])
    n, k = 64, 40
    A = r.uniform(0, 3, (n, k, 3)).astype(np.float32)
    Bm = np.empty_like(A)
    for i in range(n):
        R = O.random_rotation(r)
        Bm[i] = A[i] @ R.T + r.uniform(-1, 1, 3) + r.normal(0, 0.02, (k, 3))
    w = r.uniform(0, 1, (n, k)).astype(np.float32)
    w[:8, ::3] = 0.0                       # zero weights
    w[8:12, 1::4] = -0.5                   # negative weights (clipped in place by the reference)
    A[12:16, :, 2] = 0.3 + 0.02 * r.normal(size=(4, k))   # thin (near-planar) clouds
    Bm[12:16] = A[12:16] @ O.random_rotation(r).T.astype(np.float32) + 0.5
    Bm[16:20] = Bm[16:20] * np.array([1, 1, -1], np.float32)   # reflected target: forces det fix
    Tref = common.rigid_transform_3d(torch.from_numpy(A), torch.from_numpy(Bm), torch.from_numpy(w.copy()))
    Tref_now = common.rigid_transform_3d(torch.from_numpy(A), torch.from_numpy(Bm))
    np.savez_compressed(os.path.join(GOLD, "f6_rigid_transform.npz"), A=A, B=Bm, w=w, T=_np(Tref), T_noweight=_np(Tref_now))

    # ---------------- F8: DGR weighted_procrustes ----------------------------------
    out = {}
    for N in (10, 1000, 8000):
        r = np.random.default_rng([108, N])
        X = r.uniform(0, 3, (N, 3)).astype(np.float32)
        R = O.random_rotation(r)
        t = r.uniform(-0.5, 0.5, 3)
        Y = (X @ R.T + t).astype(np.float32)
        nout = int(N * 0.7)
        Y[:nout] = r.uniform(0, 3, (nout, 3)).astype(np.float32)
        logit = r.normal(-3.0, 1.0, (N, 1)).astype(np.float32)
        logit[nout:] = r.normal(3.0, 1.0, (N - nout, 1)).astype(np.float32)
        wgt = O.dgr_inlier_weights(torch.from_numpy(logit))
        Rr, tr = reg.weighted_procrustes(torch.from_numpy(X), torch.from_numpy(Y), wgt, np.finfo(np.float32).eps)
        out[f"X_{N}"], out[f"Y_{N}"], out[f"w_{N}"] = X, Y, _np(wgt)
        out[f"R_{N}"], out[f"t_{N}"] = _np(Rr), _np(tr)
        out[f"Rgt_{N}"], out[f"tgt_{N}"] = R.astype(np.float32), t.astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "f8_weighted_procrustes.npz"), **out)

    # ---------------- F11: image encoder (ResNet-34 -> layer2), SURVEY section 8 row f-1 -----------------
    import models.Img_Encoder as ie
    enc = ie.ImageEncoder().eval()
    shapes_ie = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    sd_ie = O.seeded_state_dict(shapes_ie, seed=111, gain=1.0)
    enc.load_state_dict(sd_ie)
    r = np.random.default_rng([111])
    img = torch.from_numpy(r.uniform(0, 1, (2, 3, 120, 160)).astype(np.float32))
    feat = enc(img)                                   # [2,128,15,20]
    tok = feat.view(2, 128, -1).permute(0, 2, 1)      # PointDSC.py:129-131
    np.savez_compressed(os.path.join(GOLD, "f11_image_encoder.npz"), seed=111, tokens=_np(tok).astype(np.float32))

    # ---------------- F12: descriptor matching (row f-2) ---------------------------------------------
    import core.knn as dgr_knn
    out = {}
    for d, N0, N1 in ((32, 500, 700), (33, 257, 300)):
        r = np.random.default_rng([112, d])
        F1 = r.normal(0, 1, (N1, d)).astype(np.float32)
        F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
        F0 = F1[r.integers(0, N1, N0)] + 0.15 * r.normal(0, 1, (N0, d)).astype(np.float32)
        F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
        F0, F1 = F0.astype(np.float32), F1.astype(np.float32)
        # PointDSC: the numpy expressions of datasets/ThreeDMatch.py:164-166
        distance = np.sqrt(2 - 2 * (F0 @ F1.T) + 1e-6)
        out[f"pdsc_idx_{d}"], out[f"pdsc_dis_{d}"] = np.argmin(distance, axis=1), np.min(distance, axis=1)
        i1, d1 = dgr_knn.find_knn_gpu(torch.from_numpy(F0), torch.from_numpy(F1), nn_max_n=250, knn=1, return_distance=True)
        i2, d2 = dgr_knn.find_knn_gpu(torch.from_numpy(F0), torch.from_numpy(F1), nn_max_n=-1, knn=1, return_distance=True)
        out[f"dgr_idx_chunk_{d}"], out[f"dgr_dis_chunk_{d}"] = _np(i1), _np(d1)
        out[f"dgr_idx_{d}"], out[f"dgr_dis_{d}"] = _np(i2), _np(d2)
        out[f"F0_{d}"], out[f"F1_{d}"] = F0, F1
    np.savez_compressed(os.path.join(GOLD, "f12_descriptor_matching.npz"), **out)

    # ---------------- state_dict surface (key names + shapes) of the reference modules -------------
    import json
    ref_full = pdsc.PointDSC(in_dim=6, num_layers=12, num_channels=128)
    keys = {"pointdsc": {k: list(v.shape) for k, v in ref_full.state_dict().items()},
            "fusion_layer_pe": {k: list(v.shape) for k, v in fl.FusionLayer(
                depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8, cross_dim_head=64,
                latent_dim_head=64, pe=True).state_dict().items()},
            "perceiver_io_128": {k: list(v.shape) for k, v in pio.PerceiverIO(
                depth=0, dim=128, latent_dim=128, cross_heads=1, latent_heads=8, cross_dim_head=64,
                latent_dim_head=64).state_dict().items()}}
    json.dump(keys, open(os.path.join(GOLD, "state_dict_keys.json"), "w"), indent=0, sort_keys=True)

    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
