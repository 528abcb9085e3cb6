"""Dev probe (GPU box): golden F19 - the HIP training step against the reference's gradients, per-parameter error summary."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd
from gmf_amd import synthetic
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "f19_training_step.npz"))
for tag in ("def", "bal"):
    cfg = g[f"{tag}_cfg"]; balanced, N, seeds = bool(cfg[0]), int(cfg[1]), [int(v) for v in cfg[2:]]
    sd = synthetic.seeded_state_dict(synthetic.pointdsc_shapes(6, 3, 128), seed=7)
    m = gmf_amd.PointDSC(num_layers=3); m.load_state_dict(sd, strict=False); m = m.cuda().train()
    b = synthetic.synthetic_batch(seeds, N=N, T=40)
    data = {k: b[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts", "p_tokens", "q_tokens")}
    gt = b["gt_labels"].cuda()
    res = m(data)
    cl = gmf_amd.ClassificationLoss(balanced=balanced)(res["final_labels"], gt)
    sm = gmf_amd.SpectralMatchingLoss(balanced=balanced)(res["M"], gt)
    loss = 1.0 * cl["loss"] + 1.0 * sm
    loss.backward()
    print(tag, "losses", float(cl["loss"]), float(sm), "ref", g[f"{tag}_losses"], "dlogit", float((res["final_labels"].detach().cpu() - torch.from_numpy(g[f"{tag}_logits"])).abs().max()))
    names = list(g[f"{tag}_grad_names"]); stats = g[f"{tag}_grad_stats"]; heads = g[f"{tag}_grad_heads"]
    params = dict(m.named_parameters())
    worst = []
    for i, n in enumerate(names):
        p = params[n]
        if p.grad is None:
            print("NO GRAD", n); continue
        gr = p.grad.double().reshape(-1).cpu()
        e_norm = abs(float(gr.norm()) - stats[i, 1]) / max(stats[i, 1], 1e-30)
        hd = gr[:16].numpy(); k = min(16, gr.numel())
        e_head = np.abs(hd[:k] - heads[i, :k]).max() / max(stats[i, 2], 1e-30)
        e_sum = abs(float(gr.sum()) - stats[i, 0]) / max(stats[i, 1], 1e-30)
        worst.append((max(e_norm, e_head), e_norm, e_head, e_sum, n, stats[i, 1]))
    worst.sort(reverse=True)
    for w in worst[:8]: print("  %.2e norm %.2e head %.2e sum %.2e  %s |g|=%.3e" % w)
    print("  median err %.2e, params %d, missing %d" % (np.median([w[0] for w in worst]), len(worst), len(names) - len(worst)))
    bn = m.encoder.blocks["NonLocal_layer_2"].fc_message[1]
    print("  bn running err", float((torch.stack([bn.running_mean, bn.running_var]).cpu() - torch.from_numpy(g[f"{tag}_bn_running"])).abs().max()))
