"""Times the validation-step kernels (row f-4, forward half) against the same math in eager PyTorch-ROCm on the GPU box:
python tools/time_validation.py [B] [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmf_amd  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    dev = "cuda:0"
    f = torch.nn.functional.normalize(torch.randn(B, N, 128, device=dev), dim=-1)
    gt = (torch.rand(B, N, device=dev) < 0.25).float()
    sigma = 0.9
    sm = gmf_amd.SpectralMatchingLoss()
    M = gmf_amd.similarity_matrix(f, sigma)
    print(f"B={B} N={N}  M = {M.numel() * 4 / 1e9:.2f} GB")
    t = timed(lambda: M.fill_(0.5))
    print(f"torch fill_ of M (write baseline)            {t:8.3f} ms   {M.numel() * 4 / t / 1e6:7.0f} GB/s")
    t = timed(lambda: M.sum())
    print(f"torch sum of M (read baseline)               {t:8.3f} ms   {M.numel() * 4 / t / 1e6:7.0f} GB/s")
    t = timed(lambda: gmf_amd.similarity_matrix(f, sigma))
    M = gmf_amd.similarity_matrix(f, sigma)
    print(f"similarity_matrix (HIP, incl. torch.empty)   {t:8.3f} ms   {M.numel() * 4 / t / 1e6:7.0f} GB/s written")
    t = timed(lambda: sm(M, gt))
    print(f"SpectralMatchingLoss(M, gt) (HIP)            {t:8.3f} ms   {M.numel() * 4 / t / 1e6:7.0f} GB/s read")
    t = timed(lambda: sm.from_features(f, sigma, gt))
    print(f"SpectralMatchingLoss.from_features (HIP)     {t:8.3f} ms")

    def torch_M():
        Mt = torch.matmul(f, f.permute(0, 2, 1))
        Mt = torch.clamp(1 - (1 - Mt) / sigma ** 2, min=0, max=1)
        idx = torch.arange(N, device=dev)
        Mt[:, idx, idx] = 0
        return Mt

    def torch_sm(Mt):
        gt_M = ((gt[:, None, :] + gt[:, :, None]) == 2).float()
        for i in range(B):
            gt_M[i].fill_diagonal_(0)
        lp = ((Mt - 1) ** 2 * gt_M).sum(-1).sum(-1) / (torch.relu(gt_M.sum(-1).sum(-1) - 1.0) + 1.0)
        ln = ((Mt - 0) ** 2 * (1 - gt_M)).sum(-1).sum(-1) / (torch.relu((1 - gt_M).sum(-1).sum(-1) - 1.0) + 1.0)
        return torch.mean(lp * 0.5 + ln * 0.5)

    # first backward slice of the training path: forward + backward of the loss from the features (M never written)
    fg = f.detach().clone().requires_grad_(True)
    sg = torch.tensor([float(sigma)], device=dev, requires_grad=True)

    def fwd_bwd():
        fg.grad = None
        sg.grad = None
        sm.from_features(fg, sg, gt).backward()
    t = timed(fwd_bwd)
    print(f"from_features forward + backward (HIP)       {t:8.3f} ms   (backward alone ~ {t - timed(lambda: sm.from_features(f, sigma, gt)):.3f} ms; "
          f"{B * 1024.0 * N * N / 1e12:.2f} TFLOP of algorithmic MFMA work)")

    def torch_fwd_bwd():
        fr = f.detach().clone().requires_grad_(True)
        sr = sg.detach().clone().requires_grad_(True)
        Mt = torch.clamp(1 - (1 - fr @ fr.permute(0, 2, 1)) / sr ** 2, min=0, max=1)
        idx = torch.arange(N, device=dev)
        Mt[:, idx, idx] = 0
        torch_sm(Mt).backward()
    if B * N * N * 4 * 12 < 200e9:
        print(f"eager torch forward + backward               {timed(torch_fwd_bwd, 2):8.3f} ms")

    if B * N * N * 4 * 6 < 200e9:
        t = timed(torch_M, 3)
        print(f"eager torch M (rocBLAS fp32 + 3 passes)      {t:8.3f} ms")
        Mt = torch_M()
        print("   max |M - M_torch| =", float((M - Mt).abs().max()))
        t = timed(lambda: torch_sm(Mt), 3)
        print(f"eager torch SpectralMatchingLoss             {t:8.3f} ms")
        print("   loss HIP / fused / torch:", float(sm(M, gt)), float(sm.from_features(f, sigma, gt)), float(torch_sm(Mt)))
    logits = torch.randn(B, N, device=dev)
    T = torch.eye(4, device=dev).repeat(B, 1, 1)
    pts = torch.rand(B, N, 3, device=dev)
    cl, tl = gmf_amd.ClassificationLoss(), gmf_amd.TransformationLoss()
    print(f"ClassificationLoss (HIP, incl. host read)    {timed(lambda: cl(logits, gt)):8.3f} ms")
    print(f"TransformationLoss (HIP, incl. host read)    {timed(lambda: tl(T, T, pts, pts, logits)):8.3f} ms")


if __name__ == "__main__":
    main()
