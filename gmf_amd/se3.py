"""SE(3) helpers with the reference's names (GMF_PointDSC/utils/SE3.py:43-112).  torch only."""
import torch


def transform(pts, trans):
    """pts [N,3] or [B,N,3]; trans [4,4] or [B,4,4] -> R @ pts + t (SE3.py:43-57)."""
    if pts.dim() == 3:
        return pts @ trans[:, :3, :3].transpose(1, 2) + trans[:, None, :3, 3]
    return pts @ trans[:3, :3].t() + trans[:3, 3]


def decompose_trans(trans):
    """-> (R, t[...,3,1]) (SE3.py:59-71)."""
    if trans.dim() == 3:
        return trans[:, :3, :3], trans[:, :3, 3:4]
    return trans[:3, :3], trans[:3, 3:4]


def integrate_trans(R, t):
    """R [.,3,3], t [.,3,1] -> [.,4,4] (SE3.py:73-96)."""
    if R.dim() == 3:
        T = torch.eye(4, device=R.device, dtype=R.dtype).repeat(R.shape[0], 1, 1)
        T[:, :3, :3] = R
        T[:, :3, 3:4] = t.reshape(-1, 3, 1)
    else:
        T = torch.eye(4, device=R.device, dtype=R.dtype)
        T[:3, :3] = R
        T[:3, 3:4] = t.reshape(3, 1)
    return T


def concatenate(trans1, trans2):
    """trans1 @ trans2 on SE(3) (SE3.py:98-112)."""
    R1, t1 = decompose_trans(trans1)
    R2, t2 = decompose_trans(trans2)
    return integrate_trans(R1 @ R2, R1 @ t2 + t1)
