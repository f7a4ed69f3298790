"""CPU restatement of the key-frame body of the reference's preprocess/knn_gaussian.py:116-132
-- TEST INFRASTRUCTURE ONLY (only tests/ may import it).  Same tensor expressions as the
reference, on the packed arrays it uses (visible Gaussians only)."""
import torch


def backproject_frame(gaussian_masks, means2d_packed, depths_packed, gaussian_ids, depth, atrb_masks, mask_valids):
    """gaussian_masks [N,M] bool is updated in place and returned."""
    H, W = depth.shape
    M = gaussian_masks.shape[1]
    xy = means2d_packed.cpu().long()  # :117
    im = ((xy >= 0) & (xy < torch.tensor([W, H]))).all(-1)  # :118
    xy = xy[im]
    delta_depth = depth[xy[:, 1], xy[:, 0]] - depths_packed[im]  # :121
    dm = (-depth[xy[:, 1], xy[:, 0]] * 0.1 < delta_depth) & (delta_depth < depth[xy[:, 1], xy[:, 0]] * 1)  # :122
    xy = xy[dm]
    mask = atrb_masks[..., :-1] & mask_valids[..., :-1][None, None, ...]  # :128
    m = mask[xy[:, 1], xy[:, 0]]  # (n, M)
    ids = gaussian_ids[im][dm]
    ids = ids[..., None].expand(-1, M)[m]
    gaussian_masks[ids, m.nonzero()[:, -1]] = True  # :132
    return gaussian_masks
