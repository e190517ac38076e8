"""Independent NumPy statement of tf.image.extract_patches on one channel (TF's documented
SAME/VALID rules), transposed to the reference's feature-major patch matrix
(scripts/quantized_network.py:158-179, :789-797).  Test helper only."""
import numpy as np


def out_dim(size, k, s, d, same):
    if same:
        return -(-size // s)
    keff = k + (k - 1) * (d - 1)
    return max(-(-(size - keff + 1) // s), 0)


def patches(act, c, kh, kw, sh, sw, rh, rw, padding):
    """act: [n][H][W][C] -> [kh*kw][n*oh*ow], column order (image, oy, ox), row order (ky, kx)."""
    n, H, W, _ = act.shape
    same = padding.upper() == "SAME"
    oh, ow = out_dim(H, kh, sh, rh, same), out_dim(W, kw, sw, rw, same)
    pt = pl = 0
    if same:
        pt = max((oh - 1) * sh + kh + (kh - 1) * (rh - 1) - H, 0) // 2
        pl = max((ow - 1) * sw + kw + (kw - 1) * (rw - 1) - W, 0) // 2
    P = np.zeros((kh * kw, n * oh * ow), dtype=np.float32)
    for b in range(n):
        for oy in range(oh):
            for ox in range(ow):
                col = (b * oh + oy) * ow + ox
                for ky in range(kh):
                    for kx in range(kw):
                        iy, ix = oy * sh + ky * rh - pt, ox * sw + kx * rw - pl
                        if 0 <= iy < H and 0 <= ix < W:
                            P[ky * kw + kx, col] = act[b, iy, ix, c]
    return P
