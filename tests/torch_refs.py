"""Plain-torch float64 reference implementations (autograd-capable) of the fused ops, used
by the GPU tests for value AND gradient parity.  They restate the published definitions
independently of both the HIP kernels and the C oracle."""
import math

import torch


def roi_align_ref(inp, rois, out_size, scale, sampling_ratio=-1, aligned=False):
    inp = inp.double()
    K = rois.shape[0]
    _, C, H, W = inp.shape
    ph = pw = out_size
    out = []
    off = 0.5 if aligned else 0.0
    for k in range(K):
        b = int(rois[k, 0].item())
        x1, y1, x2, y2 = [float(v) * scale - off for v in rois[k, 1:]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bw, bh = rw / pw, rh / ph
        gh = sampling_ratio if sampling_ratio > 0 else math.ceil(rh / ph)
        gw = sampling_ratio if sampling_ratio > 0 else math.ceil(rw / pw)
        acc = torch.zeros((C, ph, pw), dtype=torch.float64, device=inp.device)
        for p_h in range(ph):
            for p_w in range(pw):
                for iy in range(gh):
                    y = y1 + p_h * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = x1 + p_w * bw + (ix + 0.5) * bw / gw
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        yy, xx = max(y, 0.0), max(x, 0.0)
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yh = yl = H - 1; yy = float(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1; xx = float(xl)
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        hy, hx = 1 - ly, 1 - lx
                        acc[:, p_h, p_w] = acc[:, p_h, p_w] + hy * hx * inp[b, :, yl, xl] + hy * lx * inp[b, :, yl, xh] \
                            + ly * hx * inp[b, :, yh, xl] + ly * lx * inp[b, :, yh, xh]
        out.append(acc / max(gh * gw, 1))
    return torch.stack(out)


def dafm_ref(q, k, v, de, sigma, scale):
    e = torch.softmax(-(de / sigma), dim=1)
    att = torch.softmax((q @ k.T) * e * scale, dim=1)
    return att @ v, att


def gatv2_ref(x, edge_index, lin_l, lin_r, att, bias, heads, out_ch, slope=0.2, concat=False, edge_scale=None):
    """Dense-loop GATv2 with self loops re-added; edge_scale: dict {(j, i): (H,) tensor} or None."""
    n = x.shape[0]
    xl = lin_l(x).view(n, heads, out_ch)
    xr = lin_r(x).view(n, heads, out_ch)
    src, dst = edge_index[0].tolist(), edge_index[1].tolist()
    pairs = [(j, i) for j, i in zip(src, dst) if j != i] + [(i, i) for i in range(n)]
    out = []
    a = att.view(heads, out_ch)
    for i in range(n):
        js = [j for j, t in pairs if t == i]
        z = torch.nn.functional.leaky_relu(xl[js] + xr[i][None], slope)      # (deg, H, C)
        e = (z * a[None]).sum(-1)                                           # (deg, H)
        al = torch.softmax(e, dim=0)
        if edge_scale is not None:
            al = al * torch.stack([edge_scale[(j, i)] for j in js])
        out.append((al[:, :, None] * xl[js]).sum(0))
    out = torch.stack(out)
    out = out.reshape(n, heads * out_ch) if concat else out.mean(1)
    return out + bias
