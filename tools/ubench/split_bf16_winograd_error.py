"""What do split-bf16 products cost an F(4x4,3x3) convolution in accuracy?  (CPU, numpy; HISTORY §10-6)

The transform-domain operands U = G g G^T (filters) and V = B^T d B (data) are fp32.  On the bf16 matrix pipe each is written as a
sum of bf16 terms (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)) and the element-wise product U (.) V summed over
input channels becomes several bf16 x bf16 products accumulated in fp32:
    2 terms, 3 products: hi*hi + hi*mid + mid*hi                      (16/3 = 5.3x the fp32 MFMA rate)
    3 terms, 6 products: + mid*mid + hi*lo + lo*hi                    (16/6 = 2.7x)
Reference: the same convolution in float64.  Compared: fp32 direct, fp32 F(4x4) (what the kernels do today) and the two split
schemes, for a layer like the decoder's (Ci = 256, 64 output channels, 32 x 32 pixels).
"""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7fff
    return ((u + r) & 0xffff0000).view(np.float32)


G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def conv_ref(x, w):                      # x [Ci,H,W] (already padded by 1), w [Co,Ci,3,3] -> [Co,H-2,W-2], float64
    Co, Ci = w.shape[:2]
    H, W = x.shape[1] - 2, x.shape[2] - 2
    y = np.zeros((Co, H, W))
    for kh in range(3):
        for kw in range(3):
            y += np.einsum('oc,chw->ohw', w[:, :, kh, kw], x[:, kh:kh + H, kw:kw + W])
    return y


def wino(x, w, dt, split=None):
    """F(4x4,3x3) with transforms in `dt`; split = None (plain products in dt) | (terms, products)"""
    Co, Ci = w.shape[:2]
    H, W = x.shape[1] - 2, x.shape[2] - 2
    U = np.einsum('ij,ocjk,lk->ocil', G.astype(dt), w.astype(dt), G.astype(dt)).astype(dt)          # [Co,Ci,6,6]
    y = np.zeros((Co, H, W), dtype=np.float64)
    for th in range(0, H, 4):
        for tw in range(0, W, 4):
            d = x[:, th:th + 6, tw:tw + 6].astype(dt)
            V = np.einsum('ij,cjk,lk->cil', BT.astype(dt), d, BT.astype(dt)).astype(dt)             # [Ci,6,6]
            if split is None:
                M = np.einsum('ocil,cil->oil', U, V).astype(dt)
            else:
                terms, prods = split
                uh = bf16(U); um = bf16(U - uh); ul = bf16(U - uh - um)
                vh = bf16(V); vm = bf16(V - vh); vl = bf16(V - vh - vm)
                pairs = [(uh, vh), (uh, vm), (um, vh)] + ([(um, vm), (uh, vl), (ul, vh)] if prods == 6 else [])
                M = np.zeros((Co, 6, 6), dtype=np.float32)
                for a, b in pairs[::-1]:                                  # small terms first, fp32 accumulation
                    M += np.einsum('ocil,cil->oil', a.astype(np.float32), b.astype(np.float32)).astype(np.float32)
            y[:, th:th + 4, tw:tw + 4] = np.einsum('ij,ojk,lk->oil', AT.astype(dt), M.astype(dt), AT.astype(dt))
    return y


for Ci, Co, S, scale_note in ((256, 16, 32, 'decoder-like'), (64, 16, 32, 'level 1'), (1024, 8, 16, 'bottleneck concat')):
    x = rng.standard_normal((Ci, S + 2, S + 2)) * (rng.random((Ci, 1, 1)) + 0.5)
    x[:, 0, :] = x[:, -1, :] = 0; x[:, :, 0] = x[:, :, -1] = 0
    x = np.maximum(x, 0)                                      # post-ReLU activations
    w = rng.standard_normal((Co, Ci, 3, 3)) / np.sqrt(9 * Ci)
    ref = conv_ref(x, w)
    sc = np.abs(ref).max()
    x32, w32 = x.astype(np.float32), w.astype(np.float32)
    ref32in = conv_ref(x32.astype(np.float64), w32.astype(np.float64))      # what any fp32 kernel can at best return
    rows = [('fp32 direct (fp32 products, fp32 sums)', conv_ref(x32, w32).astype(np.float32) if False else None)]
    out = {}
    out['fp32 F(4x4) (today)'] = wino(x32, w32, np.float32)
    out['split bf16 F(4x4), 2 terms / 3 products'] = wino(x32, w32, np.float32, (2, 3))
    out['split bf16 F(4x4), 3 terms / 6 products'] = wino(x32, w32, np.float32, (3, 6))
    print('Ci %4d Co %3d %dx%d (%s): max |err| / max |y| against float64 of the same fp32 inputs' % (Ci, Co, S, S, scale_note))
    for k, v in out.items():
        print('    %-46s %.2e   (rms %.2e)' % (k, np.abs(v - ref32in).max() / sc, np.sqrt(np.mean((v - ref32in) ** 2)) / sc))
