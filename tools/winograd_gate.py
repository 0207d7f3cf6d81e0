"""Round 5 gate: what would Winograd F(2x2, 3x3) do to the NUMBERS of the stride-1 3x3 convolutions
(hallo/models/resnet.py:388,405) if the point-wise products ran on the bf16 / fp16 MFMA?

    python tools/winograd_gate.py            # per-convolution error at the UNet's widths, CPU, ~1 min

Model of the kernel that would be built: input transform V = B^T d B and output transform Y = A^T M A in fp32, the
transformed input V and the (offline, fp32) transformed weights U = G g G^T rounded ONCE to the storage type, products
accumulated in fp32 over Cin, the result rounded to the storage type.  Compared with the direct form the library runs today
(operands exact in the storage type, fp32 accumulation, one output rounding), both against an fp64 evaluation.
The performance side of the gate is in DESIGN.md section 7.4 (accumulator footprint x4 -> LDS operand bytes x1.78 on a
kernel that is LDS-bound already); this script is the numerical side."""
import json
import sys

import torch
import torch.nn.functional as F

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd_conv(x, w, dtype):
    """x [N, Cin, H, W] (values representable in dtype), w [Cout, Cin, 3, 3]; pad 1, stride 1; H, W even."""
    N, Ci, H, W = x.shape
    xp = F.pad(x.double(), (1, 1, 1, 1))
    # 4x4 patches at stride 2: [N, Ci, H/2, W/2, 4, 4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum("ij,nchwjk,lk->nchwil", BT, d, BT).float()          # fp32 transform (exact sums of <= 4 storage values)
    V = V.to(dtype).double()                                             # ONE rounding to the MFMA operand type
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G).float().to(dtype).double()
    M = torch.einsum("nchwil,ocil->nohwil", V, U)                        # fp32 accumulation on the MFMA (fp64 here: an upper bound on quality)
    Y = torch.einsum("ij,nohwjk,lk->nohwil", AT, M, AT)                  # [N, Co, H/2, W/2, 2, 2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, H, W)
    return Y.float().to(dtype).double()


def direct_conv(x, w, dtype):
    return F.conv2d(x.double(), w.double(), padding=1).float().to(dtype).double()


def main():
    torch.manual_seed(0)
    rows = []
    for dtype in (torch.bfloat16, torch.float16):
        for (Ci, Co, S) in ((320, 320, 32), (640, 640, 16), (1280, 1280, 8), (128, 128, 64)):
            x = F.silu(torch.randn(2, Ci, S, S)).to(dtype).float()          # post GroupNorm + SiLU statistics
            w = (torch.randn(Co, Ci, 3, 3) * (1.0 / (9 * Ci) ** 0.5)).to(dtype).float()
            ref = F.conv2d(x.double(), w.double(), padding=1)
            rel = lambda y: float((y - ref).norm() / ref.norm())
            mx = lambda y: float((y - ref).abs().max() / ref.pow(2).mean().sqrt())
            yd, yw = direct_conv(x, w, dtype), winograd_conv(x, w, dtype)
            rows.append(dict(dtype=str(dtype), Cin=Ci, Cout=Co, size=S, direct_rel_l2=rel(yd), winograd_rel_l2=rel(yw),
                             ratio=rel(yw) / rel(yd), direct_max_over_rms=mx(yd), winograd_max_over_rms=mx(yw)))
            print(rows[-1], flush=True)
    json.dump(rows, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None


if __name__ == "__main__":
    main()
