"""CPU numerics probe (no kernel): how much rounding error would a Winograd F(2x2, 3x3) form of the 3x3 spatial taps add?
Direct form: operands rounded to the storage dtype, exact products, fp32 accumulation (what conv_kernel.h does).
Winograd form: V = B^T d B and U = G g G^T evaluated in fp32 from the SAME rounded operands, then rounded to the storage dtype
(the MFMA operand format), 16 element-wise channel contractions in fp32, output transform A^T m A in fp32.
Both are compared with the fp64 convolution of the rounded operands (the error the FORM adds on top of operand rounding), on
post-SiLU-like activations and default-init weights of a 128 -> 128 layer.
usage: python tools/probes/winograd_numerics.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
B_T = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
A_T = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd(x, w, dt):
    """x [N,C,H,W] (H, W even), w [K,C,3,3], zero padding 1; operands of the 16 contractions rounded to dt"""
    N, C, H, W = x.shape
    K = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                               # [N,C,H/2,W/2,4,4]
    V = torch.einsum("ij,nchwjk,lk->nchwil", B_T.float(), d.float(), B_T.float()).to(dt).float()
    U = torch.einsum("ij,kcjl,ml->kcim", G.float(), w.float(), G.float()).to(dt).float()
    M = torch.einsum("nchwij,kcij->nkhwij", V, U)                        # fp32 accumulation over channels
    Y = torch.einsum("ij,nkhwjl,ml->nkhwim", A_T.float(), M, A_T.float())  # [N,K,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


def main():
    N, C, K, H, W = 2, 128, 128, 32, 32
    a = F.silu(torch.randn(N, C, H, W) * 1.0 + 0.1)
    w = (torch.rand(K, C, 3, 3) * 2 - 1) / (C * 9) ** 0.5
    for dt in (torch.bfloat16, torch.float16):
        ar, wr = a.to(dt), w.to(dt)
        ref = F.conv2d(ar.double(), wr.double(), padding=1)
        direct = F.conv2d(ar.float(), wr.float(), padding=1)
        wino = winograd(ar, wr, dt)
        full = F.conv2d(a.double(), w.double(), padding=1)               # unrounded operands: the error budget of the dtype itself
        rel = lambda y: float((y.double() - ref).norm() / ref.norm())
        mx = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
        print(f"{str(dt)[6:]:9s} operand rounding alone (vs unrounded operands): rel {float((ref - full).norm() / full.norm()):.2e}")
        print(f"{'':9s} direct  form, fp32 accumulation : rel {rel(direct):.2e}  max/|max| {mx(direct):.2e}")
        print(f"{'':9s} Winograd F(2x2,3x3), V and U rounded to {str(dt)[6:]}: rel {rel(wino):.2e}  max/|max| {mx(wino):.2e}")
        print(f"{'':9s} storing the output in {str(dt)[6:]} adds          : rel {rel(direct.to(dt)):.2e}")


if __name__ == "__main__":
    main()
