"""CPU numerics probe, end to end (no kernel): the sd3 encoder + decoder of the oracle with every conv's OPERAND PRECISION emulated --
(a) the fast fp32 mode's operands (fp16 + an e5m2 correction of the residual, exact products, fp32 accumulation), direct form;
(b) the same operand format on the TRANSFORMED operands of a Winograd F(2x2, 3x3) form of every stride-1 3x3(x3) conv
    (V = B^T d B and U = G g G^T evaluated in fp32 from the fp32 activations / weights, then split; output transform in fp32).
Reported: max |delta| of the posterior mean and of the reconstruction against the plain fp32 oracle, i.e. what the bound
|delta| <= 1e-3 on the latents would see.  Calibration: the real kernels of form (a) measure 1.6e-4 on cfg 3.
usage: python tools/probes/winograd_e2e_numerics.py [T H W]   (default 5 64 64; full-size networks, seeded default-init weights)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as TF

from oracle import cvvae_oracle as O
from oracle import parity as P

B_T = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
A_T = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


QBITS = int(os.environ.get("QBITS", "14"))


def q(x):
    """the operand the fast fp32 mode multiplies: an fp16 value plus a SCALED bf8 (e5m2) correction of what fp16 dropped -- 11 + 3
    significant bits; emulated as round-to-nearest at QBITS significant bits (an unscaled e5m2 cast would flush the residuals of
    small activations to zero, which the kernels' scaled correction does not)"""
    m, e = torch.frexp(x)
    return torch.ldexp(torch.round(m * (1 << QBITS)) / (1 << QBITS), e)


def wino2d(x, w):
    """x [N,C,H,W] already padded, w [K,C,3,3], 'valid' conv: [N,K,H-2,W-2]"""
    N, C, H, W = x.shape
    Ho, Wo = H - 2, W - 2
    xp = TF.pad(x, (0, Wo % 2, 0, Ho % 2))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = q(torch.einsum("ij,nchwjk,lk->nchwil", B_T, d, B_T))
    U = q(torch.einsum("ij,kcjl,ml->kcim", G, w, G))
    M = torch.einsum("nchwij,kcij->nkhwij", V, U)
    Y = torch.einsum("ij,nkhwjl,ml->nkhwim", A_T, M, A_T)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], Y.shape[2] * 2, Y.shape[3] * 2)
    return Y[:, :, :Ho, :Wo]


class Emu:
    """stands in for torch.nn.functional inside the oracle module"""

    def __init__(self, form):
        self.form = form

    def __getattr__(self, name):
        return getattr(TF, name)

    def conv2d(self, x, w, b=None, stride=1, padding=0):
        if self.form == "wino" and tuple(w.shape[2:]) == (3, 3) and stride in (1, (1, 1)):
            p = padding if isinstance(padding, int) else padding[0]
            y = wino2d(TF.pad(x, (p, p, p, p)), w)
            return y if b is None else y + b.view(1, -1, 1, 1)
        return TF.conv2d(q(x), q(w), b, stride=stride, padding=padding)

    def conv3d(self, x, w, b=None, stride=1, padding=0):
        if self.form == "wino" and tuple(w.shape[2:]) == (3, 3, 3) and stride in (1, (1, 1, 1)):
            p = padding if isinstance(padding, int) else padding[0]
            if p:
                x = TF.pad(x, (p, p, p, p, p, p))
            N, C, T, H, W = x.shape
            out = 0
            for dt in range(3):  # (time taps: separate 2-D Winograd convs, summed)
                xt = x[:, :, dt:T - 2 + dt].transpose(1, 2).reshape(N * (T - 2), C, H, W)
                out = out + wino2d(xt, w[:, :, dt])
            y = out.reshape(N, T - 2, w.shape[0], H - 2, W - 2).transpose(1, 2)
            return y if b is None else y + b.view(1, -1, 1, 1, 1)
        return TF.conv3d(q(x), q(w), b, stride=stride, padding=padding)


def main():
    T, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (5, 64, 64)
    torch.manual_seed(0)
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model()
    P.load_seeded(m, 0)
    sd = {k: v.float() for k, v in m.state_dict().items()}
    cfg = dict(m.encoder._cfg)
    g = torch.Generator().manual_seed(1000)
    x = torch.rand((1, 3, T, H, W), generator=g) * 2 - 1
    res = {}
    for form in ("exact", "direct") + (() if os.environ.get("NO_WINO") else ("wino",)):
        O.F = TF if form == "exact" else Emu(form)
        try:
            with torch.no_grad():
                mom = O.sd3_encoder(x, sd, cfg)
                rec = O.sd3_decoder(mom[:, :16], sd, dict(m.decoder._cfg))
        finally:
            O.F = TF
        res[form] = (mom[:, :16], rec)
        if form != "exact":
            dm = float((mom[:, :16] - res["exact"][0]).abs().max())
            dr = float((rec - res["exact"][1]).abs().max())
            print(f"{form:7s} operands at {QBITS} significant bits: latent max|d| {dm:.2e}  recon max|d| {dr:.2e}   "
                  f"(clip {T}x{H}x{W}, latent max {float(res['exact'][0].abs().max()):.2f})", flush=True)


if __name__ == "__main__":
    main()
