"""State-dict key -> shape tables for both families, derived from the config alone (no module
construction).  TEST INFRASTRUCTURE: lets the oracle/golden tests build seeded weights without the
reference or the product.  Key names follow the reference's on-disk contract;
tests/test_state_dict_contract.py checks them against the product's modules and, when /root/reference
is present, against the reference's own state_dict()."""


def _conv(d, pre, co, ci, k):
    d[pre + ".weight"] = (co, ci) + tuple(k)
    d[pre + ".bias"] = (co,)


def _norm(d, pre, c):
    d[pre + ".weight"] = (c,)
    d[pre + ".bias"] = (c,)


def _lin(d, pre, co, ci):
    d[pre + ".weight"] = (co, ci)
    d[pre + ".bias"] = (co,)


def sd3_shapes(cfg):
    boc = list(cfg.get("block_out_channels", [128, 256, 512, 512]))
    lpb = cfg.get("layers_per_block", 2)
    cin = cfg.get("in_channels", 3)
    z = cfg.get("out_channels", 16)
    dz = 2 * z if cfg.get("double_z", True) else z
    attn = cfg.get("mid_block_add_attention", True)
    d = {}

    def resnet(pre, ci, co):
        _norm(d, pre + ".norm1", ci)
        _conv(d, pre + ".conv1", co, ci, (3, 3, 3))
        _norm(d, pre + ".norm2", co)
        _conv(d, pre + ".conv2", co, co, (3, 3))
        if ci != co:
            _conv(d, pre + ".conv_shortcut", co, ci, (1, 1))

    def mid(pre, c):
        resnet(pre + ".resnets.0", c, c)
        if attn:
            a = pre + ".attentions.0"
            _norm(d, a + ".group_norm", c)
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                _lin(d, a + "." + n, c, c)
        resnet(pre + ".resnets.1", c, c)

    _conv(d, "encoder.conv_in", boc[0], cin, (3, 3, 3))
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(lpb):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co)
        ch = co
        if i != len(boc) - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, (3, 3, 3))
    mid("encoder.mid_block", boc[-1])
    _norm(d, "encoder.conv_norm_out", boc[-1])
    _conv(d, "encoder.conv_out", dz, boc[-1], (3, 3, 3))

    rev = list(reversed(boc))
    _conv(d, "decoder.conv_in", rev[0], z, (3, 3, 3))
    mid("decoder.mid_block", rev[0])
    ch = rev[0]
    for i, co in enumerate(rev):
        final = i == len(rev) - 1
        for j in range(lpb + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else co, co)
        ch = co
        if not final:
            up_time = 2 if i % 2 == 0 else 1
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", co * up_time, co, (3, 3, 3))
    _norm(d, "decoder.conv_norm_out", boc[0])
    _conv(d, "decoder.conv_out", cin, boc[0], (3, 3, 3))
    return d


def vae3d_shapes(cfg):
    ch = cfg.get("ch", 128)
    mult = list(cfg.get("ch_mult", [1, 2, 4, 4]))
    nrb = cfg.get("num_res_blocks", 2)
    cin = cfg.get("in_channels", 3)
    out_ch = cfg.get("out_ch", 3)
    z = cfg.get("z_channels", 4)
    dz = 2 * z if cfg.get("double_z", True) else z
    d = {}

    def resnet(pre, ci, co):
        _norm(d, pre + ".norm1", ci)
        _conv(d, pre + ".conv1", co, ci, (3, 3, 3))
        _norm(d, pre + ".norm2", co)
        _conv(d, pre + ".conv2", co, co, (3, 3))
        if ci != co:
            _conv(d, pre + ".nin_shortcut", co, ci, (1, 1, 1))

    def attn(pre, c, temporal):
        _norm(d, pre + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            _conv(d, pre + "." + n, c, c, (1, 1))
        if temporal:
            for n in ("q_t", "k_t", "v_t", "proj_out_t"):
                _lin(d, pre + "." + n, c, c)
            _norm(d, pre + ".norm_t", c)

    _conv(d, "encoder.conv_in", ch, cin, (3, 3, 3))
    in_mult = [1] + mult
    bi = ch
    for lvl in range(len(mult)):
        bi = ch * in_mult[lvl]
        bo = ch * mult[lvl]
        for j in range(nrb):
            resnet(f"encoder.down.{lvl}.block.{j}", bi, bo)
            bi = bo
        if lvl != len(mult) - 1:
            _conv(d, f"encoder.down.{lvl}.downsample.conv", bi, bi, (3, 3, 3))
    resnet("encoder.mid.block_1", bi, bi)
    attn("encoder.mid.attn_1", bi, False)
    resnet("encoder.mid.block_2", bi, bi)
    _norm(d, "encoder.norm_out", bi)
    _conv(d, "encoder.conv_out", dz, bi, (3, 3, 3))

    bi = ch * mult[-1]
    _conv(d, "decoder.conv_in", bi, z, (3, 3, 3))
    resnet("decoder.mid.block_1", bi, bi)
    attn("decoder.mid.attn_1", bi, True)
    resnet("decoder.mid.block_2", bi, bi)
    for lvl in reversed(range(len(mult))):
        bo = ch * mult[lvl]
        for j in range(nrb + 1):
            resnet(f"decoder.up.{lvl}.block.{j}", bi, bo)
            bi = bo
        if lvl != 0:
            up_time = 2 if lvl % 2 == 1 else 1
            _conv(d, f"decoder.up.{lvl}.upsample.conv", bi * up_time, bi, (3, 3, 3))
    _norm(d, "decoder.norm_out", bi)
    _conv(d, "decoder.conv_out", out_ch, bi, (3, 3, 3))
    return d


def state_dict_shapes(family, cfg=None):
    cfg = cfg or {}
    return sd3_shapes(cfg) if family == "sd3" else vae3d_shapes(cfg)


def constraint2d_shapes(cfg):
    """DecoderWith3DWrapper / Decoder (2-D, lvdm/modules/diffusionmodules/vae_models_sd3.py:196-398; blocks
    lvdm/modules/diffusionmodules/vae_blocks_sd3.py): the frozen SD3 image decoder of the training path."""
    boc = list(cfg.get("block_out_channels", [128, 256, 512, 512]))
    lpb = cfg.get("layers_per_block", 2)
    zin = cfg.get("in_channels", 16)
    cout = cfg.get("out_channels", 3)
    attn = cfg.get("mid_block_add_attention", True)
    d = {}
    top = boc[-1]
    _conv(d, "conv_in", top, zin, (3, 3))
    for j in range(2):
        pre = f"mid_block.resnets.{j}"
        _norm(d, pre + ".norm1", top); _conv(d, pre + ".conv1", top, top, (3, 3))
        _norm(d, pre + ".norm2", top); _conv(d, pre + ".conv2", top, top, (3, 3))
    if attn:
        a = "mid_block.attentions.0"
        _norm(d, a + ".group_norm", top)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            _lin(d, f"{a}.{n}", top, top)
    rev = list(reversed(boc))
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(lpb + 1):
            pre = f"up_blocks.{i}.resnets.{j}"
            ci = prev if j == 0 else c
            _norm(d, pre + ".norm1", ci); _conv(d, pre + ".conv1", c, ci, (3, 3))
            _norm(d, pre + ".norm2", c); _conv(d, pre + ".conv2", c, c, (3, 3))
            if ci != c:
                _conv(d, pre + ".conv_shortcut", c, ci, (1, 1))
        if i != len(rev) - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", c, c, (3, 3))
        prev = c
    _norm(d, "conv_norm_out", boc[0])
    _conv(d, "conv_out", cout, boc[0], (3, 3))
    return d

