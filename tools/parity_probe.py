#!/usr/bin/env python3
"""Stage-by-stage parity probe (diagnostic, GPU box): every kernel of one SelfAttentionBlock, one cross fusion and one head is
compared IN ISOLATION against the oracle evaluated on the kernel's own GPU inputs, at the real configs[1] shapes and with
the real data flow (not random operands).  Tells which stage contributes what to the end-to-end deviation.

    python tools/parity_probe.py [config] [batch]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cross-attention-vit_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import ref_cpu as R  # noqa: E402
import xvit  # noqa: E402
import xvit.functional as XF  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def q(t):
    return R.bf16_round(t.float())


def line(name, got, ref):
    print(f"  {name:34s} rel-L2 {rel(got, ref):9.3e}   max|d| {float((got.float().cpu() - ref.float().cpu()).abs().max()):9.3e}   |ref| {float(ref.float().abs().mean()):9.3e}", flush=True)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "base"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    cfg = R.make_config(name)
    sd = R.make_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, B, seed=0)
    model = xvit.ModelCross(cfg).to(dev)
    model.load_state_dict(sd)
    model.train()
    H, d = cfg.num_heads, cfg.hidden_dim
    cpu = lambda t: t.detach().float().cpu()   # noqa: E731

    with torch.no_grad():
        model._sync_flat_weights()
        toks = XF.PatchEmbedFn.apply(img.to(dev), model.patch_to_embedding.weight, model.patch_to_embedding.bias, model.cls_token, model.pos_embedding, model.patch_size, 0.0)
        with R.emulate_bf16():
            ref_toks = R.embed(sd, img, cfg)
        print("embed"); line("tokens[0]", toks[0], ref_toks[0])
        N = toks[0].shape[1]

        def probe_sab(tag, prefix, x):
            blk = model.get_submodule(prefix)
            a, f = blk.attn, blk.ffn
            sh = [XF.SHADOWS.get(w) for w in (a.fn.to_qkv.weight, a.fn.to_out[0].weight, f.fn.net[0].weight, f.fn.net[3].weight)]
            x2d = x.reshape(B * N, d).contiguous()
            x2, sv = XF.block_forward(x2d, B, N, H, a.norm.eps, (d // H) ** -0.5, a.norm.weight, a.norm.bias, sh[0], None, sh[1], a.fn.to_out[0].bias,
                                      f.norm.weight, f.norm.bias, sh[2], f.fn.net[0].bias, sh[3], f.fn.net[3].bias)
            (x_, mu1, rs1, h1, qkv, o, lse, x1, mu2, rs2, h2, z, a_) = sv
            P = prefix
            print(f"{tag} ({prefix})")
            line("LN1 -> h1 (bf16)", h1, q(R.layer_norm(cpu(x2d), sd[P + ".attn.norm.weight"], sd[P + ".attn.norm.bias"])))
            line("qkv GEMM (bf16)", qkv, q(cpu(h1) @ q(sd[P + ".attn.fn.to_qkv.weight"]).T))
            qq, kk, vv = (R._split_heads(t.reshape(B, N, d), H) for t in cpu(qkv).split(d, dim=-1))
            with R.emulate_bf16():
                o_ref, lse_ref = R.softmax_attention(qq, kk, vv, (d // H) ** -0.5)
            line("flash attention o (bf16)", o.reshape(B, N, d), q(R._merge_heads(o_ref)))
            line("lse", lse, lse_ref)
            line("out-proj + res -> x1 (f32)", x1, cpu(o) @ q(sd[P + ".attn.fn.to_out.0.weight"]).T + sd[P + ".attn.fn.to_out.0.bias"] + cpu(x2d))
            line("LN2 -> h2 (bf16)", h2, q(R.layer_norm(cpu(x1), sd[P + ".ffn.norm.weight"], sd[P + ".ffn.norm.bias"])))
            pre = cpu(h2) @ q(sd[P + ".ffn.fn.net.0.weight"]).T + sd[P + ".ffn.fn.net.0.bias"]
            if XF.AUX_MODE == 1:   # the FFN saves gelu'(z): Phi(z) + z phi(z)
                dg = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5
                line("FFN1 saved gelu'(z) (bf16)", z, q(dg))
            else:
                line("FFN1 pre-act z (bf16)", z, q(pre))
            line("FFN1 GELU a (bf16)", a_, q(R.gelu(pre)))
            line("FFN2 + res -> x2 (f32)", x2, cpu(a_) @ q(sd[P + ".ffn.fn.net.3.weight"]).T + sd[P + ".ffn.fn.net.3.bias"] + cpu(x1))
            return x2.reshape(B, N, d)

        xs = []
        for m in range(cfg.num_modalities):
            x = toks[m]
            for s in range(cfg.num_self_blocks):
                x = probe_sab(f"SAB m{m} s{s}", f"transformer.0.blocks.{m}.{s}", x) if m == 0 else model.transformer[0].blocks[m][s](x)
            xs.append(x)

        # one cross fusion: cls of 0 + patches of 1
        blk = model.transformer[0].fusion[0]
        P = "transformer.0.fusion.0"
        a, f, c = blk.attn, blk.ffn, blk.attn.fn
        sh = (XF.SHADOWS.get(c.wq.weight), XF.SHADOWS.get(c.wk.weight, c.wv.weight), XF.SHADOWS.get(c.proj.weight), XF.SHADOWS.get(f.fn.net[0].weight), XF.SHADOWS.get(f.fn.net[3].weight))
        bkv = torch.cat((c.wk.bias, c.wv.bias)).detach()
        xi, xj = xs[0].reshape(B * N, d).contiguous(), xs[1].reshape(B * N, d).contiguous()
        y2, sv = XF.cross_forward(xi, xj, B, N, H, a.norm.eps, a.norm.weight, a.norm.bias, c.wq.weight.detach(), c.wq.bias, sh[1], bkv, c.proj.weight.detach(), c.proj.bias,
                                  f.norm.weight, f.norm.bias, f.fn.net[0].weight.detach(), f.fn.net[0].bias, f.fn.net[3].weight.detach(), f.fn.net[3].bias)
        (xi_, xj_, mu, rs, hn, kv, qv, oc, pr, y, mu2, rs2, h2, z, a_) = sv
        print(f"cross fusion ({P})")
        cat = torch.cat((cpu(xs[0])[:, 0:1], cpu(xs[1])[:, 1:]), dim=1)
        line("LN(concat) -> hn (bf16)", hn.reshape(B, N, d), q(R.layer_norm(cat, sd[P + ".attn.norm.weight"], sd[P + ".attn.norm.bias"])))
        wkv = torch.cat((sd[P + ".attn.fn.wk.weight"], sd[P + ".attn.fn.wv.weight"]))
        line("kv GEMM (bf16)", kv, q(cpu(hn) @ q(wkv).T + cpu(bkv)))
        hn0 = R.layer_norm(cat, sd[P + ".attn.norm.weight"], sd[P + ".attn.norm.bias"])[:, 0]          # fp32 single-token path from here on
        line("q = wq(cls) (bf16 copy)", qv, q(hn0 @ sd[P + ".attn.fn.wq.weight"].T + sd[P + ".attn.fn.wq.bias"]))
        kk, vv = (R._split_heads(t.reshape(B, N, d), H) for t in cpu(kv).split(d, dim=-1))
        qq = R._split_heads((hn0 @ sd[P + ".attn.fn.wq.weight"].T + sd[P + ".attn.fn.wq.bias"]).reshape(B, 1, d), H)
        o_ref, _ = R.softmax_attention(qq, kk, vv, (d // H) ** -0.5)
        line("cls attention oc (bf16)", oc, q(R._merge_heads(o_ref).reshape(B, d)))
        y_ref = R._merge_heads(o_ref).reshape(B, d) @ sd[P + ".attn.fn.proj.weight"].T + sd[P + ".attn.fn.proj.bias"] + cpu(xs[0])[:, 0]
        line("proj + res -> y (f32)", y, y_ref)
        h2_ref = R.layer_norm(cpu(y), sd[P + ".ffn.norm.weight"], sd[P + ".ffn.norm.bias"])
        line("LN2 -> h2 (bf16 copy)", h2, q(h2_ref))
        pre = h2_ref @ sd[P + ".ffn.fn.net.0.weight"].T + sd[P + ".ffn.fn.net.0.bias"]
        line("FFN1 z (bf16 copy)", z, q(pre))
        line("FFN1 GELU a (bf16 copy)", a_, q(R.gelu(pre)))
        line("FFN2 + res -> y2 (f32)", y2, R.gelu(pre) @ sd[P + ".ffn.fn.net.3.weight"].T + sd[P + ".ffn.fn.net.3.bias"] + cpu(y))

        # end to end, against the emulating and the exact oracle
        cap, cap32 = {}, {}
        caps = {}
        hooks = [b.register_forward_hook(lambda m_, i_, o_, k=k: caps.__setitem__(k, [t.detach() for t in o_])) for k, b in enumerate(model.transformer)]
        logits, loss = model(img.to(dev), labels.to(dev))
        for h in hooks:
            h.remove()
        with R.emulate_bf16():
            lg_e, _ = R.model_cross_forward(sd, img, labels, cfg, capture=cap)
        lg_f, _ = R.model_cross_forward(sd, img, labels, cfg, capture=cap32)
        print("end to end (GPU vs bf16-emulating oracle | GPU vs fp32 oracle | emulating vs fp32 oracle)")
        for b in range(cfg.num_multi_blocks):
            for m in range(cfg.num_modalities):
                g, e, f32 = caps[b][m], cap[f"msb{b}"][m], cap32[f"msb{b}"][m]
                print(f"  msb{b} mod{m}: all tokens {rel(g, e):.3e} | {rel(g, f32):.3e} | {rel(e, f32):.3e}    cls rows {rel(g[:, 0], e[:, 0]):.3e} | {rel(g[:, 0], f32[:, 0]):.3e} | {rel(e[:, 0], f32[:, 0]):.3e}")
        print(f"  logits: {rel(logits, lg_e):.3e} | {rel(logits, lg_f):.3e} | {rel(lg_e, lg_f):.3e}")
        print("  logits GPU", logits.detach().cpu().tolist(), "\n  logits emu", lg_e.tolist(), "\n  logits f32", lg_f.tolist())


if __name__ == "__main__":
    main()
