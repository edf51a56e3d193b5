"""Fused engine for NestFuse / RFN-Nest (reference core/model.py:319-384, core/block.py:708-759,836-867) on
blocked-NHWC buffers: every concat is zero-copy, pooling / up-sampling / attention fusion / residual adds are HIP
kernels, the whole network is one autograd node.

Buffer plan.  Encoder features of both images share one allocation per level, E_l = [feat(img1) | feat(img2)]
(what RFN's `res` conv and the attention kernels read).  The UNet++ decoder rows are one allocation each with the
up-sampled operand FIRST, R_l = [U | f_l | x_{l,1} | x_{l,2}], so the input of every decoder block is a contiguous
prefix of R_l (the reference concatenates [f, x.., up]; the first conv of a decoder block therefore runs on
input-channel-permuted weights, ConvSpec.split).  The U slot is re-used by the row's blocks, so backward
re-materialises it (one up-sample kernel) before a block's wgrad.
"""
import torch

from . import _lib
from . import tensor as T
from .engine import ConvSpec, ModelEngine, all_bits, bits, conv_impl


def rng_bits(lo, n):
    return ((1 << n) - 1) << lo


class NestEngine(ModelEngine):
    CH = [64, 112, 160, 208]

    def __init__(self, module, rfn):
        m = module
        self.rfn = rfn

        def cs(name, layer, split=0):
            return ConvSpec(name, layer.layers[0], layer.act is not None, split)

        self.conv_in = cs("conv_in", m.conv_in)
        self.cb = [(cs(f"CB{i + 1}_0.0", b.layers[0]), cs(f"CB{i + 1}_0.1", b.layers[1])) for i, b in enumerate((m.CB1_0, m.CB2_0, m.CB3_0, m.CB4_0))]
        c = self.CH
        d = m.decode
        # (block, split = channels of the non-upsampled part of its concat)
        self.db = {k: (cs(f"decode.{k}.0", getattr(d, k).layers[0], split), cs(f"decode.{k}.1", getattr(d, k).layers[1]))
                   for k, split in (("DB1_1", c[0]), ("DB2_1", c[1]), ("DB3_1", c[2]), ("DB1_2", 2 * c[0]), ("DB2_2", 2 * c[1]), ("DB1_3", 3 * c[0]))}
        self.conv_out = cs("conv_out", m.conv_out)
        specs = [self.conv_in] + [s for p in self.cb for s in p] + [s for p in self.db.values() for s in p] + [self.conv_out]
        self.rfns = []
        if rfn:
            for i, r in enumerate((m.RFN1, m.RFN2, m.RFN3, m.RFN4)):
                d_ = dict(res=cs(f"RFN{i + 1}.res", r.res), conv1=cs(f"RFN{i + 1}.conv1", r.conv1), conv2=cs(f"RFN{i + 1}.conv2", r.conv2),
                          l0=cs(f"RFN{i + 1}.layers.0", r.layers[0]), l1=cs(f"RFN{i + 1}.layers.1", r.layers[1]), l2=cs(f"RFN{i + 1}.layers.2", r.layers[2]))
                self.rfns.append(d_)
                specs += list(d_.values())
        super().__init__(module, specs)
        self._attn_ws = None

    # ------------------------------------------------------------------------------------------
    def forward(self, img1, img2):
        if img2 is None:
            raise ValueError("NestFuse / RFNNest fuse two images")
        (img1, img2), n, h, w, dtype, impl = self.prepare((img1, img2))
        dev = img1.device
        c = self.CH
        hs = [(h, w)]
        for _ in range(3):
            hs.append((hs[-1][0] // 2, hs[-1][1] // 2))
        if hs[3][0] < 2 or hs[3][1] < 2:
            raise ValueError("NestFuse needs images of at least 16x16")
        L = self.lease((n, h, w, dtype), dev)
        L.imgs = (img1, img2)
        L.hs = hs
        B = lambda name, ch, lvl, halo=0: self.buf(L, name, n, ch, hs[lvl][0], hs[lvl][1], dtype, dev, halo)
        cb = lambda ch: ch // 8
        # one attention workspace per level, owned by the lease: the backward reuses the channel sums its forward left there
        if getattr(L, "attn_ws", None) is None or L.attn_ws[0].device != dev:
            L.attn_ws = [T.attn_workspace(n, c[l], dev) for l in range(4)]
        # ---- encoders (shared weights), features into E_l = [img1 | img2]
        E = [B(f"E{l}", 2 * c[l], l) for l in range(4)]
        for k, img in enumerate((img1, img2)):
            a0 = B(f"A0_{k}", 16, 0)
            T.image_in_fwd(img, self.conv_in.w.detach(), self.conv_in.b.detach(), a0, 16, self.conv_in.k, self.conv_in.relu)
            x = a0
            for l in range(4):
                s3, s1 = self.cb[l]
                hbuf = B(f"H{l}_{k}", s3.cout, l)
                self.c_fwd(s3, x, hbuf, impl)
                self.c_fwd(s1, hbuf, E[l].view(k * cb(c[l]), cb(c[l])), impl)
                if l < 3:
                    p = B(f"P{l + 1}_{k}", c[l], l + 1)
                    T.maxpool_fwd(E[l].view(k * cb(c[l]), cb(c[l])), p)
                    x = p
        # ---- decoder rows: R_l = [U | f_l | x_l1 | x_l2]
        R = [B("R0", c[1] + 3 * c[0], 0), B("R1", c[2] + 2 * c[1], 1), B("R2", c[3] + c[2], 2), B("F3", c[3], 3)]
        L.R, L.E = R, E
        uo = [cb(c[1]), cb(c[2]), cb(c[3]), 0]          # blocks of the U slot per row
        fslot = [R[l].view(uo[l], cb(c[l])) for l in range(4)]
        L.uo, L.fslot = uo, fslot
        # ---- fusion
        for l in range(4):
            e1, e2 = E[l].view(0, cb(c[l])), E[l].view(cb(c[l]), cb(c[l]))
            if not self.rfn:
                T.attn_fwd(e1, e2, fslot[l], T.ATTN_MODES["sca"], L.attn_ws[l])
            else:
                r = self.rfns[l]
                res = B(f"RES{l}", c[l], l)
                fc = B(f"FC{l}", 2 * c[l], l)
                t0, t1, t2 = B(f"T0_{l}", c[l], l), B(f"T1_{l}", c[l], l), B(f"T2_{l}", c[l], l)
                self.c_fwd(r["res"], E[l], res, impl)
                self.c_fwd(r["conv1"], e1, fc.view(0, cb(c[l])), impl)
                self.c_fwd(r["conv2"], e2, fc.view(cb(c[l]), cb(c[l])), impl)
                self.c_fwd(r["l0"], fc, t0, impl)
                self.c_fwd(r["l1"], t0, t1, impl)
                self.c_fwd(r["l2"], t1, t2, impl)
                T.fuse_elem_fwd(t2, res, fslot[l], _lib.FUSE_SUM)
        # ---- nested decoder
        X31, X22, X13 = B("X31", c[2], 2), B("X22", c[1], 1), B("X13", c[0], 0)
        x11, x12 = R[0].view(uo[0] + cb(c[0]), cb(c[0])), R[0].view(uo[0] + 2 * cb(c[0]), cb(c[0]))
        x21 = R[1].view(uo[1] + cb(c[1]), cb(c[1]))
        L.dec = dict(DB1_1=(0, 1, fslot[1], x11), DB2_1=(1, 1, fslot[2], x21), DB3_1=(2, 1, R[3], X31),
                     DB1_2=(0, 2, x21, x12), DB2_2=(1, 2, X31, X22), DB1_3=(0, 3, X22, X13))
        for name in ("DB1_1", "DB2_1", "DB3_1", "DB1_2", "DB2_2", "DB1_3"):
            row, nf, src, dst = L.dec[name]
            self._db_fwd(L, name, row, nf, src, dst, impl)
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
        T.image_out_fwd(X13, self.conv_out.w.detach(), self.conv_out.b.detach(), out, c[0], self.conv_out.k, self.conv_out.relu)
        L.out = out.detach() if self.conv_out.relu else None     # (an alias without the grad_fn the returned object gets)
        return out, L

    def _row_in(self, L, row, nf):
        """input view of a decoder block on row `row` that reads nf feature slots: [U | f | x.. ] prefix"""
        c = self.CH
        return L.R[row].view(0, L.uo[row] + nf * (c[row] // 8))

    def _db_fwd(self, L, name, row, nf, src, dst, impl):
        s3, s1 = self.db[name]
        n, h, w, dtype = L.key
        T.upsample_fwd(src, L.R[row].view(0, L.uo[row]))
        hd = self.buf(L, "HD_" + name, n, s3.cout, L.hs[row][0], L.hs[row][1], dtype, src.buf.device)
        self.c_fwd(s3, self._row_in(L, row, nf), hd, impl)
        self.c_fwd(s1, hd, dst, impl)

    # ------------------------------------------------------------------------------------------
    def backward(self, L, gout):
        n, h, w, dtype = L.key
        dev = gout.device
        impl = conv_impl()
        c = self.CH
        cb = lambda ch: ch // 8
        flat, grads = self._assign_grad_views(dev)
        ws = self.workspace(dev)
        hs, R, E, uo = L.hs, L.R, L.E, L.uo
        G = lambda name, ch, lvl: self.buf(L, "G_" + name, n, ch, hs[lvl][0], hs[lvl][1], dtype, dev, halo=1)
        GR = [G("R0", c[1] + 3 * c[0], 0), G("R1", c[2] + 2 * c[1], 1), G("R2", c[3] + c[2], 2), G("F3", c[3], 3)]
        GX31, GX22, GX13 = G("X31", c[2], 2), G("X22", c[1], 1), G("X13", c[0], 0)
        X31, X22, X13 = L.bufs["X31"], L.bufs["X22"], L.bufs["X13"]
        # ---- conv_out (1x1, ReLU)
        co = self.conv_out
        T.image_out_wgrad(X13, gout, L.out, co.dw, co.db, c[0], co.k, ws)
        T.image_out_dgrad(gout, L.out, co.w.detach(), X13, GX13, c[0], co.k, all_bits(GX13.cb), 0)
        gX13 = GX13.as_folded()

        def slot(row, i):  # i-th feature slot (0 = fused feature f) of row `row`, in blocks
            return uo[row] + i * cb(c[row]), cb(c[row])

        def db_bwd(name, g, accum_slots, mask_slots):
            """backward of decoder block `name` given the (masked, folded) gradient g of its output"""
            row, nf, src, dst = L.dec[name]
            s3, s1 = self.db[name]
            hd = L.bufs["HD_" + name]
            ghd = self.buf(L, "G_HD_" + name, n, s3.cout, hs[row][0], hs[row][1], dtype, dev, halo=1)
            self.c_wgrad(s1, hd, g, ws, impl)
            g_hd = self.c_dgrad(s1, g, hd, ghd, all_bits(ghd.cb), 0, impl)
            if name not in ("DB1_3", "DB2_2", "DB3_1"):           # (the LAST block of a row in the forward order still finds its own U there)
                T.upsample_fwd(src, R[row].view(0, uo[row]))      # re-materialise this block's U operand
            xin = self._row_in(L, row, nf)
            self.c_wgrad(s3, xin, g_hd, ws, impl)
            ab = mb = 0
            for i in accum_slots:
                ab |= rng_bits(*slot(row, i))
            for i in mask_slots:
                mb |= rng_bits(*slot(row, i))
            gin = GR[row].view(0, xin.cb)
            return self.c_dgrad(s3, g_hd, xin, gin, mb, ab, impl)

        # order: DB1_3, DB2_2, DB1_2, DB3_1, DB1_1, DB2_1 (every gradient is complete when it is consumed)
        gi = db_bwd("DB1_3", gX13, (), (2,))                       # x1_2 slot: only consumer -> masked here
        T.upsample_bwd(gi.view(0, uo[0]), GX22, False, relu_of=X22)      # (the ReLU mask of x2_2 rides in its only contribution)
        gi = db_bwd("DB2_2", GX22.as_folded(), (), ())
        T.upsample_bwd(gi.view(0, uo[1]), GX31, False, relu_of=X31)
        gi = db_bwd("DB1_2", GR[0].as_folded().view(*slot(0, 2)), (0, 1), (1,))     # accumulate into f0, x1_1; x1_1 done -> mask
        T.upsample_bwd(gi.view(0, uo[0]), GR[1].view(*slot(1, 1)), True, relu_of=R[1].view(*slot(1, 1)))   # + DB2_2's contribution to x2_1: the last one -> mask
        gi = db_bwd("DB3_1", GX31.as_folded(), (), ())
        T.upsample_bwd(gi.view(0, uo[2]), GR[3], False)                             # gradient of the fused level-3 feature
        gi = db_bwd("DB1_1", GR[0].as_folded().view(*slot(0, 1)), (0,), ())
        T.upsample_bwd(gi.view(0, uo[0]), GR[1].view(*slot(1, 0)), True)            # into f1
        gi = db_bwd("DB2_1", GR[1].as_folded().view(*slot(1, 1)), (0,), ())
        T.upsample_bwd(gi.view(0, uo[1]), GR[2].view(*slot(2, 0)), True)            # into f2
        gf = [GR[l].as_folded().view(*slot(l, 0)) for l in range(3)] + [GR[3].as_folded()]
        # (data parallel) the decoder's gradients are final: their all-reduce runs beside the fusion / encoder backward
        self.early_reduce(flat, [s for pr in self.db.values() for s in pr] + [self.conv_out])

        # ---- fusion backward -> G_E_l = [grad feat(img1) | grad feat(img2)]
        GE = [G(f"E{l}", 2 * c[l], l) for l in range(4)]
        for l in range(4):
            e1, e2 = E[l].view(0, cb(c[l])), E[l].view(cb(c[l]), cb(c[l]))
            g1, g2 = GE[l].view(0, cb(c[l])), GE[l].view(cb(c[l]), cb(c[l]))
            if not self.rfn:
                T.attn_bwd(e1, e2, gf[l], g1, g2, T.ATTN_MODES["sca"], False, L.attn_ws[l], cached=True)
            else:
                r = self.rfns[l]
                res, fc = L.bufs[f"RES{l}"], L.bufs[f"FC{l}"]
                t0, t1, t2 = L.bufs[f"T0_{l}"], L.bufs[f"T1_{l}"], L.bufs[f"T2_{l}"]
                gt2, gres, gt1, gt0, gfc = G(f"T2_{l}", c[l], l), G(f"RES{l}", c[l], l), G(f"T1_{l}", c[l], l), G(f"T0_{l}", c[l], l), G(f"FC{l}", 2 * c[l], l)
                T.fuse_elem_bwd(t2, res, gf[l], gt2, gres, _lib.FUSE_SUM, True)     # both addends are ReLU outputs
                gt2, gres = gt2.as_folded(), gres.as_folded()
                self.c_wgrad(r["l2"], t1, gt2, ws, impl)
                g_t1 = self.c_dgrad(r["l2"], gt2, t1, gt1, all_bits(gt1.cb), 0, impl)
                self.c_wgrad(r["l1"], t0, g_t1, ws, impl)
                g_t0 = self.c_dgrad(r["l1"], g_t1, t0, gt0, all_bits(gt0.cb), 0, impl)
                self.c_wgrad(r["l0"], fc, g_t0, ws, impl)
                g_fc = self.c_dgrad(r["l0"], g_t0, fc, gfc, all_bits(gfc.cb), 0, impl)
                self.c_wgrad(r["res"], E[l], gres, ws, impl)
                self.c_dgrad(r["res"], gres, None, GE[l], 0, 0, impl)
                self.c_wgrad(r["conv1"], e1, g_fc.view(0, cb(c[l])), ws, impl)
                self.c_dgrad(r["conv1"], g_fc.view(0, cb(c[l])), None, g1, 0, all_bits(g1.cb), impl)
                self.c_wgrad(r["conv2"], e2, g_fc.view(cb(c[l]), cb(c[l])), ws, impl)
                self.c_dgrad(r["conv2"], g_fc.view(cb(c[l]), cb(c[l])), None, g2, 0, all_bits(g2.cb), impl)
        # ---- encoders (shared weights: the second image accumulates)
        for k, img in enumerate(L.imgs):
            acc_w = k == 1
            for l in range(3, -1, -1):
                s3, s1 = self.cb[l]
                e = E[l].view(k * cb(c[l]), cb(c[l]))
                ge = GE[l].view(k * cb(c[l]), cb(c[l]))
                if l == 3:
                    T.relu_mask_(e, ge)                    # level 3 has the fusion's contribution only
                ge = ge.as_folded()                        # (levels 0..2: masked by the max-pool backward of level l + 1 below, their last contribution)
                hbuf = L.bufs[f"H{l}_{k}"]
                x = L.bufs[f"A0_{k}"] if l == 0 else L.bufs[f"P{l}_{k}"]
                self.c_wgrad(s1, hbuf, ge, ws, impl, acc_w)
                gh = self.c_dgrad(s1, ge, hbuf, G(f"H{l}_{k}", s3.cout, l), all_bits((s3.cout + 7) // 8), 0, impl)
                self.c_wgrad(s3, x, gh, ws, impl, acc_w)
                if l == 0:
                    ga0 = self.c_dgrad(s3, gh, x, G(f"A0_{k}", 16, 0), all_bits(2), 0, impl)
                    ci = self.conv_in
                    T.image_in_wgrad(img, ga0, ci.dw, ci.db, 16, ci.k, ws, acc_w)
                else:
                    gp = self.c_dgrad(s3, gh, None, G(f"P{l}_{k}", c[l - 1], l), 0, 0, impl)
                    T.maxpool_bwd(E[l - 1].view(k * cb(c[l - 1]), cb(c[l - 1])), gp, GE[l - 1].view(k * cb(c[l - 1]), cb(c[l - 1])), True, relu=True)
        return grads
