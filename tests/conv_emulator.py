"""CPU replay of the INDEX ARITHMETIC of the fused BBBConv2d kernels (csrc/conv_lrt.hip, csrc/conv_lrt_bwd.hip).

Test infrastructure, not product.  The kernels were (re)written in round 4 while the GPU pool was closed for a while;
this module transcribes them statement by statement -- workgroup / wave / lane decomposition, LDS staging of the input
patch (zero padding after the clamp, dilation in the input-gradient mode), the pre-arranged weight layouts of
bde_conv_lrt_prep, the tap -> patch-offset table, the k-split over waves, the epilogue's LDS transposition with its
float4 groups, the weight-gradient kernel's pixel table / column offsets / partial blocks -- with numpy arrays standing
in for LDS and a dense [MF, KS] x [KS, MF] product standing in for one MFMA (lane l supplies A[l % MF][l / MF] and
B[l / MF][l % MF]; the accumulator register -> (row, column) map is the documented one that round 3's kernels and the
round-4 forward, version 1, were validated with on the MI355X).  The tilings come from the library itself
(bde_conv_lrt_plan / bde_conv_lrt_bwd_weight_plan run on the host), so what is replayed is what would be launched.

It cannot see wait counts, LDS hazards or register allocation -- the GPU tests (tests/test_ops_gpu.py::test_conv_lrt_*)
do -- but it pins every address the kernels compute against torch's own conv2d and autograd.
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from beyond_deep_ensembles_amd import _lib


def _softplus_sigmoid(x):
    sp = F.softplus(torch.from_numpy(x)).numpy()
    sg = torch.sigmoid(torch.from_numpy(x)).numpy()
    return sp, sg


def pad32(v):
    return (v + 31) // 32 * 32


def prep(w_mu, w_rho):
    """conv_lrt_prep_kernel: WT_mu / WT_s2 [ktot][op], WB_mu / WB_s2 [O khw][cp], DS2 [O][ktot]."""
    O, C, KH, KW = w_mu.shape
    khw, ktot, op, cp = KH * KW, C * KH * KW, pad32(O), pad32(C)
    wt_mu, wt_s2 = np.zeros((ktot, op), np.float32), np.zeros((ktot, op), np.float32)
    wb_mu, wb_s2 = np.zeros((O * khw, cp), np.float32), np.zeros((O * khw, cp), np.float32)
    sp, sg = _softplus_sigmoid(w_rho.reshape(O, ktot))
    s2 = sp * sp
    s2c = np.maximum(s2, np.float32(1e-4))
    mu = w_mu.reshape(O, ktot)
    for o in range(O):
        for k in range(ktot):
            c, rq = k // khw, k % khw
            wt_mu[k, o] = mu[o, k]
            wt_s2[k, o] = s2c[o, k]
            kb = o * khw + (khw - 1 - rq)
            wb_mu[kb, c] = mu[o, k]
            wb_s2[kb, c] = s2c[o, k]
    ds2 = np.where(s2 >= 1e-4, 2.0 * sp * sg, 0.0).astype(np.float32)
    return dict(wt_mu=wt_mu, wt_s2=wt_s2, wb_mu=wb_mu, wb_s2=wb_s2, ds2=ds2)


def plan(which, geo):
    out = (ctypes.c_int * 16)()
    rc = _lib.load().bde_conv_lrt_plan(which, *geo, out)
    assert rc == 0, (which, geo, rc)
    names = "MF PT NI TH bands CC PH PWP WP WK tiles_per_img kcpad_max gx gy lds _".split()
    return dict(zip(names, list(out)))


def plan_wgrad(geo):
    out = (ctypes.c_int * 16)()
    rc = _lib.load().bde_conv_lrt_bwd_weight_plan(*geo, out)
    assert rc == 0, (geo, rc)
    names = "MF CT_MAX NI TH bands PS CT colgroups PH PWP cmax GP npix gy lds otiles".split()
    return dict(zip(names, list(out)))


def _acc_row(mf, r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h if mf == 32 else 4 * h + r


def conv_kernel(mode, layer_geo, a1, a2, wt_mu, wt_s2, b1, b2, eps):
    """conv_lrt_kernel<MF, PT, RNG = false, MODE>.  mode 0: a1 = x, b1 = b_mu, b2 = b_var, returns (out, var_out).
    mode 1: a1 = g, a2 = gvar, b1 = x (the layer input), returns g_x."""
    N, C, H, W, O, KH, KW, sh, sw, ph, pw = layer_geo
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    p = plan(mode, layer_geo)
    if mode == 0:
        g = dict(N=N, C=C, H=H, W=W, O=O, KH=KH, KW=KW, sh=sh, sw=sw, ph=ph, pw=pw, Ho=Ho, Wo=Wo, dh=1, dw=1, rp=pad32(O))
    else:
        g = dict(N=N, C=O, H=Ho, W=Wo, O=C, KH=KH, KW=KW, sh=1, sw=1, ph=KH - 1 - ph, pw=KW - 1 - pw, Ho=H, Wo=W, dh=sh, dw=sw,
                 rp=pad32(C))
    MF, PT = p["MF"], p["PT"]
    KS = 2 if MF == 32 else 4
    REGS = 16 if MF == 32 else 4
    RS = MF + 4
    NI, TH, bands, CC, PH, PWP, WP, WK = (p[k] for k in ("NI", "TH", "bands", "CC", "PH", "PWP", "WP", "WK"))
    tpi, kcpad_max = p["tiles_per_img"], p["kcpad_max"]
    row_elems = PH * PWP
    img_floats = CC * row_elems
    patch_floats = (NI * img_floats + 3) & ~3                  # float4 (16-byte) alignment of the weight tiles behind the patches
    assert (2 * patch_floats) % 4 == 0 and MF % 4 == 0 and RS % 4 == 0 and (2 * MF * RS) % 4 == 0 and g["rp"] % 4 == 0
    khw = g["KH"] * g["KW"]
    howo = g["Ho"] * g["Wo"]
    out = np.full(g["N"] * g["O"] * howo, np.nan, np.float32)
    var_out = np.full(g["N"] * g["O"] * howo, np.nan, np.float32)
    a1f, a2f = a1.reshape(-1), (a2.reshape(-1) if a2 is not None else None)
    lds_floats = p["lds"] // 4
    assert 2 * patch_floats + 2 * kcpad_max * MF + kcpad_max <= lds_floats
    assert 4 * 2 * MF * RS <= lds_floats
    lanes = np.arange(64)
    h_l, idx_l = lanes // MF, lanes % MF
    for bx in range(p["gx"]):
        for by in range(p["gy"]):
            xs = np.full(patch_floats, np.nan, np.float32)      # NaN = never written: any use of it poisons the result
            x2s = np.full(patch_floats, np.nan, np.float32)
            wm = np.full(kcpad_max * MF, np.nan, np.float32)
            wsv = np.full(kcpad_max * MF, np.nan, np.float32)
            kofs = np.zeros(kcpad_max, np.int64)
            img0, band = (bx // bands) * NI, bx % bands
            o0 = by * MF
            ho0 = band * TH
            th = min(TH, g["Ho"] - ho0)
            band_pixels = th * g["Wo"]
            pixoff = np.zeros((4, PT, 64), np.int64)
            for wave in range(4):
                wp = wave % WP
                for i in range(PT):
                    tile = wp * PT + i
                    img, pp = tile // tpi, (tile % tpi) * MF + idx_l
                    ok = (img < NI) & (img0 + img < g["N"]) & (pp < band_pixels)
                    hl = np.where(ok, pp // g["Wo"], 0)
                    wo = np.where(ok, pp % g["Wo"], 0)
                    pixoff[wave, i] = np.where(ok, img, 0) * img_floats + hl * g["sh"] * PWP + wo * g["sw"]
            acc_m = np.zeros((4, PT, MF, MF), np.float64)
            acc_v = np.zeros((4, PT, MF, MF), np.float64)
            for k in range(kcpad_max):
                c, rq = k // khw, k % khw
                kofs[k] = c * row_elems + (rq // g["KW"]) * PWP + (rq % g["KW"]) if c < CC else 0
            hi0 = ho0 * g["sh"] - g["ph"]
            for c0 in range(0, g["C"], CC):
                cc = min(CC, g["C"] - c0)
                kc = cc * khw
                kcpad = (kc + KS * WK - 1) // (KS * WK) * (KS * WK)
                assert kcpad <= kcpad_max
                for wave in range(4):
                    for rc in range(wave, NI * cc, 4):
                        img, c = rc // cc, rc % cc
                        img_ok = img0 + img < g["N"]
                        plane = ((img0 + img if img_ok else 0) * g["C"] + c0 + c) * g["H"] * g["W"]
                        d = img * img_floats + c * row_elems
                        for px in range(PWP):
                            wi = px - g["pw"]
                            col_ok = img_ok and wi >= 0
                            if mode == 1 and g["dw"] != 1:
                                col_ok = col_ok and wi % g["dw"] == 0
                                wi //= g["dw"]
                            col_ok = col_ok and wi < g["W"]
                            for py in range(PH):
                                hi = hi0 + py
                                ok = col_ok and hi >= 0
                                if mode == 1 and g["dh"] != 1:
                                    ok = ok and hi % g["dh"] == 0
                                    hi //= g["dh"]
                                ok = ok and hi < g["H"]
                                v = a1f[plane + hi * g["W"] + wi] if ok else np.float32(0)
                                if mode == 0:
                                    v2 = max(np.float32(v * v), np.float32(1e-4)) if ok else np.float32(0)
                                else:
                                    v2 = a2f[plane + hi * g["W"] + wi] if ok else np.float32(0)
                                xs[d + py * PWP + px] = v
                                x2s[d + py * PWP + px] = v2
                Q = MF // 4
                k_base = c0 * khw
                for e in range(kcpad * Q):
                    k, o4 = e // Q, e % Q
                    if k < kc:
                        src = (k_base + k) * g["rp"] + o0 + 4 * o4
                        assert src % 4 == 0 and (k * MF + 4 * o4) % 4 == 0
                        wm[k * MF + 4 * o4:k * MF + 4 * o4 + 4] = wt_mu.reshape(-1)[src:src + 4]
                        wsv[k * MF + 4 * o4:k * MF + 4 * o4 + 4] = wt_s2.reshape(-1)[src:src + 4]
                    else:
                        wm[k * MF + 4 * o4:k * MF + 4 * o4 + 4] = 0
                        wsv[k * MF + 4 * o4:k * MF + 4 * o4 + 4] = 0
                ksteps = kcpad // KS
                for wave in range(4):
                    wp, wk = wave % WP, wave // WP
                    ks0 = wk * (ksteps // WK)
                    ks1 = ks0 + ksteps // WK
                    for ks in range(ks0, ks1):
                        kk = ks * KS + h_l
                        ko = kofs[kk]
                        am, asv = wm[kk * MF + idx_l], wsv[kk * MF + idx_l]
                        A_m, A_v = np.zeros((MF, KS)), np.zeros((MF, KS))
                        A_m[idx_l, h_l], A_v[idx_l, h_l] = am, asv
                        for i in range(PT):
                            bop, bop2 = xs[pixoff[wave, i] + ko], x2s[pixoff[wave, i] + ko]
                            B, B2 = np.zeros((KS, MF)), np.zeros((KS, MF))
                            B[h_l, idx_l], B2[h_l, idx_l] = bop, bop2
                            acc_m[wave, i] += A_m @ B
                            acc_v[wave, i] += A_v @ B2
            # k-split reduction in wave order
            for wave in range(4):
                wp, wk = wave % WP, wave // WP
                if wk != 0:
                    continue
                for s in range(1, WK):
                    acc_m[wave] += acc_m[wp + s * WP]
                    acc_v[wave] += acc_v[wp + s * WP]
                vec = (howo & 3) == 0 and ((TH * g["Wo"]) & 3) == 0
                for i in range(PT):
                    tile = wp * PT + i
                    img, p0 = tile // tpi, (tile % tpi) * MF
                    if img >= NI or img0 + img >= g["N"] or p0 >= band_pixels:
                        continue
                    base = (img0 + img) * g["O"] * howo + ho0 * g["Wo"] + p0
                    tm = np.full(MF * RS, np.nan)
                    tv = np.full(MF * RS, np.nan)
                    if vec:
                        for r in range(REGS):
                            for lane in range(64):
                                hh, ii = lane // MF, lane % MF
                                tm[_acc_row(MF, r, hh) * RS + ii] = acc_m[wave, i][_acc_row(MF, r, hh), ii]
                                tv[_acc_row(MF, r, hh) * RS + ii] = acc_v[wave, i][_acc_row(MF, r, hh), ii]
                        Q = MF // 4
                        for ps in range(MF * Q // 64):
                            for lane in range(64):
                                ch, p4 = ps * (64 // Q) + lane // Q, lane % Q
                                o = o0 + ch
                                if o < g["O"] and p0 + 4 * p4 < band_pixels:
                                    m4, v4 = tm[ch * RS + 4 * p4:ch * RS + 4 * p4 + 4], tv[ch * RS + 4 * p4:ch * RS + 4 * p4 + 4]
                                    e = base + o * howo + 4 * p4
                                    assert e % 4 == 0
                                    if mode == 1:
                                        xv = b1.reshape(-1)[e:e + 4]
                                        out[e:e + 4] = m4 + np.where(xv * xv >= 1e-4, 2.0 * xv * v4, 0.0)
                                    else:
                                        bm = b1[o] if b1 is not None else 0.0
                                        bv = b2[o] if b2 is not None else 0.0
                                        var = v4 + bv
                                        out[e:e + 4] = (m4 + bm) + np.sqrt(var) * eps.reshape(-1)[e:e + 4]
                                        var_out[e:e + 4] = var
                    else:
                        for lane in range(64):
                            hh, ii = lane // MF, lane % MF
                            if p0 + ii >= band_pixels:
                                continue
                            for r in range(REGS):
                                o = o0 + _acc_row(MF, r, hh)
                                if o < g["O"]:
                                    e = base + o * howo + ii
                                    cm, cv = acc_m[wave, i][_acc_row(MF, r, hh), ii], acc_v[wave, i][_acc_row(MF, r, hh), ii]
                                    if mode == 1:
                                        xv = b1.reshape(-1)[e]
                                        out[e] = cm + (2.0 * xv * cv if xv * xv >= 1e-4 else 0.0)
                                    else:
                                        var = cv + (b2[o] if b2 is not None else 0.0)
                                        out[e] = cm + (b1[o] if b1 is not None else 0.0) + np.sqrt(var) * eps.reshape(-1)[e]
                                        var_out[e] = var
    shape = (g["N"], g["O"], g["Ho"], g["Wo"])
    return (out.reshape(shape), var_out.reshape(shape)) if mode == 0 else out.reshape(shape)


def wgrad_kernel(layer_geo, x, gout, gvar, w_rho):
    """conv_lrt_wgrad_kernel + conv_lrt_wgrad_finish_kernel: returns (g_wmu, g_wrho)."""
    N, C, H, W, O, KH, KW, sh, sw, ph, pw = layer_geo
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    p = plan_wgrad(layer_geo)
    MF, CT_MAX = p["MF"], p["CT_MAX"]
    KS = 2 if MF == 32 else 4
    NI, TH, bands, PS, CT, colgroups, PH, PWP, cmax, GP, npix = (p[k] for k in ("NI", "TH", "bands", "PS", "CT", "colgroups", "PH", "PWP",
                                                                                "cmax", "GP", "npix"))
    row_elems = PH * PWP
    khw, ktot = KH * KW, C * KH * KW
    howo = Ho * Wo
    part = np.zeros((PS, 2, O, ktot), np.float64)
    written = np.zeros((PS, O, ktot), bool)
    xf, gf, gvf = x.reshape(-1), gout.reshape(-1), gvar.reshape(-1)
    lanes = np.arange(64)
    h_l, idx_l = lanes // MF, lanes % MF
    assert 2 * NI * cmax * row_elems + 2 * MF * GP + npix <= p["lds"] // 4
    for bx in range(PS):
        for by in range(p["gy"]):
            otile, cg = by // colgroups, by % colgroups
            o0, col0 = otile * MF, cg * CT * MF
            col_end = min(ktot, col0 + CT * MF)
            c_lo, c_hi = col0 // khw, (col_end - 1) // khw
            cc = c_hi - c_lo + 1
            assert cc <= cmax
            img_floats = cc * row_elems
            patch_floats = NI * cmax * row_elems
            kofs = np.zeros((CT_MAX, 64), np.int64)
            for ct in range(CT_MAX):
                col = col0 + ct * MF + idx_l
                ok = (ct < CT) & (col < ktot)
                c, rq = col // khw - c_lo, col % khw
                kofs[ct] = np.where(ok, c * row_elems + (rq // KW) * PWP + rq % KW, 0)
            acc_m = np.zeros((4, CT_MAX, MF, MF), np.float64)
            acc_v = np.zeros((4, CT_MAX, MF, MF), np.float64)
            bpi = TH * Wo
            items = ((N + NI - 1) // NI) * bands
            for item in range(bx, items, PS):
                xs = np.full(patch_floats, np.nan, np.float32)
                x2s = np.full(patch_floats, np.nan, np.float32)
                gs = np.full(MF * GP, np.nan, np.float32)
                gvs = np.full(MF * GP, np.nan, np.float32)
                pixtab = np.zeros(npix, np.int64)
                img0, band = (item // bands) * NI, item % bands
                ho0 = band * TH
                th = min(TH, Ho - ho0)
                hi0 = ho0 * sh - ph
                for rc in range(NI * cc):
                    img, c = rc // cc, rc % cc
                    img_ok = img0 + img < N
                    src = ((img0 + img if img_ok else 0) * C + c_lo + c) * H * W
                    d = img * img_floats + c * row_elems
                    for px in range(PWP):
                        wi = px - pw
                        col_ok = img_ok and 0 <= wi < W
                        for py in range(PH):
                            hi = hi0 + py
                            ok = col_ok and 0 <= hi < H
                            v = xf[src + hi * W + wi] if ok else np.float32(0)
                            xs[d + py * PWP + px] = v
                            x2s[d + py * PWP + px] = max(np.float32(v * v), np.float32(1e-4)) if ok else np.float32(0)
                for ro in range(MF * NI):
                    o, img = ro // NI, ro % NI
                    ok_row = img0 + img < N and o0 + o < O
                    src0 = ((img0 + img if ok_row else 0) * O + (o0 + o if ok_row else 0)) * howo + ho0 * Wo
                    for pp in range(bpi):
                        ok = ok_row and pp < th * Wo
                        gs[o * GP + img * bpi + pp] = gf[src0 + pp] if ok else 0
                        gvs[o * GP + img * bpi + pp] = gvf[src0 + pp] if ok else 0
                padn = npix - NI * bpi
                for e in range(MF * padn):
                    o, pp = e // padn, NI * bpi + e % padn
                    gs[o * GP + pp] = 0
                    gvs[o * GP + pp] = 0
                for pp in range(npix):
                    img, q = pp // bpi, pp % bpi
                    hl, wo = q // Wo, q % Wo
                    pixtab[pp] = img * img_floats + hl * sh * PWP + wo * sw if (img < NI and hl < th) else 0
                ksteps = npix // KS
                for wave in range(4):
                    for ks in range(wave, ksteps, 4):
                        pix = ks * KS + h_l
                        po = pixtab[pix]
                        am, av = gs[idx_l * GP + pix], gvs[idx_l * GP + pix]
                        A_m, A_v = np.zeros((MF, KS)), np.zeros((MF, KS))
                        A_m[idx_l, h_l], A_v[idx_l, h_l] = am, av
                        for ct in range(CT):
                            bop, bop2 = xs[po + kofs[ct]], x2s[po + kofs[ct]]
                            B, B2 = np.zeros((KS, MF)), np.zeros((KS, MF))
                            B[h_l, idx_l], B2[h_l, idx_l] = bop, bop2
                            acc_m[wave, ct] += A_m @ B
                            acc_v[wave, ct] += A_v @ B2
            for s in range(1, 4):
                acc_m[0] += acc_m[s]
                acc_v[0] += acc_v[s]
            for ct in range(CT):
                for j in range(MF):
                    col = col0 + ct * MF + j
                    if col < ktot:
                        for i in range(MF):
                            if o0 + i < O:
                                assert not written[bx, o0 + i, col]
                                written[bx, o0 + i, col] = True
                                part[bx, 0, o0 + i, col] = acc_m[0, ct, i, j]
                                part[bx, 1, o0 + i, col] = acc_v[0, ct, i, j]
    assert written.all(), "some (share, o, column) of the partials buffer is never written: the finish pass would read garbage"
    sm, sv = part[:, 0].sum(0), part[:, 1].sum(0)
    sp, sg = _softplus_sigmoid(w_rho.reshape(O, ktot))
    g_wrho = np.where(sp * sp >= 1e-4, sv * (2.0 * sp * sg), 0.0)
    return sm.reshape(w_rho.shape).astype(np.float32), g_wrho.reshape(w_rho.shape).astype(np.float32)
